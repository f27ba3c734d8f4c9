#!/usr/bin/env python
"""EBEN train-step benchmark (BASELINE.json metric: audio-seconds/sec of the full GAN step).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = EBENLightningModule.training_step on one batch of synthetic (32, 1, 32000) float32
waveforms already resident in HBM (cut to 31968): generator fwd, MRSTFT + feature-matching + hinge,
EMA loss balancing, generator backward + Adam, discriminator fwd/backward + Adam, and (N > 1) the
RCCL gradient all-reduces.  Weak scaling: 32 clips per GPU.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from functools import partial

# RCCL's version banner (NCCL_DEBUG=VERSION in this image) goes to STDOUT, where the driver reads one JSON line
if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
    os.environ["NCCL_DEBUG"] = "WARN"

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)


def build_module(device, batch_seed):
    from vibravox_amd.lightning_modules.eben import EBENLightningModule
    from vibravox_amd.optim import FusedAdam
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales
    from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator
    from vibravox_amd.torch_modules.losses.feature_loss import FeatureLossForDiscriminatorMelganMultiScales
    from vibravox_amd.torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales
    from vibravox_amd.torch_modules.losses.mrstft_loss import MultiResolutionSTFTLoss

    torch.manual_seed(42)  # mirrors run.py:74 seed_everything(42): identical weights on every rank
    gen, disc = EBENGenerator(m=4, n=32, p=2), DiscriminatorEBENMultiScales(q=4, min_channels=24)
    opt = partial(FusedAdam, lr=3e-4, betas=(0.5, 0.9))
    mod = EBENLightningModule(
        sample_rate=16000, generator=gen.to(device), discriminator=disc.to(device), generator_optimizer=opt,
        discriminator_optimizer=opt,
        reconstructive_loss_freq_fn=MultiResolutionSTFTLoss(fft_sizes=(512, 1024, 2048), hop_sizes=(50, 120, 240),
                                                            win_lengths=(240, 600, 1200), sample_rate=16000,
                                                            perceptual_weighting=True).to(device),
        feature_matching_loss_fn=FeatureLossForDiscriminatorMelganMultiScales(),
        adversarial_loss_fn=HingeLossForDiscriminatorMelganMultiScales(),
        dynamic_loss_balancing="ema", beta_ema=0.9, update_discriminator_ratio=1)
    return mod


def pmc_traffic(batch, math):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_melgan_l4_fwd[_bf16].json; collected by tools/pmc_traffic.sh at the batch the step launches)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_melgan_l4_fwd_bf16.json" if math == "bf16" else "r01_pmc_melgan_l4_fwd.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    return rec.get("traffic_bytes_per_launch") if rec.get("launch_batch") == batch else None


def synthetic_batch(batch, length, seed, device):
    g = torch.Generator().manual_seed(seed)
    return {"audio_body_conducted": (0.1 * torch.randn(batch, 1, length, generator=g)).to(device),
            "audio_airborne": (0.1 * torch.randn(batch, 1, length, generator=g)).to(device)}


def cpu_baseline(batch, length, steps):
    """The CPU oracle (oracle/eben_oracle.py, a restatement pinned to the reference by golden
    fixtures) running the reference's as-executed step order on the host cores."""
    from oracle import eben_oracle as O
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales
    from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator

    # torch's intra-op pool.  Measured on the GPU host (2 x EPYC 9575F, 256 hardware threads) at the
    # full batch 32: 8 threads 9.9, 16 threads 14.8, 32 threads 10.6 audio-s/s, 256 threads > 20x slower
    # -- these convolutions do not scale past ~16 threads, so 16 is the fairest setting.
    avail = len(os.sched_getaffinity(0))
    cores = int(os.environ.get("EBEN_CPU_THREADS", "0")) or max(1, min(16, avail))
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    gen, disc = EBENGenerator(m=4, n=32, p=2), DiscriminatorEBENMultiScales(q=4, min_channels=24)
    trainer = O.OracleTrainer({k: v.detach() for k, v in gen.state_dict().items()}, {k: v.detach() for k, v in disc.state_dict().items()})
    data = synthetic_batch(batch, length, 1234, "cpu")
    trainer.step(data["audio_body_conducted"], data["audio_airborne"])  # warm-up (first call is ~17x slower)
    t0 = time.perf_counter()
    for _ in range(steps):
        trainer.step(data["audio_body_conducted"], data["audio_airborne"])
    dt = (time.perf_counter() - t0) / steps
    cut = length - (length + 32) % 256
    return {"value": round(batch * cut / 16000 / dt, 3), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} timed step(s) after 1 warm-up, batch {batch} x {length} samples, fp32, reference as-executed order, {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (BASELINE config 2: 32)")
    ap.add_argument("--length", type=int, default=32000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--force-ddp", action="store_true", help="run the bucketed RCCL gradient path even with one rank")
    ap.add_argument("--disc-math", default=os.environ.get("EBEN_DISC_MATH", "bf16"), choices=["bf16", "f32"],
                    help="discriminator contractions: bf16 MFMA operands with fp32 accumulate (BASELINE config 2 names bf16) or exact "
                         "fp32 products; the generator computes in fp32 either way")
    ap.add_argument("--gen-bwd-math", default=None, choices=["bf16", "f32"],
                    help="generator backward contractions (default: same as --disc-math); the generator forward is always fp32")
    ap.add_argument("--stft-math", default=None, choices=["bf16x3", "folded", "dense"],
                    help="MRSTFT windowed-DFT contractions (default: 'folded', exact fp32 on the even / odd parts of the frames; "
                         "'bf16x3' = hi/lo bf16 operand splits on the bf16 MFMA, ~2^-17 relative: measured slower on the tap-conv "
                         "kernel, whose tile staging dominates a pointwise contraction over 1800-3600 channels)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra fp32-discriminator timing reported beside a bf16 run")
    args = ap.parse_args()

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    assert torch.cuda.is_available(), "bench.py measures the HIP path: no GPU visible"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_ddp = world > 1 or args.force_ddp
    if use_ddp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from vibravox_amd import ops
    from vibravox_amd.ddp import BucketedZeroGrad, GradSync

    mod = build_module(device, 1234 + rank)
    mod.disc_math = args.disc_math
    mod.gen_backward_math = args.gen_bwd_math or args.disc_math
    mod.stft_math = args.stft_math or "folded"
    gen_bwd_math, stft_math = mod.gen_backward_math, mod.stft_math   # of the measured steps (the fp32 leg below changes the module's)
    if use_ddp:
        g_opt, d_opt = mod.optimizers()
        gs, ds = GradSync(mod.generator.parameters()), GradSync(mod.discriminator.parameters())
        g_w, d_w = BucketedZeroGrad(g_opt, gs), BucketedZeroGrad(d_opt, ds)
        mod._optimizers = [g_w, d_w]
        mod.grad_sync = {id(g_w): gs, id(d_w): ds}
    batch = synthetic_batch(args.batch, args.length, 1234 + rank, device)
    cut = args.length - (args.length + 32) % 256

    # dominant kernel: MelGAN layer 4 (1024->1024, k41, s4, g4) forward, 52 % of conv MACs sit in L3-L5
    layer = mod.discriminator.melgan_discriminator.discriminator[4][0]
    timer = ops.KernelTimer(layer.spec)
    ops.set_kernel_timer(timer)

    def barrier():
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        tw = time.perf_counter()
        mod.training_step(batch)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench] warm-up step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    # Python's cyclic GC walks every tracked object of the process on a full collection (~100 ms here, a few times per
    # 40 steps: +2..5 ms/step and most of the run-to-run spread): the objects alive after warm-up are moved to the
    # permanent generation, as a long-running training process would do once after set-up (run.py does the same)
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mod.training_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # beside a bf16 run: the same K steps with exact-fp32 discriminator products (the parity mode of tests/), reported
    # as `f32_discriminator` -- never as `value`
    dt32 = None
    if args.disc_math == "bf16" and not args.no_f32_leg:
        mod.disc_math = mod.gen_backward_math = "f32"
        mod.stft_math = "folded"
        mod.training_step(batch)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            mod.training_step(batch)
        barrier()
        dt32 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt32], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt32 = float(t.item())

    # the roofline kernel once more, ALONE on the device (inside the step it shares the GPU with the three other
    # discriminator chains and the MRSTFT GEMMs, which stretches its launch-to-finish time): reported as `isolated`
    iso_ms = None
    if rank == 0:
        import ctypes
        from vibravox_amd._lib import check, load, ptr, stream
        lib = load()
        math = ops.MATH_BF16 if args.disc_math == "bf16" else ops.MATH_F32
        lb = timer.batch or 2 * args.batch
        lx = cut
        for (k, s_, p_) in [(15, 1, 7), (41, 4, 20), (41, 4, 20), (41, 4, 20)]:
            lx = (lx + 2 * p_ - (k - 1) - 1) // s_ + 1
        d = ops.conv_desc(layer.spec, lb, lx, math)
        prm = layer.parametrizations["weight"]
        pw = ops.pack_weights(layer.spec, d, prm.original1.detach(), prm.original0.detach(), None, False)
        xin = torch.randn(lb, layer.spec.c_in, lx, device=device)
        yout = torch.empty(lb, layer.spec.c_out, d.l_out, device=device)
        torch.cuda.synchronize()
        for _ in range(3):
            check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xin), ptr(pw.wp_fwd), ptr(layer.bias), None, ptr(yout), stream()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xin), ptr(pw.wp_fwd), ptr(layer.bias), None, ptr(yout), stream()))
        e1.record()
        torch.cuda.synchronize()
        iso_ms = e0.elapsed_time(e1) / 20

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * cut / 16000 / (dt / args.steps)
        sp = layer.spec
        lx = cut
        for (_, _, k, s, p, _) in [(1, 16, 15, 1, 7, 1), (16, 64, 41, 4, 20, 4), (64, 256, 41, 4, 20, 4), (256, 1024, 41, 4, 20, 4)]:
            lx = (lx + 2 * p - (k - 1) - 1) // s + 1
        l_out = sp.out_len(lx)
        launch_batch = timer.batch or args.batch   # the discriminator engine runs enhanced + reference as one batch-2B launch
        flops = 2.0 * launch_batch * sp.c_out * (sp.c_in // sp.groups) * sp.ksize * l_out
        kms = timer.mean_ms()
        achieved = flops / (kms * 1e-3) / 1e12 if kms else None
        peak = MFMA_BF16_PEAK_TFLOPS if args.disc_math == "bf16" else MFMA_F32_PEAK_TFLOPS
        kname = "tap3_kernel<4,*> (v_mfma_f32_32x32x16_bf16)" if args.disc_math == "bf16" else "tap2_kernel<4,4,16> (v_mfma_f32_32x32x2_f32)"
        line = {
            "metric": "EBEN train-step audio-seconds/sec (gen+disc)", "value": round(value, 2), "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.disc_math, "data": "synthetic",
            "config": {"workload": f"EBEN full GAN train step (gen+disc+MRSTFT+feature-matching+hinge, EMA balancing, Adam), "
                                   f"batch {args.batch} x {args.length} samples @16kHz per GPU (cut to {cut})",
                       "global_batch": world * args.batch, "samples_per_clip": cut, "parallelism": f"dp{world}",
                       "weights": "random init, torch.manual_seed(42)",
                       "precision": (f"discriminator contractions (91 % of the step's FLOPs): bf16 MFMA operands, fp32 accumulate; generator "
                                     f"forward, losses, Adam, storage: fp32; generator backward contractions: {gen_bwd_math}; MRSTFT DFT contractions: "
                                     + ("bf16x3 (hi/lo bf16 operand splits, fp32 accumulate, ~2^-17 relative)" if stft_math == "bf16x3"
                                        else f"fp32 ({stft_math})")
                                     if args.disc_math == "bf16" else "fp32 throughout (exact fp32 MFMA products)")},
            "roofline": {"bound": "mfma", "kernel": f"eben::{kname} MelGAN L4 fwd (1024->1024 k41 s4 g4), {launch_batch} items per launch",
                         "achieved": round(achieved, 2) if achieved else None, "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4) if achieved else None, "traffic": pmc_traffic(launch_batch, args.disc_math),
                         "launch_ms": round(kms, 4) if kms else None, "flops_per_launch": flops,
                         "launches_timed": len(timer.events),
                         "isolated": {"launch_ms": round(iso_ms, 4), "achieved": round(flops / (iso_ms * 1e-3) / 1e12, 2),
                                      "frac": round(flops / (iso_ms * 1e-3) / 1e12 / peak, 4),
                                      "note": "same launch alone on the device, 20 back-to-back launches after the timed region"}},
        }
        # whole-step arithmetic rate on SURVEY section 8d's minimal algorithmic count F_min = 2*(3G + 8D) (no credit for redundant passes)
        f_min = 2.0 * (3 * 6.727e8 + 8 * 2.5255e9) * (world * args.batch * cut / 16000.0)
        line["step_work"] = {"f_min_flop": f_min, "achieved_tflops": round(f_min / (dt / args.steps) / 1e12, 1),
                             "note": "F_min = 2*(3G+8D), G = 6.727e8 and D = 2.5255e9 MACs per audio-second; 91 % of it (the discriminator passes) "
                                     "runs on bf16 MFMA operands in this mode" if args.disc_math == "bf16" else "F_min = 2*(3G+8D), all fp32"}
        if dt32 is not None:
            line["f32_discriminator"] = {"ms_per_step": round(dt32 / args.steps * 1e3, 3), "steps": args.steps,
                                         "value": round(world * args.batch * cut / 16000 / (dt32 / args.steps), 2), "unit": "audio-seconds/sec"}
        print(f"[bench] GPU: {ms:.1f} ms/step, {value:.1f} audio-s/s", file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_batch, args.length, args.cpu_steps)
    # the JSON line is the LAST thing on stdout: the process group is torn down and every C stdio buffer (RCCL's log stream) is
    # flushed first
    if use_ddp:
        barrier()
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
