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


def profile_average(fragment):
    """Average duration of the kernel whose name contains `fragment` in the latest committed rocprofv3 --kernel-trace --stats summary."""
    import csv
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{n:02d}_rocprofv3_kernel_stats.csv") for n in range(9, 0, -1)) if os.path.exists(q)), None)
    if path is None:
        return None
    try:
        for r in csv.DictReader(open(path)):
            if fragment in r["Name"]:
                return {"file": os.path.relpath(path, ROOT), "calls": int(r["Calls"]), "avg_ms": round(float(r["AverageNs"]) / 1e6, 4),
                        "min_ms": round(float(r["MinNs"]) / 1e6, 4)}
    except Exception:
        return None
    return None


def named_launch(name):
    """In-step duration of ONE named launch from the latest committed kernel trace with dispatch order (tools/step_trace.py under
    rocprofv3 --kernel-trace, tools/step_trace_report.py --named -> profiles/rNN_instep_layers.json): that launch, not the average over every
    launch of its template instantiation."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{n:02d}_instep_layers.json") for n in range(9, 5, -1)) if os.path.exists(q)), None)
    if path is None:
        return None
    try:
        with open(path) as f:
            rec = json.load(f)
        r = rec.get(name)
        if not r:
            return None
        return {"file": os.path.relpath(path, ROOT), "launch": name, "launches": r["launches"], "avg_ms": round(r["avg_us"] / 1e3, 4),
                "min_ms": round(r["min_us"] / 1e3, 4), "note": rec.get("_note")}
    except Exception:
        return None


def pmc_family(math):
    """Roofline records with HBM traffic from this round's rocprofv3 PMC passes (tools/pmc_family_bl.sh -> profiles/rNN_pmc_family.json, the latest round's:
    FETCH_SIZE / WRITE_SIZE / TCC hit + miss in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): the
    roofline kernel (MelGAN L4 forward), its weight gradient and one PQMF-band mid layer, each launched alone at the step's row
    counts.  Empty for the plans the passes were not taken in."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r{n:02d}_pmc_family.json") for n in range(9, 2, -1)) if os.path.exists(q)), None)
    if math != "bf16_bl" or path is None:
        return []
    with open(path) as f:
        return json.load(f)


def noisy_bwe_source(device, n_items=64, seed=4321):
    """BASELINE config 4: a resident pool of synthetic (speech, airborne, speechless-noise) clips of ragged lengths (1.9-3.4 s at
    16 kHz, the noise at least as long as its speech), from which every step assembles its batch ON THE DEVICE: the reference's
    collator arithmetic (noisybwe.py:219-291: random noise slice, add, random crop / pad to 2.5 s) as one gather kernel, then the
    waveform augmentation (data_augmentation.py:38-71; time masking only: the other two transforms change the clip length)."""
    g = torch.Generator().manual_seed(seed)
    items = []
    for _ in range(n_items):
        n = int(torch.randint(30400, 54400, (1,), generator=g))
        items.append({"audio_body_conducted": (0.1 * torch.randn(n, generator=g)).to(device), "audio_airborne": (0.1 * torch.randn(n, generator=g)).to(device),
                      "audio_body_conducted_speechless_noisy": (0.05 * torch.randn(n + int(torch.randint(1, 16000, (1,), generator=g)), generator=g)).to(device)})
    return items


def synthetic_batch(batch, length, seed, device):
    g = torch.Generator().manual_seed(seed)
    return {"audio_body_conducted": (0.1 * torch.randn(batch, 1, length, generator=g)).to(device),
            "audio_airborne": (0.1 * torch.randn(batch, 1, length, generator=g)).to(device)}


def cpu_baseline(batch, length, steps, threads):
    """The CPU oracle (oracle/eben_oracle.py, a restatement pinned to the reference by golden fixtures) running the
    reference's as-executed step order on the host cores: `steps` timed steps after one warm-up."""
    from oracle import eben_oracle as O
    from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales
    from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator

    torch.set_num_threads(threads)
    torch.manual_seed(42)
    gen, disc = EBENGenerator(m=4, n=32, p=2), DiscriminatorEBENMultiScales(q=4, min_channels=24)
    trainer = O.OracleTrainer({k: v.detach() for k, v in gen.state_dict().items()}, {k: v.detach() for k, v in disc.state_dict().items()})
    data = synthetic_batch(batch, length, 1234, "cpu")
    trainer.step(data["audio_body_conducted"], data["audio_airborne"])  # warm-up (first call is ~17x slower)
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        trainer.step(data["audio_body_conducted"], data["audio_airborne"])
        times.append(time.perf_counter() - t0)
    cut = length - (length + 32) % 256
    dt = sum(times) / len(times)
    return batch * cut / 16000 / dt, dt, times


def percentile(sorted_vals, q):
    if not sorted_vals:
        return None
    k = (len(sorted_vals) - 1) * q
    lo, hi = int(k), min(int(k) + 1, len(sorted_vals) - 1)
    return sorted_vals[lo] + (sorted_vals[hi] - sorted_vals[lo]) * (k - lo)


# MACs of ONE forward pass at batch 32 x 31968 samples (both scale with batch x length)
G_MACS, D_MACS = 4.3008e10, 1.6147e11
D_PQMF_MACS = 1.398e10 + 1.361e10 + 1.324e10   # the three PQMF-band discriminators (section 8a, row a10)
G_BYTES = 2.318e9                              # generator activations in + out, one pass, fp32
HBM_PEAK = 8.0e12


def step_roofline_ms(disc_math, scale):
    """Mixed roofline of the minimal step F_min = 2 (3 G + 8 D) with NO credit for work this build adds on top of it (the
    hi + lo products of the PQMF-band forwards, the three / six piece products of the generator forward): the eight discriminator
    passes at max(FLOP / MFMA peak of `dtype`, bytes / HBM), the three generator passes likewise (HBM-bound: G_BYTES per pass; half
    of it for the two backward passes on bf16 operands)."""
    f32, bf16 = MFMA_F32_PEAK_TFLOPS * 1e12, MFMA_BF16_PEAK_TFLOPS * 1e12
    d_bytes = 1.627e9   # discriminator activations in + out of one pass, fp32
    if disc_math == "f32":
        t = 8 * max(2 * D_MACS / f32, d_bytes / HBM_PEAK) + 3 * max(2 * G_MACS / f32, G_BYTES / HBM_PEAK)
    else:
        t = 8 * max(2 * D_MACS / bf16, 0.5 * d_bytes / HBM_PEAK)
        t += max(2 * G_MACS / bf16, G_BYTES / HBM_PEAK) + 2 * max(2 * G_MACS / bf16, 0.5 * G_BYTES / HBM_PEAK)
    return t * scale * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU (BASELINE config 2: 32)")
    ap.add_argument("--length", type=int, default=32000)
    ap.add_argument("--workload", default="bwe", choices=["bwe", "noisybwe"],
                    help="bwe: BASELINE config 2 (resident synthetic batch); noisybwe: config 4 (every step assembles its batch on the "
                         "device from a resident pool of ragged clips: noise mixing + crop/pad to 2.5 s + time masking, then the same step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--force-ddp", action="store_true", help="run the bucketed RCCL gradient path even with one rank")
    ap.add_argument("--disc-math", default=os.environ.get("EBEN_DISC_MATH", "bf16_bl"), choices=["bf16_bl", "bf16", "f32", "bf16_plain"],
                    help="discriminator contractions: 'bf16_bl' (default) = bf16 MFMA operands with fp32 accumulate, embeddings and stacked "
                         "gradients at rest as bf16 bundles (hi + lo planes, disc_engine_bl.py; BASELINE config 2 names bf16; the PQMF-band "
                         "discriminators' forward takes hi + lo operands, see DESIGN.md), 'bf16' = the same arithmetic on fp32 tensors at rest, "
                         "'f32' = exact fp32 products, 'bf16_plain' = every contraction on single bf16 operands; the generator's forward takes hi + lo "
                         "bf16 operands (three products, ~2^-17) inside a step whose generator backward is bf16, else fp32-grade six-product arithmetic")
    ap.add_argument("--gen-bwd-math", default=None, choices=["bf16", "f32"],
                    help="generator backward contractions (default: bf16 with a bf16 discriminator, else f32)")
    ap.add_argument("--stft-math", default=None, choices=["bf16x3", "folded", "dense", "folded_x3", "folded_x6"],
                    help="MRSTFT windowed-DFT contractions (default: 'folded', exact fp32 on the even / odd parts of the frames)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra fp32 timing reported beside a bf16 run")
    args = ap.parse_args()

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher of its N ranks (one process per GPU, RCCL;
        # counterpart of vibravox configs/trainer/ddp.yaml:5-7 `devices` / `strategy`: one command, N ranks).  Rank 0's JSON line is
        # the only thing the ranks write to stdout.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    # stdout carries ONE line, rank 0's JSON.  RCCL writes its banner and its warnings ("NCCL WARN Missing iommu=pt ...") to file
    # descriptor 1 from C: from here on fd 1 IS stderr for everything in this process, and the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from vibravox_amd._env import configure_hw_queues
    hw_queues = configure_hw_queues(world > 1 or args.force_ddp)   # before the first HIP call (vibravox_amd/_env.py)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: no GPU visible"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_ddp = world > 1 or args.force_ddp
    if use_ddp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # ([MI355X] the process group's stream at the high HIP priority -- ProcessGroupNCCL.Options(is_high_priority_stream=True) -- so that a
        # bucket's exchange does not queue behind the compute streams' resident blocks: 10.5 -> 15.2 ms/step at one rank, the seventh stream
        # collides on the hardware queues again; default priority it is)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # The CPU-baseline leg runs FIRST (rank 0, one rank only): everything after it is GPU work, so a sampler watching the device over the
    # run (the driver's rocm-smi trace) sees the timed region and not a host-only tail.
    cpu_rec = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        # torch's intra-op pool.  Measured on the GPU host (2 x EPYC 9575F, 256 hardware threads) at the full batch 32: 8 threads 9.9,
        # 16 threads 14.8, 32 threads 10.6 audio-s/s, 256 threads > 20x slower -- these convolutions do not scale past ~16 threads, so
        # 16 is the setting that is fair to the CPU; the 8-thread leg is comparable with the survey's measurement of the reference.
        avail = len(os.sched_getaffinity(0))
        cores = int(os.environ.get("EBEN_CPU_THREADS", "0")) or max(1, min(16, avail))
        threads_before = torch.get_num_threads()
        v16, dt16, times = cpu_baseline(args.cpu_batch, args.length, args.cpu_steps, cores)
        v8, dt8, _ = cpu_baseline(args.cpu_batch, args.length, 1, min(8, avail))
        torch.set_num_threads(threads_before)
        cpu_rec = {"value": round(v16, 3), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
                   "sample": f"{args.cpu_steps} timed steps after 1 warm-up ({', '.join(f'{t:.2f}' for t in times)} s), batch {args.cpu_batch} x "
                             f"{args.length} samples, fp32, reference as-executed order; host has {avail} hardware threads "
                             f"(more than ~16 threads is slower for these convolutions); run before the GPU legs",
                   "threads_8": {"value": round(v8, 3), "s_per_step": round(dt8, 2), "steps": 1}}
        print(f"[bench] CPU oracle: {v16:.2f} audio-s/s on {cores} threads", file=sys.stderr, flush=True)

    from vibravox_amd import ops
    from vibravox_amd.ddp import BucketedZeroGrad, GradSync

    mod = build_module(device, 1234 + rank)
    bf16 = args.disc_math != "f32"
    mod.disc_math = args.disc_math
    mod.gen_backward_math = args.gen_bwd_math or ("bf16" if bf16 else "f32")
    mod.stft_math = args.stft_math or ("folded_x3" if bf16 else "folded")
    gen_bwd_math, stft_math = mod.gen_backward_math, mod.stft_math   # of the measured steps (the fp32 leg below changes the module's)
    # the generator forward's arithmetic INSIDE the measured steps (lightning_modules/eben.py: forward_math scope of the engine step)
    gen_fwd_x3 = gen_bwd_math == "bf16" and bool(mod.ru_forward_x3)
    gen_fwd_desc = ("hi + lo bf16 operands, three MFMAs per product (EBEN_MATH_BF16X3, ~2^-17 per product; EBEN_RU_FWD_X3=0 selects the "
                    "fp32-grade six-product form), fp32 accumulate, fp32 activations at rest" if gen_fwd_x3 else
                    "fp32-grade (three bf16 pieces per operand, six MFMAs per product, EBEN_MATH_BF16X6), fp32 activations at rest")
    syncs = []
    if use_ddp:
        g_opt, d_opt = mod.optimizers()
        # generator: 7.8 MB of gradients in ~2 MB buckets, released per backward segment group (gen_engine.backward_train) so that only the last
        # group's exchange sits in front of its Adam; discriminator: 92.6 MB in 32 MB buckets, released per chain (largest layers first)
        gs, ds = GradSync(mod.generator.parameters(), bucket_bytes=2 << 20), GradSync(mod.discriminator.parameters())
        gs.profile = ds.profile = True
        g_w, d_w = BucketedZeroGrad(g_opt, gs), BucketedZeroGrad(d_opt, ds)
        mod._optimizers = [g_w, d_w]
        mod.grad_sync = {id(g_w): gs, id(d_w): ds}
        syncs = [("generator", gs), ("discriminator", ds)]

    if args.workload == "noisybwe":
        from vibravox_amd.augment import WaveformDataAugmentation
        from vibravox_amd.collate import noisy_bwe_collate

        pool = noisy_bwe_source(device, seed=4321 + rank)
        augment = WaveformDataAugmentation(16000, p_data_augmentation=1.0, p_speed_perturbation=0.0, p_pitch_shift=0.0, p_time_masking=1.0)
        length = 40000   # collate_strategy constant_length-2500-ms (noisybwe.yaml:7)
        counter = [0]

        def next_batch():
            i = counter[0]
            counter[0] += 1
            items = [pool[(i * args.batch + k) % len(pool)] for k in range(args.batch)]
            batch = noisy_bwe_collate(items, 16000, "constant_length-2500-ms")
            bc, air = augment(batch["audio_body_conducted"], batch["audio_airborne"])
            return {"audio_body_conducted": bc, "audio_airborne": air}
    else:
        length = args.length
        resident = synthetic_batch(args.batch, args.length, 1234 + rank, device)

        def next_batch():
            return resident
    cut = length - (length + 32) % 256
    scale = world * args.batch * cut / (32 * 31968.0)   # work relative to SURVEY's batch 32 x 31968 figures

    # roofline kernels: MelGAN layer 4 forward (1024->1024, k41, s4, g4: the launch with the most FLOPs; L3-L5 hold 52 % of the conv
    # MACs) and MelGAN layer 3's input gradient over the four stacked right-hand sides (the longest single launch of the step)
    mel = mod.discriminator.melgan_discriminator.discriminator
    layer, layer_t = mel[4][0], mel[3][0]
    timer, timer_t = ops.KernelTimer(layer.spec, "fwd"), ops.KernelTimer(layer_t.spec, "dx")
    ops.set_kernel_timers([timer, timer_t])

    def barrier():
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    host_s = [0.0]

    def timed_steps(n):
        """EXACTLY n steps between two barrier + synchronize brackets (wall clock), with a HIP event at every step boundary on the main
        stream for the per-step distribution."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        barrier()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(n):
            mod.training_step(next_batch())
            evs[i + 1].record()
        host_s[0] = (time.perf_counter() - t0) / n   # what the host needed to enqueue a step (the GPU runs behind it)
        barrier()
        dt = time.perf_counter() - t0
        per = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, per

    for i in range(args.warmup):
        tw = time.perf_counter()
        mod.training_step(next_batch())
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[bench] warm-up step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    # Launch sequences on their way into a HIP graph (ops.ReplayedChain / ReplayedPrepack: the capture of a sequence takes tens of ms and
    # the chained ones settle one after the other) finish settling before ANY leg's clock starts; reported as `settle_steps`
    def settle_graphs(done_before=0):
        n = 0
        # multi-rank: the captures are voted step by step (ops.CaptureGate: ~3 steps per level of chained sequences) and the vote must
        # have SETTLED on every rank (same step everywhere) before the clock starts -- no blocking collective inside the timed region
        gated = ops.capture_gate.active()
        while n < (60 if gated else 12) and ((gated and not ops.capture_gate.settled) or ops.graphs_pending() or done_before + n < 2):   # (no sequence has a signature before its first step)
            mod.training_step(next_batch())
            n += 1
        torch.cuda.synchronize()
        return n

    settle = settle_graphs(args.warmup)
    # Python's cyclic GC walks every tracked object of the process on a full collection (~100 ms here, a few times per
    # 40 steps: +2..5 ms/step and most of the run-to-run spread): the objects alive after warm-up are moved to the
    # permanent generation, as a long-running training process would do once after set-up (run.py does the same)
    import gc
    gc.collect()
    gc.freeze()
    for _, sy in syncs:
        sy.exposed_comm_ms()   # drop the warm-up's samples
    dt, per_step = timed_steps(args.steps)
    host_ms = host_s[0] * 1e3
    n_graphs = ops.graphs_captured()   # launch sequences replayed as HIP graphs during the timed steps
    exposed = {name: sy.exposed_comm_ms() for name, sy in syncs}
    # The two roofline launches are bracketed by HIP events on the stream they are launched on.  Events cannot sit between the nodes of
    # a replayed HIP graph, and the timed steps above replay the discriminator chains as graphs (ops.ReplayedChain): the brackets are
    # taken over a second run of the same steps with the chains launched kernel by kernel (same kernels, same streams, same batch)
    timer.enabled = timer_t.enabled = True
    timed_steps(max(5, args.steps // 4))
    timer.enabled = timer_t.enabled = False

    # beside a bf16 run: the same K steps in exact fp32 (the arithmetic of the reference), reported as value_f32 -- never as `value`
    dt32 = per32 = None
    if bf16 and not args.no_f32_leg:
        mod.disc_math = mod.gen_backward_math = "f32"
        mod.stft_math = "folded"
        settle32 = settle_graphs()   # the plan's own launch sequences are captured before its clock starts (as for the bf16 leg)
        dt32, per32 = timed_steps(args.steps)

    # ... and in the fp32-grade plan on the bf16 matrix pipe ("bf16x6": discriminator forward / input gradients and the MRSTFT
    # contractions with three bf16 pieces per operand, weight gradients and the generator backward in fp32 arithmetic): the plan whose
    # tests run at the fp32 tolerances, reported as value_fp32_grade
    dt6 = per6 = None
    if bf16 and not args.no_f32_leg:
        mod.disc_math, mod.gen_backward_math, mod.stft_math = "bf16x6", "f32", "folded_x6"
        settle6 = settle_graphs()
        dt6, per6 = timed_steps(args.steps)

    # the roofline kernel once more, ALONE on the device (inside the step it shares the GPU with the three other
    # discriminator chains and the MRSTFT GEMMs, which stretches its launch-to-finish time): reported as `isolated`
    iso_ms = None
    if rank == 0:
        import ctypes
        from vibravox_amd._lib import check, load, ptr, stream
        from vibravox_amd.lightning_modules.eben import DISC_MATH_PLANS
        lib = load()
        plan = DISC_MATH_PLANS[args.disc_math]
        bundle = isinstance(plan, dict) and plan.get("layout") == "bl"
        plan = plan["melgan"] if isinstance(plan, dict) else plan
        math = plan if isinstance(plan, int) else plan[0]
        lb = timer.batch or 2 * args.batch
        lx = cut
        for (k, s_, p_) in [(15, 1, 7), (41, 4, 20), (41, 4, 20), (41, 4, 20)]:
            lx = (lx + 2 * p_ - (k - 1) - 1) // s_ + 1
        d = ops.conv_desc(layer.spec, lb, lx, math | (0x100 if bundle else 0))
        prm = layer.parametrizations["weight"]
        pw = ops.pack_weights(layer.spec, d, prm.original1.detach(), prm.original0.detach(), None, False)
        xin = torch.randn(lb, layer.spec.c_in, lx, device=device)
        if bundle:
            from vibravox_amd.disc_engine_bl import Planes
            xpl, ypl = Planes.from_f32(xin, lo=False), Planes(lb, layer.spec.c_out, d.l_out, device)

            def launch():
                check(lib.eben_bl_conv1d_fwd(ctypes.byref(d), xpl.hi.data_ptr(), None, ptr(pw.wp_fwd), ptr(layer.bias), ypl.hi.data_ptr(), ypl.lo.data_ptr(), stream()))
        else:
            yout = torch.empty(lb, layer.spec.c_out, d.l_out, device=device)

            def launch():
                check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xin), ptr(pw.wp_fwd), ptr(layer.bias), None, ptr(yout), stream()))
        torch.cuda.synchronize()
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            launch()
        e1.record()
        torch.cuda.synchronize()
        iso_ms = e0.elapsed_time(e1) / 20

    if rank == 0:
        ms = dt / args.steps * 1e3
        audio_s = world * args.batch * cut / 16000
        value = audio_s / (dt / args.steps)
        peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TFLOPS

        def launch_record(tm, which, kname):
            sp = tm.spec
            lx = cut
            chain = [(15, 1, 7), (41, 4, 20), (41, 4, 20), (41, 4, 20)]
            depth = 4 if tm is timer else 3
            for (k, s_, p_) in chain[:depth]:
                lx = (lx + 2 * p_ - (k - 1) - 1) // s_ + 1
            l_out = sp.out_len(lx)
            items = tm.batch or args.batch
            flops = 2.0 * items * sp.c_out * (sp.c_in // sp.groups) * sp.ksize * l_out
            kms = tm.mean_ms()
            ach = flops / (kms * 1e-3) / 1e12 if kms else None
            return {"bound": "mfma", "kernel": f"eben::{kname} MelGAN L{depth} {which} ({sp.c_in}->{sp.c_out} k{sp.ksize} s{sp.stride} g{sp.groups}), {items} items per launch",
                    "achieved": round(ach, 2) if ach else None, "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4) if ach else None,
                    "launch_ms": round(kms, 4) if kms else None, "flops_per_launch": flops, "launches_timed": len(tm.events)}, flops

        stft_desc = {"folded_x6": "fp32-grade (three bf16 pieces per operand)",
                     "folded_x3": "hi + lo bf16 operands (three MFMAs per product, ~2^-17; the generator gradient against the fp32 step stays at 5.5e-3)"
                     }.get(stft_math, "fp32 (" + stft_math + ")")
        kn = ("tap4_kernel<2,2,4,2,1,4,3,2> (bigtap.hip: persistent producer / consumer block per CU, v_mfma_f32_32x32x16_bf16, bf16 bundles at rest)" if args.disc_math == "bf16_bl" else
              "tap3_kernel<4,*> (v_mfma_f32_32x32x16_bf16)" if bf16 else "tap2_kernel<4,4,16> (v_mfma_f32_32x32x2_f32)")
        roof, flops = launch_record(timer, "fwd", kn)
        family = pmc_family(args.disc_math) if (timer.batch or 2 * args.batch) == 64 else []
        roof["traffic"] = next((r["traffic_bytes"] for r in family if r["name"] == "melgan_l4_fwd"), None)
        # the counters are not collected inside this run (rocprofv3 --pmc passes are separate processes): the figure is the committed profile's
        roof["traffic_source"] = next((r.get("source") for r in family if r["name"] == "melgan_l4_fwd"), None)
        roof["launch_ms_note"] = ("HIP events around the launch on its stream, over a second run of the timed steps with the chains launched kernel by "
                                  "kernel (events cannot sit between the nodes of a replayed graph); `isolated` = the same launch alone; "
                                  "`profile` = THAT launch's duration inside the graph-replayed step, from the committed rocprofv3 --kernel-trace with "
                                  "dispatch order (named by its position on MelGAN's queue; the tracer stretches the step)")
        prof = (named_launch("melgan_l4_fwd") if args.disc_math == "bf16_bl" else None) or profile_average("tap4_kernel<2, 2, 4, 2, 1" if args.disc_math == "bf16_bl" else "tap3_kernel<4")
        if prof:
            roof["profile"] = prof
            if prof.get("launch") and prof.get("avg_ms"):
                roof["profile"]["achieved"] = round(flops / (prof["avg_ms"] * 1e-3) / 1e12, 2)
                roof["profile"]["frac"] = round(flops / (prof["avg_ms"] * 1e-3) / 1e12 / peak, 4)
        if iso_ms:
            roof["isolated"] = {"launch_ms": round(iso_ms, 4), "achieved": round(flops / (iso_ms * 1e-3) / 1e12, 2),
                                "frac": round(flops / (iso_ms * 1e-3) / 1e12 / peak, 4),
                                "note": "same launch alone on the device, 20 back-to-back launches after the timed region"}
        if roof.get("bound") == "mfma":
            roof["peak_note"] = ("`peak` is the dense bf16 MFMA figure at 2.4 GHz (MI355X_MICROARCH.md); tools/ubench/clock_probe.hip on this pool: the shader clock is "
                                 "2.38 GHz idle and 1.82-1.89 GHz with MFMA streams on every SIMD, 1.83-1.94 PFLOP/s of pure MFMA issue")
        roof_t, flops_t = launch_record(timer_t, "input gradient (two 2B-row passes per step: rows [fm | adv], then rows [fake | real])", kn)
        prof_t = named_launch("melgan_l3_dx") if args.disc_math == "bf16_bl" else None
        if prof_t and prof_t.get("avg_ms"):
            prof_t["achieved"] = round(flops_t / (prof_t["avg_ms"] * 1e-3) / 1e12, 2)
            prof_t["frac"] = round(flops_t / (prof_t["avg_ms"] * 1e-3) / 1e12 / peak, 4)
            roof_t["profile"] = prof_t
        ideal = step_roofline_ms("f32" if not bf16 else "bf16", scale)
        line = {
            "metric": "EBEN train-step audio-seconds/sec (gen+disc)", "value": round(value, 2), "unit": "audio-seconds/sec",
            "n_gpus": world, "gpu_max_hw_queues": hw_queues or "default", "steps": args.steps, "warmup": args.warmup, "settle_steps": settle, "graphs_replayed": n_graphs, "host_enqueue_ms_per_step": round(host_ms, 3), "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": (f"EBEN full GAN train step (gen+disc+MRSTFT+feature-matching+hinge, EMA balancing, Adam), batch {args.batch} x "
                                    f"{length} samples @16kHz per GPU (cut to {cut})"
                                    + ("; BASELINE config 4: each step's batch assembled on the device from a resident pool of ragged clips "
                                       "(noise slice + mix + crop/pad to 2.5 s + time masking)" if args.workload == "noisybwe" else "")),
                       "global_batch": world * args.batch, "samples_per_clip": cut, "parallelism": f"dp{world}",
                       "weights": "random init, torch.manual_seed(42)", "disc_math": args.disc_math,
                       "gen_forward_math": "bf16x3" if gen_fwd_x3 else "bf16x6", "EBEN_RU_FWD_X3": int(bool(mod.ru_forward_x3)),
                       "precision": (f"discriminator contractions on bf16 MFMA operands with fp32 accumulate (MelGAN: every pass; PQMF-band "
                                     f"discriminators: input / weight gradients -- their forward takes hi + lo bf16 operands (3 MFMAs per product), which keeps the discriminator "
                                     f"gradient within 3.4e-2 of the fp32 step's, tests/test_gpu_models.py); generator forward: {gen_fwd_desc}; losses, Adam, "
                                     f"parameters: fp32; discriminator embeddings / stacked gradients at rest: "
                                     f"{'bf16 bundles [row][channels / 8][position][8], hi + lo planes where the fp32 value is needed' if args.disc_math == 'bf16_bl' else 'fp32'}; generator backward contractions: {gen_bwd_math}; MRSTFT DFT contractions: {stft_desc}"
                                     if args.disc_math in ("bf16", "bf16_bl") else
                                     "every contraction on single bf16 MFMA operands (bf16_plain)" if bf16 else "fp32 throughout (exact fp32 MFMA products)")},
            "step_ms": {"median": round(percentile(per_step, 0.5), 3), "p10": round(percentile(per_step, 0.1), 3),
                        "p90": round(percentile(per_step, 0.9), 3), "max": round(per_step[-1], 3),
                        "mean": round(sum(per_step) / len(per_step), 3), "clock": "HIP events at the step boundaries on the main stream"},
            "roofline": roof,
            "roofline_time_dominant": roof_t,
            "roofline_family": family,
            "step_roofline": {"ideal_ms": round(ideal, 3), "frac": round(ideal / ms, 4),
                              "note": "sum over the eleven passes of F_min = 2(3G+8D) of max(FLOP / dense MFMA peak of `dtype`, SURVEY 8d bytes / 8 TB/s) / measured ms per "
                                      "step; no credit for the build's own extra products (G = 6.727e8, D = 2.5255e9 MACs per audio-second)"},
        }
        # whole-step arithmetic rate on SURVEY section 8d's minimal algorithmic count F_min = 2*(3G + 8D) (no credit for redundant passes)
        f_min = 2.0 * (3 * 6.727e8 + 8 * 2.5255e9) * audio_s
        line["step_work"] = {"f_min_flop": f_min, "achieved_tflops": round(f_min / (dt / args.steps) / 1e12, 1)}
        line["step_roofline_f_min"] = {"ideal_ms": round(f_min / (peak * 1e12) * 1e3, 3), "frac": round(f_min / (peak * 1e12) * 1e3 / ms, 4),
                                       "note": "F_min alone on the dense MFMA peak of `dtype` (no credit for the hi + lo forward products of "
                                               "the discriminators / the generator or any byte)"}
        if dt32 is not None:
            ms32 = dt32 / args.steps * 1e3
            line["value_f32"] = round(audio_s / (dt32 / args.steps), 2)
            line["ms_per_step_f32"] = round(ms32, 3)
            line["step_ms_f32"] = {"median": round(percentile(per32, 0.5), 3), "p10": round(percentile(per32, 0.1), 3), "p90": round(percentile(per32, 0.9), 3),
                                   "settle_steps": settle32}
            line["step_roofline_f32"] = {"ideal_ms": round(step_roofline_ms("f32", scale), 3), "frac": round(step_roofline_ms("f32", scale) / ms32, 4)}
        if dt6 is not None:
            ms6 = dt6 / args.steps * 1e3
            line["value_fp32_grade"] = round(audio_s / (dt6 / args.steps), 2)
            line["ms_per_step_fp32_grade"] = round(ms6, 3)
            line["step_ms_fp32_grade"] = {"median": round(percentile(per6, 0.5), 3), "p10": round(percentile(per6, 0.1), 3), "p90": round(percentile(per6, 0.9), 3),
                                          "settle_steps": settle6}
            line["fp32_grade_plan"] = ("disc_math bf16x6 + stft folded_x6 + fp32 generator backward: every contraction either exact fp32 or six bf16 piece "
                                       "products per fp32 product (dropped terms <= 2^-26); tests/test_gpu_models.py runs the golden replay in it")
        if use_ddp:
            line["comm"] = {"backend": dist.get_backend(), "ranks": world,
                            "exposed_ms_per_step": {k: (round(v, 4) if v is not None else None) for k, v in exposed.items()},
                            "grad_bytes": {name: sum(b.numel for b in sy.buckets) * 4 for name, sy in syncs},
                            "buckets": {name: len(sy.buckets) for name, sy in syncs},
                            "note": "exposed = main-stream time spent waiting for the bucket all-reduces in front of each Adam"}
        print(f"[bench] GPU: {ms:.1f} ms/step, {value:.1f} audio-s/s", file=sys.stderr, flush=True)
        if cpu_rec is not None:
            line["cpu_baseline"] = cpu_rec
    # the JSON line is the LAST thing on stdout: the process group is torn down and every C stdio buffer (RCCL's log stream) is
    # flushed first
    if use_ddp:
        barrier()
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
