"""Where the main stream spends the step: HIP events between the phases of the engine train step (mean over steps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
FORCE_DDP = os.environ.get("FORCE_DDP", "0") == "1"   # the single-rank RCCL path of bench.py --force-ddp: where its extra time goes
if FORCE_DDP:
    from vibravox_amd._env import configure_hw_queues
    configure_hw_queues(True)
dev = torch.device("cuda", 0)
if FORCE_DDP:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mod = bench.build_module(dev, 1234)
if FORCE_DDP:
    from vibravox_amd.ddp import BucketedZeroGrad, GradSync
    g_opt, d_opt = mod.optimizers()
    gs, ds = GradSync(mod.generator.parameters(), bucket_bytes=2 << 20), GradSync(mod.discriminator.parameters())
    g_w, d_w = BucketedZeroGrad(g_opt, gs), BucketedZeroGrad(d_opt, ds)
    mod._optimizers = [g_w, d_w]
    mod.grad_sync = {id(g_w): gs, id(d_w): ds}
mod.disc_math = os.environ.get("EBEN_DISC_MATH", "bf16_bl"); mod.gen_backward_math = os.environ.get("EBEN_GEN_BWD_MATH", "f32" if mod.disc_math == "f32" else "bf16")
mod.stft_math = "folded" if mod.disc_math == "f32" else "folded_x3"
if os.environ.get("NO_RECON", "0") == "1":   # the discriminator phases alone: how long the chains take with nothing beside them
    mod.reconstructive_loss_freq_fn = None
    mod.reconstructive_loss_temp_fn = None
if os.environ.get("NO_D_UPDATE", "0") == "1":   # no discriminator weight gradients / Adam: what the generator backward costs with the GPU to itself
    mod.update_discriminator_ratio = 0
batch = bench.synthetic_batch(32, 32000, 1234, dev)
for _ in range(8):
    mod.training_step(batch)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
N = 10
acc = {}
order = []
SYNC = os.environ.get("SYNC_EACH_STEP", "0") == "1"
runs = []
PRE_SLEEP = int(float(os.environ.get("PRE_SLEEP_CYCLES", "0")))   # a GPU-side delay in front of every step: the host gets that far ahead
for _ in range(N):
    if PRE_SLEEP:
        torch.cuda._sleep(PRE_SLEEP)
    mod.phase_events = []
    mod.training_step(batch)
    if SYNC:
        torch.cuda.synchronize()
    runs.append(mod.phase_events)
torch.cuda.synchronize()
for ev in runs:
    for (l0, e0), (l1, e1) in zip(ev[:-1], ev[1:]):
        if l1 not in acc:
            order.append(l1)
        acc[l1] = acc.get(l1, 0.0) + e0.elapsed_time(e1)
tot = 0.0
for k in order:
    print(f"{acc[k] / N:7.2f} ms  {k}")
    tot += acc[k] / N
print(f"{tot:7.2f} ms  total between first and last marker ({mod.disc_math})")
