"""Scratch: tools/phase_times.py with the single-rank data-parallel wrapping of bench.py --force-ddp."""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vibravox_amd.ddp import BucketedZeroGrad, GradSync
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mod = bench.build_module(dev, 1234)
mod.disc_math, mod.gen_backward_math, mod.stft_math = "bf16_bl", "bf16", "folded_x3"
g_opt, d_opt = mod.optimizers()
gs, ds = GradSync(mod.generator.parameters()), GradSync(mod.discriminator.parameters())
g_w, d_w = BucketedZeroGrad(g_opt, gs), BucketedZeroGrad(d_opt, ds)
mod._optimizers = [g_w, d_w]
mod.grad_sync = {id(g_w): gs, id(d_w): ds}
batch = bench.synthetic_batch(32, 32000, 1234, dev)
for _ in range(8):
    mod.training_step(batch)
torch.cuda.synchronize()
N, acc, order, runs = 10, {}, [], []
for _ in range(N):
    mod.phase_events = []
    mod.training_step(batch)
    runs.append(mod.phase_events)
torch.cuda.synchronize()
for ev in runs:
    for (l0, e0), (l1, e1) in zip(ev[:-1], ev[1:]):
        if l1 not in acc:
            order.append(l1)
        acc[l1] = acc.get(l1, 0.0) + e0.elapsed_time(e1)
for k in order:
    print(f"{acc[k] / N:7.2f} ms  {k}")
print(f"{sum(acc.values()) / N:7.2f} ms  total")
dist.destroy_process_group()
