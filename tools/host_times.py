"""Where the HOST spends the engine train step: time.perf_counter() between the phases (mean over steps, the GPU running behind)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
mod.disc_math = os.environ.get("EBEN_DISC_MATH", "bf16_bl"); mod.gen_backward_math = "bf16"; mod.stft_math = "folded_x3"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
for _ in range(5):
    mod.training_step(batch)
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
N = 20
acc, order = {}, []
t_all = 0.0
for _ in range(N):
    mod.phase_host = []
    t0 = time.perf_counter()
    mod.training_step(batch)
    t1 = time.perf_counter()
    t_all += t1 - t0
    ev = [("enter", t0)] + mod.phase_host + [("return (discriminator prepack, bookkeeping)", t1)]
    for (l0, a), (l1, b) in zip(ev[:-1], ev[1:]):
        if l1 not in acc:
            order.append(l1)
        acc[l1] = acc.get(l1, 0.0) + (b - a)
torch.cuda.synchronize()
for k in order:
    print(f"{1e3 * acc[k] / N:7.2f} ms  {k}")
print(f"{1e3 * t_all / N:7.2f} ms  host time per step ({mod.disc_math})")
