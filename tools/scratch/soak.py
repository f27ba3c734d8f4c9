"""Scratch: 400 train steps in the benchmarked plan with a new batch tensor every step: finite losses, stable memory, stable step time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
mod.disc_math, mod.gen_backward_math, mod.stft_math = "bf16_bl", "bf16", "folded_x3"
g = torch.Generator().manual_seed(7)
pool = [bench.synthetic_batch(32, 32000, 100 + i, dev) for i in range(4)]
mem, t0 = [], time.perf_counter()
for i in range(400):
    b = pool[i % 4]
    batch = {k: v.clone() for k, v in b.items()}   # a new tensor (new address) every step, as a data loader hands them over
    mod.training_step(batch)
    if i % 50 == 49:
        torch.cuda.synchronize()
        vals = {k: float(v) for k, v in mod.logged.items()}
        assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
        mem.append(torch.cuda.memory_allocated() / 2**20)
        print(i + 1, f"{(time.perf_counter() - t0) / 50 * 1e3:.2f} ms/step", f"{mem[-1]:.0f} MiB allocated, {torch.cuda.memory_reserved() / 2**20:.0f} reserved",
              {k.split('/')[-1]: round(v, 4) for k, v in list(vals.items())[:4]}, flush=True)
        t0 = time.perf_counter()
assert max(mem[2:]) - min(mem[2:]) < 64, mem
print("ok")
