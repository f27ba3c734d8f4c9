import sys, time, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
import os
mod.disc_math = os.environ.get("EBEN_DISC_MATH", "bf16_bl"); mod.gen_backward_math = "bf16"; mod.stft_math = "folded_x3"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
for _ in range(3):
    mod.training_step(batch)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    mod.training_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"CPU enqueue {1e3*(t1-t0)/N:.1f} ms/step, GPU complete {1e3*(t2-t0)/N:.1f} ms/step")
# single step from idle
torch.cuda.synchronize(); t0 = time.perf_counter(); mod.training_step(batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"single step from idle: CPU {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
import gc
for label, fn in (("gc enabled", lambda: None), ("gc.freeze()", lambda: (gc.collect(), gc.freeze())), ("gc.disable()", gc.disable)):
    fn()
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); mod.training_step(batch); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    print(f"{label}: single steps from idle: min {ts[0]:.1f} median {ts[15]:.1f} max {ts[-1]:.1f} ms")
