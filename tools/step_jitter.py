"""Per-step wall time over a run of back-to-back steps (one device sync per step, so a host stall shows up in the step
it happens in): does the Python cyclic GC account for the 100 ms stalls?"""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
mod.disc_math = mod.gen_backward_math = "bf16"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
for _ in range(3):
    mod.training_step(batch)
torch.cuda.synchronize()
def run(label, n=40):
    ts = []
    t_all = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter(); mod.training_step(batch); ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    tot = 1e3 * (time.perf_counter() - t_all) / n
    s = sorted(ts)
    print(f"{label}: {tot:.1f} ms/step over {n} steps; CPU time per step: median {s[n//2]:.1f}, max {s[-1]:.1f}, steps > 2x median: {sum(t > 2 * s[n//2] for t in ts)}; gc counts {gc.get_count()}")
run("gc enabled")
gc.collect(); gc.freeze()
run("gc.freeze()")
gc.disable()
run("gc.disable()")
