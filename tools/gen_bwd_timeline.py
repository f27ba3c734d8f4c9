"""Kernel timeline of the generator backward alone on the device (input-gradient chain on the main stream, weight gradients on the side
stream): start, duration, stream, kernel -- where the 3 ms of the step's generator backward go."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from vibravox_amd import ops, gen_engine
gen_engine.USE_GRAPHS = False
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
batch = bench.synthetic_batch(32, 32000, 1234, dev)
gen = mod.generator
x = gen.cut_to_valid_length(batch["audio_body_conducted"])
gp = [p for p in gen.parameters() if p.requires_grad]
def once(prof_ctx=None):
    with ops.backward_math(ops.MATH_BF16):
        y, bands = gen(x)
        seed = torch.ones_like(bands)
        torch.cuda.synchronize()
        if prof_ctx is not None:
            prof_ctx.__enter__()
        with ops.weight_grads_on_side_stream() as side:
            torch.autograd.backward(bands, seed, inputs=gp)
            side.join()
        torch.cuda.synchronize()
        if prof_ctx is not None:
            prof_ctx.__exit__(None, None, None)
        for p in gp:
            p.grad = None
for _ in range(3):
    once()
from torch.profiler import profile, ProfilerActivity
prof = profile(activities=[ProfilerActivity.CUDA])
once(prof)
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
for e in evs:
    print(f"{(e.time_range.start - t0):8.1f} us  {e.time_range.end - e.time_range.start:7.1f} us  {e.name[:100]}")
print("span", evs[-1].time_range.end - t0, "kernels", len(evs), "sum", sum(e.time_range.end - e.time_range.start for e in evs))
