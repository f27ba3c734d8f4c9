import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
batch = bench.synthetic_batch(32, 32000, 1234, dev)
gen = mod.generator
x = gen.cut_to_valid_length(batch["audio_body_conducted"])
from vibravox_amd import ops, gen_engine
gen_engine.USE_GRAPHS = False   # launch by launch: the profiler shows kernels, not a graph replay
math = ops.MATH_F32 if os.environ.get("GEN_BWD_MATH", "bf16") == "f32" else ops.MATH_BF16   # the training forward saves what THIS backward reads
# the forward arithmetic of the benchmarked step: hi + lo operands (three products) with the bf16 backward
gen_engine.set_forward_math("bf16x3" if math == ops.MATH_BF16 and mod.ru_forward_x3 else None)
for _ in range(3):
    with ops.backward_math(math):
        y, b = gen(x)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    with ops.backward_math(math):
        y, b = gen(x)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
for e in evs:
    print(f"{(e.time_range.start - t0):8.1f} us  {e.time_range.end - e.time_range.start:7.1f} us  {e.name[:90]}")
print("span", evs[-1].time_range.end - t0)
