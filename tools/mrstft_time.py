"""MRSTFT loss alone on the device: forward and forward + backward time at the bench shape, and the kernels of one pass."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
fn = mod.reconstructive_loss_freq_fn
fn.stft_math = sys.argv[1] if len(sys.argv) > 1 else "folded_x3"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
y = batch["audio_airborne"] if "audio_airborne" in batch else list(batch.values())[0]
x = (y * 0.9 + 0.01 * torch.randn_like(y)).requires_grad_(True)
def fwd():
    return fn(x, y)
def both():
    x.grad = None
    fn(x, y).backward()
for f, name in ((fwd, "forward"), (both, "forward + backward")):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms ({fn.stft_math})")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    both(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
for e in evs:
    print(f"{(e.time_range.start - t0):8.1f} us  {e.time_range.end - e.time_range.start:7.1f} us  {e.name[:100]}")
