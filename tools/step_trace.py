"""The benchmarked train step under rocprofv3 --kernel-trace, step boundaries marked on the main stream (torch.cuda._sleep = one
`spin_kernel` dispatch per boundary): tools/step_trace_report.py turns the database into per-kernel time per step, GPU busy time per step
and the per-launch durations of the named discriminator layers in dispatch order.
Usage: rocprofv3 --kernel-trace -d DIR -o p -- python tools/step_trace.py [--steps N]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vibravox_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=10); a = ap.parse_args()
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
mod.disc_math = os.environ.get("EBEN_DISC_MATH", "bf16_bl"); mod.gen_backward_math = "f32" if mod.disc_math == "f32" else "bf16"
mod.stft_math = "folded" if mod.disc_math == "f32" else "folded_x3"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
n = 0
while n < 16 and (n < 6 or ops.graphs_pending()):
    mod.training_step(batch); n += 1
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
for _ in range(a.steps):
    torch.cuda._sleep(2000)
    mod.training_step(batch)
torch.cuda._sleep(2000)
torch.cuda.synchronize()
print("steps", a.steps, "graphs", ops.graphs_captured())
