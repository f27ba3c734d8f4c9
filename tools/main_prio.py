"""Experiment: the train step with the main stream = a HIGH-priority user stream (EBEN: the generator's critical path runs there) against
torch's default stream.  Usage: MAIN_PRIO=-1 python tools/main_prio.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vibravox_amd import ops
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
mod.disc_math = "bf16_bl"; mod.gen_backward_math = "bf16"; mod.stft_math = "folded_x3"
batch = bench.synthetic_batch(32, 32000, 1234, dev)
st = torch.cuda.Stream(device=dev, priority=int(os.environ["MAIN_PRIO"])) if os.environ.get("MAIN_PRIO") is not None else torch.cuda.current_stream()
with torch.cuda.stream(st):
    n = 0
    while n < 16 and (n < 6 or ops.graphs_pending()):
        mod.training_step(batch); n += 1
    torch.cuda.synchronize()
    import gc; gc.collect(); gc.freeze()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50):
            mod.training_step(batch)
        torch.cuda.synchronize()
        print(f"MAIN_PRIO={os.environ.get('MAIN_PRIO')} {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step")
