"""Is the generator launch-bound?  Host enqueue time vs GPU time of its forward and backward, alone on the device."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vibravox_amd import ops
dev = torch.device("cuda", 0)
mod = bench.build_module(dev, 1234)
batch = bench.synthetic_batch(32, 32000, 1234, dev)
x = mod.generator.cut_to_valid_length(batch["audio_body_conducted"])
gp = [p for p in mod.generator.parameters() if p.requires_grad]
def once():
    with ops.backward_math(ops.MATH_BF16):
        torch.cuda.synchronize(); e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t0 = time.perf_counter(); e[0].record(); y, bands = mod.generator(x); e[1].record(); t1 = time.perf_counter()
        torch.cuda.synchronize()
        seed = torch.ones_like(bands)
        torch.cuda.synchronize()
        t2 = time.perf_counter(); e[2].record()
        with ops.weight_grads_on_side_stream() as side:
            torch.autograd.backward(bands, seed, inputs=gp)
            tj = time.perf_counter(); ej = torch.cuda.Event(enable_timing=True); ej.record()
            side.join()
        e[3].record(); t3 = time.perf_counter()
        torch.cuda.synchronize()
        once.chain = (1e3 * (tj - t2), e[2].elapsed_time(ej))
    for p in gp: p.grad = None
    return 1e3 * (t1 - t0), e[0].elapsed_time(e[1]), 1e3 * (t3 - t2), e[2].elapsed_time(e[3])
for _ in range(3): once()
import gc; gc.collect(); gc.freeze()
r = [once() for _ in range(10)]
m = [sorted(c)[len(c) // 2] for c in zip(*r)]
print(f"generator forward: host {m[0]:.2f} ms, GPU {m[1]:.2f} ms;  backward (dX chain + dW on the side stream, joined): host {m[2]:.2f} ms, GPU {m[3]:.2f} ms; "
      f"dX chain alone: host {once.chain[0]:.2f} ms, GPU {once.chain[1]:.2f} ms")
