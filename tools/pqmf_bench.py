"""The PQMF filter-bank kernels alone on the device at BASELINE config 2 (batch 32 x 31968 samples, 4 bands, 32 taps): us per launch and the
achieved HBM rate of each shape (vibravox/torch_modules/dsp/pqmf.py:194-213 -- analysis = strided FIR decimation `fir_decimate_kernel`,
synthesis + band sum = `fir_interp_sum_kernel`) and of their adjoints (the backward of the generator's synthesis / the balancing seeds).
Algorithmic bytes: every input sample read once, every output sample written once (4 B each).  Peak 8 TB/s (6.3 measured for a copy).
Usage: python tools/pqmf_bench.py [--iters 50]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vibravox_amd.torch_modules.dsp.pqmf import PseudoQMFBanks

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=50); ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda")
pq = PseudoQMFBanks(decimation=4, kernel_size=32).to(dev)
B, T = args.batch, 31968


def timed(fn):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


x = 0.1 * torch.randn(B, 1, T, device=dev)
print(f"{'launch':64s} {'us':>7s} {'MB':>7s} {'TB/s':>6s} {'of 8':>5s}")
rows = []
for bands in (2, 4):
    with torch.no_grad():
        y = pq(x, "analysis", bands=bands)
        us = timed(lambda: pq(x, "analysis", bands=bands))
    nb = (x.numel() + y.numel()) * 4
    rows.append((f"fir_decimate_kernel: analysis, {bands} bands ({B},1,{T}) -> {tuple(y.shape)}", us, nb))
with torch.no_grad():
    bands4 = pq(x, "analysis", bands=4)
    out = pq.synthesis_sum(bands4)
    us = timed(lambda: pq.synthesis_sum(bands4))
rows.append((f"fir_interp_sum_kernel: synthesis + band sum {tuple(bands4.shape)} -> {tuple(out.shape)}", us, (bands4.numel() + out.numel()) * 4))
# adjoints through autograd (what the generator backward / the balancing seeds launch)
bl = bands4.clone().requires_grad_(True)
o = pq.synthesis_sum(bl)
g = torch.randn_like(o)
us = timed(lambda: torch.autograd.grad(o, bl, grad_outputs=g, retain_graph=True))
rows.append((f"synthesis adjoint (fir_decimate_kernel) {tuple(o.shape)} -> {tuple(bl.shape)}", us, (o.numel() + bl.numel()) * 4))
for name, us, nb in rows:
    r = nb / us / 1e6
    assert r / 8 <= 1.0
    print(f"{name:64s} {us:7.1f} {nb / 1e6:7.1f} {r:6.2f} {r / 8:5.2f}")
