"""The PQMF filter-bank kernels alone on the device at BASELINE config 2 (batch 32 x 31968 samples, 4 bands, 32 taps): us per launch and the
achieved HBM rate of each shape (vibravox/torch_modules/dsp/pqmf.py:194-213 -- analysis = `pqmf_analysis_kernel`, synthesis + band sum =
`pqmf_synthesis_kernel`: the polyphase forms on wave shuffles; EBEN_PQMF_SHUFFLE=0 selects the LDS forms `fir_decimate_kernel` /
`fir_interp_sum_kernel`) and of their adjoints (the backward of the generator's synthesis / the balancing seeds).
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
from vibravox_amd import ops
n = pq.kernel_size
wa = pq.analysis_weights.detach().reshape(4, 32).contiguous()
ws = pq.synthesis_weights.detach().reshape(4, 32).contiguous()
with torch.no_grad():   # the launches themselves (ops._fir_*): through the module + autograd.Function a call costs ~13 us of host time
    for bands in (2, 4):
        y = pq(x, "analysis", bands=bands)
        wb = wa[:bands].contiguous()
        us = timed(lambda: ops._fir_decimate(x, wb, y.shape[2], bands, n, 4, -(n - 1)))
        rows.append((f"analysis, {bands} bands ({B},1,{T}) -> {tuple(y.shape)}", us, (x.numel() + y.numel()) * 4))
    bands4 = pq(x, "analysis", bands=4)
    out = pq.synthesis_sum(bands4)
    us = timed(lambda: ops._fir_interp_sum(bands4, ws, T, 4, n, 4, -(n - 1)))
rows.append((f"synthesis + band sum {tuple(bands4.shape)} -> {tuple(out.shape)}", us, (bands4.numel() + out.numel()) * 4))
# the adjoints as the kernels autograd launches for them (ops._FirInterpSumFn.backward = eben_fir_decimate, ops._FirDecimateFn.backward =
# eben_fir_interp_sum), HIP events around the launches -- not around torch.autograd.grad, whose ~40 us of host time the round-4 row timed
w4 = ws
g = torch.randn_like(out)
with torch.no_grad():
    us = timed(lambda: ops._fir_decimate(g, w4, bands4.shape[2], 4, n, 4, -(n - 1)))
rows.append((f"synthesis adjoint (analysis-form kernel) {tuple(out.shape)} -> {tuple(bands4.shape)}", us, (out.numel() + bands4.numel()) * 4))
gb = torch.randn_like(bands4)
with torch.no_grad():
    us = timed(lambda: ops._fir_interp_sum(gb, wa, T, 4, n, 4, -(n - 1)))
rows.append((f"analysis adjoint (synthesis-form kernel) {tuple(bands4.shape)} -> ({B}, 1, {T})", us, (gb.numel() + B * T) * 4))
for name, us, nb in rows:
    r = nb / us / 1e6
    assert r / 8 <= 1.0
    print(f"{name:64s} {us:7.1f} {nb / 1e6:7.1f} {r:6.2f} {r / 8:5.2f}")
