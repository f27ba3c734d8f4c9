"""eben_bl_fm_sums_codes alone at the benchmark's embedding shapes (64 rows: [enhanced | reference]): time and bytes per second.
Usage: python tools/fm_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vibravox_amd._lib import load
from vibravox_amd.disc_engine_bl import Planes, _fm_sums

dev = torch.device("cuda", 0)
lib = load()
half = 32
chains = {"pqmf": [(24, 7994), (48, 3997), (96, 1999), (192, 1000), (384, 500), (768, 250), (768, 250)],
          "melgan": [(16, 31968), (64, 7992), (256, 1998), (1024, 500), (1024, 125), (1024, 125)]}
for name, shapes in chains.items():
    acts = []
    for c, l in shapes:
        p = Planes(2 * half, c, l, dev)
        p.hi.copy_(torch.randn(p.hi.shape, device=dev).to(torch.bfloat16)); p.lo.copy_((torch.randn(p.lo.shape, device=dev) * 0.004).to(torch.bfloat16))
        acts.append(p)
    sums = torch.empty(2 * len(acts), device=dev)
    for _ in range(3):
        _fm_sums(lib, acts, half, sums)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _fm_sums(lib, acts, half, sums)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = sum(2 * half * c * l * 4 + half * c * l for c, l in shapes)   # hi + lo of both halves read, one code byte per enhanced element written
    print(f"{name:7s} {ms * 1e3:7.1f} us  {nbytes / 1e6:7.1f} MB  {nbytes / ms / 1e9:6.2f} TB/s   sums[0] {float(sums[0]):.6e}")
