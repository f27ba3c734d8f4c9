#!/usr/bin/env python
"""Stand-alone launches of the generator's eight non-residual convs (eben_generator.py:241-312) at BASELINE config-2 shapes (32 items),
forward in EBEN_MATH_BF16X6 (what the training forward computes in): time per launch, the matrix / HBM roofs of the launch
(six piece products per multiply-accumulate; fp32 tensors read and written once) and which kernel generation serves it.
Usage: python tools/gen_conv_bench.py [--iters 20] [--only enc3]"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vibravox_amd import ops  # noqa: E402
from vibravox_amd._lib import check, load, ptr, stream  # noqa: E402

LAYERS = [
    ("enc1", dict(c_in=32, c_out=64, ksize=4, stride=2, pad_l=1, pad_r=1, reflect=True), 7992),
    ("enc2", dict(c_in=64, c_out=128, ksize=8, stride=4, pad_l=3, pad_r=3, reflect=True), 3996),
    ("enc3", dict(c_in=128, c_out=256, ksize=16, stride=8, pad_l=7, pad_r=7, reflect=True), 999),
    ("lat1", dict(c_in=256, c_out=64, ksize=7, pad_l=3, pad_r=3, reflect=True, in_slope=0.01, out_slope=0.01), 125),
    ("lat2", dict(c_in=64, c_out=256, ksize=7, pad_l=3, pad_r=3, reflect=True, out_slope=0.01), 125),
    ("dec1", dict(c_in=256, c_out=128, ksize=16, stride=8, pad_l=4, transposed=True, out_slope=0.01), 125),
    ("dec2", dict(c_in=128, c_out=64, ksize=8, stride=4, pad_l=2, transposed=True, out_slope=0.01), 999),
    ("dec3", dict(c_in=64, c_out=32, ksize=4, stride=2, pad_l=1, transposed=True, out_slope=0.01), 3996),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = load()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tot = 0.0
    for name, kw, length in LAYERS:
        if a.only and a.only != name:
            continue
        spec = ops.ConvSpec(**kw)
        d = ops.conv_desc(spec, a.batch, length, ops.MATH_BF16X6)
        w = torch.randn(spec.weight_shape(), device=dev) * 0.05
        x = torch.randn(a.batch, spec.c_in, length, device=dev)
        l_out = spec.out_len(length)
        y = torch.empty(a.batch, spec.c_out, l_out, device=dev)
        wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 0), dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(w), None, ptr(wp), None, stream()), "pack")
        fn = lambda: check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(x), ptr(wp), None, None, ptr(y), stream()), "fwd")
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        macs = a.batch * l_out * spec.c_out * spec.c_in * spec.ksize / (spec.stride if spec.transposed else 1)
        flops = 2 * 6 * macs
        nbytes = 4 * (x.numel() + y.numel())
        gen = lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0)
        print(f"{name}  {us:7.1f} us   x6 MFMA {flops / 1e9:6.1f} GF = {flops / 2.5e15 * 1e6:5.1f} us at peak   {nbytes / 1e6:6.1f} MB = {nbytes / 8e12 * 1e6:5.1f} us at 8 TB/s   generation {gen}")
        tot += us
    print(f"total {tot:7.1f} us")


if __name__ == "__main__":
    main()
