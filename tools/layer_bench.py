#!/usr/bin/env python
"""Per-layer micro-benchmark of the conv kernels at BASELINE config-2 shapes (batch 32, T=31968).

For every conv of the generator and the four discriminators: forward, input-gradient and
weight-gradient launches timed with HIP events through the C ABI; prints ms and fp32 TFLOP/s
(2*MACs / time; 157.3 TFLOP/s = MFMA fp32 peak).  Usage: python tools/layer_bench.py [--filter melgan] [--iters 5]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vibravox_amd import ops  # noqa: E402
from vibravox_amd._lib import check, load, ptr, stream  # noqa: E402
from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales  # noqa: E402
from vibravox_amd.torch_modules.dnn.eben_generator import EBENGenerator  # noqa: E402
from vibravox_amd.torch_modules.utils import HipConv1d  # noqa: E402


def collect(batch, length):
    """(name, spec, l_in) for every conv, by tracing one forward with hooks on the CPU-side modules."""
    dev = torch.device("cuda")
    gen, disc = EBENGenerator(4, 32, 2).to(dev), DiscriminatorEBENMultiScales(q=4, min_channels=24).to(dev)
    gen.use_engine = False   # module by module: the hooks below see every conv
    rows = []

    def hook(name):
        def f(mod, inp, out):
            rows.append((name, mod.spec, inp[0].shape[2], mod.bias is not None))
        return f

    for tag, net in (("G", gen), ("D", disc)):
        for n, m in net.named_modules():
            if isinstance(m, HipConv1d):
                m.register_forward_hook(hook(f"{tag}.{n}"))
    x = torch.randn(2, 1, length, device=dev) * 0.1
    with torch.no_grad():
        enh, bands = gen(x)
        disc(bands=bands, audio=enh)
    return rows


def time_ms(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--length", type=int, default=31968)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--filter", default="")
    ap.add_argument("--min-gmacs", type=float, default=0.0)
    ap.add_argument("--math", default="f32", choices=["f32", "bf16", "bf16x2", "bf16x3", "bf16x6"], help="bf16: EBEN_MATH_BF16 descriptors; dx through eben_conv1d_bwd_dx_ex; bf16x2: EBEN_MATH_BF16X2")
    a = ap.parse_args()
    lib = load()
    dev = torch.device("cuda")
    rows = collect(a.batch, a.length)
    tot = {"fwd": 0.0, "dx": 0.0, "dw": 0.0}
    print(f"{'layer':58s} {'GMAC':>7s} | {'fwd ms':>8s} {'TF':>6s} | {'dx ms':>8s} {'TF':>6s} | {'dw ms':>8s} {'TF':>6s}")
    for name, spec, l_in, has_bias in rows:
        if a.filter and a.filter not in name:
            continue
        math = {"bf16": ops.MATH_BF16, "bf16x2": ops.MATH_BF16X2, "f32": ops.MATH_F32, "bf16x3": ops.MATH_BF16X3, "bf16x6": ops.MATH_BF16X6}[a.math]
        d = ops.conv_desc(spec, a.batch, l_in, math)
        l_out = d.l_out
        wshape = spec.weight_shape()
        macs = a.batch * wshape[0] * wshape[1] * wshape[2] * (l_in if spec.transposed else l_out)
        if macs / 1e9 < a.min_gmacs:
            continue
        x = torch.randn(a.batch, spec.c_in, l_in, device=dev)
        v = torch.randn(*wshape, device=dev) * 0.05
        g = torch.ones(wshape[0], 1, 1, device=dev)
        bias = torch.zeros(spec.c_out, device=dev) if has_bias else None
        pw = ops.pack_weights(spec, d, v, g, None, True)
        y = torch.empty(a.batch, spec.c_out, l_out, device=dev)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        st = stream()
        wsb = lib.eben_conv1d_bwd_dx_workspace(ctypes.byref(d))
        ws = torch.empty(max(1, wsb // 4), device=dev)
        nslab, rs = ctypes.c_int(0), ctypes.c_int(0)
        dwb = lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(rs))
        slabs = torch.empty(max(1, dwb // 4), device=dev)
        dv, dg = torch.empty_like(v), torch.empty_like(g)
        db = torch.empty(wshape[0], device=dev) if has_bias else None
        ymask = y if spec.out_slope != 1.0 else None
        xmask = x if spec.in_slope != 1.0 else None

        def fwd():
            check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(x), ptr(pw.wp_fwd), ptr(bias), None, ptr(y), st))

        def bdx():
            if math != ops.MATH_F32:   # the engine's form: linear conv, masks in the producer's epilogue
                return check(lib.eben_conv1d_bwd_dx(ctypes.byref(d_lin), ptr(dy), None, ptr(pw.wp_bwd), None, ptr(dx), 0, ptr(ws), wsb, st))
            check(lib.eben_conv1d_bwd_dx(ctypes.byref(d), ptr(dy), ptr(ymask), ptr(pw.wp_bwd), ptr(xmask), ptr(dx), 0, ptr(ws), wsb, st))

        if math != ops.MATH_F32:   # the engine's form: pre-masked gradient, linear conv
            import dataclasses
            d_lin = ops.conv_desc(dataclasses.replace(spec, in_slope=1.0, out_slope=1.0), a.batch, l_in, math)
            dwb = lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d_lin), ctypes.byref(nslab), ctypes.byref(rs))
            slabs = torch.empty(max(1, dwb // 4), device=dev)

        def bdw():
            if math != ops.MATH_F32:
                check(lib.eben_conv1d_bwd_dw(ctypes.byref(d_lin), ptr(dy), None, ptr(x), 1 if has_bias else 0, ptr(slabs), dwb, st))
            else:
                check(lib.eben_conv1d_bwd_dw(ctypes.byref(d), ptr(dy), ptr(ymask), ptr(x), 1 if has_bias else 0, ptr(slabs), dwb, st))
            check(lib.eben_wn_bwd(ptr(slabs), nslab.value, wshape[0] * rs.value, wshape[0], wshape[1] * wshape[2], rs.value,
                                  ptr(g), ptr(v), ptr(pw.norm), ptr(dg), ptr(dv), ptr(db), st))

        t = [time_ms(f, a.iters) for f in (fwd, bdx, bdw)]
        tf = [2 * macs / (ms * 1e-3) / 1e12 for ms in t]
        for k, ms in zip(tot, t):
            tot[k] += ms
        print(f"{name[:58]:58s} {macs/1e9:7.2f} | {t[0]:8.3f} {tf[0]:6.1f} | {t[1]:8.3f} {tf[1]:6.1f} | {t[2]:8.3f} {tf[2]:6.1f}  (nslab {nslab.value}, gen {lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0)}/{lib.eben_conv1d_kernel_generation(ctypes.byref(d), 1)})")
    print(f"TOTAL ms: fwd {tot['fwd']:.2f}  dx {tot['dx']:.2f}  dw {tot['dw']:.2f}")


if __name__ == "__main__":
    main()
