"""GPU: every HIP kernel family through the C ABI against the CPU oracle (float64) on seeded inputs.

Tolerances (fp32 kernels vs an fp64 reference): max-abs error <= 3e-5 * max|ref| for the conv
GEMMs (K up to ~5k fp32 accumulations), 1e-5 for elementwise / FIR ops.
"""
import math

import numpy as np
import pytest
import torch

from formula import formula_tensor
from oracle import eben_oracle as O

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.detach().double().cpu(), ref.detach().double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


CONV_CASES = {
    # name: (kwargs of ConvSpec, batch, length, weight_norm, bias)
    "plain_k3_bias_act": (dict(c_in=8, c_out=24, ksize=3, pad_l=1, pad_r=1, out_slope=0.2), 3, 700, True, True),
    "melgan_l1_like": (dict(c_in=16, c_out=64, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 2, 1500, True, True),
    "melgan_l2_like": (dict(c_in=64, c_out=256, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 2, 1100, True, True),
    "dense_k5_chunks": (dict(c_in=256, c_out=384, ksize=5, pad_l=2, pad_r=2, out_slope=0.2), 2, 300, True, True),
    "pqmf_disc_d3": (dict(c_in=24, c_out=48, ksize=7, stride=2, dilation=3, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 3, 1001, True, True),
    "pqmf_disc_d2_even": (dict(c_in=12, c_out=24, ksize=7, stride=2, dilation=2, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 3, 1000, True, True),
    "pqmf_disc_wide": (dict(c_in=384, c_out=768, ksize=7, stride=2, dilation=2, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 2, 260, True, True),
    "ru_dilated_reflect": (dict(c_in=32, c_out=32, ksize=3, dilation=9, pad_l=9, pad_r=9, reflect=True, in_slope=0.01), 3, 600, True, False),
    "ru_pointwise": (dict(c_in=64, c_out=64, ksize=1, out_slope=0.01), 3, 517, True, False),
    "enc_s4_reflect": (dict(c_in=64, c_out=128, ksize=8, stride=4, pad_l=3, pad_r=3, reflect=True), 2, 1000, True, False),
    "enc_s8_reflect": (dict(c_in=128, c_out=256, ksize=16, stride=8, pad_l=7, pad_r=7, reflect=True), 2, 1000, True, False),
    "latent_k7_reflect": (dict(c_in=256, c_out=64, ksize=7, pad_l=3, pad_r=3, reflect=True, out_slope=0.01), 2, 125, True, False),
    "dec_convT_s8": (dict(c_in=256, c_out=128, ksize=16, stride=8, pad_l=4, transposed=True, out_slope=0.01), 2, 125, True, False),
    "dec_convT_s2": (dict(c_in=64, c_out=32, ksize=4, stride=2, pad_l=1, transposed=True, in_slope=0.01, out_slope=0.01), 2, 500, True, False),
    "logits_m1": (dict(c_in=96, c_out=1, ksize=3, pad_l=1, pad_r=1), 3, 251, True, True),
    "first_conv_plain": (dict(c_in=2, c_out=32, ksize=3, pad_l=1, pad_r=1, reflect=True), 3, 1000, False, False),
    "last_conv_plain": (dict(c_in=32, c_out=4, ksize=3, pad_l=1, pad_r=1, reflect=True), 3, 1000, False, False),
    "melgan_l0_k15": (dict(c_in=1, c_out=16, ksize=15, out_slope=0.2), 2, 2014, True, True),
    "stft_like": (dict(c_in=1, c_out=66, ksize=24, stride=5, pad_l=12, pad_r=12, reflect=True), 4, 1003, False, False),
    "tiny_l": (dict(c_in=8, c_out=8, ksize=3, pad_l=1, pad_r=1), 1, 5, True, True),
    # second-generation kernels: channel-chunked input tiles (tile hand-over mid weight chunk), 96-row tiles,
    # phase-scatter with several tile refreshes, time chunks that do not divide the length
    "melgan_l3_like_chunked": (dict(c_in=256, c_out=512, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 2, 700, True, True),
    "melgan_l5_like_dense": (dict(c_in=320, c_out=256, ksize=5, pad_l=2, pad_r=2, out_slope=0.2), 2, 131, True, True),
    "pqmf_l6_like_96rows": (dict(c_in=384, c_out=384, ksize=5, dilation=3, pad_l=2, pad_r=2, groups=4, out_slope=0.2), 2, 140, True, True),
    "enc_s8_wide": (dict(c_in=128, c_out=256, ksize=16, stride=8, pad_l=7, pad_r=7, reflect=True, in_slope=0.01), 2, 1500, True, False),
    # direct (VALU) kernel: 1 / 4 / 6 / 12 / 16 output channels per group, strided, dilated, input gradients
    "thin_pqmf_l0": (dict(c_in=4, c_out=24, ksize=3, dilation=2, pad_l=1, pad_r=1, groups=4, out_slope=0.2), 3, 1003, True, True),
    "thin_pqmf_l1": (dict(c_in=24, c_out=48, ksize=7, stride=2, dilation=3, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 3, 999, True, True),
    "thin_melgan_l1": (dict(c_in=16, c_out=64, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 2, 2100, True, True),
}


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv_layer_fwd_bwd(hip, name):
    from vibravox_amd import ops

    kw, batch, length, wn, has_bias = CONV_CASES[name]
    spec = ops.ConvSpec(**kw)
    wshape = spec.weight_shape()
    fan_in = wshape[1] * wshape[2]
    v = formula_tensor(f"{name}/v", wshape, 1 / math.sqrt(fan_in))
    g = (v.reshape(wshape[0], -1).norm(dim=1).reshape(-1, 1, 1) * (1 + 0.3 * formula_tensor(f"{name}/g", (wshape[0], 1, 1)))) if wn else None
    bias = formula_tensor(f"{name}/b", (spec.c_out,), 0.1) if has_bias else None
    x = formula_tensor(f"{name}/x", (batch, spec.c_in, length))
    l_out = spec.out_len(length)
    seed = formula_tensor(f"{name}/dy", (batch, spec.c_out, l_out))

    # CPU float64 reference
    rx, rv = x.double().requires_grad_(True), v.double().requires_grad_(True)
    rg = g.double().requires_grad_(True) if wn else None
    rb = bias.double().requires_grad_(True) if has_bias else None
    okw = {k: val for k, val in kw.items() if k not in ("c_in", "c_out", "ksize")}
    ry = O.conv_layer(rx, rv, rg, rb, **okw)
    assert ry.shape[2] == l_out
    (ry * seed.double()).sum().backward()

    dev = torch.device("cuda")
    dx_, dv_ = x.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
    dg_ = g.to(dev).requires_grad_(True) if wn else None
    db_ = bias.to(dev).requires_grad_(True) if has_bias else None
    y = ops.conv_layer(dx_, dv_, dg_, db_, spec)
    (y * seed.to(dev)).sum().backward()
    torch.cuda.synchronize()

    tol = 3e-5
    assert rel_err(y, ry) < tol, f"forward {rel_err(y, ry)}"
    assert rel_err(dx_.grad, rx.grad) < tol, f"dx {rel_err(dx_.grad, rx.grad)}"
    assert rel_err(dv_.grad, rv.grad) < 2 * tol, f"dv {rel_err(dv_.grad, rv.grad)}"
    if wn:
        assert rel_err(dg_.grad, rg.grad) < 2 * tol, f"dg {rel_err(dg_.grad, rg.grad)}"
    if has_bias:
        assert rel_err(db_.grad, rb.grad) < 2 * tol, f"dbias {rel_err(db_.grad, rb.grad)}"


def test_kernel_generations_cover_the_eben_layers(hip):
    """Every kernel family is exercised by the shapes above and serves the layers it was written for."""
    import ctypes

    from vibravox_amd import ops
    from vibravox_amd._lib import load

    lib = load()

    def gen(name, which, batch=32):
        kw, _, length, _, _ = CONV_CASES[name]
        spec = ops.ConvSpec(**kw)
        return lib.eben_conv1d_kernel_generation(ctypes.byref(ops.conv_desc(spec, batch, length)), which)

    assert gen("melgan_l3_like_chunked", 0) == 2 and gen("melgan_l3_like_chunked", 1) == 2
    assert gen("pqmf_l6_like_96rows", 0) == 2
    assert gen("thin_pqmf_l0", 0) == 3 and gen("thin_pqmf_l1", 1) == 3 and gen("thin_melgan_l1", 1) == 3
    assert gen("logits_m1", 0) == 1 and gen("stft_like", 0) == 1
    seen = {gen(n, w) for n in CONV_CASES for w in (0, 1)}
    assert seen == {1, 2, 3}


BF16_CASES = {
    # name: (ConvSpec kwargs, length, (generation of the forward, of the input gradient) at math = bf16)
    "melgan_l3_like_chunked": (CONV_CASES["melgan_l3_like_chunked"][0], 700, (4, 4)),
    "pqmf_disc_wide": (CONV_CASES["pqmf_disc_wide"][0], 260, (4, 4)),
    "dense_k5_chunks": (CONV_CASES["dense_k5_chunks"][0], 300, (4, 4)),
    "pqmf_l6_like_96rows": (CONV_CASES["pqmf_l6_like_96rows"][0], 140, (4, 4)),
    "melgan_l2_like": (CONV_CASES["melgan_l2_like"][0], 1100, (4, 4)),   # 16 rows per group on the gradient side: half-filled 32-row tiles
    "melgan_l1_like": (CONV_CASES["melgan_l1_like"][0], 1500, (4, 4)),   # 4 channels / 4 rows per group: block-diagonal over the groups
    "pqmf_l1_like_6ch": (CONV_CASES["thin_pqmf_l1"][0], 999, (4, 4)),     # 6 channels per group: the four groups as one block-diagonal contraction
    "melgan_l0_k15": (CONV_CASES["melgan_l0_k15"][0], 2014, (None, None)),   # one input channel: direct kernel, bf16 weight gradient
    "pqmf_mid_24ch": (dict(c_in=96, c_out=192, ksize=7, stride=2, dilation=3, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 1001, (4, 4)),
    "pqmf_low_12ch": (dict(c_in=48, c_out=96, ksize=7, stride=2, dilation=1, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 2003, (None, None)),
    "ru32_dilated": (dict(c_in=32, c_out=32, ksize=3, dilation=9, pad_l=9, pad_r=9, out_slope=0.01), 900, (4, 4)),   # reduction of 96: 1.5 weight chunks
    "ru64_pointwise": (dict(c_in=64, c_out=64, ksize=1, out_slope=0.01), 517, (4, 4)),                                # reduction of 64: one chunk
    "melgan_l4_like": (dict(c_in=512, c_out=512, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 500, (4, 4)),
}


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("name", list(BF16_CASES))
def test_bf16_math_forward_and_batched_input_gradient(hip, name):
    """EBEN_MATH_BF16 (tapconv3.hip): both MFMA operands are rounded to bf16 (RNE), everything else is fp32 --
    so against an fp64 conv of the bf16-ROUNDED operands the kernel must be as tight as the fp32 kernels
    (this pins the kernel itself; what the rounding costs the train step is measured in test_gpu_models)."""
    import ctypes
    import dataclasses

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    kw, length, gens = BF16_CASES[name]
    spec = ops.ConvSpec(**kw)
    S = 2
    wshape = spec.weight_shape()
    w = formula_tensor(f"bf/{name}/w", wshape, 1 / math.sqrt(wshape[1] * wshape[2]))
    bias = formula_tensor(f"bf/{name}/b", (spec.c_out,), 0.1)
    x = formula_tensor(f"bf/{name}/x", (2 * S, spec.c_in, length))
    l_out = spec.out_len(length)
    okw = {k: v for k, v in kw.items() if k not in ("c_in", "c_out", "ksize", "in_slope", "out_slope")}
    dev = torch.device("cuda")
    wd, bd, xd = w.to(dev), bias.to(dev), x.to(dev)

    # forward: lrelu(conv(bf16(x), bf16(w)) + bias)
    d = ops.conv_desc(spec, 2 * S, length, ops.MATH_BF16)
    fgen = lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0)
    assert gens[0] is None or fgen == gens[0]
    rnd = _bf16 if fgen == 4 else (lambda t: t.double())
    ref = torch.nn.functional.leaky_relu(O.conv_layer(rnd(x), rnd(w), None, bias.double(), **okw), spec.out_slope)
    wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 0), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, ptr(wp), None, stream()), "pack")
    y = torch.empty(2 * S, spec.c_out, l_out, dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xd), ptr(wp), ptr(bd), None, ptr(y), stream()), "fwd")
    torch.cuda.synchronize()
    assert rel_err(y, ref) < 3e-5
    # ... and the rounding itself costs what bf16 should cost (2^-9 per operand, averaged over the reduction)
    exact = torch.nn.functional.leaky_relu(O.conv_layer(x.double(), w.double(), None, bias.double(), **okw), spec.out_slope)
    assert (1e-4 if fgen == 4 else 0.0) < rel_err(y, exact) < 1e-2

    # batched input gradient of the linear conv: dx[b] = (conv^T(bf16(g[b]); bf16(w)) + res[b < S]) * lrelu'(act[map(b)])
    lin = dataclasses.replace(spec, in_slope=1.0, out_slope=1.0)
    d = ops.conv_desc(lin, 4 * S, length, ops.MATH_BF16)
    got_gen = lib.eben_conv1d_kernel_generation(ctypes.byref(d), 1)
    if gens[1] is not None:
        assert got_gen == gens[1]
    g = formula_tensor(f"bf/{name}/g", (4 * S, spec.c_out, l_out))
    act = formula_tensor(f"bf/{name}/act", (2 * S, spec.c_in, length))
    res = formula_tensor(f"bf/{name}/res", (S, spec.c_in, length))
    rounded = got_gen == 4
    xr = torch.zeros(4 * S, spec.c_in, length, dtype=torch.float64, requires_grad=True)
    (O.conv_layer(xr, _bf16(w) if rounded else w.double(), None, None, **okw) * (_bf16(g) if rounded else g.double())).sum().backward()
    ref = xr.grad.clone()
    ref[:S] += res.double()
    rows = torch.tensor([0, 1, 0, 1, 0, 1, 2, 3])
    ref = ref * torch.where(act.double()[rows] > 0, 1.0, 0.2)
    wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 1), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, None, ptr(wp), stream()), "pack")
    dx = torch.empty(4 * S, spec.c_in, length, dtype=torch.float32, device=dev)
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
    gd, rd, ad = g.to(dev), res.to(dev), act.to(dev)
    check(lib.eben_conv1d_bwd_dx_ex(ctypes.byref(d), ptr(gd), ptr(wp), ptr(rd), S, ptr(ad), 0.2, S, seg_map, ptr(dx), stream()),
          "bwd_dx_ex")
    torch.cuda.synchronize()
    assert rel_err(dx, ref) < 3e-5

    # weight / bias gradient over 24 batch rows (one full group of 16 + a half-empty one): k-steps of 16 batch items
    nb = 24
    d = ops.conv_desc(lin, nb, length, ops.MATH_BF16)
    xb = formula_tensor(f"bf/{name}/xb", (nb, spec.c_in, length))
    gb = formula_tensor(f"bf/{name}/gb", (nb, spec.c_out, l_out))
    wr = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    (O.conv_layer(_bf16(xb), wr, None, None, **okw) * _bf16(gb)).sum().backward()
    nslab, row_stride = ctypes.c_int(0), ctypes.c_int(0)
    ws_bytes = lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride))
    slabs = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    xbd, gbd = xb.to(dev), gb.to(dev)
    check(lib.eben_conv1d_bwd_dw(ctypes.byref(d), ptr(gbd), None, ptr(xbd), 1, ptr(slabs), ws_bytes, stream()), "bwd_dw")
    rows, cols = wshape[0], wshape[1] * wshape[2]
    dv = torch.empty(wshape, dtype=torch.float32, device=dev)
    dbias = torch.empty(rows, dtype=torch.float32, device=dev)
    check(lib.eben_wn_bwd(ptr(slabs), nslab.value, rows * row_stride.value, rows, cols, row_stride.value, None, None, None, None,
                          ptr(dv), ptr(dbias), stream()), "wn_bwd")
    torch.cuda.synchronize()
    assert rel_err(dv, wr.grad) < 1e-4
    assert rel_err(dbias, _bf16(gb).sum(dim=(0, 2))) < 1e-4

    # autograd's form of the backward (ops.backward_math): the fused output activation is differentiated ON LOAD --
    # the gradient operand is dy * lrelu'(y), rounded after the mask
    if spec.out_slope != 1.0 and not spec.reflect:
        d = ops.conv_desc(spec, nb, length, ops.MATH_BF16)
        yb = formula_tensor(f"bf/{name}/yb", (nb, spec.c_out, l_out))
        masked = gb.double() * torch.where(yb.double() > 0, 1.0, spec.out_slope)
        gens_m = (lib.eben_conv1d_kernel_generation(ctypes.byref(d), 1), )
        rnd = _bf16 if gens_m[0] == 4 else (lambda t: t.double())
        xr = torch.zeros(nb, spec.c_in, length, dtype=torch.float64, requires_grad=True)
        (O.conv_layer(xr, rnd(w), None, None, **okw) * rnd(masked)).sum().backward()
        wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 1), dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, None, ptr(wp), stream()), "pack")
        dxm = torch.empty(nb, spec.c_in, length, dtype=torch.float32, device=dev)
        ybd = yb.to(dev)
        check(lib.eben_conv1d_bwd_dx(ctypes.byref(d), ptr(gbd), ptr(ybd), ptr(wp), None, ptr(dxm), 0, None, 0, stream()), "bwd_dx")
        wr2 = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
        (O.conv_layer(_bf16(xb), wr2, None, None, **okw) * _bf16(masked)).sum().backward()
        ws_bytes = lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride))
        slabs = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_bwd_dw(ctypes.byref(d), ptr(gbd), ptr(ybd), ptr(xbd), 1, ptr(slabs), ws_bytes, stream()), "bwd_dw")
        check(lib.eben_wn_bwd(ptr(slabs), nslab.value, rows * row_stride.value, rows, cols, row_stride.value, None, None, None, None,
                              ptr(dv), ptr(dbias), stream()), "wn_bwd")
        torch.cuda.synchronize()
        assert rel_err(dxm, xr.grad) < 3e-5
        assert rel_err(dv, wr2.grad) < 1e-4


GC_CASES = {
    # the generator's eight non-residual convs (eben_generator.py:241-312) with the MFMA B operand loaded straight from the fp32 rows
    # (gen_conv.hip, kernel generation 5): name -> (ConvSpec kwargs, batch, length, bias, residual)
    "enc_s2": (dict(c_in=32, c_out=64, ksize=4, stride=2, pad_l=1, pad_r=1, reflect=True), 3, 1000, False, False),
    "enc_s4": (dict(c_in=64, c_out=128, ksize=8, stride=4, pad_l=3, pad_r=3, reflect=True), 2, 1001, False, False),
    "enc_s8": (dict(c_in=128, c_out=256, ksize=16, stride=8, pad_l=7, pad_r=7, reflect=True), 2, 999, False, False),
    "enc_s8_act_bias": (dict(c_in=128, c_out=256, ksize=16, stride=8, pad_l=7, pad_r=7, reflect=True, in_slope=0.01, out_slope=0.2), 2, 1500, True, False),
    "latent_down": (dict(c_in=256, c_out=64, ksize=7, pad_l=3, pad_r=3, reflect=True, in_slope=0.01, out_slope=0.01), 3, 125, False, False),
    "latent_up": (dict(c_in=64, c_out=256, ksize=7, pad_l=3, pad_r=3, reflect=True, out_slope=0.01), 3, 125, False, False),
    "latent_zero_pad_res": (dict(c_in=64, c_out=64, ksize=5, pad_l=2, pad_r=2, out_slope=0.01), 2, 300, True, True),
    "dec_s8": (dict(c_in=256, c_out=128, ksize=16, stride=8, pad_l=4, transposed=True, out_slope=0.01), 2, 125, False, False),
    "dec_s4": (dict(c_in=128, c_out=64, ksize=8, stride=4, pad_l=2, transposed=True, out_slope=0.01), 2, 999, False, False),
    "dec_s2": (dict(c_in=64, c_out=32, ksize=4, stride=2, pad_l=1, transposed=True, out_slope=0.01), 2, 3996 // 4, False, False),
    "dec_s2_bias_res": (dict(c_in=64, c_out=32, ksize=4, stride=2, pad_l=1, transposed=True, in_slope=0.01, out_slope=0.01), 2, 517, True, True),
    "dec_s4_bias_res": (dict(c_in=32, c_out=16, ksize=8, stride=4, pad_l=2, transposed=True), 1, 70, True, True),
    "dec_s8_bias_res": (dict(c_in=16, c_out=8, ksize=16, stride=8, pad_l=4, transposed=True), 1, 33, True, True),
    "short_rows": (dict(c_in=16, c_out=32, ksize=8, stride=4, pad_l=3, pad_r=3, reflect=True), 5, 9, False, False),   # every tile at both ends of the row
    "one_column": (dict(c_in=16, c_out=32, ksize=7, pad_l=3, pad_r=3), 2, 1, True, False),
}


@pytest.mark.parametrize("name", list(GC_CASES))
def test_generator_direct_operand_convs(hip, name):
    """gen_conv.hip (EBEN_MATH_BF16X6 forward of the generator's strided / transposed / latent convs): fp32-grade against the fp64 conv
    of the exact operands -- 3e-5 is the bound of the fp32 kernels, the split products land at ~1e-6."""
    import ctypes

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    kw, batch, length, has_bias, has_res = GC_CASES[name]
    spec = ops.ConvSpec(**kw)
    wshape = spec.weight_shape()
    v = formula_tensor(f"gc/{name}/w", wshape, 1 / math.sqrt(wshape[1] * wshape[2]))
    g = formula_tensor(f"gc/{name}/g", (wshape[0], 1, 1)).abs() + 0.5
    bias = formula_tensor(f"gc/{name}/b", (spec.c_out,), 0.1) if has_bias else None
    x = formula_tensor(f"gc/{name}/x", (batch, spec.c_in, length))
    l_out = spec.out_len(length)
    res = formula_tensor(f"gc/{name}/r", (batch, spec.c_out, l_out)) if has_res else None
    okw = {k: val for k, val in kw.items() if k not in ("c_in", "c_out", "ksize")}
    exact = O.conv_layer(x.double(), v.double(), g.double(), None if bias is None else bias.double(), **okw)
    if res is not None:
        exact = exact + torch.nn.functional.leaky_relu(res.double(), 0.3)
    dev = torch.device("cuda")
    d = ops.conv_desc(spec, batch, length, ops.MATH_BF16X6)
    assert lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0) == 5
    vd, gd, xd = v.to(dev), g.to(dev), x.to(dev)
    scale = torch.empty(wshape[0], dtype=torch.float32, device=dev)
    norm = torch.empty_like(scale)
    check(lib.eben_wn_scale(ptr(gd), ptr(vd), wshape[0], wshape[1] * wshape[2], ptr(scale), ptr(norm), stream()), "wn_scale")
    wp = torch.full((lib.eben_conv1d_packed_floats(ctypes.byref(d), 0),), float("nan"), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(vd), ptr(scale), ptr(wp), None, stream()), "pack")
    y = torch.full((batch, spec.c_out, l_out), float("nan"), dtype=torch.float32, device=dev)
    bd = None if bias is None else bias.to(dev)
    if res is None:
        check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xd), ptr(wp), ptr(bd), None, ptr(y), stream()), "fwd")
    else:
        rd = res.to(dev)
        check(lib.eben_conv1d_fwd_res(ctypes.byref(d), ptr(xd), ptr(wp), ptr(bd), ptr(rd), 0.3, ptr(y), stream()), "fwd_res")
    torch.cuda.synchronize()
    assert torch.isfinite(wp).all()
    assert rel_err(y, exact) < 3e-6


@pytest.mark.parametrize("math_name", ["bf16x6", "bf16x3"])
@pytest.mark.parametrize("name", [n for n, c in BF16_CASES.items() if c[2][0] == 4])
def test_split_bf16_math_forward_and_batched_input_gradient(hip, name, math_name):
    """EBEN_MATH_BF16X6 / BF16X3 (tapconv3.hip with split operands): both MFMA operands enter as three / two bf16 pieces and a
    product is the sum of the six / three piece products that matter.  Against the fp64 conv of the EXACT operands:
    X6 at the bound of the fp32 kernels (3e-5) and within 2x of what the exact-fp32 kernel itself achieves on the same inputs --
    it is fp32 arithmetic; X3 at 2^-17 per product (1e-4).  Weight gradients of these modes run on the fp32 kernels."""
    import ctypes
    import dataclasses

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    kw, length, _ = BF16_CASES[name]
    spec = ops.ConvSpec(**kw)
    mm, tol = (ops.MATH_BF16X6, 3e-5) if math_name == "bf16x6" else (ops.MATH_BF16X3, 1e-4)
    S = 2
    wshape = spec.weight_shape()
    w = formula_tensor(f"bf/{name}/w", wshape, 1 / math.sqrt(wshape[1] * wshape[2]))
    bias = formula_tensor(f"bf/{name}/b", (spec.c_out,), 0.1)
    x = formula_tensor(f"bf/{name}/x", (2 * S, spec.c_in, length))
    l_out = spec.out_len(length)
    okw = {k: v for k, v in kw.items() if k not in ("c_in", "c_out", "ksize", "in_slope", "out_slope")}
    dev = torch.device("cuda")
    wd, bd, xd = w.to(dev), bias.to(dev), x.to(dev)

    def forward(math_id):
        d = ops.conv_desc(spec, 2 * S, length, math_id)
        gen = lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0)
        wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 0), dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, ptr(wp), None, stream()), "pack")
        y = torch.full((2 * S, spec.c_out, l_out), float("nan"), dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xd), ptr(wp), ptr(bd), None, ptr(y), stream()), "fwd")
        torch.cuda.synchronize()
        return y, gen

    exact = torch.nn.functional.leaky_relu(O.conv_layer(x.double(), w.double(), None, bias.double(), **okw), spec.out_slope)
    y, gen = forward(mm)
    assert gen in (4, 5)   # 5: the dense stride-1 k = 5 case is a latent-conv shape (gen_conv.hip takes it on two or three pieces per operand)
    assert rel_err(y, exact) < tol
    if mm == ops.MATH_BF16X6:
        y32, _ = forward(ops.MATH_F32)
        assert rel_err(y, exact) < 2 * rel_err(y32, exact) + 2e-7

    lin = dataclasses.replace(spec, in_slope=1.0, out_slope=1.0)
    g = formula_tensor(f"bf/{name}/g", (4 * S, spec.c_out, l_out))
    act = formula_tensor(f"bf/{name}/act", (2 * S, spec.c_in, length))
    res = formula_tensor(f"bf/{name}/res", (S, spec.c_in, length))
    xr = torch.zeros(4 * S, spec.c_in, length, dtype=torch.float64, requires_grad=True)
    (O.conv_layer(xr, w.double(), None, None, **okw) * g.double()).sum().backward()
    ref = xr.grad.clone()
    ref[:S] += res.double()
    rows = torch.tensor([0, 1, 0, 1, 0, 1, 2, 3])
    ref = ref * torch.where(act.double()[rows] > 0, 1.0, 0.2)
    gd, rd, ad = g.to(dev), res.to(dev), act.to(dev)
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)

    def input_gradient(math_id):
        d = ops.conv_desc(lin, 4 * S, length, math_id)
        gen = lib.eben_conv1d_kernel_generation(ctypes.byref(d), 1)
        wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 1), dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, None, ptr(wp), stream()), "pack")
        dx = torch.full((4 * S, spec.c_in, length), float("nan"), dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_bwd_dx_ex(ctypes.byref(d), ptr(gd), ptr(wp), ptr(rd), S, ptr(ad), 0.2, S, seg_map, ptr(dx), stream()), "bwd_dx_ex")
        torch.cuda.synchronize()
        return dx, gen

    dx, gen = input_gradient(mm)
    assert gen == 4
    assert rel_err(dx, ref) < tol
    if mm == ops.MATH_BF16X6:
        dx32, _ = input_gradient(ops.MATH_F32)
        assert rel_err(dx, ref) < 2 * rel_err(dx32, ref) + 2e-7
    # the weight gradient of these modes is the exact-fp32 kernels' (generation 2 / 1)
    d = ops.conv_desc(lin, 24, length, mm)
    nslab, row_stride = ctypes.c_int(0), ctypes.c_int(0)
    d32 = ops.conv_desc(lin, 24, length, ops.MATH_F32)
    assert lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride)) == \
        lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d32), ctypes.byref(nslab), ctypes.byref(row_stride))


# max |got - ref| / max |ref| against the fp64 unit, per math mode of the fused ResidualUnit launches: exact fp32 (0) and three bf16
# pieces per operand (4) at the fp32 bound of the conv tests; two pieces (3): 2^-17 per product; one (1): plain bf16 operands
RU_TOL = {0: 3e-5, 4: 3e-5, 3: 1e-4, 1: 2e-2}


@pytest.mark.parametrize("channels,dilation,length,batch,in_slope", [
    (32, 1, 1000, 3, 1.0), (32, 3, 517, 2, 0.01), (32, 9, 8000, 2, 1.0), (64, 9, 300, 3, 0.01), (64, 1, 4000, 2, 1.0),
    (128, 3, 1000, 2, 1.0), (128, 9, 131, 2, 0.01), (32, 9, 20, 1, 1.0)])
@pytest.mark.parametrize("math_mode", [0, 4, 3, 1])
def test_fused_residual_unit_forward(hip, channels, dilation, length, batch, in_slope, math_mode):
    """eben_ru_fwd_ex (math 0 = eben_ru_fwd, the exact-fp32 MFMA kernel; 4 / 3 / 1 = the split-bf16 kernels with three / two / one
    piece per operand, tolerances RU_TOL): y = xin + lrelu(W_pw . (W_dil (*) xin), 0.01), xin = lrelu(x, in_slope) -- eben_generator.py:287-316 in one
    launch -- against the fp64 oracle: tile interiors (float4 staging), both reflected ends, lengths that are not a multiple
    of the 128-position tile or of 4, a clip shorter than one tile; plus the two tensors kept for the backward."""
    from vibravox_amd._lib import check, load, ptr, stream

    lib, dev = load(), torch.device("cuda")
    c = channels
    vd = formula_tensor(f"ru/{c}/{dilation}/vd", (c, c, 3), 1 / math.sqrt(3 * c))
    vp = formula_tensor(f"ru/{c}/{dilation}/vp", (c, c, 1), 1 / math.sqrt(c))
    sd = 0.5 + formula_tensor(f"ru/{c}/sd", (c,), 0.4).abs()
    sp = 0.5 + formula_tensor(f"ru/{c}/sp", (c,), 0.4).abs()
    x = formula_tensor(f"ru/{c}/{dilation}/{length}/x", (batch, c, length))
    xin = torch.nn.functional.leaky_relu(x.double(), in_slope)
    wd, wp_ = vd.double() * sd.double().reshape(c, 1, 1), vp.double() * sp.double().reshape(c, 1, 1)
    h_ref = torch.nn.functional.conv1d(torch.nn.functional.pad(xin, (dilation, dilation), mode="reflect"), wd, dilation=dilation)
    u_ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(h_ref, wp_), 0.01)
    y_ref = xin + u_ref
    mm = math_mode
    assert lib.eben_ru_supported(c, dilation, mm) == 1
    img = torch.empty(lib.eben_ru_packed_floats_ex(c, mm), dtype=torch.float32, device=dev)
    assert img.numel() == {0: 4, 1: 2, 3: 4, 4: 6}[mm] * c * c
    vdd, vpd, sdd, spd, xd = vd.to(dev), vp.to(dev), sd.to(dev), sp.to(dev), x.to(dev)
    check(lib.eben_ru_pack_ex(c, mm, 0, ptr(vdd), ptr(sdd), ptr(vpd), ptr(spd), ptr(img), stream()), "ru_pack")
    y, h, u = torch.full_like(xd, float("nan")), torch.full_like(xd, float("nan")), torch.full_like(xd, float("nan"))
    check(lib.eben_ru_fwd_ex(mm, batch, c, length, dilation, ptr(xd), in_slope, 0.01, ptr(img), ptr(y), ptr(h), ptr(u), stream()), "ru_fwd")
    torch.cuda.synchronize()
    tol = RU_TOL[mm]
    assert rel_err(h, h_ref) < tol and rel_err(u, u_ref) < tol and rel_err(y, y_ref) < tol
    if mm == 4:   # three pieces per operand: as close to the fp64 result as the exact-fp32 kernel, not just inside its bound
        img0 = torch.empty(lib.eben_ru_packed_floats(c), dtype=torch.float32, device=dev)
        check(lib.eben_ru_pack(c, ptr(vdd), ptr(sdd), ptr(vpd), ptr(spd), ptr(img0), stream()), "ru_pack")
        y0, h0 = torch.empty_like(xd), torch.empty_like(xd)
        check(lib.eben_ru_fwd(batch, c, length, dilation, ptr(xd), in_slope, 0.01, ptr(img0), ptr(y0), ptr(h0), None, stream()), "ru_fwd")
        torch.cuda.synchronize()
        assert rel_err(h, h_ref) < 2 * rel_err(h0, h_ref) + 1e-7 and rel_err(y, y_ref) < 2 * rel_err(y0, y_ref) + 1e-7
    # inference form: no h / u written; same y bit for bit
    y2 = torch.empty_like(xd)
    check(lib.eben_ru_fwd_ex(mm, batch, c, length, dilation, ptr(xd), in_slope, 0.01, ptr(img), ptr(y2), None, None, stream()), "ru_fwd")
    torch.cuda.synchronize()
    assert torch.equal(y, y2)
    # a 4-byte-aligned (not 16-byte-aligned) input takes the scalar staging path: same values
    buf = torch.empty(xd.numel() + 1, dtype=torch.float32, device=dev)
    xo = buf[1:].view_as(xd)
    xo.copy_(xd)
    y3 = torch.empty_like(xd)
    check(lib.eben_ru_fwd_ex(mm, batch, c, length, dilation, ptr(xo), in_slope, 0.01, ptr(img), ptr(y3), None, None, stream()), "ru_fwd")
    torch.cuda.synchronize()
    assert torch.equal(y, y3)


@pytest.mark.parametrize("channels,dilation,length,batch,in_slope,post", [
    (32, 1, 1000, 3, 1.0, False), (32, 3, 517, 2, 0.01, True), (32, 9, 8000, 2, 1.0, False), (64, 9, 300, 3, 0.01, True),
    (64, 1, 4000, 2, 1.0, False), (128, 3, 1000, 2, 1.0, True), (128, 9, 131, 2, 0.01, False), (32, 9, 20, 1, 1.0, False),
    (32, 9, 110, 1, 1.0, False), (64, 3, 123, 2, 1.0, False), (128, 9, 999, 2, 1.0, False), (128, 1, 47, 1, 1.0, True),
    (128, 9, 999, 32, 1.0, True)])   # BASELINE config 2's widest units: the bf16 launch takes the five-wave window (320 -> 256 blocks)
@pytest.mark.parametrize("math_mode", [0, 4, 3, 1])
def test_fused_residual_unit_backward(hip, channels, dilation, length, batch, in_slope, post, math_mode):
    """eben_ru_bwd_ex (math modes as the forward's): g_h = W_pw^T (g_y * lrelu'(u)) and g_x = (g_y + fold(W_dil^T g_h)) * lrelu'(x) + post in one launch, against
    fp64 autograd of the unit: interior tiles, both reflect folds (also when they land in the same or in neighbouring tiles),
    windows that overhang the signal, clips of exactly one tile and shorter."""
    from vibravox_amd._lib import check, load, ptr, stream

    lib, dev = load(), torch.device("cuda")
    c = channels
    vd = formula_tensor(f"rub/{c}/{dilation}/vd", (c, c, 3), 1 / math.sqrt(3 * c))
    vp = formula_tensor(f"rub/{c}/{dilation}/vp", (c, c, 1), 1 / math.sqrt(c))
    sd = 0.5 + formula_tensor(f"rub/{c}/sd", (c,), 0.4).abs()
    sp = 0.5 + formula_tensor(f"rub/{c}/sp", (c,), 0.4).abs()
    x = formula_tensor(f"rub/{c}/{dilation}/{length}/x", (batch, c, length))
    gy = formula_tensor(f"rub/{c}/{dilation}/{length}/gy", (batch, c, length))
    pt = formula_tensor(f"rub/{c}/{dilation}/{length}/post", (batch, c, length))
    xr = x.double().requires_grad_(True)
    xin = torch.nn.functional.leaky_relu(xr, in_slope)
    wd, wp_ = vd.double() * sd.double().reshape(c, 1, 1), vp.double() * sp.double().reshape(c, 1, 1)
    h = torch.nn.functional.conv1d(torch.nn.functional.pad(xin, (dilation, dilation), mode="reflect"), wd, dilation=dilation)
    h.retain_grad()
    u = torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(h, wp_), 0.01)
    ((xin + u) * gy.double()).sum().backward()
    gx_ref = xr.grad + (pt.double() if post else 0.0)
    mm = math_mode
    img = torch.empty(lib.eben_ru_packed_floats_ex(c, mm), dtype=torch.float32, device=dev)
    vdd, vpd, sdd, spd = vd.to(dev), vp.to(dev), sd.to(dev), sp.to(dev)
    check(lib.eben_ru_pack_ex(c, mm, 1, ptr(vdd), ptr(sdd), ptr(vpd), ptr(spd), ptr(img), stream()), "ru_pack_bwd")
    xd, gyd, ud, ptd = x.to(dev), gy.to(dev), u.detach().float().to(dev), pt.to(dev)
    gx, gh = torch.full_like(xd, float("nan")), torch.full_like(xd, float("nan"))
    check(lib.eben_ru_bwd_ex(mm, batch, c, length, dilation, ptr(gyd), ptr(ud), 0.01, ptr(xd) if in_slope != 1.0 else None, in_slope,
                             ptr(ptd) if post else None, ptr(img), ptr(gx), ptr(gh), stream()), "ru_bwd")
    torch.cuda.synchronize()
    assert rel_err(gh, h.grad) < RU_TOL[mm]
    assert rel_err(gx, gx_ref) < RU_TOL[mm]


@pytest.mark.parametrize("channels,dilation,length,batch,in_slope", [
    (32, 1, 1000, 3, 1.0), (32, 3, 517, 2, 0.01), (32, 9, 2100, 2, 1.0), (64, 9, 300, 3, 0.01), (64, 1, 1999, 2, 1.0),
    (128, 3, 1000, 2, 1.0), (128, 9, 131, 2, 0.01), (32, 9, 20, 1, 1.0), (64, 3, 15, 2, 1.0)])
@pytest.mark.parametrize("math_mode", [4, 1])
def test_fused_residual_unit_weight_gradients(hip, channels, dilation, length, batch, in_slope, math_mode):
    """eben_ru_dw: dW_pw = sum g_z h^T and dW_dil[j] = sum g_h xin(. + (j - 1) d)^T (reflected ends) of one unit in one launch, summed by
    eben_wn_bwd, against fp64 autograd of the weight-normalised unit (dv and dg of both convs): K slabs that end inside a row, rows
    whose length is not a multiple of 4 or 16 (unaligned 16-byte operand loads), clips shorter than one k-step, the fused input
    activation.  math 4: three bf16 pieces per operand (fp32 bound); 1: single bf16 operands."""
    from vibravox_amd._lib import check, load, ptr, stream

    lib, dev = load(), torch.device("cuda")
    c = channels
    vd = formula_tensor(f"rudw/{c}/{dilation}/vd", (c, c, 3), 1 / math.sqrt(3 * c))
    vp = formula_tensor(f"rudw/{c}/{dilation}/vp", (c, c, 1), 1 / math.sqrt(c))
    gd = vd.reshape(c, -1).norm(dim=1).reshape(c, 1, 1) * (1 + 0.3 * formula_tensor(f"rudw/{c}/gd", (c, 1, 1)))
    gp = vp.reshape(c, -1).norm(dim=1).reshape(c, 1, 1) * (1 + 0.3 * formula_tensor(f"rudw/{c}/gp", (c, 1, 1)))
    x = formula_tensor(f"rudw/{c}/{dilation}/{length}/x", (batch, c, length))
    gy = formula_tensor(f"rudw/{c}/{dilation}/{length}/gy", (batch, c, length))
    rvd, rvp, rgd, rgp = (t.double().requires_grad_(True) for t in (vd, vp, gd, gp))
    wd = rvd * (rgd / rvd.flatten(1).norm(dim=1).reshape(c, 1, 1))
    wp_ = rvp * (rgp / rvp.flatten(1).norm(dim=1).reshape(c, 1, 1))
    xin = torch.nn.functional.leaky_relu(x.double(), in_slope)
    h = torch.nn.functional.conv1d(torch.nn.functional.pad(xin, (dilation, dilation), mode="reflect"), wd, dilation=dilation)
    h.retain_grad()
    u = torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(h, wp_), 0.01)
    ((xin + u) * gy.double()).sum().backward()
    mm = math_mode
    nslab = lib.eben_ru_dw_slabs(batch, c, length)
    assert nslab >= batch
    sp = torch.full((nslab * c * c,), float("nan"), dtype=torch.float32, device=dev)
    sdil = torch.full((nslab * c * 3 * c,), float("nan"), dtype=torch.float32, device=dev)
    xd, gyd, ud, hd, ghd = x.to(dev), gy.to(dev), u.detach().float().to(dev), h.detach().float().to(dev), h.grad.float().to(dev)
    check(lib.eben_ru_dw(mm, batch, c, length, dilation, ptr(gyd), ptr(ud), 0.01, ptr(hd), ptr(ghd), ptr(xd), in_slope, ptr(sp), ptr(sdil), stream()), "ru_dw")
    tol = {4: 3e-5, 1: 2e-2}[mm]
    for slabs, cols, v, g, rv, rg in ((sp, c, vp, gp, rvp, rgp), (sdil, 3 * c, vd, gd, rvd, rgd)):
        vdev, gdev = v.to(dev), g.to(dev)
        scale, norm = torch.empty(c, device=dev), torch.empty(c, device=dev)
        check(lib.eben_wn_scale(ptr(gdev), ptr(vdev), c, cols, ptr(scale), ptr(norm), stream()), "wn_scale")
        dv, dg = torch.empty_like(vdev), torch.empty_like(gdev)
        check(lib.eben_wn_bwd(ptr(slabs), nslab, c * cols, c, cols, cols, ptr(gdev), ptr(vdev), ptr(norm), ptr(dg), ptr(dv), None, stream()), "wn_bwd")
        torch.cuda.synchronize()
        assert rel_err(dv, rv.grad) < 2 * tol, (cols, rel_err(dv, rv.grad))
        assert rel_err(dg, rg.grad) < 2 * tol, (cols, rel_err(dg, rg.grad))


def _to_bundles(t: torch.Tensor) -> torch.Tensor:
    """(batch, C, L) fp32 -> bf16 bundle plane [batch][C / 8][L][8] (RNE), include/eben_hip.h "ResidualUnit ... AT REST AS bf16 BUNDLES"."""
    b, c, l = t.shape
    return t.view(b, c // 8, 8, l).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)


def _from_bundles(p: torch.Tensor) -> torch.Tensor:
    b, cb, l, _ = p.shape
    return p.float().permute(0, 1, 3, 2).reshape(b, cb * 8, l).contiguous()


def _sign_plane(t: torch.Tensor) -> torch.Tensor:
    b, c, l = t.shape
    bits = (t.view(b, c // 8, 8, l) > 0).to(torch.int32)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=t.device).view(1, 1, 8, 1)
    return (bits * w).sum(dim=2).to(torch.uint8).contiguous()


@pytest.mark.parametrize("channels,dilation,length,batch,in_slope,post", [
    (32, 1, 1000, 3, 1.0, False), (32, 3, 517, 2, 0.01, True), (32, 9, 8000, 2, 1.0, False), (64, 9, 300, 3, 0.01, True),
    (64, 1, 4000, 2, 1.0, False), (128, 3, 1000, 2, 1.0, True), (128, 9, 131, 2, 0.01, False), (32, 9, 20, 1, 1.0, False),
    (32, 9, 110, 1, 1.0, False), (64, 3, 123, 2, 1.0, False), (128, 9, 999, 2, 1.0, False), (128, 1, 47, 1, 1.0, True),
    (64, 9, 3996, 2, 1.0, False)])
def test_bundle_layout_residual_unit(hip, channels, dilation, length, batch, in_slope, post):
    """The ResidualUnit with its saved tensors at rest as bf16 bundles (csrc/ru_bl.hip; eben_generator.py:287-316 forward and backward):
      * eben_rubl_fwd: y bit-identical to eben_ru_fwd_ex(EBEN_MATH_BF16X6); the planes it saves are exactly bf16(lrelu(x)), bf16(h) of the
        fp32 kernel's h and the sign bits of its u;
      * eben_rubl_bwd: g_x bit-identical to eben_ru_bwd_ex(EBEN_MATH_BF16) (same roundings, same accumulation order) and within its
        tolerance of fp64 autograd; the planes it writes are bf16(g_y lrelu'(u)) and bf16 of that kernel's g_h;
      * eben_rubl_dw + eben_wn_bwd: dv / dg of both convs against fp64 autograd at the bf16 tolerance, and against the fp64 contraction of
        the SAME bf16 operands at fp32 accumulation accuracy (the reduction itself is exact up to summation order).
    Tile interiors, both reflect folds, windows that overhang the signal, K slabs that end inside a chunk, lengths that are not a
    multiple of anything."""
    from vibravox_amd._lib import check, load, ptr, stream

    lib, dev = load(), torch.device("cuda")
    c = channels
    assert lib.eben_rubl_supported(c, dilation) == 1
    vd = formula_tensor(f"rubl/{c}/{dilation}/vd", (c, c, 3), 1 / math.sqrt(3 * c))
    vp = formula_tensor(f"rubl/{c}/{dilation}/vp", (c, c, 1), 1 / math.sqrt(c))
    gd = vd.reshape(c, -1).norm(dim=1).reshape(c, 1, 1) * (1 + 0.3 * formula_tensor(f"rubl/{c}/gd", (c, 1, 1)))
    gp = vp.reshape(c, -1).norm(dim=1).reshape(c, 1, 1) * (1 + 0.3 * formula_tensor(f"rubl/{c}/gp", (c, 1, 1)))
    x = formula_tensor(f"rubl/{c}/{dilation}/{length}/x", (batch, c, length))
    gy = formula_tensor(f"rubl/{c}/{dilation}/{length}/gy", (batch, c, length))
    pt = formula_tensor(f"rubl/{c}/{dilation}/{length}/post", (batch, c, length))
    # fp64 autograd of the weight-normalised unit
    rvd, rvp, rgd, rgp = (t.double().requires_grad_(True) for t in (vd, vp, gd, gp))
    wd = rvd * (rgd / rvd.flatten(1).norm(dim=1).reshape(c, 1, 1))
    wp_ = rvp * (rgp / rvp.flatten(1).norm(dim=1).reshape(c, 1, 1))
    xr = x.double().requires_grad_(True)
    xin = torch.nn.functional.leaky_relu(xr, in_slope)
    h = torch.nn.functional.conv1d(torch.nn.functional.pad(xin, (dilation, dilation), mode="reflect"), wd, dilation=dilation)
    h.retain_grad()
    u = torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(h, wp_), 0.01)
    ((xin + u) * gy.double()).sum().backward()
    gx_ref = xr.grad + (pt.double() if post else 0.0)

    xd, gyd, ptd = x.to(dev), gy.to(dev), pt.to(dev)
    scales = {}
    for name, v, g, cols in (("d", vd, gd, 3 * c), ("p", vp, gp, c)):
        vdev, gdev = v.to(dev), g.to(dev)
        scale, norm = torch.empty(c, device=dev), torch.empty(c, device=dev)
        check(lib.eben_wn_scale(ptr(gdev), ptr(vdev), c, cols, ptr(scale), ptr(norm), stream()), "wn_scale")
        scales[name] = (vdev, gdev, scale, norm)
    img_f = torch.empty(lib.eben_ru_packed_floats_ex(c, 4), dtype=torch.float32, device=dev)
    img_b = torch.empty(lib.eben_ru_packed_floats_ex(c, 1), dtype=torch.float32, device=dev)
    check(lib.eben_ru_pack_ex(c, 4, 0, ptr(scales["d"][0]), ptr(scales["d"][2]), ptr(scales["p"][0]), ptr(scales["p"][2]), ptr(img_f), stream()), "ru_pack")
    check(lib.eben_ru_pack_ex(c, 1, 1, ptr(scales["d"][0]), ptr(scales["d"][2]), ptr(scales["p"][0]), ptr(scales["p"][2]), ptr(img_b), stream()), "ru_pack")

    # ---- forward
    y0, h0, u0 = torch.empty_like(xd), torch.empty_like(xd), torch.empty_like(xd)
    check(lib.eben_ru_fwd_ex(4, batch, c, length, dilation, ptr(xd), in_slope, 0.01, ptr(img_f), ptr(y0), ptr(h0), ptr(u0), stream()), "ru_fwd")
    y1 = torch.full_like(xd, float("nan"))
    xb = torch.full((batch, c // 8, length, 8), float("nan"), dtype=torch.bfloat16, device=dev)
    hb = torch.full_like(xb, float("nan"))
    um = torch.full((batch, c // 8, length), 0xAA, dtype=torch.uint8, device=dev)
    check(lib.eben_rubl_fwd(4, batch, c, length, dilation, ptr(xd), in_slope, 0.01, ptr(img_f), ptr(y1), xb.data_ptr(), hb.data_ptr(), um.data_ptr(), stream()), "rubl_fwd")
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(xb, _to_bundles(torch.nn.functional.leaky_relu(xd, in_slope)))
    assert torch.equal(hb, _to_bundles(h0))
    assert torch.equal(um, _sign_plane(u0))

    # ---- input gradients
    gx0, gh0 = torch.empty_like(xd), torch.empty_like(xd)
    check(lib.eben_ru_bwd_ex(1, batch, c, length, dilation, ptr(gyd), ptr(u0), 0.01, ptr(xd) if in_slope != 1.0 else None, in_slope,
                             ptr(ptd) if post else None, ptr(img_b), ptr(gx0), ptr(gh0), stream()), "ru_bwd")
    gx1 = torch.full_like(xd, float("nan"))
    gzb, ghb = torch.full_like(xb, float("nan")), torch.full_like(xb, float("nan"))
    check(lib.eben_rubl_bwd(batch, c, length, dilation, ptr(gyd), um.data_ptr(), 0.01, ptr(xd) if in_slope != 1.0 else None, in_slope,
                            ptr(ptd) if post else None, ptr(img_b), ptr(gx1), gzb.data_ptr(), ghb.data_ptr(), stream()), "rubl_bwd")
    torch.cuda.synchronize()
    assert torch.equal(gzb, _to_bundles(gyd * torch.where(u0 > 0, 1.0, 0.01)))
    assert torch.equal(ghb, _to_bundles(gh0))
    assert torch.equal(gx0, gx1)
    assert rel_err(gx1, gx_ref) < RU_TOL[1] and rel_err(_from_bundles(ghb), h.grad) < RU_TOL[1]

    # ---- weight gradients
    nslab = lib.eben_rubl_dw_slabs(batch, c, length)
    assert nslab >= batch
    sp = torch.full((nslab * c * c,), float("nan"), dtype=torch.float32, device=dev)
    sdil = torch.full((nslab * c * 3 * c,), float("nan"), dtype=torch.float32, device=dev)
    check(lib.eben_rubl_dw(batch, c, length, dilation, gzb.data_ptr(), hb.data_ptr(), ghb.data_ptr(), xb.data_ptr(), ptr(sp), ptr(sdil), stream()), "rubl_dw")
    torch.cuda.synchronize()
    # the contraction of the same bf16 operands in fp64
    gz64, h64, gh64, x64 = (_from_bundles(t).double().cpu() for t in (gzb, hb, ghb, xb))
    dwp = torch.einsum("bmt,bct->mc", gz64, h64)
    xpad = torch.nn.functional.pad(x64, (dilation, dilation), mode="reflect")
    dwd = torch.stack([torch.einsum("bmt,bct->mc", gh64, xpad[:, :, j * dilation: j * dilation + length]) for j in range(3)], dim=2)
    got_p = sp.view(nslab, c, c).double().sum(0).cpu()
    got_d = sdil.view(nslab, c, c, 3).double().sum(0).cpu()
    assert rel_err(got_p, dwp) < 2e-6 and rel_err(got_d, dwd) < 2e-6, (rel_err(got_p, dwp), rel_err(got_d, dwd))
    for slabs, cols, key, rv, rg in ((sp, c, "p", rvp, rgp), (sdil, 3 * c, "d", rvd, rgd)):
        vdev, gdev, _, norm = scales[key]
        dv, dg = torch.empty_like(vdev), torch.empty_like(gdev)
        check(lib.eben_wn_bwd(ptr(slabs), nslab, c * cols, c, cols, cols, ptr(gdev), ptr(vdev), ptr(norm), ptr(dg), ptr(dv), None, stream()), "wn_bwd")
        torch.cuda.synchronize()
        assert rel_err(dv, rv.grad) < 4e-2 and rel_err(dg, rg.grad) < 4e-2, (cols, rel_err(dv, rv.grad), rel_err(dg, rg.grad))


def test_fused_residual_unit_rejects_unsupported_shapes(hip):
    from vibravox_amd._lib import load

    lib = load()
    assert lib.eben_ru_packed_floats(48) == 0
    assert lib.eben_ru_fwd(1, 48, 100, 1, None, 1.0, 0.01, None, None, None, None, None) < 0     # channels
    assert lib.eben_ru_fwd(1, 32, 5, 9, None, 1.0, 0.01, None, None, None, None, None) < 0       # reflect pad >= length
    assert lib.eben_ru_packed_floats_ex(48, 4) == 0 and lib.eben_ru_packed_floats_ex(64, 9) == 0
    assert lib.eben_ru_supported(64, 12, 4) == 0 and lib.eben_ru_supported(64, 12, 0) == 1   # split kernels: dilation <= 9
    assert lib.eben_ru_fwd_ex(4, 1, 64, 100, 12, None, 1.0, 0.01, None, None, None, None, None) < 0
    assert lib.eben_ru_bwd_ex(7, 1, 64, 100, 3, None, None, 0.01, None, 1.0, None, None, None, None, None) < 0   # unknown math
    assert lib.eben_ru_dw_slabs(2, 48, 100) == 0
    assert lib.eben_ru_dw(0, 2, 64, 100, 3, None, None, 0.01, None, None, None, 1.0, None, None, None) < 0         # fp32 MFMA form: not built


def test_input_gradient_with_residual_joins(hip):
    """eben_conv1d_bwd_dx_res: dx = (conv^T(dy * lrelu_out'(y)) + res_pre) * lrelu_in'(x) + res_post on a reflect-padded dilated
    conv (the ResidualUnit's gradient join) and res_pre alone on a pointwise conv (no fold pass)."""
    import ctypes

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib, dev = load(), torch.device("cuda")
    for name, kw, use_post in (("dil", dict(c_in=32, c_out=32, ksize=3, dilation=3, pad_l=3, pad_r=3, reflect=True, in_slope=0.01), True),
                               ("pw", dict(c_in=64, c_out=64, ksize=1, out_slope=0.01), False)):
        spec = ops.ConvSpec(**kw)
        b, l = 2, 333
        w = formula_tensor(f"dxr/{name}/w", spec.weight_shape(), 0.1)
        x, dy = formula_tensor(f"dxr/{name}/x", (b, spec.c_in, l)), formula_tensor(f"dxr/{name}/dy", (b, spec.c_out, l))
        pre, post = formula_tensor(f"dxr/{name}/pre", (b, spec.c_in, l)), formula_tensor(f"dxr/{name}/post", (b, spec.c_in, l))
        yv = formula_tensor(f"dxr/{name}/y", (b, spec.c_out, l))
        okw = {k: v for k, v in kw.items() if k not in ("c_in", "c_out", "ksize", "in_slope", "out_slope")}
        xr = torch.zeros(b, spec.c_in, l, dtype=torch.float64, requires_grad=True)
        masked = dy.double() * torch.where(yv.double() > 0, 1.0, spec.out_slope)
        (O.conv_layer(xr, w.double(), None, None, **okw) * masked).sum().backward()
        ref = (xr.grad + pre.double()) * torch.where(x.double() > 0, 1.0, spec.in_slope)
        if use_post:
            ref = ref + post.double()
        d = ops.conv_desc(spec, b, l)
        wd = w.to(dev)
        wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 1), dtype=torch.float32, device=dev)
        check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, None, ptr(wp), stream()), "pack")
        ws_bytes = lib.eben_conv1d_bwd_dx_workspace(ctypes.byref(d))
        ws = torch.empty(max(1, ws_bytes // 4), dtype=torch.float32, device=dev)
        xd, dyd, pred, postd, yd = x.to(dev), dy.to(dev), pre.to(dev), post.to(dev), yv.to(dev)
        dx = torch.empty_like(xd)
        check(lib.eben_conv1d_bwd_dx_res(ctypes.byref(d), ptr(dyd), ptr(yd) if spec.out_slope != 1.0 else None, ptr(wp),
                                         ptr(xd) if spec.in_slope != 1.0 else None, ptr(pred), ptr(postd) if use_post else None, ptr(dx),
                                         ptr(ws), ws_bytes, stream()), "bwd_dx_res")
        torch.cuda.synchronize()
        assert rel_err(dx, ref) < 3e-5
        if not use_post:   # an addend behind the mask needs the fold pass
            assert lib.eben_conv1d_bwd_dx_res(ctypes.byref(d), ptr(dyd), ptr(yd), ptr(wp), None, None, ptr(postd), ptr(dx), None, 0, stream()) < 0


@pytest.mark.parametrize("name", list(BF16_CASES))
def test_bf16x2_math_keeps_the_activation_operand(hip, name):
    """EBEN_MATH_BF16X2: the activation operand enters the forward and the weight gradient as hi + lo (two bf16 MFMAs per
    k-step), weights / gradients as single bf16.  Against an fp64 conv of the EXACT x and the bf16-rounded other operand
    the kernels must be as tight as the fp32 ones (x = hi + lo to 2^-17 per element); the input gradient is EBEN_MATH_BF16's."""
    import ctypes
    import dataclasses

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    kw, length, _ = BF16_CASES[name]
    spec = ops.ConvSpec(**kw)
    wshape = spec.weight_shape()
    w = formula_tensor(f"bf/{name}/w", wshape, 1 / math.sqrt(wshape[1] * wshape[2]))
    bias = formula_tensor(f"bf/{name}/b", (spec.c_out,), 0.1)
    # a large common component plus a small item-dependent one: what the discriminator's activations look like, and the
    # case single-bf16 rounding gets wrong (the item-dependent part sits below the 2^-9 grid of the common one)
    common = formula_tensor(f"x2/{name}/c", (1, spec.c_in, length))
    x = common + 2e-3 * formula_tensor(f"x2/{name}/x", (4, spec.c_in, length))
    l_out = spec.out_len(length)
    okw = {k: v for k, v in kw.items() if k not in ("c_in", "c_out", "ksize", "in_slope", "out_slope")}
    dev = torch.device("cuda")
    wd, bd, xd = w.to(dev), bias.to(dev), x.to(dev)
    d = ops.conv_desc(spec, 4, length, ops.MATH_BF16X2)
    fgen = lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0)
    rw = _bf16 if fgen == 4 else (lambda t: t.double())
    xin = torch.nn.functional.leaky_relu(x.double(), spec.in_slope)
    ref = torch.nn.functional.leaky_relu(O.conv_layer(xin, rw(w), None, bias.double(), **{**okw}), spec.out_slope) if spec.in_slope == 1.0 else None
    wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 0), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, ptr(wp), None, stream()), "pack")
    y = torch.empty(4, spec.c_out, l_out, dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xd), ptr(wp), ptr(bd), None, ptr(y), stream()), "fwd")
    torch.cuda.synchronize()
    if ref is not None:
        assert rel_err(y, ref) < 3e-5
        # the item-dependent part of the output (differences between items) survives: single bf16 loses it
        diff, rdiff = (y[1] - y[0]).double().cpu(), ref[1] - ref[0]
        assert float((diff - rdiff).norm() / rdiff.norm()) < 2e-3
        if fgen == 4:
            d1 = ops.conv_desc(spec, 4, length, ops.MATH_BF16)
            wp1 = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d1), 0), dtype=torch.float32, device=dev)
            check(lib.eben_conv1d_pack(ctypes.byref(d1), ptr(wd), None, ptr(wp1), None, stream()), "pack")
            y1 = torch.empty_like(y)
            check(lib.eben_conv1d_fwd(ctypes.byref(d1), ptr(xd), ptr(wp1), ptr(bd), None, ptr(y1), stream()), "fwd")
            torch.cuda.synchronize()
            d1v = (y1[1] - y1[0]).double().cpu()
            assert float((d1v - rdiff).norm() / rdiff.norm()) > 10 * float((diff - rdiff).norm() / rdiff.norm())

    # weight / bias gradient: x exact, gradient operand rounded
    lin = dataclasses.replace(spec, in_slope=1.0, out_slope=1.0)
    nb = 24
    d = ops.conv_desc(lin, nb, length, ops.MATH_BF16X2)
    xb = common + 2e-3 * formula_tensor(f"x2/{name}/xb", (nb, spec.c_in, length))
    gb = formula_tensor(f"bf/{name}/gb", (nb, spec.c_out, l_out))
    nslab, row_stride = ctypes.c_int(0), ctypes.c_int(0)
    ws_bytes = lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride))
    wr = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    (O.conv_layer(xb.double(), wr, None, None, **okw) * gb.double()).sum().backward()
    exact = wr.grad.clone()
    wr.grad = None
    (O.conv_layer(xb.double(), wr, None, None, **okw) * _bf16(gb)).sum().backward()
    slabs = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=dev)
    xbd, gbd = xb.to(dev), gb.to(dev)
    check(lib.eben_conv1d_bwd_dw(ctypes.byref(d), ptr(gbd), None, ptr(xbd), 1, ptr(slabs), ws_bytes, stream()), "bwd_dw")
    rows, cols = wshape[0], wshape[1] * wshape[2]
    dv = torch.empty(wshape, dtype=torch.float32, device=dev)
    dbias = torch.empty(rows, dtype=torch.float32, device=dev)
    check(lib.eben_wn_bwd(ptr(slabs), nslab.value, rows * row_stride.value, rows, cols, row_stride.value, None, None, None, None,
                          ptr(dv), ptr(dbias), stream()), "wn_bwd")
    torch.cuda.synchronize()
    # either the bf16 kernel with the split X operand (== fp64 of exact x, rounded g) or the fp32 fallback (== exact)
    assert min(rel_err(dv, wr.grad), rel_err(dv, exact)) < 1e-4


@pytest.mark.parametrize("name", ["pqmf_disc_wide", "melgan_l2_like", "thin_pqmf_l1", "dense_k5_chunks"])
def test_batched_input_gradient_ex(hip, name):
    """eben_conv1d_bwd_dx_ex: four stacked right-hand sides [fm | adv | fake | real] against activations
    [enhanced | reference]: dx[b] = (conv^T(g[b]) + (b < S ? res[b] : 0)) * lrelu'(act[map(b)])."""
    import ctypes
    import dataclasses

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    kw, _, length, _, _ = CONV_CASES[name]
    spec = dataclasses.replace(ops.ConvSpec(**kw), in_slope=1.0, out_slope=1.0)
    S = 2
    wshape = spec.weight_shape()
    w = formula_tensor(f"ex/{name}/w", wshape, 1 / math.sqrt(wshape[1] * wshape[2]))
    l_out = spec.out_len(length)
    g = formula_tensor(f"ex/{name}/g", (4 * S, spec.c_out, l_out))
    act = formula_tensor(f"ex/{name}/act", (2 * S, spec.c_in, length))
    res = formula_tensor(f"ex/{name}/res", (S, spec.c_in, length))
    # fp64 reference: the adjoint of the (linear) conv applied to every row
    xr = torch.zeros(4 * S, spec.c_in, length, dtype=torch.float64, requires_grad=True)
    okw = {k: v for k, v in kw.items() if k not in ("c_in", "c_out", "ksize", "in_slope", "out_slope")}
    (O.conv_layer(xr, w.double(), None, None, **okw) * g.double()).sum().backward()
    ref = xr.grad.clone()
    ref[:S] += res.double()
    rows = torch.tensor([0, 1, 0, 1, 0, 1, 2, 3])
    ref = ref * torch.where(act.double()[rows] > 0, 1.0, 0.2)

    dev = torch.device("cuda")
    d = ops.conv_desc(spec, 4 * S, length)
    wd = w.to(dev)
    wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 1), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, None, ptr(wp), stream()), "pack")
    dx = torch.empty(4 * S, spec.c_in, length, dtype=torch.float32, device=dev)
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
    gd, rd, ad = g.to(dev), res.to(dev), act.to(dev)   # keep them alive: ptr() of a temporary dangles once it is freed
    check(lib.eben_conv1d_bwd_dx_ex(ctypes.byref(d), ptr(gd), ptr(wp), ptr(rd), S, ptr(ad), 0.2, S, seg_map, ptr(dx), stream()),
          "bwd_dx_ex")
    torch.cuda.synchronize()
    assert rel_err(dx, ref) < 3e-5


@pytest.mark.parametrize("name,mathmode", [("thin_pqmf_l1", 0), ("pqmf_disc_wide", 1), ("melgan_l2_like", 1), ("dense_k5_chunks", 1),
                                           ("pqmf_disc_wide", 0)])
def test_batched_input_gradient_with_feature_matching_epilogue(hip, name, mathmode):
    """eben_conv1d_bwd_dx_fm: the feature-matching gradient of the embedding pair (feature_loss.py:40-47) formed in the
    input-gradient epilogue == eben_fm_bwd into a buffer + eben_conv1d_bwd_dx_ex reading it, bit for bit, and == fp64; kernel
    generations without that epilogue refuse."""
    import ctypes
    import dataclasses

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    kw, _, length, _, _ = CONV_CASES[name]
    spec = dataclasses.replace(ops.ConvSpec(**kw), in_slope=1.0, out_slope=1.0)
    S = 2
    wshape = spec.weight_shape()
    w = formula_tensor(f"fmx/{name}/w", wshape, 1 / math.sqrt(wshape[1] * wshape[2]))
    l_out = spec.out_len(length)
    g = formula_tensor(f"fmx/{name}/g", (4 * S, spec.c_out, l_out))
    act = formula_tensor(f"fmx/{name}/act", (2 * S, spec.c_in, length))
    dev = torch.device("cuda")
    d = ops.conv_desc(spec, 4 * S, length, mathmode)
    gen = lib.eben_conv1d_kernel_generation(ctypes.byref(d), 1)
    wd, gd, ad = w.to(dev), g.to(dev), act.to(dev)
    wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 1), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_pack(ctypes.byref(d), ptr(wd), None, None, ptr(wp), stream()), "pack")
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
    # sums of the pair (s1 = sum|a - b|, s2 = sum|a|) as eben_fm_sums leaves them, and the gradient buffer of the two-kernel form
    a, b = ad[:S], ad[S:]
    sums = torch.stack(((a - b).abs().sum(), a.abs().sum())).to(torch.float32).contiguous()
    gs = 1.0 / 7.0
    one = torch.ones(1, dtype=torch.float32, device=dev)
    da = torch.empty_like(a)
    pairs = (ctypes.c_void_p * 2)(ptr(a), ptr(b))
    outs = (ctypes.c_void_p * 1)(ptr(da))
    numel = (ctypes.c_int64 * 1)(a.numel())
    check(lib.eben_fm_bwd(pairs, outs, numel, 1, ptr(sums), ptr(one), gs, stream()), "fm_bwd")
    two = torch.empty(4 * S, spec.c_in, length, dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_bwd_dx_ex(ctypes.byref(d), ptr(gd), ptr(wp), ptr(da), S, ptr(ad), 0.2, S, seg_map, ptr(two), stream()), "bwd_dx_ex")
    one_pass = torch.empty_like(two)
    rc = lib.eben_conv1d_bwd_dx_fm(ctypes.byref(d), ptr(gd), ptr(wp), ptr(b), S, ptr(sums), gs, ptr(ad), 0.2, S, seg_map, ptr(one_pass), stream())
    if gen not in (3, 4):
        assert rc == -3   # EBEN_EUNSUPPORTED
        return
    check(rc, "bwd_dx_fm")
    torch.cuda.synchronize()
    assert torch.equal(one_pass, two)
    # fp64 of the same expression
    xr = torch.zeros(4 * S, spec.c_in, length, dtype=torch.float64, requires_grad=True)
    okw = {k: v for k, v in kw.items() if k not in ("c_in", "c_out", "ksize", "in_slope", "out_slope")}
    (O.conv_layer(xr, w.double(), None, None, **okw) * g.double()).sum().backward()
    ref = xr.grad.clone()
    a64, b64 = act.double()[:S], act.double()[S:]
    s1, s2 = (a64 - b64).abs().sum(), a64.abs().sum()
    ref[:S] += gs * (torch.sign(a64 - b64) / s2 - s1 * torch.sign(a64) / s2 ** 2)
    rows = torch.tensor([0, 1, 0, 1, 0, 1, 2, 3])
    ref = ref * torch.where(act.double()[rows] > 0, 1.0, 0.2)
    assert rel_err(one_pass, ref) < (3e-5 if mathmode == 0 else 2e-2)


@pytest.mark.parametrize("reflect,masked", [(True, False), (False, True), (False, False)])
def test_space_to_depth_turns_a_strided_conv_into_a_stride_one_conv(hip, reflect, masked):
    """eben_space_to_depth: out[row][r][q] = xp[row][S q + r + off] (bit-exact gather, reflection / zeros beyond the signal, optional
    lrelu' mask), and the stride-8 k-16 conv of EncBlock (eben_generator.py:251-254) == the two-tap stride-1 conv over the C*S
    channels of `out` with the weights viewed (M, C, 2, 8) -> (M, C, 8, 2)."""
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    b, c, m, length, s_, k, pad = 2, 6, 5, 203, 8, 16, 7 if reflect else 4
    x = formula_tensor(f"s2d/x/{reflect}", (b, c, length))
    mk = formula_tensor(f"s2d/m/{reflect}", (b, c, length))
    w = formula_tensor(f"s2d/w/{reflect}", (m, c, k), 0.1)
    l_out = (length + 2 * pad - k) // s_ + 1
    l_q = l_out + k // s_ - 1
    dev = torch.device("cuda")
    xd, md = x.to(dev), mk.to(dev)
    out = torch.empty((b, c * s_, l_q), dtype=torch.float32, device=dev)
    check(lib.eben_space_to_depth(ptr(xd), ptr(md) if masked else None, 0.2, ptr(out), b * c, length, s_, -pad, l_q, 1 if reflect else 0,
                                  stream()), "space_to_depth")
    torch.cuda.synchronize()
    src = x * torch.where(mk > 0, 1.0, 0.2) if masked else x
    if reflect:
        xp = torch.nn.functional.pad(src, (pad, pad), mode="reflect")
        xp = torch.nn.functional.pad(xp, (0, max(0, s_ * l_q - xp.shape[-1])))
    else:
        xp = torch.nn.functional.pad(src, (pad, max(0, s_ * l_q - length - pad)))
    ref = xp[..., :s_ * l_q].reshape(b, c, l_q, s_).permute(0, 1, 3, 2).reshape(b, c * s_, l_q)
    got = out.cpu()
    n_ok = (length + 2 * pad) // s_ if reflect else l_q   # reflect side: the positions inside the padded signal (all the conv reads)
    assert torch.equal(got[..., :n_ok], ref[..., :n_ok])
    # the conv identity (fp64 on the CPU)
    y_ref = torch.nn.functional.conv1d(torch.nn.functional.pad(src.double(), (pad, pad), mode="reflect" if reflect else "constant"), w.double(), stride=s_)
    w2 = w.double().view(m, c, k // s_, s_).permute(0, 1, 3, 2).reshape(m, c * s_, k // s_)
    y_alt = torch.nn.functional.conv1d(got.double(), w2)
    assert y_alt.shape == y_ref.shape and rel_err(y_alt, y_ref) < 1e-12


@pytest.mark.parametrize("mathmode,tol", [(1, 2e-2), (3, 1e-4), (4, 3e-5)])
@pytest.mark.parametrize("g,m,k,n", [(2, 257, 120, 1000), (2, 120, 257, 333), (1, 33, 16, 129), (2, 513, 300, 4100), (3, 31, 50, 64)])
def test_grouped_gemm_on_split_bf16_operands(hip, g, m, k, n, mathmode, tol):
    """eben_gemm_fwd (the folded windowed-DFT contraction of the MRSTFT loss and its transpose): y[g] = W[g] x[g] against fp64, in
    the three operand splits; ragged M / K / N (tiles of 64 rows, 16 reduction rows, 128 columns)."""
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    w = formula_tensor(f"gemm/w/{g}/{m}/{k}", (g * m, k), 1 / math.sqrt(k))
    x = formula_tensor(f"gemm/x/{g}/{k}/{n}", (g * k, n))
    dev = torch.device("cuda")
    wd, xd = w.to(dev), x.to(dev)
    wp = torch.empty(lib.eben_gemm_packed_floats(mathmode, g, m, k), dtype=torch.float32, device=dev)
    check(lib.eben_gemm_pack(mathmode, g, m, k, ptr(wd), ptr(wp), stream()), "gemm_pack")
    y = torch.full((g * m, n), float("nan"), dtype=torch.float32, device=dev)
    check(lib.eben_gemm_fwd(mathmode, g, m, k, n, ptr(xd), ptr(wp), ptr(y), stream()), "gemm_fwd")
    torch.cuda.synchronize()
    ref = torch.bmm(w.double().view(g, m, k), x.double().view(g, k, n)).reshape(g * m, n)
    assert rel_err(y, ref) < tol
    assert lib.eben_gemm_fwd(0, g, m, k, n, ptr(xd), ptr(wp), ptr(y), stream()) != 0   # exact fp32 is not a mode of this kernel


@pytest.mark.parametrize("batch,length,n", [(3, 1000, 3), (2, 8000, 3), (1, 257, 1), (2, 2047, 4), (4, 37, 2)])
def test_last_conv_gradient_norms_in_one_pass(hip, batch, length, n):
    """ops.last_conv_grad_norms (eben_last_conv_norms): the balancing norms ||dL_i / d last_conv.weight|| (eben.py:222-229) from the seeds
    dL_i / d bands in one pass -- against autograd through tanh_lift + the HIP conv, and against float64: position ranges that end inside a
    chunk, a clip shorter than one chunk, 1..4 losses."""
    from vibravox_amd import ops
    from vibravox_amd.torch_modules.utils import HipConv1d

    dev = torch.device("cuda")
    conv = HipConv1d(32, 4, 3, padding="same", bias=False, padding_mode="reflect", weight_norm=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(formula_tensor("lcn/w", (4, 32, 3), 0.2))
    pre = formula_tensor(f"lcn/{batch}/{length}/pre", (batch, 32, length)).to(dev)
    lift = formula_tensor(f"lcn/{batch}/{length}/lift", (batch, 2, length)).to(dev)
    seeds = [formula_tensor(f"lcn/{batch}/{length}/s{i}", (batch, 4, length), 10.0 ** (i - 1)).to(dev) for i in range(n)]
    bands = ops.tanh_lift(conv(pre), lift)
    want = [torch.norm(torch.autograd.grad(bands, conv.weight, grad_outputs=s, retain_graph=True)[0]) for s in seeds]
    got = ops.last_conv_grad_norms(seeds, bands, pre, conv)
    assert got is not None and len(got) == n
    # float64: dW = sum s (1 - bands^2) (*) reflect_pad(pre)
    bd, pd = bands.detach().double().cpu(), torch.nn.functional.pad(pre.double().cpu(), (1, 1), mode="reflect")
    for i in range(n):
        g = seeds[i].double().cpu() * (1 - bd * bd)
        dw = torch.stack([torch.einsum("bot,bct->oc", g, pd[:, :, j:j + length]) for j in range(3)], dim=2)
        ref = float(dw.norm())
        assert abs(float(got[i]) - ref) <= 2e-5 * ref, (i, float(got[i]), ref)
        assert abs(float(want[i]) - ref) <= 1e-4 * ref
    # a conv the kernel is not built for: the caller falls back to autograd
    other = HipConv1d(32, 4, 5, padding="same", bias=False, padding_mode="reflect", weight_norm=False).to(dev)
    assert ops.last_conv_grad_norms(seeds, bands, pre, other) is None


@pytest.mark.parametrize("mode,first", [("ema", True), ("ema", False), ("simple", False)])
def test_fused_balancing_matches_the_torch_arithmetic_bit_for_bit(hip, mode, first):
    """eben_balance + eben_weighted_sum == the one-element torch kernels of EBENLightningModule._update_lambdas and the weighted seed
    (eben.py:229-240): EMA with the first-call quirk, clamp(1 / (ema + 1e-4), 0, 1e4), sum loss_i lambda_i, sum s_i lambda_i."""
    import ctypes

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib = load()
    dev = torch.device("cuda")
    n, beta = 3, 0.9
    norms = [formula_tensor(f"bal/n/{mode}/{i}", (1,), 1.0).abs().reshape(()).to(dev) * (10.0 ** (i - 1)) + 1e-3 for i in range(n)]
    norms[2] = norms[2] * 0 + 3e-5   # lambda = 1 / 1.3e-4 = 7692: below the clamp; and one above it
    losses = [formula_tensor(f"bal/l/{mode}/{i}", (1,), 1.0).reshape(()).to(dev) for i in range(n)]
    old0 = [formula_tensor(f"bal/o/{mode}/{i}", (1,), 1.0).abs().reshape(()).to(dev) + 0.5 for i in range(n)]
    seeds = [formula_tensor(f"bal/s/{mode}/{i}", (2, 4, 1003)).to(dev) for i in range(n)]
    # torch, as _update_lambdas writes it
    old = list(norms) if (first or mode == "simple") else list(old0)
    if mode == "ema":
        old = [beta * o + (1 - beta) * nv for o, nv in zip(old, norms)]
    lam = [torch.clamp(1 / (o + 1e-4), min=0.0, max=1e4) for o in old]
    bp = sum(l * w for l, w in zip(losses, lam))
    seed = None
    for s_, w in zip(seeds, lam):
        seed = s_ * w if seed is None else seed + s_ * w
    # fused
    state = torch.stack(old0).contiguous()
    out = torch.empty(n + 1, dtype=torch.float32, device=dev)
    check(lib.eben_balance((ctypes.c_void_p * n)(*[ptr(t) for t in norms]), (ctypes.c_void_p * n)(*[ptr(t) for t in losses]), n, ptr(state),
                           1 if (first or mode == "simple") else 0, 1 if mode == "ema" else 0, beta, 1 - beta, ptr(out), ptr(out[n:]), stream()),
          "balance")
    got_seed = ops.weighted_sum(seeds, out[:n])
    torch.cuda.synchronize()
    assert torch.equal(state, torch.stack(old))
    assert torch.equal(out[:n], torch.stack(lam))
    assert torch.equal(out[n], bp)
    assert torch.equal(got_seed, seed)


def test_conv_bad_descriptor_raises(hip):
    from vibravox_amd import _lib, ops

    spec = ops.ConvSpec(c_in=8, c_out=8, ksize=3, pad_l=1, pad_r=1)
    x = torch.zeros(1, 4, 10, device="cuda")
    v = torch.zeros(8, 8, 3, device="cuda")
    with pytest.raises(_lib.EbenError):
        ops.conv_layer(x, v, None, None, spec)
    with pytest.raises(_lib.EbenError):
        ops.conv_layer(torch.zeros(1, 8, 10), v, None, None, spec)  # CPU tensor: no fallback


@pytest.mark.parametrize("bands,length", [(4, 8160), (2, 1000), (4, 224), (1, 37)])
def test_pqmf_analysis_synthesis(hip, golden, bands, length):
    from vibravox_amd import ops

    ana = torch.from_numpy(golden["pqmf/analysis_4_32"])
    syn = torch.from_numpy(golden["pqmf/synthesis_4_32"])
    x = formula_tensor(f"pq/{length}", (3, 1, length))
    ref = O.pqmf_analysis(x.double(), ana.double(), bands=bands)
    dev = torch.device("cuda")
    xd = x.to(dev).requires_grad_(True)
    got = ops.fir_decimate(xd, ana[:bands].reshape(bands, 32).to(dev), ref.shape[2], 4, -31)
    assert rel_err(got, ref) < 1e-5
    # adjoint: <A x, s> gradient equals interp_sum
    s = formula_tensor(f"pq/s/{length}", tuple(ref.shape))
    (got * s.to(dev)).sum().backward()
    rx = x.double().requires_grad_(True)
    (O.pqmf_analysis(rx, ana.double(), bands=bands) * s.double()).sum().backward()
    assert rel_err(xd.grad, rx.grad) < 1e-5
    if bands == 4 and length > 64:
        b4 = formula_tensor(f"pq/b/{length}", (3, 4, ref.shape[2])).requires_grad_(True)
        rsyn = O.pqmf_synthesis(b4.double(), syn.double()).sum(1, keepdim=True)
        bd = b4.detach().to(dev).requires_grad_(True)
        gsyn = ops.fir_interp_sum(bd, syn.reshape(4, 32).to(dev), rsyn.shape[2], 4, -31)
        assert rel_err(gsyn, rsyn) < 1e-5
        w = formula_tensor(f"pq/w/{length}", tuple(rsyn.shape))
        (gsyn * w.to(dev)).sum().backward()
        (rsyn * w.double()).sum().backward()
        assert rel_err(bd.grad, b4.grad) < 1e-5


def test_pqmf_roundtrip_snr_on_device(hip, golden):
    """size-independent property at full length: analysis -> synthesis reconstructs (> 50 dB)."""
    from vibravox_amd import ops

    dev = torch.device("cuda")
    ana = torch.from_numpy(golden["pqmf/analysis_4_32"]).reshape(4, 32).to(dev)
    syn = torch.from_numpy(golden["pqmf/synthesis_4_32"]).reshape(4, 32).to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(32, 1, 31968, generator=g).to(dev)
    bands = ops.fir_decimate(x, ana, 8000, 4, -31)
    rec = ops.fir_interp_sum(bands, syn, 31968, 4, -31)
    snr = 10 * torch.log10((rec ** 2).mean() / ((x - rec) ** 2).mean()).item()
    assert snr > 50.0


def test_elementwise_and_pad(hip):
    from vibravox_amd import ops

    dev = torch.device("cuda")
    x = formula_tensor("ew/x", (3, 5, 1001))
    y = formula_tensor("ew/y", (3, 5, 1001))
    s = formula_tensor("ew/s", (3, 5, 1001))
    for slope in (0.01, 0.2):
        xd = x.to(dev).requires_grad_(True)
        out = ops.leaky_relu(xd, slope)
        (out * s.to(dev)).sum().backward()
        rx = x.double().requires_grad_(True)
        ro = torch.nn.functional.leaky_relu(rx, slope)
        (ro * s.double()).sum().backward()
        assert rel_err(out, ro) < 1e-6 and rel_err(xd.grad, rx.grad) < 1e-6
    assert rel_err(ops.add(x.to(dev), y.to(dev)), x.double() + y.double()) < 1e-6
    lift = formula_tensor("ew/l", (3, 2, 1001))
    xd = x.to(dev).requires_grad_(True)
    out = ops.tanh_lift(xd, lift.to(dev))
    (out * s.to(dev)).sum().backward()
    rx = x.double().requires_grad_(True)
    ro = torch.tanh(rx + torch.cat((lift.double(), torch.zeros(3, 3, 1001, dtype=torch.float64)), 1))
    (ro * s.double()).sum().backward()
    assert rel_err(out, ro) < 1e-5 and rel_err(xd.grad, rx.grad) < 1e-5
    for pad in (1, 7):
        xd = x.to(dev).requires_grad_(True)
        out = ops.reflect_pad(xd, pad, pad)
        sp = formula_tensor(f"ew/p{pad}", tuple(out.shape))
        (out * sp.to(dev)).sum().backward()
        rx = x.double().requires_grad_(True)
        ro = torch.nn.functional.pad(rx, (pad, pad), mode="reflect")
        (ro * sp.double()).sum().backward()
        assert rel_err(out, ro) == 0.0 and rel_err(xd.grad, rx.grad) < 1e-6


def test_feature_and_hinge_losses(hip):
    from vibravox_amd.torch_modules.losses.feature_loss import FeatureLossForDiscriminatorMelganMultiScales
    from vibravox_amd.torch_modules.losses.hinge_loss import HingeLossForDiscriminatorMelganMultiScales

    dev = torch.device("cuda")
    shapes = [[(2, 4, 300), (2, 24, 302), (2, 48, 151), (2, 1, 151)], [(2, 1, 1000), (2, 16, 1000), (2, 64, 250), (2, 256, 63), (2, 1, 63)]]
    ea = [[formula_tensor(f"fl/a{i}{j}", s) for j, s in enumerate(sc)] for i, sc in enumerate(shapes)]
    eb = [[formula_tensor(f"fl/b{i}{j}", s) for j, s in enumerate(sc)] for i, sc in enumerate(shapes)]
    ra = [[t.double().requires_grad_(True) for t in sc] for sc in ea]
    rb = [[t.double() for t in sc] for sc in eb]
    r_fm, r_hp, r_hm = O.feature_loss(ra, rb), O.hinge_loss(ra, 1), O.hinge_loss(ra, -1)
    (r_fm * 0.7 + r_hp * 1.3 + r_hm * 0.4).backward()
    da = [[t.to(dev).requires_grad_(True) for t in sc] for sc in ea]
    db = [[t.to(dev) for t in sc] for sc in eb]
    fm, hinge = FeatureLossForDiscriminatorMelganMultiScales(), HingeLossForDiscriminatorMelganMultiScales()
    g_fm, g_hp, g_hm = fm(da, db), hinge(embeddings=da, target=1), hinge(embeddings=da, target=-1)
    assert g_fm.dim() == 0 and g_hp.dim() == 0  # reference tests: loss is a 0-dim tensor
    (g_fm * 0.7 + g_hp * 1.3 + g_hm * 0.4).backward()
    np.testing.assert_allclose(g_fm.item(), r_fm.item(), rtol=1e-5)
    np.testing.assert_allclose(g_hp.item(), r_hp.item(), rtol=1e-5)
    np.testing.assert_allclose(g_hm.item(), r_hm.item(), rtol=1e-5)
    for sa, sr in zip(da, ra):
        for i, (t, r) in enumerate(zip(sa, sr)):
            if r.grad is None:
                assert t.grad is None
            else:
                assert rel_err(t.grad, r.grad) < 1e-5


def test_folded_framing_split_and_adjoint(hip):
    """eben_stft_frames_folded against the dense framing (E / O parts about the window centre), its bf16x3 form
    (hi + lo exact in bf16, hi + lo == v to 2^-17), eben_split3, and eben_overlap_add_folded as its exact adjoint."""
    from vibravox_amd._lib import load
    from vibravox_amd.ops import ptr, stream

    lib, dev = load(), torch.device("cuda")
    rows, t, win, hop = 3, 1000, 240, 50
    h, pad = win // 2, win // 2
    frames = (t + 2 * pad - win) // hop + 1
    cols = rows * frames
    g = torch.Generator().manual_seed(5)
    sig = torch.randn(rows, t, generator=g).to(dev)
    dense = torch.empty(win, cols, device=dev)
    assert lib.eben_stft_frames(ptr(sig), ptr(dense), rows, t, win, hop, pad, frames, stream()) == 0
    fold = torch.empty(2 * h, cols, device=dev)
    assert lib.eben_stft_frames_folded(ptr(sig), ptr(fold), rows, t, win, hop, pad, frames, 0, stream()) == 0
    m = torch.arange(1, h, device=dev)
    want_e = torch.cat((dense[h:h + 1], dense[h + m] + dense[h - m]))
    want_o = torch.cat((torch.zeros_like(dense[:1]), dense[h + m] - dense[h - m]))
    assert torch.equal(fold[:h], want_e) and torch.equal(fold[h:], want_o)

    sp = torch.empty(6 * h, cols, device=dev)
    assert lib.eben_stft_frames_folded(ptr(sig), ptr(sp), rows, t, win, hop, pad, frames, 1, stream()) == 0
    for grp, want in ((0, want_e), (1, want_o)):
        hi, lo, hi2 = sp[grp * 3 * h:grp * 3 * h + h], sp[grp * 3 * h + h:grp * 3 * h + 2 * h], sp[grp * 3 * h + 2 * h:(grp + 1) * 3 * h]
        assert torch.equal(hi, hi2)
        assert torch.equal(hi, want.to(torch.bfloat16).float()) and torch.equal(lo, (want - hi).to(torch.bfloat16).float())
        assert float((hi.double() + lo.double() - want.double()).abs().max()) <= 2.0 ** -16 * float(want.abs().max())
    s3 = torch.empty(6 * h, cols, device=dev)
    assert lib.eben_split3(ptr(fold), ptr(s3), 2, h, cols, stream()) == 0
    assert torch.equal(s3, sp)

    # adjoint: <frames_folded(sig), D> == <sig, overlap_add_folded(D)> for every D
    d = torch.randn(2 * h, cols, generator=g).to(dev)
    back = torch.empty(rows, t, device=dev)
    assert lib.eben_overlap_add_folded(ptr(d), ptr(back), rows, t, win, frames, hop, pad, 0, frames, cols, stream()) == 0
    lhs = float((fold.double() * d.double()).sum())
    rhs = float((sig.double() * back.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs)), (lhs, rhs)
    twice = back.clone()
    assert lib.eben_overlap_add_folded(ptr(d), ptr(twice), rows, t, win, frames, hop, pad, 1, frames, cols, stream()) == 0
    assert torch.equal(twice, 2 * back)


# "bf16x3": hi / lo bf16 operand splits on the bf16 MFMA, ~2^-17 relative per product (the bf16 train step's setting)
# (loss rtol, gradient relative L2): the gradient of |log X - log Y| carries sign(log X - log Y), which flips wherever the two
# magnitudes agree to within the arithmetic's noise -- 2e-3 of the gradient's L2 at fp32, 5e-3 at 2^-17
STFT_MATH_TOL = {"folded": (2e-5, 2e-3), "dense": (2e-5, 2e-3), "bf16x3": (5e-5, 1e-2)}


@pytest.mark.parametrize("stft_math", ["folded", "dense", "bf16x3"])
@pytest.mark.parametrize("perceptual", [True, False])
def test_mrstft_loss_and_grad(hip, perceptual, stft_math):
    from vibravox_amd.torch_modules.losses.mrstft_loss import MultiResolutionSTFTLoss

    dev = torch.device("cuda")
    from formula import formula_audio

    x, y = formula_audio("mr_x", 2, 4000), formula_audio("mr_y", 2, 4000)
    loss = MultiResolutionSTFTLoss(fft_sizes=(512, 1024, 2048), hop_sizes=(50, 120, 240), win_lengths=(240, 600, 1200),
                                   sample_rate=16000, perceptual_weighting=perceptual).to(dev)
    loss.stft_math = stft_math
    xd = x.to(dev).requires_grad_(True)
    got = loss(xd, y.to(dev))
    got.backward()
    rx = x.double().requires_grad_(True)
    fir = O.a_weighting_fir(16000).double() if perceptual else None
    ref = O.mrstft_loss(rx, y.double(), perceptual_weighting=perceptual, fir=fir)
    ref.backward()
    rtol, gtol = STFT_MATH_TOL[stft_math]
    np.testing.assert_allclose(got.item(), ref.item(), rtol=rtol)
    err = float((xd.grad.double().cpu() - rx.grad).norm() / rx.grad.norm())
    assert err < gtol, err  # sign(log X - log Y) flips at fp32 noise: compare in L2


@pytest.mark.parametrize("stft_math", ["folded", "bf16x3"])
def test_mrstft_ragged_length_and_batch(hip, stft_math):
    """The flat frame matrix (win, rows*frames) at a length / batch where neither the frame count nor the column count
    is a multiple of anything convenient (3 x 4321 samples: 87 / 37 / 19 frames per item)."""
    from formula import formula_audio

    from vibravox_amd.torch_modules.losses.mrstft_loss import MultiResolutionSTFTLoss

    dev = torch.device("cuda")
    x, y = formula_audio("mr2_x", 3, 4321), formula_audio("mr2_y", 3, 4321)
    loss = MultiResolutionSTFTLoss(fft_sizes=(512, 1024, 2048), hop_sizes=(50, 120, 240), win_lengths=(240, 600, 1200),
                                   sample_rate=16000, perceptual_weighting=True).to(dev)
    loss.stft_math = stft_math
    xd = x.to(dev).requires_grad_(True)
    got = loss(xd, y.to(dev))
    got.backward()
    rx = x.double().requires_grad_(True)
    ref = O.mrstft_loss(rx, y.double(), perceptual_weighting=True, fir=O.a_weighting_fir(16000).double())
    ref.backward()
    rtol, gtol = STFT_MATH_TOL[stft_math]
    np.testing.assert_allclose(got.item(), ref.item(), rtol=rtol)
    assert float((xd.grad.double().cpu() - rx.grad).norm() / rx.grad.norm()) < gtol


def test_adam_matches_torch(hip):
    from vibravox_amd.optim import FusedAdam

    dev = torch.device("cuda")
    shapes = [(7,), (33, 5, 3), (1, 1, 1), (1024, 64)]
    ps = [formula_tensor(f"ad/p{i}", s) for i, s in enumerate(shapes)]
    ref = [p.clone().double().requires_grad_(True) for p in ps]
    got = [p.clone().to(dev).requires_grad_(True) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=3e-4, betas=(0.5, 0.9))
    o_got = FusedAdam(got, lr=3e-4, betas=(0.5, 0.9))
    for step in range(3):
        for i, (r, g) in enumerate(zip(ref, got)):
            gr = formula_tensor(f"ad/g{step}/{i}", shapes[i], 0.01)
            r.grad, g.grad = gr.double(), gr.to(dev)
        o_ref.step()
        o_got.step()
    for r, g in zip(ref, got):
        assert rel_err(g, r) < 1e-6
    sd = o_got.state_dict()
    assert set(sd["state"][0].keys()) >= {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("math_name", ["f32", "bf16"])
def test_full_size_melgan_layer4_forward_against_fp64(hip, math_name):
    """One full-size layer of BASELINE config 2 against the fp64 oracle: MelGAN layer 4 (1024 -> 1024, k 41, stride 4, 4 groups,
    melgan_discriminator.py:89-156) on the 32 x 1024 x 500 activations the step feeds it -- the launch bench.py quotes its roofline
    on.  fp32: 3e-5 of max|ref| (the fp32-GEMM bound of the small cases, K = 10 496); bf16: the same bound against an fp64
    convolution of the bf16-rounded operands (what the kernel is specified to compute) and 1e-2 against the exact one."""
    import ctypes

    from vibravox_amd import ops
    from vibravox_amd._lib import check, load, ptr, stream

    lib, dev = load(), torch.device("cuda")
    spec = ops.ConvSpec(c_in=1024, c_out=1024, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2)
    b, l_in = 32, 500
    v = formula_tensor("full/l4/v", (1024, 256, 41), 1 / math.sqrt(256 * 41))
    g = 0.5 + formula_tensor("full/l4/g", (1024, 1, 1), 0.4).abs()
    bias = formula_tensor("full/l4/b", (1024,), 0.1)
    x = formula_tensor("full/l4/x", (b, 1024, l_in))
    math_id = ops.MATH_F32 if math_name == "f32" else ops.MATH_BF16
    d = ops.conv_desc(spec, b, l_in, math_id)
    vd, gd, bd, xd = v.to(dev), g.to(dev), bias.to(dev), x.to(dev)
    pw = ops.pack_weights(spec, d, vd, gd, None, False)
    y = torch.empty((b, 1024, d.l_out), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(xd), ptr(pw.wp_fwd), ptr(bd), None, ptr(y), stream()), "conv1d_fwd")
    torch.cuda.synchronize()
    w = (v.double() * (g.double() / v.double().flatten(1).norm(dim=1).reshape(-1, 1, 1)))

    def ref_conv(xx, ww):
        return torch.nn.functional.leaky_relu(torch.nn.functional.conv1d(xx, ww, bias.double(), stride=4, padding=20, groups=4), 0.2)

    exact = ref_conv(x.double(), w)
    assert tuple(y.shape) == tuple(exact.shape) == (b, 1024, 125)
    scale = float(exact.abs().max())
    if math_name == "f32":
        assert float((y.cpu().double() - exact).abs().max()) < 3e-5 * scale
    else:
        rounded = ref_conv(x.bfloat16().double(), w.float().bfloat16().double())
        assert float((y.cpu().double() - rounded).abs().max()) < 3e-5 * scale
        assert float((y.cpu().double() - exact).abs().max()) < 1e-2 * scale


@pytest.mark.parametrize("ntaps,length,batch", [(101, 31968, 3), (101, 1000, 2), (7, 37, 1), (100, 2049, 2), (4, 5, 1)])
def test_single_band_fir_and_adjoint(hip, ntaps, length, batch):
    """The stride-1 single-band FIR kernel (direct.hip fir1_kernel: the A-weighting prefilter of the MRSTFT loss, auraloss FIRFilter "aw",
    and its adjoint in the backward) against float64, the adjoint identity <A x, y> = <x, A^T y>, and -- through EBEN_FIR1=0 in a second
    process -- bit for bit against the generic FIR-bank kernels it replaces on this shape."""
    import subprocess
    import sys

    import torch.nn.functional as F

    from vibravox_amd import ops

    DEV = torch.device("cuda")
    g = torch.Generator().manual_seed(ntaps * 1000 + length)
    x = torch.randn(batch, 1, length, generator=g)
    y = torch.randn(batch, 1, length, generator=g)
    w = torch.randn(1, ntaps, generator=g) / ntaps
    off0 = -(ntaps // 2)
    xd, yd, wd = x.to(DEV), y.to(DEV), w.to(DEV)
    ax = ops._fir_decimate(xd, wd, length, 1, ntaps, 1, off0)
    aty = ops._fir_interp_sum(yd, wd, length, 1, ntaps, 1, off0)
    ref = F.conv1d(F.pad(x.double(), (-off0, ntaps - 1 + off0)), w.double().reshape(1, 1, ntaps))
    assert float((ax.cpu().double() - ref).abs().max()) < 2e-6 * float(ref.abs().max()) + 1e-7
    lhs, rhs = float((ax.double() * yd.double()).sum()), float((xd.double() * aty.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * (abs(lhs) + 1.0)
    code = ("import sys, torch; sys.path.insert(0, %r); from vibravox_amd import ops; d = torch.device('cuda');"
            "g = torch.Generator().manual_seed(%d); x = torch.randn(%d, 1, %d, generator=g); y = torch.randn(%d, 1, %d, generator=g);"
            "w = torch.randn(1, %d, generator=g) / %d;"
            "a = ops._fir_decimate(x.to(d), w.to(d), %d, 1, %d, 1, %d); b = ops._fir_interp_sum(y.to(d), w.to(d), %d, 1, %d, 1, %d);"
            "torch.save((a.cpu(), b.cpu()), sys.argv[1])"
            % (ROOT, ntaps * 1000 + length, batch, length, batch, length, ntaps, ntaps, length, ntaps, off0, length, ntaps, off0))
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "generic.pt")
        subprocess.run([sys.executable, "-c", code, path], check=True, env={**os.environ, "EBEN_FIR1": "0"}, timeout=300)
        ga, gb = torch.load(path)
    assert torch.equal(ga, ax.cpu()) and torch.equal(gb, aty.cpu())


@pytest.mark.parametrize("win,hop,n_fft,t,rows", [(240, 50, 512, 31968, 3), (600, 120, 1024, 31968, 2), (1200, 240, 2048, 31968, 2), (240, 50, 512, 1300, 2),
                                                  (600, 120, 1024, 2500, 1), (1200, 240, 2048, 2600, 1)])
def test_folded_framing_and_overlap_add_tiled_kernels_are_bit_identical(hip, win, hop, n_fft, t, rows):
    """eben_stft_frames_folded through the LDS transpose and eben_overlap_add_folded from the LDS tile (direct.hip) against the gather kernels
    they replace (EBEN_STFT_FRAMES_T=0 / EBEN_OLA_TILED=0 in a second process): the same sums of the same samples, bit for bit -- full-size
    clips and clips a few tiles long (both reflected ends inside one or two blocks)."""
    import os
    import subprocess
    import sys
    import tempfile

    code = ("import sys, ctypes, torch; sys.path.insert(0, %r)\n"
            "from vibravox_amd._lib import check, load, ptr, stream\n"
            "lib = load(); d = torch.device('cuda'); win, hop, t, rows = %d, %d, %d, %d\n"
            "pad = win // 2; frames = 1 + t // hop\n"
            "g = torch.Generator().manual_seed(win + t); sig = torch.randn(rows, t, generator=g).to(d)\n"
            "fr = torch.empty(win * rows * frames, device=d)\n"
            "check(lib.eben_stft_frames_folded(ptr(sig), ptr(fr), rows, t, win, hop, pad, frames, 0, stream()), 'frames')\n"
            "dfr = torch.randn(win * rows * frames, generator=g).to(d); dsig = torch.zeros(rows, t, device=d)\n"
            "check(lib.eben_overlap_add_folded(ptr(dfr), ptr(dsig), rows, t, win, frames, hop, pad, 0, frames, rows * frames, stream()), 'ola')\n"
            "acc = torch.ones(rows, t, device=d)\n"
            "check(lib.eben_overlap_add_folded(ptr(dfr), ptr(acc), rows, t, win, frames, hop, pad, 1, frames, rows * frames, stream()), 'ola')\n"
            "torch.cuda.synchronize(); torch.save((fr.cpu(), dsig.cpu(), acc.cpu()), sys.argv[1])\n" % (ROOT, win, hop, t, rows))
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for k, env in enumerate(({}, {"EBEN_STFT_FRAMES_T": "0", "EBEN_OLA_TILED": "0"})):
            path = os.path.join(td, f"o{k}.pt")
            subprocess.run([sys.executable, "-c", code, path], check=True, env={**os.environ, **env}, timeout=300)
            outs.append(torch.load(path))
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    # adjointness of the pair: <frames(s), g> = <s, overlap_add(g)> (float64 sums of the fp32 results)
    fr, dsig, _ = outs[0]
    g = torch.Generator().manual_seed(win + t)
    sig = torch.randn(rows, t, generator=g)
    dfr = torch.randn(win * rows * (1 + t // hop), generator=g)
    lhs, rhs = float((fr.double() * dfr.double()).sum()), float((sig.double() * dsig.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * (abs(lhs) + abs(rhs) + 1.0), (lhs, rhs)
