"""GPU: the bundle-layout (bf16 at rest) kernels of the discriminator engine through the C ABI -- ``eben_bl_*`` of include/eben_hip.h.

Two kinds of checks:
  * against float64 torch restatements of the layer (the same functions the CPU oracle is built from) on the values the kernels
    actually read (hi + lo planes converted back), fp32 tolerances;
  * against the fp32-at-rest kernels of the same arithmetic plan, BIT FOR BIT where the launch plans coincide: the bundle layout
    must hand the MFMA the same bf16 operands (the same round-to-nearest-even of the same fp32 values).
"""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from formula import formula_tensor

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
BL = 0x100


def planes_of(x, lo=True):
    from vibravox_amd.disc_engine_bl import Planes

    return Planes.from_f32(x.to(DEV), lo)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape and torch.isfinite(got).all()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def bf16_hi(x):
    return x.to(torch.bfloat16).to(torch.float32)


def test_bundle_planes_round_trip(hip):
    x = formula_tensor("bl/rt", (3, 24, 1001)).to(DEV) * 3.7
    p = planes_of(x)
    hi = p.hi.permute(0, 1, 3, 2).reshape(3, 24, 1001).float()
    lo = p.lo.permute(0, 1, 3, 2).reshape(3, 24, 1001).float()
    assert torch.equal(hi, bf16_hi(x))                       # round to nearest even, the tap-conv staging's own rounding
    assert torch.equal(lo, bf16_hi(x - hi))
    back = p.to_f32()
    assert torch.equal(back, hi + lo)
    assert float(((back - x).abs() / x.abs().clamp_min(1e-20)).max()) < 2.0 ** -15


HEADS = {
    # name: (c_in, c_out, ksize, dilation, pad, reflect_pad, batch, length)
    "pqmf_d1": (4, 24, 3, 1, 1, 1, 3, 1003),
    "pqmf_d2": (4, 24, 3, 2, 1, 1, 3, 1003),
    "pqmf_d3": (4, 24, 3, 3, 1, 1, 2, 700),
    "melgan": (1, 16, 15, 1, 0, 7, 2, 2014),
    "melgan_odd_length": (1, 16, 15, 1, 0, 7, 3, 1535),   # rows that end inside a two-position pair of the input-gradient kernel
    "melgan_d2": (1, 16, 15, 2, 0, 7, 2, 1200),           # dilation 2: the one-position input-gradient kernel
}


def head_ref(x, w, bias, c_in, dil, pad, rpad, slope):
    xp = F.pad(x, (rpad, rpad), mode="reflect")
    return F.leaky_relu(F.conv1d(xp, w, bias, dilation=dil, padding=pad, groups=c_in), slope)


def head_job(lib, x, v, scale, bias, out, c_in, c_out, l_in, k, dil, pad, rpad, slope):
    from vibravox_amd._lib import EbenBlHeadJob

    j = EbenBlHeadJob()
    j.x, j.v, j.scale, j.bias = (x.data_ptr() if x is not None else None), v.data_ptr(), scale.data_ptr(), bias.data_ptr()
    j.y_hi, j.y_lo = out.hi.data_ptr(), (out.lo.data_ptr() if out.lo is not None else None)
    j.c_in, j.c_out, j.l_in, j.l_out, j.ksize, j.dilation, j.pad, j.reflect_pad, j.out_slope = c_in, c_out, l_in, out.length, k, dil, pad, rpad, slope
    return j


@pytest.mark.parametrize("name", list(HEADS))
def test_chain_head_forward_backward(hip, name):
    from vibravox_amd._lib import EbenBlHeadJob, check
    from vibravox_amd.disc_engine_bl import Planes

    c_in, c_out, k, dil, pad, rpad, batch, length = HEADS[name]
    v = formula_tensor(f"blh/{name}/v", (c_out, 1, k), 1 / math.sqrt(k)).to(DEV)
    scale = (1 + 0.3 * formula_tensor(f"blh/{name}/s", (c_out,))).to(DEV)
    bias = formula_tensor(f"blh/{name}/b", (c_out,), 0.1).to(DEV)
    x = formula_tensor(f"blh/{name}/x", (batch, c_in, length)).to(DEV)
    l_out = length + 2 * rpad + 2 * pad - dil * (k - 1)
    st = torch.cuda.current_stream().cuda_stream
    out = Planes(batch, c_out, l_out, DEV)
    jobs = (EbenBlHeadJob * 1)(head_job(hip, x, v, scale, bias, out, c_in, c_out, length, k, dil, pad, rpad, 0.2))
    check(hip.eben_bl_head_fwd(jobs, 1, batch, st), "head_fwd")
    w = (v * scale.reshape(-1, 1, 1)).double()
    xr = x.double().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = bias.double().requires_grad_(True)
    ref = head_ref(xr, wr, br, c_in, dil, pad, rpad, 0.2)
    assert rel_err(out.to_f32(), ref) < 1e-5
    # backward: gradient planes -> input gradient (with the reflect fold) and weight / bias gradient
    gy = formula_tensor(f"blh/{name}/gy", (batch, c_out, l_out)).to(DEV)
    gp = planes_of(gy)
    (ref * gp.to_f32().double()).sum().backward()          # the gradient the kernels read: hi + lo
    dx = torch.empty_like(x)
    jobs = (EbenBlHeadJob * 1)(head_job(hip, None, v, scale, bias, gp, c_in, c_out, length, k, dil, pad, rpad, 0.2))
    # d/dx of the pre-activation: feed the kernel the gradient already masked by the LeakyReLU derivative, as the engine does
    masked = planes_of(gp.to_f32() * torch.where(ref.detach().float() > 0, 1.0, 0.2))
    jobs[0].y_hi, jobs[0].y_lo = masked.hi.data_ptr(), masked.lo.data_ptr()
    check(hip.eben_bl_head_dx(jobs, 1, batch, dx.data_ptr(), st), "head_dx")
    # reference for the masked gradient: autograd through conv only
    xr2 = x.double().requires_grad_(True)
    pre = F.conv1d(F.pad(xr2, (rpad, rpad), mode="reflect"), w, None, dilation=dil, padding=pad, groups=c_in)
    (pre * masked.to_f32().double()).sum().backward()
    assert rel_err(dx, xr2.grad) < 1e-5
    # weight gradient of the same masked gradient (hi plane only feeds it) against autograd on the rounded gradient
    hi_only = Planes.__new__(Planes)
    hi_only.codes = None
    hi_only.hi, hi_only.lo, hi_only.rows, hi_only.channels, hi_only.length = masked.hi, None, batch, c_out, l_out
    wj = head_job(hip, x, v, scale, bias, hi_only, c_in, c_out, length, k, dil, pad, rpad, 0.2)
    nslab, rs = ctypes.c_int(0), ctypes.c_int(0)
    nbytes = hip.eben_bl_head_dw_workspace(ctypes.byref(wj), batch, ctypes.byref(nslab), ctypes.byref(rs))
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV)
    check(hip.eben_bl_head_dw(ctypes.byref(wj), batch, slabs.data_ptr(), nbytes, st), "head_dw")
    got = slabs.reshape(nslab.value, c_out, rs.value).double().sum(0)
    wr3 = w.clone().requires_grad_(True)
    pre3 = F.conv1d(F.pad(x.double(), (rpad, rpad), mode="reflect"), wr3, None, dilation=dil, padding=pad, groups=c_in)
    g_hi = hi_only.hi.permute(0, 1, 3, 2).reshape(batch, c_out, l_out).double()
    (pre3 * g_hi).sum().backward()
    assert rel_err(got[:, :k], wr3.grad.reshape(c_out, k)) < 2e-5
    assert rel_err(got[:, k], g_hi.sum(dim=(0, 2))) < 2e-5


def test_pqmf_heads_as_one_launch_and_summed_input_gradient(hip):
    """The three PQMF-band chains (dilation 1, 2, 3) read the same bands: one forward launch, one backward launch whose result is the SUM."""
    from vibravox_amd._lib import EbenBlHeadJob, check
    from vibravox_amd.disc_engine_bl import Planes

    batch, length = 3, 1500
    x = formula_tensor("blh3/x", (batch, 4, length)).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    par, outs, refs = [], [], []
    for dil in (1, 2, 3):
        v = formula_tensor(f"blh3/{dil}/v", (24, 1, 3), 0.5).to(DEV)
        scale = (1 + 0.3 * formula_tensor(f"blh3/{dil}/s", (24,))).to(DEV)
        bias = formula_tensor(f"blh3/{dil}/b", (24,), 0.1).to(DEV)
        l_out = length + 2 + 2 - dil * 2
        out = Planes(batch, 24, l_out, DEV)
        par.append((v, scale, bias, dil))
        outs.append(out)
        refs.append(head_ref(x.double(), (v * scale.reshape(-1, 1, 1)).double(), bias.double(), 4, dil, 1, 1, 0.2))
    jobs = (EbenBlHeadJob * 3)(*[head_job(hip, x, v, s, b, o, 4, 24, length, 3, d, 1, 1, 0.2) for (v, s, b, d), o in zip(par, outs)])
    check(hip.eben_bl_head_fwd(jobs, 3, batch, st), "head_fwd x3")
    for o, r in zip(outs, refs):
        assert rel_err(o.to_f32(), r) < 1e-5
    gs = [planes_of(formula_tensor(f"blh3/{i}/g", (batch, 24, o.length)).to(DEV)) for i, o in enumerate(outs)]
    jobs = (EbenBlHeadJob * 3)(*[head_job(hip, None, v, s, b, g, 4, 24, length, 3, d, 1, 1, 0.2) for (v, s, b, d), g in zip(par, gs)])
    dx = torch.empty_like(x)
    check(hip.eben_bl_head_dx(jobs, 3, batch, dx.data_ptr(), st), "head_dx x3")
    xr = x.double().requires_grad_(True)
    total = 0
    for (v, s, b, d), g in zip(par, gs):
        pre = F.conv1d(F.pad(xr, (1, 1), mode="reflect"), (v * s.reshape(-1, 1, 1)).double(), None, dilation=d, padding=1, groups=4)
        total = total + (pre * g.to_f32().double()).sum()
    total.backward()
    assert rel_err(dx, xr.grad) < 1e-5


@pytest.mark.parametrize("channels,length,k", [(768, 251, 3), (1024, 125, 3), (96, 33, 3), (96, 40, 5)])   # k = 5: the kernels for any tap count
def test_chain_tail_forward_backward(hip, channels, length, k):
    from vibravox_amd._lib import check
    from vibravox_amd.disc_engine_bl import Planes

    half = 2
    rows2, rows4 = 2 * half, 4 * half
    pad = k // 2
    st = torch.cuda.current_stream().cuda_stream
    v = formula_tensor(f"blt/{channels}/{k}/v", (1, channels, k), 1 / math.sqrt(channels * k)).to(DEV)
    scale = torch.tensor([1.3], device=DEV)
    bias = torch.tensor([0.05], device=DEV)
    act = planes_of(formula_tensor(f"blt/{channels}/x", (rows2, channels, length)))
    a = act.to_f32().double()
    w = (v * scale).double()
    logits = torch.empty((rows2, 1, length), dtype=torch.float32, device=DEV)
    check(hip.eben_bl_tail_fwd(act.hi.data_ptr(), act.lo.data_ptr(), rows2, channels, length, k, pad, v.data_ptr(), scale.data_ptr(), bias.data_ptr(),
                               1.0, logits.data_ptr(), st), "tail_fwd")
    assert rel_err(logits, F.conv1d(a, w, bias.double(), padding=pad)) < 1e-5
    # input gradient of four stacked seeds, feature-matching term on the first `half` rows, mask from the embedding rows (0, 0, 0, 1)
    seeds = formula_tensor(f"blt/{channels}/seed", (rows4, 1, length)).to(DEV)
    sums = torch.tensor([3.5, 11.0], device=DEV)
    fm_gs = 0.37
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
    g = Planes(rows4, channels, length, DEV)
    check(hip.eben_bl_tail_dx(seeds.data_ptr(), rows4, channels, length, k, pad, v.data_ptr(), scale.data_ptr(), act.hi.data_ptr(), act.lo.data_ptr(), 0.2,
                              half, seg_map, half, half, sums.data_ptr(), fm_gs, g.hi.data_ptr(), g.lo.data_ptr(), st), "tail_dx")
    base = F.conv_transpose1d(seeds.double(), w, padding=pad)                      # (rows4, channels, length)
    enh, ref = a[:half], a[half:]
    fm = fm_gs * (torch.sign(enh - ref) / 11.0 - 3.5 * torch.sign(enh) / 11.0 ** 2)
    base[:half] += fm
    a_hi = act.hi.permute(0, 1, 3, 2).reshape(rows2, channels, length).double()
    mask_rows = torch.cat((a_hi[:half], a_hi[:half], a_hi[:half], a_hi[half:]), dim=0)
    want = base * torch.where(mask_rows > 0, 1.0, 0.2)
    assert rel_err(g.to_f32(), want) < 2e-5
    # weight gradient of the two hinge branches in one launch: seed rows [fake | real] against embedding rows [enhanced | reference]
    nslab, rs = ctypes.c_int(0), ctypes.c_int(0)
    nbytes = hip.eben_bl_tail_dw_workspace(half, channels, length, k, 2, ctypes.byref(nslab), ctypes.byref(rs))
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV)
    sd = seeds[2 * half:].contiguous()
    check(hip.eben_bl_tail_dw(sd.data_ptr(), act.hi.data_ptr(), act.lo.data_ptr(), half, 2, channels, length, k, pad, slabs.data_ptr(), nbytes, st), "tail_dw")
    got = slabs.reshape(2, nslab.value, rs.value).double().sum(1)
    for br in range(2):
        wr = w.clone().requires_grad_(True)
        sb = sd[br * half:(br + 1) * half].double()
        (F.conv1d(a[br * half:(br + 1) * half], wr, None, padding=pad) * sb).sum().backward()
        assert rel_err(got[br, :-1], wr.grad.reshape(-1)) < 2e-5
        assert abs(float(got[br, -1]) - float(sb.sum())) < 1e-4 * float(sb.abs().sum())


def test_feature_matching_sums_over_planes(hip):
    from vibravox_amd._lib import check

    half = 3
    ps = [planes_of(formula_tensor(f"blfm/{i}", (2 * half, c, l))) for i, (c, l) in enumerate([(24, 1000), (96, 251), (768, 63)])]
    n = len(ps)
    ptrs = (ctypes.c_void_p * (2 * n))(*[q for p in ps for q in (p.hi.data_ptr(), p.lo.data_ptr())])
    units = (ctypes.c_int64 * n)(*[half * (p.channels // 8) * p.length for p in ps])
    ws_bytes = hip.eben_bl_fm_sums_workspace(n)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=DEV)
    sums = torch.empty(2 * n, dtype=torch.float32, device=DEV)
    check(hip.eben_bl_fm_sums(ptrs, units, n, ws.data_ptr(), ws_bytes, sums.data_ptr(), torch.cuda.current_stream().cuda_stream), "bl_fm_sums")
    for i, p in enumerate(ps):
        a = p.to_f32().double()
        np.testing.assert_allclose(sums[2 * i].item(), float((a[:half] - a[half:]).abs().sum()), rtol=2e-5)
        np.testing.assert_allclose(sums[2 * i + 1].item(), float(a[:half].abs().sum()), rtol=2e-5)


MID_LAYERS = {
    # name: (ConvSpec kwargs, batch rows of the forward (2 * half), length, same launch plan as the fp32-at-rest kernels in both directions?)
    "pqmf_l3_d2": (dict(c_in=96, c_out=192, ksize=7, stride=2, dilation=2, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 4, 1001, True),
    "pqmf_l5_d3": (dict(c_in=384, c_out=768, ksize=7, stride=2, dilation=3, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 4, 260, True),
    "pqmf_l6_d1": (dict(c_in=768, c_out=768, ksize=5, stride=1, dilation=1, pad_l=2, pad_r=2, groups=4, out_slope=0.2), 4, 140, True),
    "pqmf_l1_d1_dense": (dict(c_in=24, c_out=48, ksize=7, stride=2, dilation=1, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 4, 2003, False),
    "pqmf_l2_d3_dense": (dict(c_in=48, c_out=96, ksize=7, stride=2, dilation=3, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 4, 1501, False),
    "melgan_l1_dense": (dict(c_in=16, c_out=64, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 4, 2100, False),
    "melgan_l2": (dict(c_in=64, c_out=256, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 4, 1100, True),
    "melgan_l4_like": (dict(c_in=512, c_out=512, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 4, 500, True),
    "melgan_l5_like": (dict(c_in=512, c_out=512, ksize=5, stride=1, pad_l=2, pad_r=2, groups=1, out_slope=0.2), 4, 125, True),
    # 192 rows and 192 channels per group, nine taps: the 192-row tile of the persistent kernel (bigtap.hip) in both directions and both arithmetics
    "wide_192rows_k9": (dict(c_in=768, c_out=768, ksize=9, stride=1, pad_l=4, pad_r=4, groups=4, out_slope=0.2), 4, 140, False),
    # ragged edges: clips shorter than one tile, one position past a tile boundary, a single pair of rows, an odd half batch
    "ragged_short_dense": (dict(c_in=24, c_out=48, ksize=7, stride=2, dilation=2, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 2, 37, False),
    "ragged_tile_plus_one": (dict(c_in=96, c_out=192, ksize=7, stride=2, dilation=1, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 6, 263, True),
    "ragged_deep_short": (dict(c_in=256, c_out=1024, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 2, 97, True),
    "ragged_s1_short": (dict(c_in=768, c_out=768, ksize=5, stride=1, dilation=3, pad_l=2, pad_r=2, groups=4, out_slope=0.2), 6, 19, True),
}


def _layer_setup(name):
    from vibravox_amd import ops

    kw, rows, length, same = MID_LAYERS[name]
    spec = ops.ConvSpec(**kw)
    wshape = spec.weight_shape()
    v = formula_tensor(f"blm/{name}/v", wshape, 1 / math.sqrt(wshape[1] * wshape[2])).to(DEV)
    scale = (1 + 0.3 * formula_tensor(f"blm/{name}/s", (wshape[0],))).to(DEV)
    bias = formula_tensor(f"blm/{name}/b", (spec.c_out,), 0.1).to(DEV)
    return spec, rows, length, same, v, scale, bias


def _pack(hip, d, v, scale, which):
    from vibravox_amd import ops

    wp = torch.empty(hip.eben_conv1d_packed_floats(ctypes.byref(d), which), dtype=torch.float32, device=DEV)
    ops.conv1d_pack(d, v, scale, wp if which == 0 else None, wp if which == 1 else None)
    return wp


@pytest.mark.parametrize("math_name", ["bf16", "bf16x3"])
@pytest.mark.parametrize("name", list(MID_LAYERS))
def test_bundle_conv_forward(hip, name, math_name):
    """Forward on bundle planes: against float64 on the operands the MFMA multiplies, and bit for bit against the fp32-at-rest kernel of
    the same math where the two run the same launch plan (the hi plane of the output = RNE of that kernel's fp32 output)."""
    from vibravox_amd import ops
    from vibravox_amd._lib import check
    from vibravox_amd.disc_engine_bl import Planes

    spec, rows, length, same, v, scale, bias = _layer_setup(name)
    math_id = {"bf16": ops.MATH_BF16, "bf16x3": ops.MATH_BF16X3}[math_name]
    x = formula_tensor(f"blm/{name}/x", (rows, spec.c_in, length)).to(DEV)
    xp = planes_of(x)
    d = ops.conv_desc(spec, rows, length, math_id | BL)
    wp = _pack(hip, d, v, scale, 0)
    y = Planes(rows, spec.c_out, d.l_out, DEV)
    st = torch.cuda.current_stream().cuda_stream
    check(hip.eben_bl_conv1d_fwd(ctypes.byref(d), xp.hi.data_ptr(), xp.lo.data_ptr() if math_id == ops.MATH_BF16X3 else None, wp.data_ptr(), bias.data_ptr(),
                                 y.hi.data_ptr(), y.lo.data_ptr(), st), "bl_conv1d_fwd")
    w = (v * scale.reshape(-1, 1, 1))
    if math_id == ops.MATH_BF16:
        xin, wq, tol = bf16_hi(x), bf16_hi(w), 2e-5
    else:
        xin, wq, tol = xp.to_f32(), bf16_hi(w) + bf16_hi(w - bf16_hi(w)), 3e-5   # three of the four piece products: lo x lo (2^-18) is dropped
    ref = F.leaky_relu(F.conv1d(xin.double(), wq.double(), bias.double(), stride=spec.stride, padding=spec.pad_l, dilation=spec.dilation, groups=spec.groups), 0.2)
    got = y.to_f32()
    assert rel_err(got, ref) < tol, rel_err(got, ref)
    # the fp32-at-rest kernel of the same math on the same fp32 input
    d32 = ops.conv_desc(spec, rows, length, math_id)
    wp32 = _pack(hip, d32, v, scale, 0)
    y32 = torch.empty((rows, spec.c_out, d.l_out), dtype=torch.float32, device=DEV)
    check(hip.eben_conv1d_fwd(ctypes.byref(d32), x.data_ptr(), wp32.data_ptr(), bias.data_ptr(), None, y32.data_ptr(), st), "conv1d_fwd")
    # bundle-layout launches served by the persistent whole-panel kernel (generation 6) sum the channel chunks in another order
    if same and hip.eben_conv1d_kernel_generation(ctypes.byref(d), 0) == hip.eben_conv1d_kernel_generation(ctypes.byref(d32), 0):
        hi = y.hi.permute(0, 1, 3, 2).reshape(rows, spec.c_out, d.l_out).float()
        assert torch.equal(hi, bf16_hi(y32))
        assert torch.equal(got, bf16_hi(y32) + bf16_hi(y32 - bf16_hi(y32)))
    else:
        assert rel_err(got, y32) < 1e-4     # block-diagonal form here, grouped form there: other summation order, other zero padding


@pytest.mark.parametrize("name", list(MID_LAYERS))
def test_bundle_conv_input_gradient(hip, name):
    """The engine's stacked input gradient on bundle planes (mask from the saved embedding, feature-matching term on the first rows):
    against float64, and bit for bit against eben_conv1d_bwd_dx_fm on fp32 tensors where the launch plans coincide."""
    from vibravox_amd import ops
    from vibravox_amd._lib import check
    from vibravox_amd.disc_engine_bl import Planes

    spec, rows2, length, same, v, scale, bias = _layer_setup(name)
    half = rows2 // 2
    rows4 = 4 * half
    lin = ops.ConvSpec(**{**MID_LAYERS[name][0], "out_slope": 1.0})
    l_out = spec.out_len(length)
    g = formula_tensor(f"blm/{name}/g", (rows4, spec.c_out, l_out)).to(DEV)
    gp = planes_of(g, lo=False)
    act = planes_of(formula_tensor(f"blm/{name}/act", (rows2, spec.c_in, length)))
    a = act.to_f32()                                  # exactly hi + lo: what both kernels see
    sums = torch.tensor([2.5, 7.0], device=DEV)
    fm_gs = 0.41
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
    d = ops.conv_desc(lin, rows4, length, ops.MATH_BF16 | BL)
    wp = _pack(hip, d, v, scale, 1)
    dx = Planes(rows4, spec.c_in, length, DEV)
    st = torch.cuda.current_stream().cuda_stream
    check(hip.eben_bl_conv1d_bwd_dx(ctypes.byref(d), gp.hi.data_ptr(), wp.data_ptr(), act.hi.data_ptr(), act.lo.data_ptr(), 0.2, half, seg_map, half, half,
                                    sums.data_ptr(), fm_gs, dx.hi.data_ptr(), dx.lo.data_ptr(), st), "bl_conv1d_bwd_dx")
    # the same launch with the feature-matching code plane of the embedding in place of act_lo and the reference rows: bit for bit
    dxc = Planes(rows4, spec.c_in, length, DEV)
    codes = fm_codes_of(hip, act, half)
    check(hip.eben_bl_conv1d_bwd_dx_c(ctypes.byref(d), gp.hi.data_ptr(), wp.data_ptr(), act.hi.data_ptr(), act.lo.data_ptr(), codes.data_ptr(), 0.2, half, seg_map,
                                      half, half, sums.data_ptr(), fm_gs, dxc.hi.data_ptr(), dxc.lo.data_ptr(), st), "bl_conv1d_bwd_dx_c")
    assert torch.equal(dxc.hi.view(torch.int16), dx.hi.view(torch.int16)) and torch.equal(dxc.lo.view(torch.int16), dx.lo.view(torch.int16))
    w = (v * scale.reshape(-1, 1, 1))
    base = F.conv_transpose1d(bf16_hi(g).double(), bf16_hi(w).double(), stride=spec.stride, padding=spec.pad_l, dilation=spec.dilation, groups=spec.groups,
                              output_padding=length - ((l_out - 1) * spec.stride - 2 * spec.pad_l + spec.dilation * (spec.ksize - 1) + 1))
    ad = a.double()
    base[:half] += fm_gs * (torch.sign(ad[:half] - ad[half:]) / 7.0 - 2.5 * torch.sign(ad[:half]) / 49.0)
    a_hi = act.hi.permute(0, 1, 3, 2).reshape(rows2, spec.c_in, length).double()
    mask_rows = torch.cat((a_hi[:half], a_hi[:half], a_hi[:half], a_hi[half:]), dim=0)
    want = base * torch.where(mask_rows > 0, 1.0, 0.2)
    got = dx.to_f32()
    assert rel_err(got, want) < 3e-5, rel_err(got, want)
    d32 = ops.conv_desc(lin, rows4, length, ops.MATH_BF16)
    wp32 = _pack(hip, d32, v, scale, 1)
    dx32 = torch.empty((rows4, spec.c_in, length), dtype=torch.float32, device=DEV)
    check(hip.eben_conv1d_bwd_dx_fm(ctypes.byref(d32), g.data_ptr(), wp32.data_ptr(), a[half:].contiguous().data_ptr(), half, sums.data_ptr(), fm_gs, a.data_ptr(), 0.2,
                                    half, seg_map, dx32.data_ptr(), st), "conv1d_bwd_dx_fm")
    if same and hip.eben_conv1d_kernel_generation(ctypes.byref(d32), 1) == 4 and hip.eben_conv1d_kernel_generation(ctypes.byref(d), 1) == 4:
        # same launch plan, same operands: the accumulators agree bit for bit; the epilogue's feature-matching arithmetic may be
        # contracted differently by the compiler in the two kernels (one ulp of fp32), so: hi + lo within 2^-15 of the fp32 output
        # everywhere, and identical on all but a handful of elements
        split = bf16_hi(dx32) + bf16_hi(dx32 - bf16_hi(dx32))
        assert float((got - dx32).abs().max()) <= 2.0 ** -15 * float(dx32.abs().max())
        assert float((got != split).float().mean()) < 1e-3
    else:
        assert rel_err(got, dx32) < 1e-4


def fm_codes_of(hip, act, half):
    """The feature-matching code plane of an embedding (eben_bl_fm_sums_codes): (half, C / 8, L, 8) bytes, checked here against the signs."""
    from vibravox_amd._lib import check

    codes = torch.zeros((half, act.channels // 8, act.length, 8), dtype=torch.uint8, device=DEV)
    ptrs = (ctypes.c_void_p * 2)(act.hi.data_ptr(), act.lo.data_ptr())
    units = (ctypes.c_int64 * 1)(half * (act.channels // 8) * act.length)
    cp = (ctypes.c_void_p * 1)(codes.data_ptr())
    nbytes = hip.eben_bl_fm_sums_workspace(1)
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=DEV)
    sums = torch.empty(2, dtype=torch.float32, device=DEV)
    check(hip.eben_bl_fm_sums_codes(ptrs, units, cp, 1, ws.data_ptr(), nbytes, sums.data_ptr(), torch.cuda.current_stream().cuda_stream), "bl_fm_sums_codes")
    a = act.to_f32()
    want = (torch.sign(a[:half] - a[half:]) + 1).to(torch.uint8) | ((torch.sign(a[:half]) + 1).to(torch.uint8) << 2)
    got = codes.permute(0, 1, 3, 2).reshape(half, act.channels, act.length)
    assert torch.equal(got, want)
    np.testing.assert_allclose(sums.cpu().numpy(), [float((a[:half] - a[half:]).abs().sum()), float(a[:half].abs().sum())], rtol=2e-5)
    return codes


PR_LAYERS = {
    **{k: MID_LAYERS[k] for k in ("melgan_l1_dense", "melgan_l2")},
    # lengths that are not a multiple of the stride (the last positions exist for some phases only), one more / one fewer output than S q
    "melgan_l1_ragged": (dict(c_in=16, c_out=64, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 4, 2103, False),
    "melgan_l2_ragged": (dict(c_in=64, c_out=256, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 2, 1097, True),
    # the wide layers: rows ordered (channel bundle, phase, channel in bundle) on the persistent whole-panel kernel (generation 6), whole
    # 64-byte position groups per lane pair -- lengths that end inside a group of four, one and four row tiles per group
    "melgan_l3_wide": (dict(c_in=256, c_out=1024, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 2, 1030, True),
    "melgan_l4_wide": (dict(c_in=1024, c_out=1024, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4, out_slope=0.2), 4, 301, True),
    "stride8_k16": (dict(c_in=8, c_out=64, ksize=16, stride=8, pad_l=7, pad_r=7, groups=1, out_slope=0.2), 2, 1001, True),
    "stride4_g2": (dict(c_in=32, c_out=128, ksize=23, stride=4, pad_l=11, pad_r=11, groups=2, out_slope=0.2), 2, 777, True),
    # stride 2 (the PQMF-band layers): bundle-major rows, two bundles per 32-row tile, up to dilation 2 (EBEN_PR2_S2_MAX_DIL); an odd
    # bundle count (l3: 48 primed rows) ends inside a tile; dilation 3 is declined and skipped here
    "pqmf_l1_d1_dense": MID_LAYERS["pqmf_l1_d1_dense"],
    "pqmf_l3_d2": MID_LAYERS["pqmf_l3_d2"],
    "pqmf_l2_d3_dense": MID_LAYERS["pqmf_l2_d3_dense"],
    "pqmf_l4_d3": (dict(c_in=192, c_out=384, ksize=7, stride=2, dilation=3, pad_l=3, pad_r=3, groups=4, out_slope=0.2), 2, 517, True),
}


@pytest.mark.parametrize("name", list(PR_LAYERS))
def test_bundle_conv_input_gradient_phases_as_rows(hip, name):
    """eben_bl_conv1d_bwd_dx_pr: the input gradient of a strided layer as ONE stride-1 contraction whose rows are (phase, channel), stored
    depth-to-space (melgan_discriminator.py:97-130 backward).  Same operands and epilogue as eben_bl_conv1d_bwd_dx (mask from the saved
    embedding, feature-matching term on the first rows): against float64 on the MFMA operands and against the phase-scatter kernel."""
    from vibravox_amd import ops
    from vibravox_amd._lib import EbenConv1dDesc, check
    from vibravox_amd.disc_engine_bl import Planes

    kw, rows2, length, _ = PR_LAYERS[name]
    spec = ops.ConvSpec(**kw)
    wshape = spec.weight_shape()
    v = formula_tensor(f"blpr/{name}/v", wshape, 1 / math.sqrt(wshape[1] * wshape[2])).to(DEV)
    scale = (1 + 0.3 * formula_tensor(f"blpr/{name}/s", (wshape[0],))).to(DEV)
    half = rows2 // 2
    rows4 = 4 * half
    lin = ops.ConvSpec(**{**kw, "out_slope": 1.0})
    l_out = spec.out_len(length)
    g = formula_tensor(f"blpr/{name}/g", (rows4, spec.c_out, l_out)).to(DEV)
    gp = planes_of(g, lo=False)
    act = planes_of(formula_tensor(f"blpr/{name}/act", (rows2, spec.c_in, length)))
    a = act.to_f32()
    sums = torch.tensor([2.5, 7.0], device=DEV)
    fm_gs = 0.41
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
    d = ops.conv_desc(lin, rows4, length, ops.MATH_BF16 | BL)
    dq = EbenConv1dDesc()
    if hip.eben_bl_dx_pr_desc(ctypes.byref(d), ctypes.byref(dq)) != 0 and (spec.stride < 4 or spec.dilation > 1):
        pytest.skip("stride-2 / dilated layers take the phases-as-rows form only under EBEN_PR_MIN_STRIDE=2 EBEN_PR_MAX_DIL=3")
    assert hip.eben_bl_dx_pr_desc(ctypes.byref(d), ctypes.byref(dq)) == 0
    assert (dq.c_in, dq.c_out, dq.stride, dq.l_in, dq.l_out) == (spec.c_out, spec.stride * spec.c_in, 1, l_out, -(-length // spec.stride))
    wq = torch.empty(dq.c_out * (dq.c_in // dq.groups) * dq.ksize, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    check(hip.eben_bl_dx_pr_weights(ctypes.byref(d), v.data_ptr(), scale.data_ptr(), wq.data_ptr(), st), "bl_dx_pr_weights")
    img = torch.empty(hip.eben_conv1d_packed_floats(ctypes.byref(dq), 0), dtype=torch.float32, device=DEV)
    ops.conv1d_pack(dq, wq, None, img, None)
    dx = Planes(rows4, spec.c_in, length, DEV)
    dx.hi.fill_(float("nan")); dx.lo.fill_(float("nan"))
    check(hip.eben_bl_conv1d_bwd_dx_pr(ctypes.byref(d), gp.hi.data_ptr(), img.data_ptr(), act.hi.data_ptr(), act.lo.data_ptr(), 0.2, half, seg_map, half, half,
                                       sums.data_ptr(), fm_gs, dx.hi.data_ptr(), dx.lo.data_ptr(), st), "bl_conv1d_bwd_dx_pr")
    # ... and with the feature-matching code plane in place of act_lo and the reference rows: bit for bit
    dxc = Planes(rows4, spec.c_in, length, DEV)
    dxc.hi.fill_(float("nan")); dxc.lo.fill_(float("nan"))
    codes = fm_codes_of(hip, act, half)
    check(hip.eben_bl_conv1d_bwd_dx_pr_c(ctypes.byref(d), gp.hi.data_ptr(), img.data_ptr(), act.hi.data_ptr(), act.lo.data_ptr(), codes.data_ptr(), 0.2, half,
                                         seg_map, half, half, sums.data_ptr(), fm_gs, dxc.hi.data_ptr(), dxc.lo.data_ptr(), st), "bl_conv1d_bwd_dx_pr_c")
    assert torch.equal(dxc.hi.view(torch.int16), dx.hi.view(torch.int16)) and torch.equal(dxc.lo.view(torch.int16), dx.lo.view(torch.int16))
    w = (v * scale.reshape(-1, 1, 1))
    base = F.conv_transpose1d(bf16_hi(g).double(), bf16_hi(w).double(), stride=spec.stride, padding=spec.pad_l, dilation=spec.dilation, groups=spec.groups,
                              output_padding=length - ((l_out - 1) * spec.stride - 2 * spec.pad_l + spec.dilation * (spec.ksize - 1) + 1))
    ad = a.double()
    base[:half] += fm_gs * (torch.sign(ad[:half] - ad[half:]) / 7.0 - 2.5 * torch.sign(ad[:half]) / 49.0)
    a_hi = act.hi.permute(0, 1, 3, 2).reshape(rows2, spec.c_in, length).double()
    mask_rows = torch.cat((a_hi[:half], a_hi[:half], a_hi[:half], a_hi[half:]), dim=0)
    want = base * torch.where(mask_rows > 0, 1.0, 0.2)
    got = dx.to_f32()
    assert rel_err(got, want) < 3e-5, rel_err(got, want)
    # the phase-scatter kernel on the same operands: same products, another order of the fp32 accumulation
    wp = _pack(hip, d, v, scale, 1)
    dx1 = Planes(rows4, spec.c_in, length, DEV)
    check(hip.eben_bl_conv1d_bwd_dx(ctypes.byref(d), gp.hi.data_ptr(), wp.data_ptr(), act.hi.data_ptr(), act.lo.data_ptr(), 0.2, half, seg_map, half, half,
                                    sums.data_ptr(), fm_gs, dx1.hi.data_ptr(), dx1.lo.data_ptr(), st), "bl_conv1d_bwd_dx")
    assert rel_err(got, dx1.to_f32()) < 2e-5


def test_phases_as_rows_declines_what_it_does_not_cover(hip):
    from vibravox_amd import ops
    from vibravox_amd._lib import EbenConv1dDesc

    dq = EbenConv1dDesc()
    for kw in (dict(c_in=96, c_out=192, ksize=7, stride=2, dilation=3, pad_l=3, pad_r=3, groups=4),     # dilation 3 (10 primed taps for 4 + 3)
               dict(c_in=768, c_out=768, ksize=5, stride=1, pad_l=2, pad_r=2, groups=4),                # not strided
               dict(c_in=12, c_out=64, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4),               # channels not in bundles
               dict(c_in=256, c_out=1024, ksize=16, stride=8, pad_l=7, pad_r=7, groups=4)):             # wide layers (full row tiles either way) at stride 4 only
        d = ops.conv_desc(ops.ConvSpec(**kw), 4, 400, ops.MATH_BF16 | BL)
        assert hip.eben_bl_dx_pr_desc(ctypes.byref(d), ctypes.byref(dq)) != 0
    d = ops.conv_desc(ops.ConvSpec(c_in=64, c_out=256, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4), 4, 400, ops.MATH_BF16)   # fp32 at rest
    assert hip.eben_bl_dx_pr_desc(ctypes.byref(d), ctypes.byref(dq)) != 0


@pytest.mark.parametrize("name", list(MID_LAYERS))
def test_bundle_conv_weight_gradient(hip, name):
    """dW / dbias from bundle planes (bl_dw.hip: reduction along time, LDS-DMA tiles, transposing reads) against float64 on the bf16
    operands, through the slab reduction with the bundle-major column order."""
    from vibravox_amd import ops
    from vibravox_amd._lib import check

    spec, rows, length, same, v, scale, bias = _layer_setup(name)
    lin = ops.ConvSpec(**{**MID_LAYERS[name][0], "out_slope": 1.0})
    l_out = spec.out_len(length)
    dy = formula_tensor(f"blm/{name}/dy", (rows, spec.c_out, l_out)).to(DEV)
    x = formula_tensor(f"blm/{name}/xw", (rows, spec.c_in, length)).to(DEV)
    dyp, xp = planes_of(dy, lo=False), planes_of(x, lo=False)
    d = ops.conv_desc(lin, rows, length, ops.MATH_BF16 | BL)
    nslab, rs, perm = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    nbytes = hip.eben_bl_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(rs), ctypes.byref(perm))
    assert nbytes > 0
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    check(hip.eben_bl_conv1d_bwd_dw(ctypes.byref(d), dyp.hi.data_ptr(), xp.hi.data_ptr(), 1, slabs.data_ptr(), nbytes, st), "bl_conv1d_bwd_dw")
    wshape = spec.weight_shape()
    dv, dbias = torch.empty(wshape, dtype=torch.float32, device=DEV), torch.empty(wshape[0], dtype=torch.float32, device=DEV)
    ops.wn_bwd_multi([(slabs, nslab.value, wshape[0] * rs.value, wshape[0], wshape[1] * wshape[2], rs.value, None, dv, None, None, dv, dbias, perm.value)])
    wr = torch.zeros(wshape, dtype=torch.float64, device=DEV).requires_grad_(True)
    br = torch.zeros(wshape[0], dtype=torch.float64, device=DEV).requires_grad_(True)
    out = F.conv1d(bf16_hi(x).double(), wr, br, stride=spec.stride, padding=spec.pad_l, dilation=spec.dilation, groups=spec.groups)
    (out * bf16_hi(dy).double()).sum().backward()
    assert rel_err(dv, wr.grad) < 3e-5, rel_err(dv, wr.grad)
    assert rel_err(dbias, br.grad) < 3e-5


@pytest.mark.parametrize("shape", ["pqmf_l1", "pqmf_l4", "pqmf_l6", "mixed"])
def test_weight_gradients_of_three_chains_in_one_launch(hip, shape):
    """eben_bl_conv1d_bwd_dw_multi: the same layer index of the three PQMF-band discriminators (dilation 1 / 2 / 3, their own lengths,
    operands and slabs) as one launch -- slab for slab the bytes of three eben_bl_conv1d_bwd_dw launches.  "mixed": problems of different
    tile shapes in one call fall apart into one launch per run of equal shapes."""
    from vibravox_amd import ops
    from vibravox_amd._lib import check

    kws = {"pqmf_l1": [dict(c_in=24, c_out=48, ksize=7, stride=2, dilation=d, pad_l=3, pad_r=3, groups=4) for d in (1, 2, 3)],
           "pqmf_l4": [dict(c_in=192, c_out=384, ksize=7, stride=2, dilation=d, pad_l=3, pad_r=3, groups=4) for d in (1, 2, 3)],
           "pqmf_l6": [dict(c_in=768, c_out=768, ksize=5, stride=1, dilation=d, pad_l=2, pad_r=2, groups=4) for d in (1, 2, 3)],
           "mixed": [dict(c_in=24, c_out=48, ksize=7, stride=2, dilation=1, pad_l=3, pad_r=3, groups=4),
                     dict(c_in=24, c_out=48, ksize=7, stride=2, dilation=2, pad_l=3, pad_r=3, groups=4),
                     dict(c_in=64, c_out=256, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4)]}[shape]
    rows = 4
    lengths = [1003, 997, 1001] if shape != "pqmf_l6" else [140, 134, 129]
    st = torch.cuda.current_stream().cuda_stream
    k = len(kws)
    descs = (ctypes.POINTER(ops.EbenConv1dDesc) * k)()
    dys, xs, sl = (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)()
    nbs = (ctypes.c_size_t * k)()
    keep, single = [], []
    for j, (kw, length) in enumerate(zip(kws, lengths)):
        spec = ops.ConvSpec(**kw)
        l_out = spec.out_len(length)
        dy = planes_of(formula_tensor(f"blmulti/{shape}/{j}/dy", (rows, spec.c_out, l_out)).to(DEV), lo=False)
        x = planes_of(formula_tensor(f"blmulti/{shape}/{j}/x", (rows, spec.c_in, length)).to(DEV), lo=False)
        d = ops.conv_desc(spec, rows, length, ops.MATH_BF16 | BL)
        nbytes = hip.eben_bl_conv1d_bwd_dw_workspace(ctypes.byref(d), None, None, None)
        assert nbytes > 0
        one = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=DEV)
        check(hip.eben_bl_conv1d_bwd_dw(ctypes.byref(d), dy.hi.data_ptr(), x.hi.data_ptr(), 1, one.data_ptr(), nbytes, st), "bl_conv1d_bwd_dw")
        many = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=DEV)
        descs[j] = ctypes.pointer(d)
        dys[j], xs[j], sl[j], nbs[j] = dy.hi.data_ptr(), x.hi.data_ptr(), many.data_ptr(), nbytes
        keep.append((d, dy, x, many))
        single.append(one)
    check(hip.eben_bl_conv1d_bwd_dw_multi(descs, dys, xs, 1, sl, nbs, k, st), "bl_conv1d_bwd_dw_multi")
    torch.cuda.synchronize()
    for one, (_, _, _, many) in zip(single, keep):
        # the slabs hold every weight's partial sums; columns no weight owns stay what they were (NaN here, both ways)
        assert torch.equal(torch.nan_to_num(one, nan=-7.0), torch.nan_to_num(many, nan=-7.0))


@pytest.mark.parametrize("kw,length", [(dict(c_in=16, c_out=32, ksize=5, stride=1, pad_l=2, pad_r=2), 300),       # 64-row tile, 128 columns
                                       (dict(c_in=8, c_out=16, ksize=3, stride=1, pad_l=1, pad_r=1), 257),        # 64-row tile, 64 columns
                                       (dict(c_in=16, c_out=128, ksize=5, stride=2, pad_l=2, pad_r=2), 301),      # 128-row tile, 128 columns
                                       (dict(c_in=8, c_out=128, ksize=3, stride=1, dilation=2, pad_l=2, pad_r=2), 190)])   # 128-row tile, 64 columns
def test_bundle_conv_weight_gradient_small_tiles(hip, kw, length):
    """The 64- and 128-column tiles of bl_dw.hip (layers with few column bundles: none of EBEN's, which take the 192- / 256-column ones)
    against float64 on the bf16 operands."""
    from vibravox_amd import ops
    from vibravox_amd._lib import check

    spec = ops.ConvSpec(**kw)
    rows = 4
    l_out = spec.out_len(length)
    tag = f"bldw_small/{spec.c_in}_{spec.c_out}_{spec.ksize}"
    dy = formula_tensor(tag + "/dy", (rows, spec.c_out, l_out)).to(DEV)
    x = formula_tensor(tag + "/x", (rows, spec.c_in, length)).to(DEV)
    dyp, xp = planes_of(dy, lo=False), planes_of(x, lo=False)
    d = ops.conv_desc(spec, rows, length, ops.MATH_BF16 | BL)
    nslab, rs, perm = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    nbytes = hip.eben_bl_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(rs), ctypes.byref(perm))
    assert nbytes > 0
    slabs = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    check(hip.eben_bl_conv1d_bwd_dw(ctypes.byref(d), dyp.hi.data_ptr(), xp.hi.data_ptr(), 1, slabs.data_ptr(), nbytes, st), "bl_conv1d_bwd_dw")
    wshape = spec.weight_shape()
    dv, dbias = torch.empty(wshape, dtype=torch.float32, device=DEV), torch.empty(wshape[0], dtype=torch.float32, device=DEV)
    ops.wn_bwd_multi([(slabs, nslab.value, wshape[0] * rs.value, wshape[0], wshape[1] * wshape[2], rs.value, None, dv, None, None, dv, dbias, perm.value)])
    wr = torch.zeros(wshape, dtype=torch.float64, device=DEV).requires_grad_(True)
    br = torch.zeros(wshape[0], dtype=torch.float64, device=DEV).requires_grad_(True)
    out = F.conv1d(bf16_hi(x).double(), wr, br, stride=spec.stride, padding=spec.pad_l, dilation=spec.dilation, groups=spec.groups)
    (out * bf16_hi(dy).double()).sum().backward()
    assert rel_err(dv, wr.grad) < 3e-5, rel_err(dv, wr.grad)
    assert rel_err(dbias, br.grad) < 3e-5


@pytest.mark.parametrize("kw,rows,length,same_split", [
    (dict(c_in=64, c_out=256, ksize=41, stride=4, pad_l=20, pad_r=20, groups=4), 8, 1203, True),      # 64-row tiles: contiguous X rows only
    (dict(c_in=256, c_out=512, ksize=41, stride=4, pad_l=20, pad_r=20, groups=2), 6, 610, False),     # 128-row tiles: + the eight-wave tile
])
def test_stride4_weight_gradient_forms_agree(hip, kw, rows, length, same_split):
    """bl_dw.hip's stride-4 forms (X rows copied contiguously and de-interleaved by the fragment addresses; 2 x 4 waves on one A tile)
    against the phase-row 2 x 2 form (EBEN_BLDW_XC=0 EBEN_BLDW_WIDE=0 in a second process): the same products in the same order -- bit for
    bit where the split-K factor is the same, to the slab sum's rounding where the wider tile changes it."""
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, ctypes, torch; sys.path.insert(0, %r)\n"
            "from vibravox_amd import ops\n"
            "from vibravox_amd._lib import check, load\n"
            "lib = load(); d = torch.device('cuda'); rows, length = %d, %d\n"
            "spec = ops.ConvSpec(**%r); lo = spec.out_len(length)\n"
            "g = torch.Generator().manual_seed(11)\n"
            "dy = torch.randn(rows, spec.c_out // 8, lo, 8, generator=g).bfloat16().to(d)\n"
            "x = torch.randn(rows, spec.c_in // 8, length, 8, generator=g).bfloat16().to(d)\n"
            "desc = ops.conv_desc(spec, rows, length, ops.MATH_BF16 | 0x100)\n"
            "ns, rs, pk = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)\n"
            "nb = lib.eben_bl_conv1d_bwd_dw_workspace(ctypes.byref(desc), ctypes.byref(ns), ctypes.byref(rs), ctypes.byref(pk))\n"
            "slabs = torch.zeros(nb // 4, dtype=torch.float32, device=d)\n"
            "check(lib.eben_bl_conv1d_bwd_dw(ctypes.byref(desc), dy.data_ptr(), x.data_ptr(), 1, slabs.data_ptr(), nb, torch.cuda.current_stream().cuda_stream), 'dw')\n"
            "torch.cuda.synchronize(); torch.save((slabs.view(ns.value, -1).cpu(), ns.value), sys.argv[1])\n" % (root, rows, length, kw))
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for k, env in enumerate(({}, {"EBEN_BLDW_XC": "0", "EBEN_BLDW_WIDE": "0"})):
            path = os.path.join(td, f"o{k}.pt")
            subprocess.run([sys.executable, "-c", code, path], check=True, env={**os.environ, **env}, timeout=300)
            outs.append(torch.load(path))
    (a, na), (b, nb_) = outs
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    if same_split:
        assert na == nb_ and torch.equal(a, b)
    else:
        sa, sb = a.double().sum(0), b.double().sum(0)
        assert float((sa - sb).abs().max()) <= 1e-5 * float(sb.abs().max())
