#!/usr/bin/env python
"""Per-layer micro-benchmark of the discriminator's kernels in the bf16 bundle layout (plan "bf16_bl") at BASELINE config-2 shapes,
with the MIXED roofline of every launch: roof = max(FLOP / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s).

For every layer of the four sub-discriminators, stand-alone (one launch at a time, HIP events over `--iters` launches):
  fwd   2B rows (enhanced + reference), PQMF-band chains on hi + lo operands (three MFMAs per product), MelGAN on single bf16;
  dx    4B stacked rows [fm | adv | fake | real] with the engine's epilogue (mask + feature-matching term from the saved embedding);
  dw    2B rows [fake | real] against [enhanced | reference], slab reduction + weight-norm chain rule included.
Columns: ms, TFLOP/s, algorithmic MB, bound (mfma / hbm), roof us, fraction of the roof.
Usage: python tools/layer_bench_bl.py [--batch 32] [--iters 10] [--filter melgan]
"""
import argparse
import ctypes
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vibravox_amd import ops  # noqa: E402
from vibravox_amd._lib import EbenBlHeadJob, check, load, ptr, stream  # noqa: E402
from vibravox_amd.disc_engine_bl import BL, FM_CODES, Planes, _ChainBL, _fm_sums  # noqa: E402
from vibravox_amd.lightning_modules.eben import DISC_MATH_PLANS  # noqa: E402
from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales  # noqa: E402

PEAK, HBM = 2.5e15, 8.0e12


def time_ms(fn, iters):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cell(ms, flops, nbytes):
    roof = max(flops / PEAK, nbytes / HBM)
    bound = "mfma" if flops / PEAK >= nbytes / HBM else "hbm"
    return f"{ms:7.3f} {flops / (ms * 1e-3) / 1e12:6.0f} {nbytes / 1e6:7.1f} {bound:>4s} {roof * 1e6:6.1f} {roof / (ms * 1e-3):5.2f}", roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32, help="clips per step (the forward runs 2x, the stacked input gradients 4x this many rows)")
    ap.add_argument("--length", type=int, default=31968)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--filter", default="")
    ap.add_argument("--only", default="", help="one layer, e.g. melgan.4 (the PMC passes of tools/pmc_family_bl.sh)")
    ap.add_argument("--no-fm-rows", action="store_true", help="time the stacked input gradients without feature-matching rows (what those rows' four operand loads cost)")
    a = ap.parse_args()
    lib = load()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    disc = DiscriminatorEBENMultiScales(q=4, min_channels=24).to(dev)
    plan = DISC_MATH_PLANS["bf16_bl"]
    half = a.batch
    r2, r4 = 2 * half, 4 * half
    st = stream()
    seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
    sums = torch.tensor([3.0, 7.0], device=dev)
    hdr = f"{'layer':44s} | {'fwd ms':>7s} {'TF':>6s} {'MB':>7s} {'bnd':>4s} {'roof':>6s} {'frac':>5s} | {'dx ms':>7s} {'TF':>6s} {'MB':>7s} {'bnd':>4s} {'roof':>6s} {'frac':>5s} | {'dw ms':>7s} {'TF':>6s} {'MB':>7s} {'bnd':>4s} {'roof':>6s} {'frac':>5s}"
    print(hdr)
    tot = {"fwd": [0.0, 0.0], "dx": [0.0, 0.0], "dw": [0.0, 0.0]}
    chains = [("pqmf%d" % i, _ChainBL(d.discriminator, plan["pqmf"]), 4, a.length // 4) for i, d in enumerate(disc.pqmf_discriminators)]
    chains.append(("melgan", _ChainBL(disc.melgan_discriminator.discriminator, plan["melgan"]), 1, a.length))
    for cname, ch, c_in, l_in in chains:
        if (a.filter and a.filter not in cname) or (a.only and a.only.split(".")[0] != cname):
            continue
        n = len(ch.layers)
        x_in = torch.randn(r2, c_in, l_in, device=dev) * 0.1
        act0 = Planes(r2, ch.layers[0].spec.c_out, ch.head_out_len(l_in), dev)
        # ---- head
        sp = ch.layers[0].spec
        job_f = (EbenBlHeadJob * 1)(ch.head_job(x_in, l_in, act0))
        g0 = Planes.from_f32(torch.randn(r4, sp.c_out, act0.length, device=dev), lo=True)
        g0v = Planes.__new__(Planes)
        g0v.codes = None
        g0v.hi, g0v.lo, g0v.rows, g0v.channels, g0v.length = g0.hi[:r2], g0.lo[:r2], r2, g0.channels, g0.length
        job_b = (EbenBlHeadJob * 1)(ch.head_job(None, l_in, g0v))
        dxh = torch.empty_like(x_in)
        skip = bool(a.only) and a.only != cname + ".0"
        t_f = 1.0 if skip else time_ms(lambda: check(lib.eben_bl_head_fwd(job_f, 1, r2, st)), a.iters)
        t_b = 1.0 if skip else time_ms(lambda: check(lib.eben_bl_head_dx(job_b, 1, r2, ptr(dxh), st)), a.iters)
        t_w = 1.0 if skip else time_ms(lambda: ch.weight_grads([], x_in, g0.rows_slice(r2, r4), half), a.iters)
        macs2 = r2 * sp.c_out * sp.ksize * act0.length
        el = sp.c_out * act0.length
        cf, rf = cell(t_f, 2.0 * macs2, r2 * (4 * c_in * l_in + 4 * el))
        cb, rb = cell(t_b, 2.0 * macs2, r2 * (4 * c_in * l_in + 4 * el))
        cw, rw = cell(t_w, 2.0 * macs2, r2 * (4 * c_in * l_in + 2 * el))
        print(f"{cname + '.0 head ' + str(sp.c_in) + '->' + str(sp.c_out) + ' k' + str(sp.ksize):44s} | {cf} | {cb} | {cw}")
        for k, (t, r) in zip(tot, ((t_f, rf), (t_b, rb), (t_w, rw))):
            tot[k][0] += t
            tot[k][1] += r
        # ---- tap-conv layers
        acts = [act0]
        cur = act0
        for i in range(1, n - 1):
            lay = ch.layers[i]
            sp = lay.spec
            d = ops.conv_desc(sp, r2, cur.length, lay.math_fwd)
            y = Planes(r2, sp.c_out, d.l_out, dev)
            xin = Planes.from_f32(torch.randn(r2, sp.c_in, cur.length, device=dev))
            split = (lay.math_fwd & 0xff) == ops.MATH_BF16X3
            if a.only and a.only != f"{cname}.{i}":
                acts.append(y)
                cur = y
                continue
            _, _, bias = lay.params()
            wp = lay.packed(0, r2, cur.length)
            t_f = time_ms(lambda: check(lib.eben_bl_conv1d_fwd(ctypes.byref(d), xin.hi.data_ptr(), xin.lo.data_ptr() if split else None, ptr(wp), ptr(bias),
                                                               y.hi.data_ptr(), y.lo.data_ptr(), st)), a.iters)
            # the stacked input gradients as the engine launches them since round 6: two passes of 2B rows (rows [fm | adv] with the
            # feature-matching term, then rows [fake | real]); the column is their sum (rounds 1-5: one 4B-row launch)
            d4 = ops.conv_desc(lay.spec_lin, r2, cur.length, lay.math_dx)
            g = Planes.from_f32(torch.randn(r4, sp.c_out, d.l_out, device=dev), lo=False)
            gp = Planes(r4, sp.c_in, cur.length, dev, lo=False)
            pr = lay.pr_desc(r2, cur.length) is not None   # strided MelGAN layers: the phases-as-rows form (as the engine launches it)
            wpb = lay.packed(2 if pr else 1, r2, cur.length)
            dx_fn = lib.eben_bl_conv1d_bwd_dx_pr if pr else lib.eben_bl_conv1d_bwd_dx
            fm_rows = 0 if a.no_fm_rows else half
            # as the engine launches it: the feature-matching code plane of the embedding (written by its sums pass) instead of three operand planes
            codes = None
            if FM_CODES and fm_rows:
                _fm_sums(lib, [xin], half, torch.empty(2, device=dev))
                codes = xin.codes.data_ptr()
            dx_fn = lib.eben_bl_conv1d_bwd_dx_pr_c if pr else lib.eben_bl_conv1d_bwd_dx_c
            seg_gen, seg_disc = (ctypes.c_int * 4)(0, 0, 0, 0), (ctypes.c_int * 4)(0, 1, 0, 1)

            def two_passes():
                check(dx_fn(ctypes.byref(d4), g.hi.data_ptr(), ptr(wpb), xin.hi.data_ptr(), xin.lo.data_ptr(), codes, 0.2, half, seg_gen,
                            fm_rows, half, ptr(sums), 0.1, gp.hi.data_ptr(), None, st))
                check(dx_fn(ctypes.byref(d4), g.hi[r2:].data_ptr(), ptr(wpb), xin.hi.data_ptr(), xin.lo.data_ptr(), None, 0.2, half, seg_disc,
                            0, half, ptr(sums), 0.1, gp.hi[r2:].data_ptr(), None, st))
            t_b = time_ms(two_passes, a.iters)
            t_w = time_ms(lambda: ch.weight_grads([(i, g.rows_slice(r2, r4), xin)], x_in, None, half), a.iters)
            wshape = sp.weight_shape()
            wel = wshape[0] * wshape[1] * wshape[2]
            macs1 = wel * d.l_out                      # per batch row
            nin, nout = sp.c_in * cur.length, sp.c_out * d.l_out
            cf, rf = cell(t_f, 2.0 * r2 * macs1 * (3 if split else 1), r2 * ((4 if split else 2) * nin + 4 * nout) + (4 if split else 2) * wel)
            # dx: gradient in (2 B), mask (2 B; + lo and the reference rows' hi / lo on the B feature-matching rows), gradient out (2 B)
            cb, rb = cell(t_b, 2.0 * r4 * macs1, r4 * 2 * nout + r4 * 2 * nin + half * 6 * nin + r4 * 2 * nin + 2 * wel)
            cw, rw = cell(t_w, 2.0 * r2 * macs1, r2 * 2 * (nin + nout) + 4 * wel * 2)
            print(f"{cname + '.' + str(i) + ' ' + str(sp.c_in) + '->' + str(sp.c_out) + ' k' + str(sp.ksize) + ' s' + str(sp.stride) + ' d' + str(sp.dilation) + ' g' + str(sp.groups) + ' L' + str(cur.length):44s} | {cf} | {cb} | {cw}")
            for k, (t, r) in zip(tot, ((t_f, rf), (t_b, rb), (t_w, rw))):
                tot[k][0] += t
                tot[k][1] += r
            acts.append(y)
            cur = y
        if a.only and a.only != f"{cname}.{n - 1}":
            continue
        # ---- tail
        tail = ch.layers[-1]
        sp = tail.spec
        v, _, bias = tail.params()
        tail.ensure_scale()
        xin = Planes.from_f32(torch.randn(r2, cur.channels, cur.length, device=dev))
        logits = torch.empty((r2, 1, cur.length), device=dev)
        seeds = torch.randn(r4, 1, cur.length, device=dev)
        gt = Planes(r4, cur.channels, cur.length, dev, lo=False)
        t_f = time_ms(lambda: check(lib.eben_bl_tail_fwd(xin.hi.data_ptr(), xin.lo.data_ptr(), r2, cur.channels, cur.length, sp.ksize, sp.pad_l, ptr(v.detach()),
                                                         ptr(tail.scale), ptr(bias.detach()), 1.0, ptr(logits), st)), a.iters)
        seg_gen, seg_disc = (ctypes.c_int * 4)(0, 0, 0, 0), (ctypes.c_int * 4)(0, 1, 0, 1)

        def tail_two_passes():
            check(lib.eben_bl_tail_dx(ptr(seeds), r2, cur.channels, cur.length, sp.ksize, sp.pad_l, ptr(v.detach()), ptr(tail.scale), xin.hi.data_ptr(),
                                      xin.lo.data_ptr(), 0.2, half, seg_gen, half, half, ptr(sums), 0.1, gt.hi.data_ptr(), None, st))
            check(lib.eben_bl_tail_dx(ptr(seeds[r2:]), r2, cur.channels, cur.length, sp.ksize, sp.pad_l, ptr(v.detach()), ptr(tail.scale), xin.hi.data_ptr(),
                                      xin.lo.data_ptr(), 0.2, half, seg_disc, 0, half, ptr(sums), 0.1, gt.hi[r2:].data_ptr(), None, st))
        t_b = time_ms(tail_two_passes, a.iters)
        t_w = time_ms(lambda: ch.weight_grads([(n - 1, seeds[r2:], xin)], x_in, None, half), a.iters)
        nin = cur.channels * cur.length
        macs1 = nin * sp.ksize
        cf, rf = cell(t_f, 2.0 * r2 * macs1, r2 * 4 * nin)
        cb, rb = cell(t_b, 2.0 * r4 * macs1, r4 * 2 * nin + r4 * 2 * nin + half * 6 * nin)
        cw, rw = cell(t_w, 2.0 * r2 * macs1, r2 * 4 * nin)
        print(f"{cname + '.' + str(n - 1) + ' tail ' + str(cur.channels) + '->1 k' + str(sp.ksize) + ' L' + str(cur.length):44s} | {cf} | {cb} | {cw}")
        for k, (t, r) in zip(tot, ((t_f, rf), (t_b, rb), (t_w, rw))):
            tot[k][0] += t
            tot[k][1] += r
    print("TOTAL " + "  ".join(f"{k} {v[0]:.3f} ms (roof {v[1] * 1e3:.3f} ms, {v[1] * 1e3 / v[0]:.2f})" for k, v in tot.items()))


if __name__ == "__main__":
    main()
