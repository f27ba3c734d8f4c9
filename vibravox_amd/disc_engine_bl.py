"""The batched discriminator passes of ``disc_engine.DiscriminatorEngine`` with every embedding and every stacked gradient at rest in
the bf16 BUNDLE LAYOUT (``include/eben_hip.h``: bf16 ``[rows][channels / 8][length][8]``, planes hi = bf16(v) and lo = bf16(v - hi)).

Same step, same seeds, same arithmetic plan as the fp32-at-rest engine with bf16 contractions (``DISC_MATH_PLANS["bf16"]``): the values the
MFMAs see are the values they saw there -- the same round-to-nearest-even of the same fp32 number, applied by the producer's epilogue
instead of by every consumer's tile staging.  What changes is the traffic and the instruction stream around the contractions:

  * the tap-conv's input tile is a copy of 16-byte units (``tapconv3.hip``, flag BL): an eighth of the loads, no conversion;
  * its epilogue reads the LeakyReLU mask / the feature-matching operands and writes its output as 8-byte halves of units;
  * the weight gradients read both operands as they are (``bl_dw.hip``: LDS-DMA tiles, transposing LDS reads) -- no ``dw3_pack_a``
    pre-pass, no fp32 -> bf16 staging;
  * the two ends of every chain are streaming kernels of their own (``bl_edge.hip``): the heads of the three PQMF-band chains
    (``eben_discriminator.py:66-76``) run as ONE launch forward and ONE backward (they read the same bands, their input gradients
    are summed in the launch), the logits layers (``eben_discriminator.py:150-157``, ``melgan_discriminator.py:147-156``) read
    bundles and produce / consume fp32 logits.

The feature-matching sums and gradient read hi + lo (16 mantissa bits, ~2^-17 relative: the loss value moves by ~1e-6, a sign only
where two embeddings agree to 16 bits).
"""
from __future__ import annotations

import ctypes
import dataclasses
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import EbenBlHeadJob, check, load, ptr
from ._lib import stream as _stream
from .disc_engine import DiscriminatorEngine, _Layer

BL = 0x100   # EBEN_LAYOUT_BL
#: split forward (forward_reference): how much of MelGAN's reference half runs underneath the generator forward.  "0": none (MelGAN whole,
#: behind the generator: rounds 4-5); "1": all of it ([MI355X, same box, 3 x alternated, 100 steps] 9.03 / 9.08 / 8.99 -> 8.96 / 9.03 /
#: 8.94 ms per step, 8.90 / 8.92 / 8.88 together with EBEN_D_BWD_SPREAD=0 -- but layers 3-5 then run as two 32-row launches at 0.26 of
#: the MFMA peak instead of one at 0.47); "thin" (default): its head and thin layers only
SPLIT_MELGAN = __import__("os").environ.get("EBEN_SPLIT_MELGAN", "thin")
#: "thin": MelGAN's head and layers 1 .. MELGAN_SPLIT_DEPTH - 1 (the HBM-bound ones) run per half like the PQMF-band chains; the MFMA-bound
#: layers behind them keep ONE 2B-row launch (the persistent tile kernel fills the machine exactly once at 64 rows)
MELGAN_SPLIT_DEPTH = int(__import__("os").environ.get("EBEN_SPLIT_MELGAN_DEPTH", "3"))
#: the four stacked seed blocks of a chain's backward from one launch (eben_hinge_bwd_stacked); 0: a memset + three launches
STACKED_SEEDS = __import__("os").environ.get("EBEN_STACKED_SEEDS", "1") != "0"
#: the feature-matching rows of the stacked input gradients read a one-byte code plane of each embedding (signs of a - r and of a, written by
#: the feature-matching sums pass) instead of three more operand planes; 0: the four-operand form (bit-identical results either way)
FM_CODES = __import__("os").environ.get("EBEN_FM_CODES", "1") != "0"


def _fm_sums(lib, acts, half: int, sums_out: torch.Tensor) -> None:
    """eben_bl_fm_sums(_codes) over the embeddings `acts` (2 half rows each: enhanced rows, then reference rows) into `sums_out`."""
    k = len(acts)
    ptrs = (ctypes.c_void_p * (2 * k))()
    units = (ctypes.c_int64 * k)()
    codes = (ctypes.c_void_p * k)()
    for j, pl in enumerate(acts):
        ptrs[2 * j], ptrs[2 * j + 1] = _addr(pl.hi), _addr(pl.lo)
        units[j] = half * (pl.channels // 8) * pl.length
        if FM_CODES:
            if pl.codes is None or pl.codes.shape[0] != half:
                pl.codes = torch.empty((half, pl.channels // 8, pl.length, 8), dtype=torch.uint8, device=pl.hi.device)
            codes[j] = pl.codes.data_ptr()
    ws_bytes = lib.eben_bl_fm_sums_workspace(k)
    ws = torch.empty(max(1, (ws_bytes + 3) // 4), dtype=torch.float32, device=acts[0].hi.device)
    if FM_CODES:
        check(lib.eben_bl_fm_sums_codes(ptrs, units, codes, k, ptr(ws), ws_bytes, ptr(sums_out), _stream()), "bl_fm_sums_codes")
    else:
        check(lib.eben_bl_fm_sums(ptrs, units, k, ptr(ws), ws_bytes, ptr(sums_out), _stream()), "bl_fm_sums")


def _addr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address of a bundle plane (bf16 tensor (rows, channels / 8, length, 8)) or of a view of its leading rows."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and t.dtype is torch.bfloat16
    return t.data_ptr()


class Planes:
    """hi / lo planes of one (rows, channels, length) tensor in the bundle layout."""

    __slots__ = ("hi", "lo", "rows", "channels", "length", "codes")

    def __init__(self, rows: int, channels: int, length: int, device, lo: bool = True):
        assert channels % 8 == 0
        self.rows, self.channels, self.length = rows, channels, length
        #: feature-matching code plane of an embedding (one byte per element of its first rows / 2 rows: the signs the stacked input
        #: gradients' feature-matching term needs; written by eben_bl_fm_sums_codes, attached by the engine)
        self.codes = None
        self.hi = torch.empty((rows, channels // 8, length, 8), dtype=torch.bfloat16, device=device)
        self.lo = torch.empty_like(self.hi) if lo else None

    def rows_slice(self, r0: int, r1: int) -> "Planes":
        """Rows [r0, r1) of the same storage (the leading dimension: still contiguous)."""
        v = Planes.__new__(Planes)
        v.rows, v.channels, v.length = r1 - r0, self.channels, self.length
        v.codes = None
        v.hi = self.hi[r0:r1]
        v.lo = None if self.lo is None else self.lo[r0:r1]
        return v

    def to_f32(self) -> torch.Tensor:
        """(rows, channels, length) fp32 = hi + lo (tests / tools)."""
        out = torch.empty((self.rows, self.channels, self.length), dtype=torch.float32, device=self.hi.device)
        check(load().eben_bl_to_f32(_addr(self.hi), _addr(self.lo), self.rows, self.channels, self.length, ptr(out), _stream()), "bl_to_f32")
        return out

    @staticmethod
    def from_f32(x: torch.Tensor, lo: bool = True) -> "Planes":
        x = x.contiguous()
        p = Planes(x.shape[0], x.shape[1], x.shape[2], x.device, lo)
        check(load().eben_bl_from_f32(ptr(x), p.rows, p.channels, p.length, _addr(p.hi), _addr(p.lo), _stream()), "bl_from_f32")
        return p


class _ChainBL:
    """A sub-discriminator in the bundle layout: head (ReflectionPad1d + layer 0), tap-conv layers 1 .. n-2, tail (logits layer)."""

    def __init__(self, modules, math):
        self.layers: List[_Layer] = []
        self.pad = 0
        convs = []
        for m in modules:
            if isinstance(m, torch.nn.Sequential):
                for sub in m:
                    if hasattr(sub, "padding") and not hasattr(sub, "spec"):
                        self.pad = int(sub.padding)
                    else:
                        convs.append(sub)
            else:
                convs.append(m)
        n = len(convs)
        for i, conv in enumerate(convs):
            mth = math(i, n) if callable(math) else math
            mth = (mth, mth, mth) if isinstance(mth, int) else tuple(mth)
            lay = _Layer(conv, tuple(v | BL for v in mth))
            lay.keep_scale = i == 0 or i == n - 1   # head / tail: no packed image, the kernels take (v, scale)
            self.layers.append(lay)
        head, tail = self.layers[0].spec, self.layers[-1].spec
        if not (head.groups == head.c_in and head.stride == 1 and head.pad_l == head.pad_r and head.c_out % 8 == 0 and head.ksize <= 16 and head.c_in <= 4):
            raise ops._lib.EbenError("bundle-layout engine: unexpected chain head")
        if not (tail.c_out == 1 and tail.groups == 1 and tail.stride == 1 and tail.dilation == 1 and tail.ksize <= 8 and tail.pad_l == tail.pad_r):
            raise ops._lib.EbenError("bundle-layout engine: unexpected logits layer")

    # ---- pieces the engine assembles into multi-job launches -------------------------------------------------------------------
    def head_job(self, x: Optional[torch.Tensor], l_in: int, out: Planes) -> EbenBlHeadJob:
        lay = self.layers[0]
        sp = lay.spec
        v, _, bias = lay.params()
        lay.ensure_scale()
        j = EbenBlHeadJob()
        j.x, j.v, j.scale, j.bias = ptr(x) if x is not None else None, ptr(v.detach()), ptr(lay.scale), ptr(bias.detach()) if bias is not None else None
        j.y_hi, j.y_lo = _addr(out.hi), _addr(out.lo)
        j.c_in, j.c_out, j.l_in, j.l_out = sp.c_in, sp.c_out, l_in, out.length
        j.ksize, j.dilation, j.pad, j.reflect_pad, j.out_slope = sp.ksize, sp.dilation, sp.pad_l, self.pad, sp.out_slope
        return j

    def head_out_len(self, l_in: int) -> int:
        sp = self.layers[0].spec
        return l_in + 2 * self.pad + 2 * sp.pad_l - sp.dilation * (sp.ksize - 1)

    # ---- forward: layers 1 .. n-1 from the head's output ------------------------------------------------------------------------
    def out_shapes(self, l0: int):
        """(channels, length) of every embedding behind the head's output of length l0, and the logits' length."""
        shapes, cur = [], l0
        for lay in self.layers[1:-1]:
            cur = lay.spec.out_len(cur)
            shapes.append((lay.spec.c_out, cur))
        sp = self.layers[-1].spec
        return shapes, cur + sp.pad_l + sp.pad_r - (sp.ksize - 1)

    def forward_body(self, act0: Planes, outs: Optional[List[Planes]] = None, logits_out: Optional[torch.Tensor] = None, start: int = 1,
                     stop: Optional[int] = None):
        """The layers behind the head on the rows of ``act0``; ``outs`` / ``logits_out``: write into these (row views of the full batch's
        planes: the two halves of the batch run as two launches sequences) instead of allocating.  ``start`` / ``stop``: layers
        [start, stop) only (``outs`` required; layer ``start`` reads ``outs[start - 2]``; the logits layer runs when stop is None)."""
        lib = load()
        acts = [act0]
        rows = act0.rows
        cur = act0 if start == 1 else outs[start - 2]
        n = len(self.layers)
        for i in range(start, n - 1 if stop is None else stop):
            lay = self.layers[i]
            d = ops.conv_desc(lay.spec, rows, cur.length, lay.math_fwd)
            y = outs[i - 1] if outs is not None else Planes(rows, lay.spec.c_out, d.l_out, act0.hi.device)
            _, _, bias = lay.params()
            wp = lay.packed(0, rows, cur.length)
            tm = ops.kernel_timer_for(lay.spec, "fwd")
            e0 = tm.start() if tm is not None else None
            split = (lay.math_fwd & 0xff) == ops.MATH_BF16X3
            check(lib.eben_bl_conv1d_fwd(ctypes.byref(d), _addr(cur.hi), _addr(cur.lo) if split else None, ptr(wp), ptr(bias), _addr(y.hi), _addr(y.lo),
                                         _stream()), "bl_conv1d_fwd")
            if tm is not None:
                tm.stop(e0, rows)
            acts.append(y)
            cur = y
        if stop is not None:
            return acts, None
        tail = self.layers[-1]
        sp = tail.spec
        v, _, bias = tail.params()
        tail.ensure_scale()
        l_out = cur.length + sp.pad_l + sp.pad_r - (sp.ksize - 1)
        logits = logits_out if logits_out is not None else torch.empty((rows, 1, l_out), dtype=torch.float32, device=act0.hi.device)
        check(lib.eben_bl_tail_fwd(_addr(cur.hi), _addr(cur.lo), rows, cur.channels, cur.length, sp.ksize, sp.pad_l, ptr(v.detach()), ptr(tail.scale),
                                   ptr(bias.detach()) if bias is not None else None, sp.out_slope, ptr(logits), _stream()), "bl_tail_fwd")
        return acts, logits

    # ---- backward: stacked input gradients down to the head's output; weight-gradient jobs ---------------------------------------
    def backward_body(self, acts: List[Planes], seeds: torch.Tensor, half: int, want_param_grads: bool, fm_sums_addr: int, fm_gs: float,
                      part: str = "all"):
        """seeds (4 half, 1, L) rows [fm | adv | fake | real]; returns (gradient planes at the head's output, 4 half rows, and the
        weight-gradient jobs (layer index, gradient rows [fake | real] at the layer's output, the layer's input)).
        ``part``: "all" = the four row blocks in one pass; "gen" = rows [fm | adv] only (``seeds`` = those 2 half rows: what the
        generator's backward waits for); "disc" = rows [fake | real] only (what the weight gradients read) -- the same launches over half
        the rows each, so that the second pass leaves the step's critical path (DiscriminatorEngineBL.backward_launch)."""
        lib = load()
        st = _stream()
        n = len(self.layers)
        if part == "all":
            rows, seg_map, fm_rows, d0 = 4 * half, (ctypes.c_int * 4)(0, 0, 0, 1), half, 2 * half
        elif part == "gen":
            rows, seg_map, fm_rows, d0 = 2 * half, (ctypes.c_int * 4)(0, 0, 0, 0), half, None
        else:
            rows, seg_map, fm_rows, d0 = 2 * half, (ctypes.c_int * 4)(0, 1, 0, 1), 0, 0
        assert seeds.shape[0] == rows
        want_param_grads = want_param_grads and d0 is not None
        jobs = []
        tail = self.layers[-1]
        sp = tail.spec
        x_in = acts[n - 2]
        v, _, _ = tail.params()
        if want_param_grads:
            jobs.append((n - 1, seeds[d0:], x_in))
        g = Planes(rows, x_in.channels, x_in.length, seeds.device, lo=False)
        check(lib.eben_bl_tail_dx(ptr(seeds), rows, x_in.channels, x_in.length, sp.ksize, sp.pad_l, ptr(v.detach()), ptr(tail.scale), _addr(x_in.hi),
                                  _addr(x_in.lo), self.layers[n - 2].spec.out_slope, half, seg_map, fm_rows, half, fm_sums_addr + 8 * (n - 2), fm_gs,
                                  _addr(g.hi), None, st), "bl_tail_dx")
        for i in range(n - 2, 0, -1):
            lay = self.layers[i]
            x_in = acts[i - 1]
            if want_param_grads:
                jobs.append((i, g.rows_slice(d0, rows), x_in))
            d = ops.conv_desc(lay.spec_lin, rows, x_in.length, lay.math_dx)
            # the head's input gradient (rows [fm | adv]) reads hi + lo
            gp = Planes(rows, x_in.channels, x_in.length, seeds.device, lo=(i == 1 and part != "disc"))
            pr = lay.pr_desc(rows, x_in.length) is not None
            wp = lay.packed(2 if pr else 1, rows, x_in.length)
            tm = ops.kernel_timer_for(lay.spec, "dx")
            e0 = tm.start() if tm is not None else None
            dx_fn = lib.eben_bl_conv1d_bwd_dx_pr_c if pr else lib.eben_bl_conv1d_bwd_dx_c
            codes = x_in.codes.data_ptr() if (fm_rows and FM_CODES and x_in.codes is not None and x_in.codes.shape[0] == half) else None
            check(dx_fn(ctypes.byref(d), _addr(g.hi), ptr(wp), _addr(x_in.hi), _addr(x_in.lo), codes, self.layers[i - 1].spec.out_slope, half,
                        seg_map, fm_rows, half, fm_sums_addr + 8 * (i - 1), fm_gs, _addr(gp.hi), _addr(gp.lo), st), "bl_conv1d_bwd_dx")
            if tm is not None:
                tm.stop(e0, rows)
            g = gp
        return g, jobs, (None if d0 is None else g.rows_slice(d0, rows))

    def weight_grads(self, jobs, x_full: torch.Tensor, g0: Optional[Planes], half: int, sink=None):
        """Weight gradients of every layer (rows [fake | real] of the stacked gradients -- what ``backward_body`` put into the jobs and
        ``g0`` -- against the layer inputs [enhanced | reference]): head from (g0, the chain's fp32 input), tap-conv layers by
        ``eben_bl_conv1d_bwd_dw``, the logits layer per branch."""
        lib = load()
        st = _stream()
        n = len(self.layers)
        grads = [None] * n
        wn_jobs = []
        logits = None
        for i, g, x_in in jobs:
            lay = self.layers[i]
            if i == n - 1:
                gf, gr = self._tail_dw(lay, g, x_in, half, st, wn_jobs)
                logits = (i, gf, gr)
            else:
                grads[i] = self._mid_dw(lay, g, x_in, half, st, wn_jobs, sink)
        if g0 is not None:
            grads[0] = self._head_dw(g0, x_full, half, st, wn_jobs, sink)
        ops.wn_bwd_multi(wn_jobs)
        if logits is not None:
            i, gf, gr = logits
            outs = [None if sink is None or p is None else sink.grad_buffer(p) for p in self.layers[i].params()]
            grads[i] = tuple(None if a is None else (a + b if o is None else torch.add(a, b, out=o)) for a, b, o in zip(gf, gr, outs))
        return grads

    @staticmethod
    def weight_grads_group(chains, jobs_list, x_fulls, g0s, half: int, sink=None):
        """``weight_grads`` of several chains of ONE structure (the three PQMF-band discriminators: same channels and taps per layer, their
        own dilation and lengths) in one launch sequence: the mid layers of one index as ONE ``eben_bl_conv1d_bwd_dw_multi`` launch (a
        thin layer's weight gradient is mostly fixed cost: [MI355X] 64 rows 37-40 us, 192 rows 66-87 us), one slab reduction / weight-norm
        pass for all of them.  Same slabs, same results as chain by chain.  Returns the chains' gradient lists."""
        lib = load()
        st = _stream()
        n = len(chains[0].layers)
        assert all(len(ch.layers) == n for ch in chains)
        grads = [[None] * n for _ in chains]
        wn_jobs, logits, by_layer = [], [], {}
        for ci, (ch, jobs) in enumerate(zip(chains, jobs_list)):
            for i, g, x_in in jobs:
                if i == n - 1:
                    gf, gr = ch._tail_dw(ch.layers[i], g, x_in, half, st, wn_jobs)
                    logits.append((ci, i, gf, gr))
                else:
                    by_layer.setdefault(i, []).append((ci, g, x_in))
        for i in sorted(by_layer, reverse=True):
            items = by_layer[i]
            outs = _ChainBL._mid_dw_multi([(chains[ci].layers[i], g, x_in) for ci, g, x_in in items], half, st, wn_jobs, sink)
            for (ci, _, _), o in zip(items, outs):
                grads[ci][i] = o
        for ci, ch in enumerate(chains):
            if g0s[ci] is not None:
                grads[ci][0] = ch._head_dw(g0s[ci], x_fulls[ci], half, st, wn_jobs, sink)
        ops.wn_bwd_multi(wn_jobs)
        for ci, i, gf, gr in logits:
            outs = [None if sink is None or p is None else sink.grad_buffer(p) for p in chains[ci].layers[i].params()]
            grads[ci][i] = tuple(None if a is None else (a + b if o is None else torch.add(a, b, out=o)) for a, b, o in zip(gf, gr, outs))
        return grads

    @staticmethod
    def _mid_dw_multi(items, half: int, st: int, wn_jobs: list, sink=None):
        """items: (layer, gradient planes rows [fake | real], input planes) of the SAME layer index of several chains."""
        lib = load()
        k = len(items)
        descs = (ctypes.POINTER(ops.EbenConv1dDesc) * k)()
        dys, xs, slabs_p = (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)(), (ctypes.c_void_p * k)()
        nbs = (ctypes.c_size_t * k)()
        keep, outs = [], []
        has_bias = None
        for j, (lay, g, x_in) in enumerate(items):
            v, gain, bias = lay.params()
            hb = 1 if bias is not None else 0
            assert has_bias in (None, hb)
            has_bias = hb
            d = ops.conv_desc(lay.spec_lin, 2 * half, x_in.length, lay.math_dw)
            ws = getattr(d, "_bl_dw_ws", None)
            if ws is None:
                nslab, row_stride, perm = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
                nbytes = lib.eben_bl_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride), ctypes.byref(perm))
                if nbytes == 0:
                    raise ops._lib.EbenError(f"bundle-layout weight gradient does not cover {lay.spec}")
                ws = d._bl_dw_ws = (nbytes, nslab.value, row_stride.value, perm.value)
            nbytes, nslab, row_stride, perm = ws
            slabs = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=g.hi.device)
            descs[j] = ctypes.pointer(d)
            dys[j], xs[j], slabs_p[j], nbs[j] = _addr(g.hi), _addr(x_in.hi), slabs.data_ptr(), nbytes
            keep.append((d, slabs))
            dv, dg, dbias = _ChainBL._outputs(lay, sink, g.hi.device)
            rows = v.shape[0]
            wn_jobs.append((slabs, nslab, rows * row_stride, rows, v.numel() // rows, row_stride, gain.detach(), v.detach(), lay.norm, dg, dv, dbias, perm))
            outs.append((dv, dg, dbias))
        check(lib.eben_bl_conv1d_bwd_dw_multi(descs, dys, xs, has_bias, slabs_p, nbs, k, st), "bl_conv1d_bwd_dw_multi")
        return outs

    @staticmethod
    def _outputs(lay: _Layer, sink, device):
        v, gain, bias = lay.params()
        dv = dg = dbias = None
        if sink is not None:
            dv, dg, dbias = sink.grad_buffer(v), sink.grad_buffer(gain), (sink.grad_buffer(bias) if bias is not None else None)
        dv = torch.empty_like(v) if dv is None else dv
        dg = torch.empty_like(gain) if dg is None else dg
        if bias is not None and dbias is None:
            dbias = torch.empty(v.shape[0], dtype=torch.float32, device=device)
        return dv, dg, dbias

    def _mid_dw(self, lay: _Layer, g: Planes, x_in: Planes, half: int, st: int, wn_jobs: list, sink=None):
        lib = load()
        v, gain, bias = lay.params()
        d = ops.conv_desc(lay.spec_lin, 2 * half, x_in.length, lay.math_dw)
        ws = getattr(d, "_bl_dw_ws", None)
        if ws is None:
            nslab, row_stride, perm = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
            nbytes = lib.eben_bl_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride), ctypes.byref(perm))
            if nbytes == 0:
                raise ops._lib.EbenError(f"bundle-layout weight gradient does not cover {lay.spec}")
            ws = d._bl_dw_ws = (nbytes, nslab.value, row_stride.value, perm.value)
        nbytes, nslab, row_stride, perm = ws
        slabs = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=g.hi.device)
        check(lib.eben_bl_conv1d_bwd_dw(ctypes.byref(d), _addr(g.hi), _addr(x_in.hi), 1 if bias is not None else 0, ptr(slabs), nbytes, st),
              "bl_conv1d_bwd_dw")
        dv, dg, dbias = self._outputs(lay, sink, g.hi.device)
        rows = v.shape[0]
        wn_jobs.append((slabs, nslab, rows * row_stride, rows, v.numel() // rows, row_stride, gain.detach(), v.detach(), lay.norm, dg, dv, dbias, perm))
        return dv, dg, dbias

    def _head_dw(self, g0: Planes, x_full: torch.Tensor, half: int, st: int, wn_jobs: list, sink=None):
        lib = load()
        lay = self.layers[0]
        v, gain, bias = lay.params()
        view = Planes.__new__(Planes)
        view.codes = None
        view.hi, view.lo, view.rows, view.channels, view.length = g0.hi, None, 2 * half, g0.channels, g0.length
        job = self.head_job(x_full, x_full.shape[2], view)
        nslab, row_stride = ctypes.c_int(0), ctypes.c_int(0)
        nbytes = lib.eben_bl_head_dw_workspace(ctypes.byref(job), 2 * half, ctypes.byref(nslab), ctypes.byref(row_stride))
        slabs = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x_full.device)
        check(lib.eben_bl_head_dw(ctypes.byref(job), 2 * half, ptr(slabs), nbytes, st), "bl_head_dw")
        dv, dg, dbias = self._outputs(lay, sink, x_full.device)
        rows = v.shape[0]
        wn_jobs.append((slabs, nslab.value, rows * row_stride.value, rows, v.numel() // rows, row_stride.value, gain.detach(), v.detach(), lay.norm, dg, dv,
                        dbias))
        return dv, dg, dbias

    @staticmethod
    def _tail_dw(lay: _Layer, seeds: torch.Tensor, x_in: Planes, half: int, st: int, wn_jobs: list):
        """Logits layer, the two hinge branches by ONE launch: seed rows [fake | real] (2 half rows) against embedding rows [enhanced |
        reference]; the branches stay separate results (the engine adds them: see ``disc_engine._Chain.weight_grads``)."""
        lib = load()
        v, gain, bias = lay.params()
        sp = lay.spec
        l_out = seeds.shape[2]
        nslab, row_stride = ctypes.c_int(0), ctypes.c_int(0)
        nbytes = lib.eben_bl_tail_dw_workspace(half, x_in.channels, l_out, sp.ksize, 2, ctypes.byref(nslab), ctypes.byref(row_stride))
        slabs = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=seeds.device)
        check(lib.eben_bl_tail_dw(ptr(seeds), _addr(x_in.hi), _addr(x_in.lo), half, 2, x_in.channels, x_in.length, sp.ksize, sp.pad_l, ptr(slabs), nbytes, st),
              "bl_tail_dw")
        outs = []
        per = nslab.value * row_stride.value
        for br in range(2):
            dv, dg = torch.empty_like(v), torch.empty_like(gain)
            dbias = torch.empty(1, dtype=torch.float32, device=seeds.device) if bias is not None else None
            wn_jobs.append((slabs[br * per:(br + 1) * per], nslab.value, row_stride.value, 1, v.numel(), row_stride.value, gain.detach(), v.detach(), lay.norm,
                            dg, dv, dbias))
            outs.append((dv, dg, dbias))
        return outs[0], outs[1]


class DiscriminatorEngineBL(DiscriminatorEngine):
    """``DiscriminatorEngine`` over bundle-layout tensors (selected by ``math = {..., "layout": "bl"}``)."""

    @staticmethod
    def unsupported(disc) -> Optional[str]:
        """Why this discriminator cannot run in the bundle layout (None: it can): the chain heads / logits layers have dedicated kernels
        built for EBEN's configured shapes (4 -> 24 k 3 and 1 -> 16 k 15 heads, C -> 1 k <= 8 logits layers)."""
        for m in list(disc.pqmf_discriminators) + [disc.melgan_discriminator]:
            convs = [sub for part in m.discriminator for sub in (part if isinstance(part, torch.nn.Sequential) else [part]) if hasattr(sub, "spec")]
            if len(convs) < 3:
                return "a sub-discriminator with fewer than three conv layers"
            head, tail = convs[0].spec, convs[-1].spec
            if (head.c_in, head.c_out, head.ksize) not in ((4, 24, 3), (1, 16, 15)) or head.groups != head.c_in or head.stride != 1:
                return f"no head kernel for {head.c_in} -> {head.c_out}, k {head.ksize}"
            if not (tail.c_out == 1 and tail.groups == 1 and tail.stride == 1 and tail.dilation == 1 and tail.ksize <= 8 and tail.c_in % 8 == 0):
                return f"no logits kernel for {tail.c_in} -> {tail.c_out}, k {tail.ksize}"
            if any(c.spec.c_in % 8 or c.spec.c_out % 8 for c in convs[1:-1]):
                return "mid-layer channels that are not whole bundles of 8"
        return None

    def __init__(self, disc, math):
        self.disc = disc
        self.q = disc.q
        self.math = math
        self.chains = [_ChainBL(d.discriminator, math["pqmf"]) for d in disc.pqmf_discriminators] + [
            _ChainBL(disc.melgan_discriminator.discriminator, math["melgan"])]
        self._streams = None
        self._state = None
        self._prepack_graph = ops.ReplayedPrepack()
        self.seed_weights = (1.0, 1.0, 1.0)
        n = len(self.chains)
        # graph replay of each chain's three launch sequences (ops.ReplayedChain) and the buffers they start from, per input shape
        self._graphs = {k: [ops.ReplayedChain() for _ in range(n)] for k in ("fwd", "bwd", "dw")}
        self._static: Dict[tuple, dict] = {}

    def _static_for(self, half: int, bands, audio) -> dict:
        """Persistent buffers of one (batch, lengths) shape: the chain inputs [enhanced | reference], the heads' outputs, the
        feature-matching sums -- what the replayed launch sequences read at fixed addresses."""
        key = (half, tuple(bands.shape[1:]), tuple(audio.shape[1:]), str(audio.device))
        st = self._static.get(key)
        if st is None:
            if len(self._static) >= 4:   # variable clip lengths: keep the buffers (and graphs) of the recent shapes only
                self._static.pop(next(iter(self._static)))
            dev = audio.device
            rows = 2 * half
            sub = torch.empty((rows, self.q) + tuple(bands.shape[2:]), dtype=torch.float32, device=dev)
            wav = torch.empty((rows,) + tuple(audio.shape[1:]), dtype=torch.float32, device=dev)
            inputs = [sub] * (len(self.chains) - 1) + [wav]
            act0 = [Planes(rows, ch.layers[0].spec.c_out, ch.head_out_len(inputs[i].shape[2]), dev) for i, ch in enumerate(self.chains)]
            npairs = sum(len(ch.layers) - 1 for ch in self.chains)
            st = self._static[key] = dict(sub=sub, wav=wav, inputs=inputs, act0=act0, one=torch.ones(1, dtype=torch.float32, device=dev),
                                          fm_sums=torch.empty(2 * npairs, dtype=torch.float32, device=dev))
        return st

    @staticmethod
    def _mark_used(ch: "_ChainBL", which: int) -> None:
        """A replayed sequence reads the chain's packed images without passing through ``_Layer.packed``: tell ``prepack`` they are in
        use (it drops the images the last step did not touch)."""
        for lay in ch.layers:
            lay.used.update(slot for slot in lay.packs if slot[0] == which or (which == 1 and slot[0] == 2))   # 2: phases-as-rows images

    @staticmethod
    def _chain_sig(ch: "_ChainBL", which: int):
        """Everything a chain's launches depend on besides tensor values: per layer the parameter storage, the weight-norm scale buffer,
        the packed images of direction `which` (address + whether they are current) and the arithmetic."""
        out = []
        for lay in ch.layers:
            v, g, b = lay.params()
            wkey = lay._weights_key()
            imgs = tuple(sorted((slot, wp.data_ptr(), key == wkey) for slot, (key, wp) in lay.packs.items() if slot[0] == which or (which == 1 and slot[0] == 2))) if which >= 0 else ()
            out.append((v.data_ptr(), g.data_ptr(), 0 if b is None else b.data_ptr(), 0 if lay.scale is None else lay.scale.data_ptr(),
                        lay.scale_key == wkey, imgs, lay.math_fwd, lay.math_dx, lay.math_dw))
        return tuple(out)

    # ---- forward ---------------------------------------------------------------------------------------------------------------
    def _split_planes(self, st: dict, half: int):
        """Planes of every embedding / the logits of the full 2B-row batch, allocated once per shape: with the forward split into its
        reference and enhanced halves two launch sequences write into them (row views)."""
        if "planes" not in st:
            dev = st["wav"].device
            planes, logits = [], []
            for i, ch in enumerate(self.chains):
                shapes, l_logits = ch.out_shapes(st["act0"][i].length)
                planes.append([Planes(2 * half, c, l, dev) for c, l in shapes])
                logits.append(torch.empty((2 * half, 1, l_logits), dtype=torch.float32, device=dev))
            st["planes"], st["logits"] = planes, logits
        return st["planes"], st["logits"]

    def _forward_rows(self, st: dict, half: int, r0: int, r1: int, fm: bool, key: str):
        """Heads + chain bodies of batch rows [r0, r1) on the chains' streams (not joined).  fm: the feature-matching sums of each chain's
        embedding pairs behind its layers (the second of the two halves)."""
        lib = load()
        sub, wav, act0 = st["sub"], st["wav"], st["act0"]
        planes, logits = self._split_planes(st, half)
        n = len(self.chains)
        nrows = r1 - r0
        fm_first = [sum(len(c.layers) - 1 for c in self.chains[:i]) for i in range(n)]
        fm_sums = st["fm_sums"]
        head_done = [None]

        # MelGAN (the last chain) keeps ONE 2B-row launch per layer, behind the generator: its heavy layers fill the machine exactly once at
        # 64 rows (512 blocks = two per CU) and lose a third of their rate as two 32-row launches ([MI355X] L4 forward alone 0.35 -> 0.29 of
        # the bf16 peak); the thin PQMF-band chains are what gains from running beside the generator forward
        whole_last = SPLIT_MELGAN == "0"
        depth = MELGAN_SPLIT_DEPTH if SPLIT_MELGAN == "thin" else None   # layers [1, depth) of the last chain per half, the rest on 2B rows
        if depth is not None and not (1 < depth < len(self.chains[n - 1].layers) - 1):
            depth = None

        def rng(i):
            if whole_last and i == n - 1:
                return (0, 2 * half) if fm else None
            return r0, r1

        def body(i):
            q0, q1 = rng(i)
            a0 = act0[i].rows_slice(q0, q1)
            if i == n - 1 and depth is not None:
                self.chains[i].forward_body(a0, [p.rows_slice(q0, q1) for p in planes[i]], None, stop=depth)
                if fm:   # both halves of layer depth - 1 are there: the heavy layers once, on all 2B rows
                    self.chains[i].forward_body(act0[i], planes[i], logits[i], start=depth)
            else:
                self.chains[i].forward_body(a0, [p.rows_slice(q0, q1) for p in planes[i]], logits[i][q0:q1])
            if fm:
                _fm_sums(lib, [act0[i]] + planes[i], half, fm_sums[2 * fm_first[i]:])
            return True

        def run(i):
            ch = self.chains[i]
            if rng(i) is None:   # MelGAN in the reference pass: nothing yet
                return [act0[i]] + planes[i], logits[i]
            if i == n - 1:
                q0, q1 = rng(i)
                jobs = (EbenBlHeadJob * 1)(ch.head_job(wav[q0:q1], wav.shape[2], act0[i].rows_slice(q0, q1)))
                check(lib.eben_bl_head_fwd(jobs, 1, q1 - q0, _stream()), "bl_head_fwd")
            elif i == 0:
                jobs = (EbenBlHeadJob * (n - 1))(*[self.chains[k].head_job(sub[r0:r1], sub.shape[2], act0[k].rows_slice(r0, r1)) for k in range(n - 1)])
                check(lib.eben_bl_head_fwd(jobs, n - 1, nrows, _stream()), "bl_head_fwd")
                head_done[0] = torch.cuda.Event()
                head_done[0].record()
            else:
                torch.cuda.current_stream().wait_event(head_done[0])
            sig = (rng(i), fm, act0[i].hi.data_ptr(), act0[i].lo.data_ptr(), act0[i].length, fm_sums.data_ptr(), planes[i][0].hi.data_ptr(),
                   self._chain_sig(ch, 0))
            self._graphs[key][i].run(sig, lambda: body(i), torch.cuda.current_stream())
            self._mark_used(ch, 0)
            return [act0[i]] + planes[i], logits[i]

        return self._launch_on_streams(run, forward=True, order=[n - 1] + list(range(n - 1)))

    def _join_pending(self):
        """Weight-gradient work of an earlier step that nobody collected (an exception between ``backward_finish`` and
        ``collect_param_grads``, a skipped batch): its kernels still read the static planes -- the current stream waits for the chains'
        streams and the work is dropped, instead of failing every later step."""
        if getattr(self, "_pending", None) is not None:
            main = torch.cuda.current_stream()
            for st in set(self._streams or ()):
                main.wait_stream(st)
            if getattr(self, "_used_streams", None):
                for st in self._used_streams:
                    main.wait_stream(st)
            self._pending = None

    @torch.no_grad()
    def forward_reference(self, bands_ref, audio_ref):
        """The reference half of the batch (rows B .. 2B: it does not depend on the generator) on the chains' streams, to run underneath
        the generator forward -- a chain of latency-bound launches that leaves most of the GPU idle.  ``forward`` then runs the enhanced
        half only."""
        half = bands_ref.shape[0]
        st = self._static_for(half, bands_ref, audio_ref)
        # the planes are static per shape: every consumer of the previous step's embeddings (stacked input gradients, weight gradients)
        # must have been joined before they are rewritten
        self._join_pending()
        st["sub"][half:].copy_(bands_ref[:, -self.q:, :])
        st["wav"][half:].copy_(audio_ref)
        if "fwd_ref" not in self._graphs:
            n = len(self.chains)
            self._graphs["fwd_ref"] = [ops.ReplayedChain() for _ in range(n)]
            self._graphs["fwd_enh"] = [ops.ReplayedChain() for _ in range(n)]
        self._forward_rows(st, half, half, 2 * half, False, "fwd_ref")
        self._ref_done = (half, st)

    @torch.no_grad()
    def forward(self, bands, audio, bands_ref, audio_ref, join: bool = True):
        lib = load()
        half = bands.shape[0]
        st = self._static_for(half, bands, audio)
        sub, wav, inputs, act0 = st["sub"], st["wav"], st["inputs"], st["act0"]
        pre = getattr(self, "_ref_done", None)
        self._ref_done = None
        if pre is None:
            self._join_pending()
        if pre is not None and pre[0] == half and pre[1] is st:
            # the reference rows are on their way (forward_reference): the enhanced rows + the feature-matching sums behind them
            sub[:half].copy_(bands[:, -self.q:, :])
            wav[:half].copy_(audio)
            res = self._forward_rows(st, half, 0, half, True, "fwd_enh")
            if join:
                self._join_streams()
            self._state = dict(half=half, acts=[r[0] for r in res], logits=[r[1] for r in res], inputs=inputs, bands_shape=tuple(bands.shape), static=st)
            return self._state
        sub[:half].copy_(bands[:, -self.q:, :])
        sub[half:].copy_(bands_ref[:, -self.q:, :])
        wav[:half].copy_(audio)
        wav[half:].copy_(audio_ref)
        n = len(self.chains)
        rows = 2 * half
        self._head_done = None
        fm_first = [sum(len(c.layers) - 1 for c in self.chains[:i]) for i in range(n)]
        fm_sums = st["fm_sums"]

        def body(i):
            # the chain's layers, then the feature-matching sums of ITS embedding pairs on its own stream: the HBM-bound pass over the
            # embeddings (1.6 GB at 64 rows) runs beside the other chains' MFMA-bound layers instead of alone behind the join
            # ([MI355X] one launch over all 35 pairs after the join: 0.27 ms of the step's critical path)
            acts, logits = self.chains[i].forward_body(act0[i])
            _fm_sums(lib, acts, half, fm_sums[2 * fm_first[i]:])
            return acts, logits

        def run(i):
            # heads (eager, so that the chains that share one can be released by an event): the PQMF-band chains' in one launch on the
            # stream of the first of them, MelGAN's on its own stream; then the chain's body, replayed as a graph once it has settled
            ch = self.chains[i]
            if i == n - 1:
                jobs = (EbenBlHeadJob * 1)(ch.head_job(wav, wav.shape[2], act0[i]))
                check(lib.eben_bl_head_fwd(jobs, 1, rows, _stream()), "bl_head_fwd")
            elif i == 0:
                jobs = (EbenBlHeadJob * (n - 1))(*[self.chains[k].head_job(sub, sub.shape[2], act0[k]) for k in range(n - 1)])
                check(lib.eben_bl_head_fwd(jobs, n - 1, rows, _stream()), "bl_head_fwd")
                self._head_done = torch.cuda.Event()
                self._head_done.record()
            else:
                torch.cuda.current_stream().wait_event(self._head_done)   # chains 1, 2 may run on another stream than chain 0
            sig = (rows, act0[i].hi.data_ptr(), act0[i].lo.data_ptr(), act0[i].length, fm_sums.data_ptr(), self._chain_sig(ch, 0))
            out = self._graphs["fwd"][i].run(sig, lambda: body(i), torch.cuda.current_stream())
            self._mark_used(ch, 0)
            return out

        res = self._launch_on_streams(run, forward=True, order=[n - 1] + list(range(n - 1)))
        if join:
            self._join_streams()
        self._state = dict(half=half, acts=[r[0] for r in res], logits=[r[1] for r in res], inputs=inputs, bands_shape=tuple(bands.shape), static=st)
        return self._state

    # ---- losses ----------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def losses(self) -> Dict[str, torch.Tensor]:
        lib = load()
        s = self._state
        half = s["half"]
        dev = s["logits"][0].device
        n = sum(len(acts) for acts in s["acts"])                  # embedding pairs: every embedding but the inputs and the logits
        sums = s["static"]["fm_sums"]   # fixed address, filled chain by chain at the end of each forward body
        nch = len(self.chains)
        inv = 1.0 / (nch * len(s["acts"][-1]))
        s.update(fm_sums=sums, fm_inv=inv, fm_first=[sum(len(a) for a in s["acts"][:i]) for i in range(nch)])
        hinge = torch.empty(3 * nch, dtype=torch.float32, device=dev)
        terms = [(rows, target) for lg in s["logits"] for rows, target in ((lg[:half], 1.0), (lg[:half], -1.0), (lg[half:], 1.0))]
        nt = len(terms)
        check(lib.eben_hinge_fwd_multi((ctypes.c_void_p * nt)(*[ptr(r) for r, _ in terms]), (ctypes.c_int64 * nt)(*[r.numel() for r, _ in terms]),
                                       (ctypes.c_float * nt)(*[t for _, t in terms]), nt, ptr(hinge), _stream()), "hinge_fwd_multi")
        vals = torch.empty(4, dtype=torch.float32, device=dev)
        check(lib.eben_disc_losses(ptr(sums), n, inv, ptr(hinge), nch, ptr(vals), _stream()), "disc_losses")
        return {"feature_matching_loss": vals[0], "adv_loss_gen": vals[1], "fake_loss": vals[2], "real_loss": vals[3]}

    # ---- backward --------------------------------------------------------------------------------------------------------------
    #: The stacked backward as TWO passes of 2B rows: rows [fm | adv] -- the generator's gradient signals, what the step's critical path
    #: (balancing, generator backward, generator Adam) waits for -- first; rows [fake | real] and the weight gradients they feed behind
    #: them on the same chain streams, joined only in front of the discriminator's Adam.  Same launches, same values (every row of a
    #: stacked pass is independent of the others); the second half of the input-gradient work (config 2: ~1.2 of 2.4 ms) moves underneath
    #: the generator backward.  EBEN_SPLIT_BWD=0: one 4B-row pass (rounds 2-5).
    split_backward: bool = __import__("os").environ.get("EBEN_SPLIT_BWD", "1") != "0"
    #: with the two-pass backward the three PQMF-band chains stay in series on ONE stream (the fp32-at-rest engines move the last of them
    #: to the side stream): [MI355X, same box, 3 x alternated] 9.03 / 9.08 / 8.99 -> 8.97 / 8.98 / 8.93 ms per step
    spread_backward = __import__("os").environ.get("EBEN_D_BWD_SPREAD", "0") != "0"

    def _seeds(self, lib, s, i, dev):
        """rows [fm | adv | fake | real] of chain i: zeros and the three hinge derivatives (targets +1, -1 on the enhanced rows, +1 on the
        reference rows)"""
        half = s["half"]
        lg = s["logits"][i]
        one = s["static"]["one"]
        inv_scales = 1.0 / len(self.chains)
        per = lg[:half].numel()
        if STACKED_SEEDS:
            seeds = torch.empty((4 * half,) + tuple(lg.shape[1:]), dtype=torch.float32, device=dev)
            check(lib.eben_hinge_bwd_stacked(ptr(lg[:half]), ptr(lg[half:]), per, ptr(one), inv_scales * self.seed_weights[0],
                                             inv_scales * self.seed_weights[1], inv_scales * self.seed_weights[2], ptr(seeds), _stream()), "hinge_bwd_stacked")
        else:   # a memset and one launch per seed block (bisecting aid: the same values)
            seeds = torch.zeros((4 * half,) + tuple(lg.shape[1:]), dtype=torch.float32, device=dev)
            flat = seeds.reshape(-1)
            for k2, (rows, target) in enumerate(((lg[:half], 1.0), (lg[:half], -1.0), (lg[half:], 1.0))):
                check(lib.eben_hinge_bwd(ptr(rows), rows.numel(), target, ptr(one), inv_scales * self.seed_weights[k2], ptr(flat[(k2 + 1) * per:]),
                                         _stream()), "hinge_bwd")
        return seeds

    def _launch_weight_grads(self, s, i, jobs, g0, half):
        """Chain i's weight gradients on the current (the chain's) stream; returns what ``collect_param_grads`` reads."""
        # with a data-parallel sink the launches write straight into its gradient buckets: static addresses, part of the signature
        sink = self._sink
        sink_sig = None if sink is None else tuple(0 if (b := sink.grad_buffer(p)) is None else b.data_ptr()
                                                   for lay in self.chains[i].layers for p in lay.params() if p is not None)
        jobs_sig = tuple((k, (g.hi.data_ptr() if isinstance(g, Planes) else g.data_ptr()), x.hi.data_ptr(), x.lo.data_ptr()) for k, g, x in jobs)
        sig = (half, s["inputs"][i].data_ptr(), g0.hi.data_ptr(), jobs_sig, self._chain_sig(self.chains[i], -1), sink_sig)
        dw_body = lambda: self.chains[i].weight_grads(jobs, s["inputs"][i], g0, half, sink)
        # A replay rewrites the gradients of the previous replay in place, and without a sink those tensors BECOME ``.grad``
        # (inject_grads): while a parameter still holds a gradient (accumulation over steps, zero_grad(set_to_none=False), a
        # caller that does not zero) the launches run eagerly into fresh tensors, as gen_engine._deferrable has it.
        held = sink is None and any(p.grad is not None for lay in self.chains[i].layers for p in lay.params() if p is not None)
        out = dw_body() if held else self._graphs["dw"][i].run(sig, dw_body, torch.cuda.current_stream())
        if sink is not None:
            ready = [p for lay in self.chains[i].layers for p in lay.params() if p is not None and p.requires_grad]
            if self.defer_mark_ready:
                self._deferred_ready = (getattr(self, "_deferred_ready", None) or []) + ready
            else:
                sink.mark_ready(ready)
        return out

    #: data-parallel runs: report the discriminator's gradients to the bucket exchange when the main stream joins the weight gradients
    #: (``collect_param_grads``) instead of from each chain's stream right behind its launches.  Reported early, a bucket's collective sits
    #: on the communication stream WAITING for the chain's events for ~3 ms (second backward pass + weight gradients), and a waiting stream
    #: stalls whatever shares its hardware queue: [MI355X, single-rank RCCL, 7 queues] the main stream did not start the generator's seeds
    #: until the discriminator's last weight gradient had run -- 10.07 ms per step against 9.18 deferred (9.06 with the generator's deferred
    #: too, 8.7 without the exchange).  The price at N > 1 is the discriminator's exchange (92.6 MB) exposed in front of its Adam instead
    #: of under the tail of its weight gradients.  EBEN_D_DEFER_READY=0: the early form.
    defer_mark_ready: bool = __import__("os").environ.get("EBEN_D_DEFER_READY", "1") != "0"

    def collect_param_grads(self):
        sink = getattr(self, "_sink", None)
        ready, self._deferred_ready = getattr(self, "_deferred_ready", None), None
        if ready and sink is not None and getattr(self, "_pending", None) is not None:
            main = torch.cuda.current_stream()
            for st in set(self._streams) | set(getattr(self, "_used_streams", None) or ()):
                main.wait_stream(st)
            sink.mark_ready(ready)
        return super().collect_param_grads()

    #: weight gradients of the three PQMF-band chains as one launch sequence (``_ChainBL.weight_grads_group``); 0: chain by chain
    group_weight_grads: bool = __import__("os").environ.get("EBEN_DW_GROUP", "0") != "0"

    def _launch_weight_grads_group(self, s, group, keep, half):
        """``_launch_weight_grads`` for the chains ``group`` (one stream, one structure) as ONE replayed sequence."""
        sink = self._sink
        chains = [self.chains[i] for i in group]
        params = [p for ch in chains for lay in ch.layers for p in lay.params() if p is not None]
        sink_sig = None if sink is None else tuple(0 if (b := sink.grad_buffer(p)) is None else b.data_ptr() for p in params)
        jobs_sig = tuple(tuple((k, (g.hi.data_ptr() if isinstance(g, Planes) else g.data_ptr()), x.hi.data_ptr(), x.lo.data_ptr()) for k, g, x in keep[i][1])
                         for i in group)
        sig = (half, tuple(s["inputs"][i].data_ptr() for i in group), tuple(keep[i][2].hi.data_ptr() for i in group), jobs_sig,
               tuple(self._chain_sig(ch, -1) for ch in chains), sink_sig)
        body = lambda: _ChainBL.weight_grads_group(chains, [keep[i][1] for i in group], [s["inputs"][i] for i in group], [keep[i][2] for i in group],
                                                   half, sink)
        held = sink is None and any(p.grad is not None for p in params)
        if "dw_group" not in self._graphs:
            self._graphs["dw_group"] = [ops.ReplayedChain()]
        out = body() if held else self._graphs["dw_group"][0].run(sig, body, torch.cuda.current_stream())
        if sink is not None:
            sink.mark_ready([p for p in params if p.requires_grad])
        return out

    @torch.no_grad()
    def backward_launch(self, want_param_grads: bool = True, sink=None):
        self._sink = sink
        lib = load()
        s = self._state
        half = s["half"]
        dev = s["logits"][0].device
        one = s["static"]["one"]
        sums_ptr = ptr(s["fm_sums"])
        split = self.split_backward
        n = len(self.chains)

        chain_sigs = {}   # both passes of a chain see the same weights and images: one walk of its layers per step

        def sig_of(i, tag):
            if i not in chain_sigs:
                chain_sigs[i] = self._chain_sig(self.chains[i], 1)
            return (half, tag, tuple(self.seed_weights), s["fm_inv"], sums_ptr, s["logits"][i].data_ptr(),
                    tuple((a.hi.data_ptr(), a.lo.data_ptr(), a.length, None if a.codes is None else a.codes.data_ptr()) for a in s["acts"][i]),
                    chain_sigs[i])

        def body(i):
            seeds = self._seeds(lib, s, i, dev)
            if split:
                g, _, _ = self.chains[i].backward_body(s["acts"][i], seeds[:2 * half], half, False, sums_ptr + 8 * s["fm_first"][i], s["fm_inv"], "gen")
                return g, [], None, seeds
            return self.chains[i].backward_body(s["acts"][i], seeds, half, want_param_grads, sums_ptr + 8 * s["fm_first"][i], s["fm_inv"]) + (seeds,)

        def run(i):
            out = self._graphs["bwd"][i].run(sig_of(i, (want_param_grads, split)), lambda: body(i), torch.cuda.current_stream())
            self._mark_used(self.chains[i], 1)
            if split:
                ev = torch.cuda.Event()
                ev.record()
                return out + (ev,)
            return out

        res = self._launch_on_streams(run)
        self._pending = None
        if split and want_param_grads:
            # second pass, not waited for by the main stream: rows [fake | real] down every chain, then the chain's weight gradients
            if "bwd_d" not in self._graphs:
                self._graphs["bwd_d"] = [ops.ReplayedChain() for _ in range(n)]
            pend = [None] * n
            keep = [None] * n

            def body_d(i):
                seeds = res[i][3]
                g, jobs, g0 = self.chains[i].backward_body(s["acts"][i], seeds[2 * half:], half, True, sums_ptr + 8 * s["fm_first"][i], s["fm_inv"], "disc")
                return g, jobs, g0

            group = list(range(n - 1)) if (self.group_weight_grads and n >= 3 and len({self._second_pass_stream(i, res[i][4]) for i in range(n - 1)}) == 1) else []
            for i in [n - 1] + list(range(n - 1)):
                with torch.cuda.stream(self._second_pass_stream(i, res[i][4])):
                    sig = sig_of(i, "disc") + (res[i][3].data_ptr(),)
                    out = self._graphs["bwd_d"][i].run(sig, lambda i=i: body_d(i), torch.cuda.current_stream())
                    keep[i] = out
                    if i not in group:
                        pend[i] = self._launch_weight_grads(s, i, out[1], out[2], half)
            if group:
                # the PQMF-band chains share a stream and a structure: their weight gradients layer index by layer index
                with torch.cuda.stream(self._second_pass_stream(group[0], res[group[0]][4])):
                    for i, grads in zip(group, self._launch_weight_grads_group(s, group, keep, half)):
                        pend[i] = grads
            self._pending = (pend, s, res, keep)   # keeps the saved activations and the stacked gradients alive until the kernels have run
        self._bwd = (res, want_param_grads, (one,))

    def _second_pass_stream(self, i, after: "torch.cuda.Event"):
        """The stream chain i's second pass runs on: the chain's own ([MI355X] streams of their own, at either HIP priority, cost more than
        any ordering gave: 9.18 -> 9.44 ms per step at the default priority, 16.8 at the high one -- more streams than hardware queues)."""
        return self._chain_stream(i)

    def _chain_stream(self, i):
        """The stream chain i's backward was launched on by the last ``_launch_on_streams`` (spread_backward may move one chain)."""
        streams = list(self._streams)
        if self.spread_backward and len(self.chains) >= 3:
            streams[len(self.chains) - 2] = ops.aux_stream(2, streams[0].device)
        return streams[i]

    @torch.no_grad()
    def backward_finish(self):
        lib = load()
        res, want_param_grads, _keep = self._bwd
        split = len(res[0]) == 5
        main = torch.cuda.current_stream()
        if split:
            for r in res:
                main.wait_event(r[4])   # rows [fm | adv] of every chain; what follows on the chains' streams is joined by collect_param_grads
        else:
            self._join_streams()
        self._bwd = _keep = None
        s = self._state
        half = s["half"]
        dev = s["logits"][0].device
        n = len(self.chains)
        for r in res:
            r[0].hi.record_stream(main)
            if r[0].lo is not None:
                r[0].lo.record_stream(main)
        # input gradients of the heads, rows [fm | adv]: the PQMF-band chains share the bands (one launch sums them), MelGAN reads the waveform
        rows = 2 * half
        sub, wav = s["inputs"][0], s["inputs"][-1]
        bshape = s["bands_shape"]
        gb = torch.zeros((rows,) + bshape[1:], dtype=torch.float32, device=dev) if bshape[1] != self.q else None
        gsub = torch.empty((rows, self.q, sub.shape[2]), dtype=torch.float32, device=dev)
        jobs = (EbenBlHeadJob * (n - 1))(*[self.chains[k].head_job(None, sub.shape[2], res[k][0]) for k in range(n - 1)])
        check(lib.eben_bl_head_dx(jobs, n - 1, rows, ptr(gsub), _stream()), "bl_head_dx")
        if gb is None:
            gb = gsub
        else:
            gb[:, -self.q:, :] = gsub
        ga = torch.empty((rows, 1, wav.shape[2]), dtype=torch.float32, device=dev)
        jobs = (EbenBlHeadJob * 1)(self.chains[-1].head_job(None, wav.shape[2], res[-1][0]))
        check(lib.eben_bl_head_dx(jobs, 1, rows, ptr(ga), _stream()), "bl_head_dx")
        if want_param_grads and not split:
            pend = [None] * n
            for i in [n - 1] + list(range(n - 1)):   # the longest chain first
                with torch.cuda.stream(self._streams[i]):
                    pend[i] = self._launch_weight_grads(s, i, res[i][1], res[i][2], half)
            self._pending = (pend, s, res)   # keeps the saved activations and the stacked gradients alive until the kernels have run
        self._state = None
        return gb[:half], ga[:half], gb[half:], ga[half:]
