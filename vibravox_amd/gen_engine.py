"""Forward / backward of the EBEN generator's convolutional core as explicit kernel chains.

``EBENGenerator.forward`` (``vibravox/torch_modules/dnn/eben_generator.py:168-213``) is a chain of 45 small-channel conv
layers; through ``torch.autograd`` every ResidualUnit (:287-316) costs three launches and seven tensor passes forward
(dilated conv, pointwise conv, add) and, backward, two input-gradient launches, a reflect fold, two weight-gradient
launches and the adds autograd inserts where a gradient fans in.  This engine runs the core -- PQMF analysis, first conv,
encoder blocks, latent convs, decoder blocks: everything up to the input of ``last_conv`` -- outside autograd:

  * forward: one fused launch per ResidualUnit (``eben_ru_fwd``: x read once; h = dilated(x) and u = lrelu(pointwise(h)) kept
    for the backward), the shared LeakyReLU of the block inputs (:187-189) folded into that launch's tile staging;
  * backward: per ResidualUnit ``g_h = pointwise^T(g_y * lrelu'(u))`` and ``g_x = (g_y + fold(dilated^T(g_h))) * lrelu'(x) +
    skip`` -- the residual / skip joins ride in the reflect-fold pass (``eben_conv1d_bwd_dx_res``), no add kernel runs; the
    weight gradients are launched beside the input-gradient chain (``ops.weight_grads``: side stream, straight into ``.grad``
    or the data-parallel buckets);
  * ``last_conv`` (the leaf of the loss balancing, eben.py:223), the tanh recomposition and the PQMF synthesis stay in
    autograd: the step differentiates them three extra times with ``retain_graph`` and they are cheap.

Arithmetic: the forward (training, evaluation and the exported output alike) runs on the bf16 matrix pipe with every fp32 operand
entering as three bf16 pieces (``RU_FWD_MATH`` / ``CONV_FWD_MATH`` = EBEN_MATH_BF16X6: six piece products, dropped terms <= 2^-26 of
a product, fp32 accumulation) -- fp32-grade, held by ``tests/test_gpu_ops.py`` to the fp32 MFMA kernels' own bound against fp64;
``EBEN_RU_FWD_MATH=f32 EBEN_GEN_CONV_FWD_MATH=f32`` selects the ``v_mfma_f32_*_f32`` kernels.  The module tree, parameters and
``state_dict`` are untouched -- this is only how ``EBENGenerator.forward`` executes.

The core's parameter gradients are produced by the engine as a side effect of ``backward`` (written to ``.grad`` / the data-parallel
buckets, never returned to autograd): ``torch.autograd.grad(loss, core parameters)`` returns nothing useful for them -- differentiate
with ``backward()`` and read ``.grad``, as the train step does (only ``last_conv``, outside the core, is reached through
``torch.autograd.grad``: eben.py:223-227).
"""
from __future__ import annotations

import ctypes
import dataclasses
import contextlib
import os
from typing import Dict, List, Optional

import torch

from . import ops
from ._lib import check, load, ptr, stream

USE_FUSED_RU = os.environ.get("EBEN_RU_FUSED", "1") != "0"
USE_FUSED_RU_BWD = os.environ.get("EBEN_RU_FUSED_BWD", "1") != "0"
USE_FUSED_RU_DW = os.environ.get("EBEN_RU_FUSED_DW", "1") != "0"
#: bf16 generator backward: the units' saved tensors at rest as bf16 bundles + sign bytes, written by the forward, and the bundle-layout
#: backward / weight-gradient launches (csrc/ru_bl.hip) -- 32 bytes per element and unit instead of 52-56
USE_RU_BL = os.environ.get("EBEN_RU_BL", "1") != "0"
#: data-parallel runs: report the generator's gradients to the bucket exchange at the join of its weight gradients instead of group by group
#: from the side stream (see DiscriminatorEngineBL.defer_mark_ready: a collective issued early waits on the communication stream and stalls
#: its hardware queue's other streams; [MI355X, single-rank RCCL] 9.18 -> 9.06 ms per step).  EBEN_G_DEFER_READY=0: the early form.
DEFER_MARK_READY = os.environ.get("EBEN_G_DEFER_READY", "1") != "0"
USE_GRAPHS = os.environ.get("EBEN_GEN_GRAPHS", "1") != "0"   # training forward / backward sequences replayed as HIP graphs
#: backward segments (3 decoder blocks, latent convs, 3 encoder blocks) per replayed graph.  A group's weight gradients start on the side
#: stream when its whole input-gradient graph has been issued, and what the main stream waits for in front of the generator's Adam is the
#: LAST group's weight gradients: one segment per graph keeps that tail at one encoder block ([MI355X, same box, 2 x alternated] "2,2,3"
#: 8.839 / 8.842, "2,2,2,1" 8.865 / 8.845, "3,3,1" 9.008 / 8.982, one per graph 8.795 / 8.809 ms per step; rounds 3-5 were bound by the host's
#: enqueue time -- a replay costs it ~70 us -- and ran "2,2,3")
BWD_GROUPS = os.environ.get("EBEN_GEN_BWD_GROUPS", "1,1,1,1,1,1,1")
# arithmetic of the fused ResidualUnit launches (include/eben_hip.h, eben_ru_*_ex): the forward and the fp32 backward run on the bf16
# matrix pipe with three bf16 pieces per operand (EBEN_MATH_BF16X6: every mantissa bit of the fp32 operands, fp32 accumulate --
# fp32 arithmetic at 6/16 of the fp32 MFMA's cost); "f32" selects the v_mfma_f32_32x32x2_f32 kernels (bisecting aid).
_RU_MATH = {"f32": ops.MATH_F32, "bf16x6": ops.MATH_BF16X6, "bf16x3": ops.MATH_BF16X3, "bf16": ops.MATH_BF16}
RU_FWD_MATH = _RU_MATH[os.environ.get("EBEN_RU_FWD_MATH", "bf16x6")]
# the arithmetic of the ResidualUnit forwards right now: EBENLightningModule's engine train step of the bf16-mixed plan runs inside
# ``forward_math("bf16x3")`` (hi + lo operands, three piece products; the weight images are cached per arithmetic and the prepack behind
# the optimiser step -- inside the same block -- rebuilds the ones of the step that just ran); everything outside a train step --
# validation, prediction, from_pretrained inference -- runs RU_FWD_MATH
_ru_fwd_math = [RU_FWD_MATH]


def ru_forward_math() -> int:
    return _ru_fwd_math[0]


def set_ru_forward_math(name=None) -> None:
    """``None`` restores the default (``EBEN_RU_FWD_MATH``, fp32-grade six-product arithmetic)."""
    _ru_fwd_math[0] = RU_FWD_MATH if name is None else _RU_MATH[name]
RU_BWD_F32_MATH = _RU_MATH[os.environ.get("EBEN_RU_BWD_F32_MATH", "bf16x6")]
# forward of the other conv layers of the core (first / strided / latent / transposed convs): the same fp32-grade split form
CONV_FWD_MATH = _RU_MATH[os.environ.get("EBEN_GEN_CONV_FWD_MATH", "bf16x6")]
_conv_fwd_math = [CONV_FWD_MATH]   # like _ru_fwd_math: the strided / transposed / latent convs of the generator


def conv_forward_math() -> int:
    return _conv_fwd_math[0]


def set_forward_math(name=None) -> None:
    """Arithmetic of the generator's forward from now on: ``"bf16x3"`` (hi + lo operands: the bf16-mixed plan) or ``None`` = the defaults
    (``EBEN_RU_FWD_MATH`` / ``EBEN_GEN_CONV_FWD_MATH``: fp32-grade six-product arithmetic)."""
    set_ru_forward_math(name)
    _conv_fwd_math[0] = CONV_FWD_MATH if name is None else _RU_MATH[name]


@contextlib.contextmanager
def forward_math(name=None):
    """``set_forward_math(name)`` for the duration of a ``with`` block, the previous arithmetic restored on exit (also on an exception):
    the train step of the bf16-mixed plan runs inside one, so that a validation / prediction forward of the same process computes in the
    default fp32-grade arithmetic whether or not a train step ran before it."""
    saved = (_ru_fwd_math[0], _conv_fwd_math[0])
    set_forward_math(name)
    try:
        yield
    finally:
        _ru_fwd_math[0], _conv_fwd_math[0] = saved


#: strided convs from this stride up run as space-to-depth + stride-1 tap-conv where the tap-conv does not cover them directly (0: never)
S2D_MIN_STRIDE = int(os.environ.get("EBEN_GEN_S2D_MIN_STRIDE", "8"))


def _params(m):
    return ops.conv_params(m)


class _ConvRec:
    """What the backward of one conv launch needs."""

    __slots__ = ("m", "spec", "d", "x", "y", "wp_bwd", "norm", "scale")

    def __init__(self, m, spec, d, x, y, wp_bwd, norm, scale=None):
        self.m, self.spec, self.d, self.x, self.y, self.wp_bwd, self.norm, self.scale = m, spec, d, x, y, wp_bwd, norm, scale


class GeneratorEngine:
    def __init__(self, gen):
        self.gen = gen
        self._spec_cache: Dict[tuple, ops.ConvSpec] = {}
        self._ru_images: Dict[int, dict] = {}   # id(ResidualUnit) -> {key, fwd image, bwd images per math, scales}
        self._s2d_images: Dict[int, tuple] = {}  # id(conv) -> (key, image of the space-to-depth form, permuted weights)
        self._prepack_graph = ops.ReplayedPrepack()
        self._ru_batch = None
        # the training forward and, per backward segment, the input-gradient and the weight-gradient launches as replayed HIP graphs (ops.ReplayedChain)
        self._fwd_graph = ops.ReplayedChain()
        self._dx_graphs: List[ops.ReplayedChain] = []   # per backward segment (decoder block / latent / encoder block)
        self._dw_graphs: List[ops.ReplayedChain] = []
        self._static: Dict[tuple, torch.Tensor] = {}   # (name, shape) -> buffer the graphs read their per-step input from
        self._dwq: Optional[list] = None               # while the input-gradient chain runs in queue mode: its weight-gradient work items
        self._core_convs = None

    # ---- helpers ------------------------------------------------------------------------------------
    def _spec(self, m, in_slope=None, out_slope=None) -> ops.ConvSpec:
        if in_slope is None and out_slope is None:
            return m.spec
        key = (id(m), in_slope, out_slope)
        s = self._spec_cache.get(key)
        if s is None:
            kw = {}
            if in_slope is not None:
                kw["in_slope"] = float(in_slope)
            if out_slope is not None:
                kw["out_slope"] = float(out_slope)
            s = self._spec_cache[key] = dataclasses.replace(m.spec, **kw)
        return s

    def _pack(self, m, spec, batch, l_in, train):
        d = ops.conv_desc(spec, batch, l_in, conv_forward_math())   # layers the split bf16 tap-conv does not cover run their fp32 kernel
        d_bwd = ops.conv_desc(spec, batch, l_in, ops._backward_math[0])
        v, g = _params(m)
        pw = ops.pack_weights(m.spec, d, v.detach(), None if g is None else g.detach(), m._packed, train, d_bwd)
        return d, d_bwd, pw

    # ---- stride >= 8 as space-to-depth ---------------------------------------------------------------------------------------
    # The split-operand tap-conv stages the input tile of 128 outputs through a per-thread prefetch that a stride-8 layer's 1032
    # positions exceed: EncBlock's last strided conv and the input gradient of DecBlock's first transposed conv fell to the exact-fp32
    # kernel (0.15 ms each, the longest launches of the generator).  With the time axis folded into channels (`eben_space_to_depth`:
    # channel (c, r) = samples S q + r) the same contraction is a stride-1 conv with k / S taps over C S channels, which it covers.
    def _s2d_route(self, m, spec, batch: int, l_q: int, kq: int, math: int, c: int, rows_out: int, in_slope: float, out_slope: float, scale):
        """(descriptor, packed image) of the stride-1 form, or None when the tap-conv does not cover it either."""
        lib = load()
        alt = self._spec_cache.get((id(m), "s2d", in_slope, out_slope))
        if alt is None:
            alt = self._spec_cache[(id(m), "s2d", in_slope, out_slope)] = ops.ConvSpec(
                c_in=c * spec.stride, c_out=rows_out, ksize=kq, in_slope=float(in_slope), out_slope=float(out_slope))
        d = ops.conv_desc(alt, batch, l_q, math)
        if lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0) != 4:
            return None
        v, g = _params(m)
        e = ops._storage_epoch
        key = (v.data_ptr(), v._version, e.get(v.data_ptr(), 0), None if g is None else (g.data_ptr(), g._version), e.get(-1, 0), batch, l_q, math)
        hit = self._s2d_images.get(id(m))
        if hit is None or hit[0] != key:
            vv = v.detach().view(rows_out, c, kq, spec.stride).permute(0, 1, 3, 2).reshape(rows_out, c * spec.stride, kq).contiguous()
            wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 0), dtype=torch.float32, device=v.device)
            ops.conv1d_pack(d, vv, scale, wp, None)
            hit = self._s2d_images[id(m)] = (key, wp, vv)
        return d, hit[1]

    @staticmethod
    def _s2d_wanted(spec, math: int) -> bool:
        return (S2D_MIN_STRIDE > 0 and spec.stride >= S2D_MIN_STRIDE and spec.ksize % spec.stride == 0 and spec.groups == 1
                and spec.dilation == 1 and math != ops.MATH_F32)

    def _conv(self, m, x, train, in_slope=None, recs=None):
        """y = conv layer ``m`` on x (its own fused output activation; ``in_slope``: LeakyReLU on load)."""
        lib = load()
        spec = self._spec(m, in_slope=in_slope)
        b, _, l_in = x.shape
        d, d_bwd, pw = self._pack(m, spec, b, l_in, train)
        y = torch.empty((b, spec.c_out, d.l_out), dtype=torch.float32, device=x.device)
        route = None
        if not spec.transposed and self._s2d_wanted(spec, conv_forward_math()) and lib.eben_conv1d_kernel_generation(ctypes.byref(d), 0) not in (4, 5):
            kq = spec.ksize // spec.stride
            l_q = d.l_out + kq - 1
            route = self._s2d_route(m, spec, b, l_q, kq, conv_forward_math(), spec.c_in, spec.c_out, spec.in_slope, spec.out_slope, pw.scale)
        if route is not None:
            xq = torch.empty((b, spec.c_in * spec.stride, l_q), dtype=torch.float32, device=x.device)
            check(lib.eben_space_to_depth(ptr(x), None, 1.0, ptr(xq), b * spec.c_in, l_in, spec.stride, -spec.pad_l, l_q,
                                          1 if spec.reflect else 0, stream()), "space_to_depth")
            check(lib.eben_conv1d_fwd(ctypes.byref(route[0]), ptr(xq), ptr(route[1]), ptr(m.bias), None, ptr(y), stream()), "conv1d_fwd")
        else:
            check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(x), ptr(pw.wp_fwd), ptr(m.bias), None, ptr(y), stream()), "conv1d_fwd")
        if recs is not None:
            recs.append(_ConvRec(m, spec, d_bwd, x, y if spec.out_slope != 1.0 else None, pw.wp_bwd, pw.norm, pw.scale))
        return y

    @staticmethod
    def _ru_bwd_math() -> int:
        """Math of the fused backward launch: plain bf16 operands next to a bf16 generator backward (``ops.backward_math``), else
        the fp32-grade split."""
        return ops.MATH_BF16 if ops._backward_math[0] != ops.MATH_F32 else RU_BWD_F32_MATH

    def _ru_image(self, ru, which: int = 0) -> torch.Tensor:
        """Weight images of the fused unit (both convs, weight-norm scales folded in): 0 forward, 1 backward (transposed, in the
        math of the current backward).  Rebuilt when a parameter changed: the forward image and the backward image of every math
        this unit has been differentiated in (so that ``prepack``, which runs outside the step's math context, rebuilds the
        right ones)."""
        lib = load()
        vd, gd = _params(ru.dilated_conv)
        vp, gp = _params(ru.pointwise_conv)
        e = ops._storage_epoch
        key = tuple((t.data_ptr(), t._version, e.get(t.data_ptr(), 0)) for t in (vd, gd, vp, gp)) + (e.get(-1, 0), ru_forward_math())
        hit = self._ru_images.get(id(ru))
        bm = self._ru_bwd_math() if which == 1 else None
        if hit is not None and hit["key"] == key and (bm is None or bm in hit["bwd"]):
            return hit["fwd"] if which == 0 else hit["bwd"][bm]
        c = vd.shape[0]
        dev = vd.device
        batch = self._ru_batch   # inside prepack(): the launches of all units are gathered into one wn_scale / one pack call
        reuse = batch is not None   # prepack(): rebuild into the buffers held (a graph replay writes the ones it captured)
        if hit is None or hit["key"] != key:
            # scale / norm of the dilated, then of the pointwise conv
            scales = ops._buffer(None if hit is None else hit["scales"], 4 * c, vd, reuse).view(4, c)
            wn = [(gd.detach(), vd.detach(), c, vd.numel() // c, scales[0], scales[1]), (gp.detach(), vp.detach(), c, vp.numel() // c, scales[2], scales[3])]
            if batch is not None:
                batch["wn"].extend(wn)
            else:
                ops.wn_scale_multi(wn)
            img = ops._buffer(None if hit is None else hit["fwd"], lib.eben_ru_packed_floats_ex(c, ru_forward_math()), vd, reuse)
            self._ru_pack(c, ru_forward_math(), 0, vd, scales[0], vp, scales[2], img)
            maths = set(hit["bwd"]) if hit is not None else set()
            old_bwd = hit["bwd"] if hit is not None else {}
            hit = self._ru_images[id(ru)] = {"key": key, "fwd": img, "bwd": {}, "scales": scales}
        else:
            maths, old_bwd = set(), {}
        if bm is not None:
            maths.add(bm)
        scales = hit["scales"]
        for m in maths:
            img_b = ops._buffer(old_bwd.get(m), lib.eben_ru_packed_floats_ex(c, m), vd, reuse)
            self._ru_pack(c, m, 1, vd, scales[0], vp, scales[2], img_b)
            hit["bwd"][m] = img_b
        return hit["fwd"] if which == 0 else hit["bwd"][bm]

    def _ru_pack(self, c, math, which, vd, sd, vp, sp, img) -> None:
        if self._ru_batch is not None and math != ops.MATH_F32:
            self._ru_batch["packs"].append((c, math, which, vd.detach(), sd, vp.detach(), sp, img))
            return
        check(load().eben_ru_pack_ex(c, math, which, ptr(vd.detach()), ptr(sd), ptr(vp.detach()), ptr(sp), ptr(img), stream()), "ru_pack")

    def _ru_flush(self) -> None:
        """Issues what ``_ru_image`` gathered: the weight-norm scales of all units as one launch, then their images as one."""
        batch, self._ru_batch = self._ru_batch, None
        if batch is None:
            return
        ops.wn_scale_multi(batch["wn"])
        if batch["packs"]:
            from ._lib import EbenRuPackJob

            table = (EbenRuPackJob * len(batch["packs"]))()
            for it, (c, math, which, vd, sd, vp, sp, img) in zip(table, batch["packs"]):
                it.channels, it.math, it.which = c, math, which
                it.v_dil, it.scale_dil, it.v_pw, it.scale_pw, it.wimg = ptr(vd), ptr(sd), ptr(vp), ptr(sp), ptr(img)
            check(load().eben_ru_pack_multi(table, len(table), stream()), "ru_pack_multi")

    def prepack(self) -> None:
        """Rebuilds the fused units' weight images on the side stream (called with ``ops.prepack`` after the optimiser step);
        replayed as a graph once the sequence has settled (``ops.ReplayedPrepack``)."""
        if not self._ru_images:
            return
        units = [ru for blk in list(self.gen.encoder_blocks) + list(self.gen.decoder_blocks) for ru in blk.residuals]
        dev = _params(units[0].dilated_conv)[0].device
        main = torch.cuda.current_stream(dev)
        side = ops._side_stream(dev)
        side.wait_stream(main)

        def key_of(ru):
            e = ops._storage_epoch
            ts = _params(ru.dilated_conv) + _params(ru.pointwise_conv)
            return tuple((t.data_ptr(), t._version, e.get(t.data_ptr(), 0)) for t in ts) + (e.get(-1, 0), ru_forward_math())

        def body():
            self._ru_batch = {"wn": [], "packs": []}
            try:
                for ru in units:
                    self._ru_image(ru)
            finally:
                self._ru_flush()

        def entry(ru):
            # what the launch sequence depends on besides the weights' values, the buffers it writes included (ops.ReplayedPrepack)
            hit = self._ru_images.get(id(ru))
            return (id(ru), _params(ru.dilated_conv)[0].data_ptr(),
                    None if hit is None else (tuple((m, hit["bwd"][m].data_ptr()) for m in sorted(hit["bwd"])), hit["fwd"].data_ptr(),
                                              hit["scales"].data_ptr(), hit["key"] != key_of(ru)))

        sig = tuple(entry(ru) for ru in units) + (ops._storage_epoch.get(-1, 0),)
        with torch.cuda.stream(side), torch.no_grad():
            if self._prepack_graph.run(sig, body, side):
                for ru in units:   # replayed: the images are current, the cache keys are not
                    self._ru_images[id(ru)]["key"] = key_of(ru)
            ev = torch.cuda.Event()
            ev.record()
        self._prepacked = ev

    def _residual_unit(self, ru, x, in_slope, train, recs):
        """y = xin + lrelu(pointwise(dilated(xin))), xin = lrelu(x, in_slope).  Returns y; records (x, h, u) for the backward."""
        lib = load()
        b, c, l = x.shape
        dil, pwc = ru.dilated_conv, ru.pointwise_conv
        if train:
            # the backward images of the two convs (and their weight-norm norms) come from the per-layer caches
            spec_d = self._spec(dil, in_slope=in_slope if in_slope != 1.0 else None)
            _, dd_bwd, pw_d = self._pack(dil, spec_d, b, l, True)
            _, dp_bwd, pw_p = self._pack(pwc, pwc.spec, b, l, True)
        fwd_math = ru_forward_math()
        fusable = dil.spec.ksize == 3 and dil.spec.reflect and lib.eben_ru_supported(c, dil.spec.dilation, fwd_math) == 1
        if (train and USE_RU_BL and USE_FUSED_RU and USE_FUSED_RU_BWD and USE_FUSED_RU_DW and fusable and fwd_math in (ops.MATH_BF16X6, ops.MATH_BF16X3)
                and self._ru_bwd_math() == ops.MATH_BF16 and lib.eben_rubl_supported(c, dil.spec.dilation) == 1 and self._ru_bl_params_ok(ru, l)):
            # what the bf16 backward reads, written once in the layout its MFMA operands want (csrc/ru_bl.hip)
            img = self._ru_image(ru)
            y = torch.empty_like(x)
            xb = torch.empty((b, c // 8, l, 8), dtype=torch.bfloat16, device=x.device)
            hb = torch.empty_like(xb)
            um = torch.empty((b, c // 8, l), dtype=torch.uint8, device=x.device)
            check(lib.eben_rubl_fwd(fwd_math, b, c, l, dil.spec.dilation, ptr(x), float(in_slope), float(pwc.spec.out_slope), ptr(img), ptr(y),
                                    xb.data_ptr(), hb.data_ptr(), um.data_ptr(), stream()), "rubl_fwd")
            recs.append((("bl", self._ru_image(ru, 1), xb, hb, um), _ConvRec(dil, spec_d, dd_bwd, x if in_slope != 1.0 else None, None, pw_d.wp_bwd, pw_d.norm),
                         _ConvRec(pwc, pwc.spec, dp_bwd, None, None, pw_p.wp_bwd, pw_p.norm)))
            return y
        if USE_FUSED_RU and fusable:
            img = self._ru_image(ru)
            y = torch.empty_like(x)
            h = torch.empty_like(x) if train else None
            u = torch.empty_like(x) if train else None
            check(lib.eben_ru_fwd_ex(fwd_math, b, c, l, dil.spec.dilation, ptr(x), float(in_slope), float(pwc.spec.out_slope), ptr(img), ptr(y),
                                     ptr(h), ptr(u), stream()), "ru_fwd")
        else:   # layer by layer (bisecting aid, EBEN_RU_FUSED=0)
            xin = x
            if in_slope != 1.0:
                xin = torch.empty_like(x)
                check(lib.eben_lrelu_fwd(ptr(x), ptr(xin), x.numel(), float(in_slope), stream()), "lrelu_fwd")
            h = self._conv(dil, xin, train)
            u = self._conv(pwc, h, train)
            y = torch.empty_like(x)
            check(lib.eben_add(ptr(xin), ptr(u), ptr(y), x.numel(), stream()), "add")
        if train:
            bm = self._ru_bwd_math()
            fused = USE_FUSED_RU_BWD and dil.spec.ksize == 3 and dil.spec.reflect and lib.eben_ru_supported(c, dil.spec.dilation, bm) == 1
            recs.append(((self._ru_image(ru, 1), bm) if fused else None, _ConvRec(dil, spec_d, dd_bwd, x, None, pw_d.wp_bwd, pw_d.norm),
                         _ConvRec(pwc, pwc.spec, dp_bwd, h, u, pw_p.wp_bwd, pw_p.norm)))
        return y

    # ---- forward -------------------------------------------------------------------------------------
    def forward(self, cut_audio: torch.Tensor, train: bool):
        """Returns (input of last_conv, first_bands, records for ``backward`` or None)."""
        gen = self.gen
        lib = load()
        ev = getattr(self, "_prepacked", None)
        if ev is not None:
            self._prepacked = None
            torch.cuda.current_stream().wait_event(ev)
        slope = gen.nl.negative_slope
        first_bands = gen.pqmf(cut_audio, "analysis", bands=gen.p).detach()
        saved = {"enc": [], "dec": [], "misc": []} if train else None
        a = self._conv(gen.first_conv, first_bands, train, recs=saved["misc"] if train else None)
        skips = []
        for blk in gen.encoder_blocks:
            recs = [] if train else None
            cur = a
            for k, ru in enumerate(blk.residuals):
                cur = self._residual_unit(ru, cur, slope if k == 0 else 1.0, train, recs)
            a = self._conv(blk.conv, cur, train, recs=recs)
            skips.append(a)
            if train:
                saved["enc"].append(recs)
        lat = [] if train else None
        l1 = self._conv(gen.latent_conv[1], a, train, in_slope=slope, recs=lat)
        cur = self._conv(gen.latent_conv[3], l1, train, recs=lat)
        if train:
            saved["latent"] = lat
        for blk, skip in zip(gen.decoder_blocks, reversed(skips)):
            recs = [] if train else None
            s = torch.empty_like(cur)
            check(lib.eben_add(ptr(cur), ptr(skip), ptr(s), cur.numel(), stream()), "add")
            cur = self._conv(blk.conv_trans, s, train, recs=recs)
            for ru in blk.residuals:
                cur = self._residual_unit(ru, cur, 1.0, train, recs)
            if train:
                saved["dec"].append(recs)
        return cur, first_bands, saved

    # ---- backward ------------------------------------------------------------------------------------
    def _dx(self, rec: _ConvRec, dy, res_pre=None, res_post=None):
        lib = load()
        d = rec.d
        spec = rec.spec
        if (spec.transposed and res_pre is None and res_post is None and spec.in_slope == 1.0 and spec.output_padding == 0
                and self._s2d_wanted(spec, d.math) and lib.eben_conv1d_kernel_generation(ctypes.byref(d), 1) != 4):
            # input gradient of ConvTranspose1d = stride-S conv of the (masked) output gradient with the same (in, out, k) weights
            b, c_dy, l_dy = dy.shape
            kq = spec.ksize // spec.stride
            l_q = rec.x.shape[2] + kq - 1
            route = self._s2d_route(rec.m, spec, b, l_q, kq, d.math, c_dy, spec.c_in, 1.0, 1.0, rec.scale)
            if route is not None:
                gq = torch.empty((b, c_dy * spec.stride, l_q), dtype=torch.float32, device=dy.device)
                check(lib.eben_space_to_depth(ptr(dy), ptr(rec.y) if spec.out_slope != 1.0 else None, spec.out_slope, ptr(gq), b * c_dy, l_dy,
                                              spec.stride, -spec.pad_l, l_q, 0, stream()), "space_to_depth")
                dx = torch.empty_like(rec.x)
                check(lib.eben_conv1d_fwd(ctypes.byref(route[0]), ptr(gq), ptr(route[1]), None, None, ptr(dx), stream()), "conv1d_fwd")
                return dx
        ws_bytes = getattr(d, "_dx_ws", None)
        if ws_bytes is None:
            ws_bytes = d._dx_ws = lib.eben_conv1d_bwd_dx_workspace(ctypes.byref(d))
        ws = ops._empty(ws_bytes, dy) if ws_bytes else None
        dx = torch.empty_like(rec.x)
        xmask = rec.x if rec.spec.in_slope != 1.0 else None
        check(lib.eben_conv1d_bwd_dx_res(ctypes.byref(d), ptr(dy), ptr(rec.y), ptr(rec.wp_bwd), ptr(xmask), ptr(res_pre), ptr(res_post), ptr(dx),
                                         ptr(ws), ws_bytes, stream()), "conv1d_bwd_dx_res")
        return dx

    def _dw(self, rec: _ConvRec, dy):
        if ops._skip_weight_grads[0]:
            return
        if self._dwq is not None:
            self._dwq.append(("conv", rec, dy))
            return
        v, g = _params(rec.m)
        if not v.requires_grad:
            return
        grads = ops.weight_grads(rec.d, dy, rec.y, rec.x, v, g, rec.m.bias, rec.norm)
        for p, t in zip((v, g, rec.m.bias), grads):   # not deferred (no side-stream context): accumulate like autograd would
            if p is not None and t is not None and p.requires_grad:
                p.grad = t if p.grad is None else p.grad.add_(t)
                for hook in list((getattr(p, "_post_accumulate_grad_hooks", None) or {}).values()):
                    hook(p)   # e.g. ddp.GradSync's bucket accounting

    @staticmethod
    def _accumulate(params, grads) -> None:
        for p, t in zip(params, grads):   # not deferred (no side-stream context): accumulate like autograd would
            if p is not None and t is not None and p.requires_grad:
                p.grad = t if p.grad is None else p.grad.add_(t)
                for hook in list((getattr(p, "_post_accumulate_grad_hooks", None) or {}).values()):
                    hook(p)   # e.g. ddp.GradSync's bucket accounting

    def _ru_dw(self, dil: _ConvRec, pwc: _ConvRec, gy, gh, bm: int) -> None:
        """Weight gradients of both convs of a fused unit: one ``eben_ru_dw`` launch (reduction along time, operands straight from
        the activation tensors) where both are weight-normalised bias-free parameters, else layer by layer."""
        if ops._skip_weight_grads[0]:
            return
        if self._dwq is not None:
            self._dwq.append(("ru", dil, pwc, gy, gh, bm))
            return
        (vd, gd), (vp, gp) = _params(dil.m), _params(pwc.m)
        fusable = (USE_FUSED_RU_DW and gd is not None and gp is not None and dil.m.bias is None and pwc.m.bias is None
                   and vd.requires_grad and vp.requires_grad and gd.requires_grad and gp.requires_grad and bm in (ops.MATH_BF16, ops.MATH_BF16X6))
        res = False
        if fusable:
            res = ops.weight_grads_ru(bm, dil.spec.dilation, gy, pwc.y, float(pwc.spec.out_slope), pwc.x, gh, dil.x, float(dil.spec.in_slope),
                                      (vp, gp, pwc.norm), (vd, gd, dil.norm))
        if res is False:
            self._dw(pwc, gy)
            self._dw(dil, gh)
            return
        self._accumulate((vp, gp, None), res[0])
        self._accumulate((vd, gd, None), res[1])

    @staticmethod
    def _ru_bl_params_ok(ru, length: int) -> bool:
        """The bundle-layout weight-gradient launch serves weight-normalised, bias-free, trainable convs (every unit of the reference) whose
        gradients take the SAME route (both or neither hold a gradient / carry a hook: the unit's two weight gradients are one launch) on
        rows longer than the dilation (the kernels' reflect window); anything else keeps the fp32-at-rest path, which can fall back conv
        by conv -- the bundle path saves only bf16 planes and cannot."""
        keys = []
        for m in (ru.dilated_conv, ru.pointwise_conv):
            v, g = _params(m)
            if g is None or m.bias is not None or not (v.requires_grad and g.requires_grad):
                return False
            keys.append((v.grad is None and g.grad is None, ops._no_grad_hooks(v) and ops._no_grad_hooks(g)))
        return keys[0] == keys[1] and ru.dilated_conv.spec.dilation < length

    def _ru_backward_bl(self, rec, gy, res_post=None):
        (_, img_b, xb, hb, um), dil, pwc = rec
        b, c, l = gy.shape
        gx = torch.empty_like(gy)
        gzb, ghb = torch.empty_like(xb), torch.empty_like(xb)
        ins = dil.spec.in_slope
        check(load().eben_rubl_bwd(b, c, l, dil.spec.dilation, ptr(gy), um.data_ptr(), float(pwc.spec.out_slope), ptr(dil.x) if ins != 1.0 else None,
                                   float(ins), ptr(res_post), ptr(img_b), ptr(gx), gzb.data_ptr(), ghb.data_ptr(), stream()), "rubl_bwd")
        if not ops._skip_weight_grads[0]:
            if self._dwq is not None:
                self._dwq.append(("rubl", dil, pwc, gzb, hb, ghb, xb))
            else:
                (vd, gd), (vp, gp) = _params(dil.m), _params(pwc.m)
                res = ops.weight_grads_ru_bl(dil.spec.dilation, gzb, hb, ghb, xb, (vp, gp, pwc.norm), (vd, gd, dil.norm))
                if res is False:
                    raise RuntimeError("bundle-layout ResidualUnit: the two convs' weight gradients are routed differently (one of them holds a gradient "
                                       "or a hook the other does not); set EBEN_RU_BL=0 for this pattern")
                self._accumulate((vp, gp, None), res[0])
                self._accumulate((vd, gd, None), res[1])
        return gx

    def _ru_backward(self, rec, gy, res_post=None):
        fused, dil, pwc = rec
        if fused is not None and fused[0] == "bl":
            return self._ru_backward_bl(rec, gy, res_post)
        if fused is not None:   # one launch: g_h and g_x = (g_y + fold(dilated^T g_h)) * lrelu'(x) + skip gradient
            b, c, l = gy.shape
            gx, gh = torch.empty_like(gy), torch.empty_like(gy)
            ins = dil.spec.in_slope
            img_b, bm = fused
            check(load().eben_ru_bwd_ex(bm, b, c, l, dil.spec.dilation, ptr(gy), ptr(pwc.y), float(pwc.spec.out_slope),
                                        ptr(dil.x) if ins != 1.0 else None, float(ins), ptr(res_post), ptr(img_b), ptr(gx), ptr(gh), stream()), "ru_bwd")
            self._ru_dw(dil, pwc, gy, gh, bm)
            return gx
        gh = self._dx(pwc, gy)                       # pointwise^T(gy * lrelu'(u))
        self._dw(pwc, gy)
        gx = self._dx(dil, gh, res_pre=gy, res_post=res_post)   # (gy + fold(dilated^T gh)) * lrelu'(x) + skip gradient
        self._dw(dil, gh)
        return gx

    def _n_segments(self, saved) -> int:
        return len(saved["dec"]) + 1 + len(saved["enc"])

    def _bwd_segment(self, saved, k: int, g: torch.Tensor, skip_grads: list) -> torch.Tensor:
        """Segment k of the backward -- decoder blocks (last first), the latent convs, encoder blocks (last first; the first block's
        segment ends with first_conv's weight gradient): takes the gradient at the segment's output, returns the one at its input.
        ``skip_grads`` collects the decoder segments' results (the gradients that join the encoder outputs)."""
        n_dec, n_enc = len(saved["dec"]), len(saved["enc"])
        if k < n_dec:                                 # decoder blocks, last first: [conv_trans, ru, ru, ru]
            recs = saved["dec"][n_dec - 1 - k]
            for rec in reversed(recs[1:]):
                g = self._ru_backward(rec, g)
            ct = recs[0]
            gs = self._dx(ct, g)                      # gradient at (x + skip)
            self._dw(ct, g)
            skip_grads.append(gs)                     # dec2 -> a1, dec1 -> a2, dec0 -> a3
            return gs
        if k == n_dec:
            c1, c2 = saved["latent"]
            gl1 = self._dx(c2, g)
            self._dw(c2, g)
            g = self._dx(c1, gl1, res_post=skip_grads[-1])   # a3: latent path (masked by lrelu'(a3)) + the decoder's skip
            self._dw(c1, gl1)
            return g
        i = n_enc - 1 - (k - n_dec - 1)               # encoder blocks, last first: [ru, ru, ru, conv]
        recs = saved["enc"][i]
        conv = recs[-1]
        gy = self._dx(conv, g)
        self._dw(conv, g)
        for rec in reversed(recs[1:-1]):
            gy = self._ru_backward(rec, gy)
        # the block's input a_i is also a decoder block's skip: that gradient joins behind the activation mask
        post = skip_grads[i - 1] if i >= 1 else None   # skip_grads: [a1 (dec2), a2 (dec1), a3 (dec0)]
        g = self._ru_backward(recs[0], gy, res_post=post)
        if i == 0:
            self._dw(saved["misc"][0], g)             # first_conv: its input is data, only the weight gradient
        return g

    @torch.no_grad()
    def backward(self, saved, g_pre: torch.Tensor) -> None:
        """Input-gradient chain on the current stream, weight gradients through ``ops.weight_grads`` (inside
        ``ops.weight_grads_on_side_stream()``: beside the chain, joined by the caller)."""
        g = g_pre.contiguous()
        skip_grads: List[Optional[torch.Tensor]] = []
        for k in range(self._n_segments(saved)):
            g = self._bwd_segment(saved, k, g, skip_grads)

    # ---- the training step's three launch sequences as replayed graphs ----------------------------------------------------------
    # ~36 launches forward, ~50 along the input-gradient chain and ~30 weight-gradient launches per step, each a few microseconds of
    # GPU-side dispatch but 15-30 us of Python + ctypes on the host: 1.2 + 2.2 + 0.5 ms of host time per step ([MI355X]
    # tools/host_times.py) on a 12 ms step whose host side took 11.3 ms in all.  Every per-step input is copied into a static
    # buffer, every tensor the bodies allocate lives in the graph's pool (rewritten in place by the next replay: valid until then,
    # which is as long as the step needs it), and the weight images are the ones prepack() rebuilds in place.
    @staticmethod
    def _graphs_usable() -> bool:
        return USE_GRAPHS and ops.ReplayedChain.enabled and not ops.ReplayedPrepack._multi_rank() and not ops._timers_enabled()

    def _static_buf(self, name: str, like: torch.Tensor) -> torch.Tensor:
        key = (name, tuple(like.shape), like.dtype, like.device)
        buf = self._static.get(key)
        if buf is None:
            if len(self._static) >= 8:   # variable clip lengths: keep the buffers of the recent shapes only
                self._static.pop(next(iter(self._static)))
            buf = self._static[key] = torch.empty_like(like, memory_format=torch.contiguous_format)
        return buf

    def _fwd_sig(self, buf: torch.Tensor) -> tuple:
        """Everything the captured forward depends on besides values: input buffer, arithmetic, and per layer the weight-image
        buffers and whether they are current (a stale image makes the eager path rebuild it -- a replay would not)."""
        if self._core_convs is None:
            gen = self.gen
            from .torch_modules.utils import HipConv1d
            self._core_convs = [m for part in (gen.first_conv, gen.encoder_blocks, gen.latent_conv, gen.decoder_blocks)
                                for m in part.modules() if isinstance(m, HipConv1d)]
            self._core_units = [ru for blk in list(gen.encoder_blocks) + list(gen.decoder_blocks) for ru in blk.residuals]
        addr = lambda t: 0 if t is None else t.data_ptr()
        parts = [buf.data_ptr(), tuple(buf.shape), ops._backward_math[0], conv_forward_math(), ru_forward_math()]
        for m in self._core_convs:
            pw = m._packed
            if pw is None or pw.last is None:
                parts.append(None)
                continue
            v, g = _params(m)
            parts.append((addr(pw.wp_fwd), addr(pw.wp_bwd), addr(pw.scale), addr(pw.norm), pw.key == ops._pack_key(v, g, pw.last[1], pw.last[2])))
        e = ops._storage_epoch
        for ru in self._core_units:
            hit = self._ru_images.get(id(ru))
            if hit is None:
                parts.append(None)
                continue
            ts = _params(ru.dilated_conv) + _params(ru.pointwise_conv)
            key = tuple((t.data_ptr(), t._version, e.get(t.data_ptr(), 0)) for t in ts) + (e.get(-1, 0), ru_forward_math())
            parts.append((hit["fwd"].data_ptr(), tuple((k, hit["bwd"][k].data_ptr()) for k in sorted(hit["bwd"])), hit["scales"].data_ptr(), hit["key"] == key))
        return tuple(parts)

    def forward_train(self, x: torch.Tensor):
        """``forward(x, True)``, replayed as a graph once the sequence has settled.

        A replay REWRITES the outputs and saved activations of the previous replay in place.  Plain PyTorch semantics -- two
        grad-enabled forwards, then a backward through both -- therefore need a guard: every call gets a serial number; while the
        latest replayed forward has not been differentiated (and its autograd node is still alive) the next forward runs eagerly on
        fresh tensors, and ``backward_train`` refuses saved state whose serial is no longer the graph's (a second backward with
        ``retain_graph`` after a newer replay) instead of returning gradients of the wrong forward."""
        self._fwd_replayed = False
        self._serial = getattr(self, "_serial", 0) + 1
        if not self._graphs_usable() or self._replay_outstanding():
            return self.forward(x, True)
        ops.join_prepack()
        ev = getattr(self, "_prepacked", None)
        if ev is not None:   # the waits stay outside the graph
            self._prepacked = None
            torch.cuda.current_stream().wait_event(ev)
        buf = self._static_buf("x", x)
        buf.copy_(x)
        out = self._fwd_graph.run(self._fwd_sig(buf), lambda: self.forward(buf, True), None)
        self._fwd_replayed = self._fwd_graph.graph is not None
        if self._fwd_replayed:
            self._graph_serial = self._serial   # whose forward the graph's buffers hold
        return out

    def _replay_outstanding(self) -> bool:
        """The graph's buffers hold a forward that an autograd node still alive has not been differentiated through."""
        ref = getattr(self, "_outstanding", None)
        if ref is None:
            return False
        if ref() is None:
            self._outstanding = None
            return False
        return True

    def _note_ctx(self, ctx) -> None:
        import weakref
        ctx.serial = self._serial
        self._outstanding = weakref.ref(ctx) if self._fwd_replayed else None

    def _deferrable(self) -> bool:
        """Weight gradients of the whole core go through ``weight_grads_on_side_stream().join()``'s assignment (no gradient held, no
        hook, no data-parallel sink)."""
        if not ops._side["enabled"] or ops._skip_weight_grads[0]:
            return False
        sink = ops._side["sink"]
        for m in self._core_convs:
            for p in _params(m) + (m.bias,):
                if p is None:
                    continue
                if not p.requires_grad:
                    return False
                if sink is not None:
                    if sink.grad_buffer(p) is None:   # data-parallel: every gradient goes straight into its bucket view
                        return False
                elif p.grad is not None or not ops._no_grad_hooks(p):
                    return False
        return True

    def _dw_body(self, queue: list, sink=None) -> list:
        """The queued weight-gradient work on the current stream: every layer's kernel, then ONE slab-sum / weight-norm launch.
        Returns [(parameter, gradient)] (with a data-parallel ``sink``: the gradients are its bucket views)."""
        jobs, assign = [], []
        with ops.collect_wn_jobs(jobs, sink):
            for item in queue:
                if item[0] == "conv":
                    _, rec, dy = item
                    v, g = _params(rec.m)
                    grads = ops.weight_grads(rec.d, dy, rec.y, rec.x, v, g, rec.m.bias, rec.norm)
                    assign.extend((p, t) for p, t in zip((v, g, rec.m.bias), grads) if p is not None and t is not None)
                elif item[0] == "rubl":
                    _, dil, pwc, gzb, hb, ghb, xb = item
                    (vd, gd), (vp, gp) = _params(dil.m), _params(pwc.m)
                    res = ops.weight_grads_ru_bl(dil.spec.dilation, gzb, hb, ghb, xb, (vp, gp, pwc.norm), (vd, gd, dil.norm))
                    assert res is not False
                    assign.extend(zip((vp, gp), res[0][:2]))
                    assign.extend(zip((vd, gd), res[1][:2]))
                else:
                    _, dil, pwc, gy, gh, bm = item
                    (vd, gd), (vp, gp) = _params(dil.m), _params(pwc.m)
                    res = False
                    if USE_FUSED_RU_DW and gd is not None and gp is not None and dil.m.bias is None and pwc.m.bias is None and bm in (ops.MATH_BF16, ops.MATH_BF16X6):
                        res = ops.weight_grads_ru(bm, dil.spec.dilation, gy, pwc.y, float(pwc.spec.out_slope), pwc.x, gh, dil.x, float(dil.spec.in_slope),
                                                  (vp, gp, pwc.norm), (vd, gd, dil.norm))
                    if res is False:
                        for rec, dy in ((pwc, gy), (dil, gh)):
                            v, g = _params(rec.m)
                            grads = ops.weight_grads(rec.d, dy, rec.y, rec.x, v, g, rec.m.bias, rec.norm)
                            assign.extend((p, t) for p, t in zip((v, g, rec.m.bias), grads) if p is not None and t is not None)
                    else:
                        assign.extend(zip((vp, gp), res[0][:2]))
                        assign.extend(zip((vd, gd), res[1][:2]))
        ops.wn_bwd_multi(jobs)
        return assign

    @torch.no_grad()
    def backward_train(self, saved, g_pre: torch.Tensor, serial: Optional[int] = None) -> None:
        """``backward(saved, g_pre)``; behind a replayed forward and inside ``ops.weight_grads_on_side_stream()``: per segment (decoder
        block / latent convs / encoder block) the input-gradient launches as one graph on the current stream and the segment's
        weight-gradient launches as one graph on the side stream behind it -- the weight gradients of segment k run beside the
        input-gradient chain of segment k + 1, as the launch-by-launch schedule has them.  Results are assigned by the context's
        ``join()``.  ([MI355X] as two graphs -- the whole chain, then all weight gradients -- the step went 12.0 -> 12.5 ms: the chain
        alone fills a fraction of the GPU.)"""
        from_graph = self._fwd_graph.out is not None and saved is self._fwd_graph.out[2]
        if from_graph and serial is not None and serial != getattr(self, "_graph_serial", None):
            raise RuntimeError("EBEN generator engine: this forward's saved activations were rewritten by a later replayed forward (backward "
                               "through an old graph after a new training forward); set EBEN_GEN_GRAPHS=0 for that pattern")
        self._outstanding = None   # differentiated: the next forward may replay over these buffers
        if not (self._graphs_usable() and getattr(self, "_fwd_replayed", False) and from_graph and self._deferrable()):
            return self.backward(saved, g_pre)
        gbuf = self._static_buf("g", g_pre)
        gbuf.copy_(g_pre)
        n = self._n_segments(saved)
        # segments per graph: a replay costs ~70 us of host time, a graph per segment (14 replays) 1.35 ms per step against 0.74 for two
        # graphs; three groups keep the weight gradients beside the chain at 0.5 ms
        sizes = [int(t) for t in BWD_GROUPS.split(",")] if BWD_GROUPS else []
        if sum(sizes) != n:
            sizes = [1] * n
        bounds = [sum(sizes[:i]) for i in range(len(sizes) + 1)]
        while len(self._dx_graphs) < len(sizes):
            self._dx_graphs.append(ops.ReplayedChain())
            self._dw_graphs.append(ops.ReplayedChain())
        dev = gbuf.device
        main = torch.cuda.current_stream(dev)
        side = ops._side_stream(dev)
        g, skip_grads = gbuf, []
        for k in range(len(sizes)):
            def dx_body(k=k, g=g):
                self._dwq = []
                sk = list(skip_grads)
                try:
                    out = g
                    for seg in range(bounds[k], bounds[k + 1]):
                        out = self._bwd_segment(saved, seg, out, sk)
                finally:
                    queue, self._dwq = self._dwq, None
                return out, sk[len(skip_grads):], queue

            # one signature for all segments: they settle, and are captured, in the same step, each on the results of the one before
            g, new_skips, queue = self._dx_graphs[k].run((self._fwd_graph.captures, gbuf.data_ptr(), ops._backward_math[0]), dx_body, None)
            skip_grads.extend(new_skips)
            if not queue:
                continue
            side.wait_stream(main)
            sink = ops._side["sink"]
            with torch.cuda.stream(side):
                if self._dx_graphs[k].graph is not None:
                    assign = self._dw_graphs[k].run((self._dx_graphs[k].captures, id(sink)), lambda: self._dw_body(queue, sink), side)
                else:
                    assign = self._dw_body(queue, sink)
                    ops._side["keep"].append(queue)   # eager tensors: referenced until join()
            if sink is None:
                ops._side["assign"].extend((p, t) for p, t in assign if p.requires_grad)
            else:
                # data-parallel: this group's gradients sit in their bucket views once the side stream has run the launches above -- report
                # them from THAT stream now (GradSync.mark_ready records its event on the current stream), so that a bucket filled by
                # an early group is exchanged underneath the later groups' input-gradient chain instead of behind join()
                if DEFER_MARK_READY:
                    ops._side["sunk"].extend(p for p, _ in assign)   # reported by join(), from the main stream, once it has waited for the side stream
                else:
                    with torch.cuda.stream(side):
                        sink.mark_ready([p for p, _ in assign])


class _CoreFn(torch.autograd.Function):
    """The engine behind autograd: outputs (input of last_conv, first_bands); the parameters are arguments only so that the
    graph knows the output depends on them -- their gradients are produced by the engine, not returned."""

    @staticmethod
    def forward(ctx, x, engine, *params):
        pre, first_bands, saved = engine.forward_train(x)
        ctx.engine, ctx.saved = engine, saved
        engine._note_ctx(ctx)
        # fresh tensor objects: behind a replayed forward `pre` is the SAME object every step, and autograd writes this call's history
        # into what it is handed
        pre, first_bands = pre.detach(), first_bands.detach()
        ctx.mark_non_differentiable(first_bands)
        return pre, first_bands

    @staticmethod
    def backward(ctx, g_pre, _g_fb):
        ctx.engine.backward_train(ctx.saved, g_pre, getattr(ctx, "serial", None))
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


def core(gen, cut_audio: torch.Tensor):
    """(input of ``gen.last_conv``, first_bands) through the engine; differentiable w.r.t. the generator's parameters."""
    engine = getattr(gen, "_engine", None)
    if engine is None:
        engine = GeneratorEngine(gen)
        object.__setattr__(gen, "_engine", engine)   # not a submodule / buffer: invisible to state_dict
    # the parameters the core owns: NOT last_conv's -- the balancing passes differentiate the losses w.r.t. last_conv.weight alone
    # (eben.py:223-227) and must not reach into the core
    params = [p for m in (gen.first_conv, gen.encoder_blocks, gen.latent_conv, gen.decoder_blocks) for p in ops.parameters_of(m) if p.requires_grad]
    if torch.is_grad_enabled() and params:
        return _CoreFn.apply(cut_audio.contiguous(), engine, *params)
    pre, first_bands, _ = engine.forward(cut_audio.contiguous(), False)
    return pre, first_bands
