"""Batched forward / backward of ``DiscriminatorEBENMultiScales`` for the fused EBEN train step.

One GAN step (``vibravox/lightning_modules/eben.py:82-130,184-240``) sends FOUR gradient signals
through the discriminators, all linear in their seed and all sharing the activations of two
forward passes:

  * feature matching  (``feature_loss.py:37-50``)  -- through the *enhanced* branch, enters at every layer;
  * adversarial, generator side (``hinge_loss.py:35-43``, target +1) -- enhanced branch, enters at the logits;
  * ``fake_loss`` (target -1)                       -- enhanced branch, logits, wanted at the weights;
  * ``real_loss`` (target +1)                       -- *reference* branch, logits, wanted at the weights.

Autograd runs them as four passes of ~80 kernel launches each.  This engine runs the two forwards as
ONE batch-2B pass ([enhanced; reference]) and the four backwards as ONE pass over gradients stacked
along the batch axis (rows [fm | adv | fake | real], 4B rows):

  * input gradients: ``eben_conv1d_bwd_dx_ex`` -- the producer's epilogue adds the feature-matching
    gradient of that layer (first B rows only) and applies the LeakyReLU derivative of the saved
    activation (row b reads activation row map(b)), so no kernel on the path reads a mask on load
    and no separate add kernel runs;
  * weight gradients: rows [fake | real] against activations [enhanced | reference] are exactly a
    batch-2B ``eben_conv1d_bwd_dw`` (K doubles instead of a second launch + sum).

Same arithmetic per sample as the four separate passes (convolutions are per-sample); only the fp32
summation order inside the weight gradients changes.  The four sub-discriminators run on their own
HIP streams.  No autograd graph is built for the discriminator; parameter gradients become ``.grad`` directly, or
are handed to autograd through ``inject_grads`` where ``.grad`` accumulation hooks (torch DDP) must fire.
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from ._lib import check, load, ptr


from ._lib import stream as _stream  # noqa: E402  (raw handle of the current HIP stream)


#: input gradients of the strided MelGAN layers as "phases as rows" (one stride-1 contraction for all output phases, csrc/tapconv.hip)
USE_DX_PR = os.environ.get("EBEN_DX_PR", "1") != "0"


class _Layer:
    """One weight-normalised conv of a sub-discriminator with the packed copies the engine needs."""

    MAX_PACKS = 6   # forward at 2B rows + input gradient at 4B / 2B rows, for two (batch, length) shapes

    def __init__(self, conv, math=ops.MATH_F32):
        """math: one EBEN_MATH_* for every contraction of the layer, or (forward, input gradient, weight gradient)."""
        self.conv = conv
        self.math_fwd, self.math_dx, self.math_dw = (math, math, math) if isinstance(math, int) else tuple(math)
        self.spec: ops.ConvSpec = conv.spec
        self.spec_lin = dataclasses.replace(conv.spec, in_slope=1.0, out_slope=1.0)   # gradients arrive pre-masked
        self.packs: Dict[Tuple[int, int, int], Tuple[tuple, torch.Tensor]] = {}
        self.used = set()   # slots touched since the last prepack(): the only ones it rebuilds
        self.scale_key = None
        self.scale = self.norm = None
        self.reuse = False   # inside prepack(): rebuild into the buffers already held (ops._buffer)
        self.keep_scale = False   # the layer's kernels take (v, scale) directly (bundle-layout heads / tails): prepack refreshes the scale
        self.pr_weights: Dict[tuple, torch.Tensor] = {}   # slot -> primed weights of the phases-as-rows input gradient
        self._pr_descs: Dict[tuple, object] = {}
        self._params = None

    def ensure_scale(self) -> None:
        """Weight-norm scale g / ||v|| and norm ||v|| of the current weights (what ``packed`` computes on the way)."""
        v, g, _ = self.params()
        v, g = v.detach(), g.detach()
        wkey = self._weights_key()
        if self.scale_key != wkey:
            rows = v.shape[0]
            self.scale = ops._buffer(self.scale, rows, v, self.reuse)
            self.norm = ops._buffer(self.norm, rows, v, self.reuse)
            check(load().eben_wn_scale(ptr(g), ptr(v), rows, v.numel() // rows, ptr(self.scale), ptr(self.norm), _stream()), "wn_scale")
            self.scale_key = wkey

    def params(self):
        """(v, g, bias).  Looked up ~440 times a step: the Parameter objects are kept until ``ops.bump_weights_epoch()`` (no arguments)
        announces replaced parameters (three module ``__getattr__`` walks per call otherwise)."""
        epoch = ops._storage_epoch.get(-1, 0)
        hit = self._params
        if hit is None or hit[0] != epoch:
            prm = self.conv.parametrizations["weight"]
            hit = self._params = (epoch, (prm.original1, prm.original0, self.conv.bias))
        return hit[1]

    def _weights_key(self):
        v, g, _ = self.params()
        e = ops._storage_epoch
        return (v.data_ptr(), v._version, e.get(v.data_ptr(), 0), g.data_ptr(), g._version, e.get(g.data_ptr(), 0), e.get(-1, 0))

    def packed(self, which: int, batch: int, l_in: int) -> torch.Tensor:
        """MFMA / direct layout of the weights for the forward (0) or input-gradient (1) launch of this
        exact descriptor (the first-generation layout depends on the batch size)."""
        lib = load()
        v, g, _ = self.params()
        v, g = v.detach(), g.detach()
        wkey = self._weights_key()
        if self.scale_key != wkey:
            rows = v.shape[0]
            self.scale = ops._buffer(self.scale, rows, v, self.reuse)
            self.norm = ops._buffer(self.norm, rows, v, self.reuse)
            check(lib.eben_wn_scale(ptr(g), ptr(v), rows, v.numel() // rows, ptr(self.scale), ptr(self.norm), _stream()), "wn_scale")
            self.scale_key = wkey
        slot = (which, batch, l_in)
        self.used.add(slot)
        hit = self.packs.get(slot)
        if hit is not None and hit[0] == wkey:
            return hit[1]
        if hit is None and len(self.packs) >= self.MAX_PACKS:
            # variable clip lengths / a short last batch: every new (batch, length) would otherwise keep another packed copy of
            # the weights alive and be re-packed after every optimiser step -- drop the slots the current step has not touched
            for old in [k for k in self.packs if k not in self.used] or list(self.packs)[:1]:
                del self.packs[old]
        d = ops.conv_desc(self.spec, batch, l_in, self.math_fwd if which == 0 else self.math_dx)
        if which == 2:
            # "phases as rows" form of the input gradient (include/eben_hip.h, eben_bl_dx_pr_*): the primed stride-1 layer's weights are a
            # gather of this layer's (scale folded in), packed as that layer's forward image
            dq = self.pr_desc(batch, l_in)
            wq = self.pr_weights.get(slot)
            n_w = dq.c_out * (dq.c_in // dq.groups) * dq.ksize
            if wq is None or wq.numel() != n_w:
                wq = self.pr_weights[slot] = torch.empty(n_w, dtype=torch.float32, device=v.device)
            check(lib.eben_bl_dx_pr_weights(ctypes.byref(d), ptr(v), ptr(self.scale), ptr(wq), _stream()), "bl_dx_pr_weights")
            wp = ops._buffer(None if hit is None else hit[1], lib.eben_conv1d_packed_floats(ctypes.byref(dq), 0), v, self.reuse)
            ops.conv1d_pack(dq, wq, None, wp, None)
            self.packs[slot] = (wkey, wp)
            return wp
        wp = ops._buffer(None if hit is None else hit[1], lib.eben_conv1d_packed_floats(ctypes.byref(d), which), v, self.reuse)
        ops.conv1d_pack(d, v, self.scale, wp if which == 0 else None, wp if which == 1 else None)
        self.packs[slot] = (wkey, wp)
        return wp

    def pr_desc(self, batch: int, l_in: int):
        """Descriptor of the primed layer of ``eben_bl_dx_pr_desc`` for this layer's input gradient at (batch, l_in), or None when the
        input gradient has no such form (cached per shape)."""
        key = (batch, l_in)
        if key not in self._pr_descs:
            from ._lib import EbenConv1dDesc
            d = ops.conv_desc(self.spec, batch, l_in, self.math_dx)
            dq = EbenConv1dDesc()
            ok = USE_DX_PR and (self.math_dx & 0x100) != 0 and load().eben_bl_dx_pr_desc(ctypes.byref(d), ctypes.byref(dq)) == 0
            if len(self._pr_descs) > 16:
                self._pr_descs.clear()
            self._pr_descs[key] = dq if ok else None
        return self._pr_descs[key]


class _Chain:
    """A sub-discriminator: ReflectionPad1d(pad) followed by the conv stack (last conv = logits)."""

    def __init__(self, modules, math=ops.MATH_F32):
        """math: EBEN_MATH_* or (forward, input gradient, weight gradient), or a callable layer index -> one of those."""
        self.layers: List[_Layer] = []
        self.pad = 0
        self._fm_ok: Dict[tuple, bool] = {}
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
        for i, conv in enumerate(convs):
            self.layers.append(_Layer(conv, math(i, len(convs)) if callable(math) else math))

    # ---- forward on a (2B, C, L) batch: returns [input, out_0, ..., logits] and the padded input
    def forward(self, x: torch.Tensor):
        bufs = self.forward_rows(x.contiguous(), None, 0, x.shape[0])
        return bufs["emb"], bufs["xp"]

    def forward_rows(self, x_full: torch.Tensor, bufs: Optional[dict], r0: int, r1: int) -> dict:
        """Runs the chain on batch rows [r0, r1) of ``x_full`` and writes rows [r0, r1) of full-batch activation tensors
        (allocated on the first call, passed back as ``bufs`` for the remaining rows): the reference half of the batch
        does not depend on the generator, so the engine runs it underneath the generator forward."""
        lib = load()
        nb, c, l = x_full.shape
        b = r1 - r0
        if bufs is None:
            xp_full = torch.empty((nb, c, l + 2 * self.pad), dtype=torch.float32, device=x_full.device) if self.pad else x_full
            bufs = {"xp": xp_full, "outs": [None] * len(self.layers)}
        xp_full = bufs["xp"]
        if self.pad:
            check(lib.eben_reflect_pad_fwd(ptr(x_full[r0:r1]), ptr(xp_full[r0:r1]), b * c, l, self.pad, self.pad, _stream()), "reflect_pad_fwd")
        cur = xp_full[r0:r1]
        for i, lay in enumerate(self.layers):
            d = ops.conv_desc(lay.spec, b, cur.shape[2], lay.math_fwd)
            if bufs["outs"][i] is None:
                bufs["outs"][i] = torch.empty((nb, lay.spec.c_out, d.l_out), dtype=torch.float32, device=x_full.device)
            y = bufs["outs"][i][r0:r1]
            _, _, bias = lay.params()
            wp = lay.packed(0, b, cur.shape[2])
            tm = ops.kernel_timer_for(lay.spec, "fwd")   # bench.py: HIP events around the roofline kernel, on the stream it is launched on
            e0 = tm.start() if tm is not None else None
            check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(cur), ptr(wp), ptr(bias), None, ptr(y), _stream()), "conv1d_fwd")
            if tm is not None:
                tm.stop(e0, b)
            cur = y
        bufs["emb"] = [x_full] + bufs["outs"]
        return bufs

    # ---- backward of the four stacked right-hand sides
    def fm_epilogue_ok(self, emb: List[torch.Tensor], xp: torch.Tensor, half: int) -> bool:
        """Whether every input-gradient launch of this chain runs on a kernel that can form the feature-matching gradient of the
        embedding it produces the gradient for in its epilogue (eben_conv1d_bwd_dx_fm: the thin and the bf16 tap-conv kernels)."""
        outs = emb[1:]
        key = (half,) + tuple(int(o.shape[2]) for o in outs[:-1])
        hit = self._fm_ok.get(key)
        if hit is None:
            lib = load()
            hit = True
            for i in range(1, len(self.layers)):
                lay = self.layers[i]
                d = ops.conv_desc(lay.spec_lin, 4 * half, outs[i - 1].shape[2], lay.math_dx)
                if self.layers[i - 1].spec.out_slope == 1.0 or lib.eben_conv1d_kernel_generation(ctypes.byref(d), 1) not in (3, 4):
                    hit = False
            if len(self._fm_ok) > 64:
                self._fm_ok.clear()
            self._fm_ok[key] = hit
        return hit

    def backward(self, emb: List[torch.Tensor], xp: torch.Tensor, fm_grads: Optional[List[Optional[torch.Tensor]]], seeds: torch.Tensor,
                 half: int, want_param_grads: bool, fm_fused: Optional[Tuple[int, float]] = None):
        """emb = [input, out_0..out_{L-1}] with 2*half rows; fm_grads[i] = d(fm)/d(out_i) (half rows) or None -- or, with
        ``fm_fused`` = (device address of this chain's (s1, s2) sums, d loss / d term), no gradient buffers at all: the
        input-gradient launches form the feature-matching gradient from the two halves of the embedding in their epilogue;
        seeds = (4*half, 1, L_logits) rows [fm | adv | fake | real].  Returns (d_input (2*half rows: fm | adv),
        [(dv, dg, dbias) per layer] or None)."""
        lib = load()
        st = _stream()
        outs = emb[1:]
        n = len(self.layers)
        seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
        jobs = []   # (layer index, stacked gradient at the layer output, layer input): weight gradients, run later
        g = seeds.contiguous()
        for i in range(n - 1, -1, -1):
            lay = self.layers[i]
            x_in = outs[i - 1] if i > 0 else xp
            l_in = x_in.shape[2]
            if want_param_grads:
                jobs.append((i, g, x_in))
            if i > 0:
                rows = 4 * half
                d = ops.conv_desc(lay.spec_lin, rows, l_in, lay.math_dx)
                gp = torch.empty((rows, lay.spec.c_in, l_in), dtype=torch.float32, device=g.device)
                prev_slope = self.layers[i - 1].spec.out_slope
                wp = lay.packed(1, rows, l_in)
                tm = ops.kernel_timer_for(lay.spec, "dx")
                e0 = tm.start() if tm is not None else None
                if fm_fused is not None:
                    check(lib.eben_conv1d_bwd_dx_fm(ctypes.byref(d), ptr(g), ptr(wp), ptr(outs[i - 1][half:]), half,
                                                    fm_fused[0] + 8 * (i - 1), fm_fused[1], ptr(outs[i - 1]), prev_slope, half, seg_map,
                                                    ptr(gp), st), "conv1d_bwd_dx_fm")
                else:
                    res = fm_grads[i - 1]
                    check(lib.eben_conv1d_bwd_dx_ex(ctypes.byref(d), ptr(g), ptr(wp), ptr(res), half if res is not None else 0,
                                                    ptr(outs[i - 1]) if prev_slope != 1.0 else None, prev_slope, half, seg_map, ptr(gp), st),
                          "conv1d_bwd_dx_ex")
                if tm is not None:
                    tm.stop(e0, rows)
                g = gp
            else:
                rows = 2 * half   # only the generator-side signals reach the discriminator input
                d = ops.conv_desc(lay.spec_lin, rows, l_in, lay.math_dx)
                gp = torch.empty((rows, lay.spec.c_in, l_in), dtype=torch.float32, device=g.device)
                check(lib.eben_conv1d_bwd_dx_ex(ctypes.byref(d), ptr(g), ptr(lay.packed(1, rows, l_in)), None, 0, None, 1.0, 0, None, ptr(gp), st),
                      "conv1d_bwd_dx_ex")
                g = gp
        if self.pad:
            b2, c, lp = g.shape
            dx = torch.empty((b2, c, lp - 2 * self.pad), dtype=torch.float32, device=g.device)
            check(lib.eben_reflect_pad_bwd(ptr(g), ptr(dx), b2 * c, lp - 2 * self.pad, self.pad, self.pad, st), "reflect_pad_bwd")
            g = dx
        return g, (jobs if want_param_grads else None)

    def weight_grads(self, jobs, half: int, sink=None):
        """Second half of the backward: the weight gradients of every layer from the stacked gradients kept by
        ``backward`` (rows [fake | real] against the layer inputs [enhanced | reference]).  Nothing on the
        generator side depends on them, so the engine launches them after the input-gradient chain and lets
        them run underneath the generator backward."""
        st = _stream()
        n = len(self.layers)
        grads = [None] * n
        wn_jobs = []   # slab sums + weight-norm chain rule of the whole chain: one multi-tensor launch at the end
        logits = None
        for i, g, x_in in jobs:
            lay = self.layers[i]
            if i == n - 1:
                # Logits layer: fake and real branches separately, then added -- the reference's structure
                # (two autograd graphs accumulating into one .grad).  While every hinge term is active the
                # two bias gradients are -c*N and +c*N summed in the SAME order, i.e. they cancel exactly and
                # Adam leaves the bias alone; one sum over both branches leaves a rounding residue that Adam
                # (m / sqrt(v)) turns into a full-size step.
                gf = self._weight_grads(lay, g[2 * half:3 * half], x_in[:half], st, wn_jobs)
                gr = self._weight_grads(lay, g[3 * half:], x_in[half:], st, wn_jobs)
                logits = (i, gf, gr)
            else:
                grads[i] = self._weight_grads(lay, g[2 * half:], x_in, st, wn_jobs, sink)
        ops.wn_bwd_multi(wn_jobs)
        if logits is not None:
            i, gf, gr = logits
            outs = [None if sink is None or p is None else sink.grad_buffer(p) for p in self.layers[i].params()]
            grads[i] = tuple(None if a is None else (a + b if o is None else torch.add(a, b, out=o)) for a, b, o in zip(gf, gr, outs))
        return grads

    @staticmethod
    def _weight_grads(lay: _Layer, g2: torch.Tensor, x_in: torch.Tensor, st: int, wn_jobs: list, sink=None):
        """sink (ddp.GradSync): the gradients are written straight into its bucket views."""
        lib = load()
        v, gain, bias = lay.params()
        rows_b = g2.shape[0]
        d = ops.conv_desc(lay.spec_lin, rows_b, x_in.shape[2], lay.math_dw)
        nslab, row_stride = ctypes.c_int(0), ctypes.c_int(0)
        ws_bytes = lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride))
        slabs = torch.empty(max(1, (ws_bytes + 3) // 4), dtype=torch.float32, device=g2.device)
        check(lib.eben_conv1d_bwd_dw(ctypes.byref(d), ptr(g2), None, ptr(x_in), 1 if bias is not None else 0, ptr(slabs), ws_bytes, st),
              "conv1d_bwd_dw")
        wrows = v.shape[0]
        cols = v.numel() // wrows
        dv = dg = dbias = None
        if sink is not None:
            dv, dg, dbias = sink.grad_buffer(v), sink.grad_buffer(gain), (sink.grad_buffer(bias) if bias is not None else None)
        dv = torch.empty_like(v) if dv is None else dv
        dg = torch.empty_like(gain) if dg is None else dg
        if bias is not None and dbias is None:
            dbias = torch.empty(wrows, dtype=torch.float32, device=g2.device)
        wn_jobs.append((slabs, nslab.value, wrows * row_stride.value, wrows, cols, row_stride.value, gain.detach(), v.detach(), lay.norm,
                        dg, dv, dbias))
        return dv, dg, dbias


_DIRECT_INJECT = os.environ.get("EBEN_DIRECT_INJECT", "1") != "0"


class _InjectGrads(torch.autograd.Function):
    """Hands externally computed parameter gradients to autograd (so accumulation hooks fire)."""

    @staticmethod
    def forward(ctx, grads, *params):
        ctx.grads = grads
        return params[0].new_zeros(())

    @staticmethod
    def backward(ctx, gout):
        return (None, *ctx.grads)


def inject_grads(params: Sequence[torch.nn.Parameter], grads: Sequence[torch.Tensor]) -> None:
    live = [(p, g) for p, g in zip(params, grads) if p.requires_grad and g is not None]
    if not live:
        return
    ps, gs = zip(*live)
    # Through autograd every gradient is CLONED into ``.grad`` (the list above keeps a second reference, so AccumulateGrad cannot
    # steal it): 93 device copies in front of the discriminator's Adam.  The gradients are fresh tensors of the parameters'
    # own layout, so without accumulation hooks to fire (torch DDP's reducer, ``register_post_accumulate_grad_hook`` users) they
    # become ``.grad`` directly.
    if _DIRECT_INJECT and not (torch.distributed.is_available() and torch.distributed.is_initialized()) and all(
            not getattr(p, "_post_accumulate_grad_hooks", None) and not p._backward_hooks for p in ps):
        for p, g in live:
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        return
    _InjectGrads.apply(list(gs), *ps).backward()


class DiscriminatorEngine:
    def __new__(cls, disc, math=ops.MATH_F32):
        if cls is DiscriminatorEngine and isinstance(math, dict) and math.get("layout") == "bl":
            from .disc_engine_bl import DiscriminatorEngineBL   # embeddings / gradients at rest as bf16 bundles

            why = DiscriminatorEngineBL.unsupported(disc)
            if why is None:
                return super().__new__(DiscriminatorEngineBL)
            # e.g. DiscriminatorEBENMultiScales at the reference's class default q = 3 (heads 3 -> 24): the chain-edge kernels are built for
            # the configured shapes (csrc/bl_edge.hip bl_head_shape).  The same arithmetic plan on fp32 tensors at rest covers every shape.
            import warnings

            warnings.warn(f"bundle-layout discriminator engine: {why}; running the same plan on fp32 tensors at rest ('bf16')")
        return super().__new__(cls)

    def __init__(self, disc, math=ops.MATH_F32):
        """math: what the contractions of the layers tapconv3.hip / conv_dw3.hip cover compute in (fp32 accumulation, storage
        and element-wise stages either way) -- one EBEN_MATH_* for everything, a (forward, input gradient, weight gradient)
        triple, a callable (layer index, layers in the chain) -> one of those, or a dict {"pqmf": ..., "melgan": ...} of
        those per sub-discriminator family."""
        self.disc = disc
        self.q = disc.q
        self.math = math
        per = math if isinstance(math, dict) else {"pqmf": math, "melgan": math}
        self.chains = [_Chain(d.discriminator, per["pqmf"]) for d in disc.pqmf_discriminators] + [
            _Chain(disc.melgan_discriminator.discriminator, per["melgan"])]
        self._streams = None
        self._state = None
        self._prepack_graph = ops.ReplayedPrepack()
        #: weights of the (adversarial, fake, real) hinge seeds of the stacked backward: (1, 1, 1) is the train step; (1, 1, 0) /
        #: (1, 0, 1) give the discriminator gradient of fake_loss / real_loss alone (the parity tests bound each branch
        #: separately -- their sum cancels to a fraction of a percent at initialisation)
        self.seed_weights = (1.0, 1.0, 1.0)

    @staticmethod
    def supports(disc) -> bool:
        return hasattr(disc, "pqmf_discriminators") and hasattr(disc, "melgan_discriminator") and all(
            hasattr(m, "discriminator") for m in list(disc.pqmf_discriminators) + [disc.melgan_discriminator])

    #: forward pass only: the last PQMF-band chain on the side stream (idle between the weight pre-packing and the generator's
    #: weight gradients) instead of behind the other two -- the PQMF chains' exact-fp32 forward ("bf16" plan) is the longest
    #: stream of that phase otherwise
    spread_forward = os.environ.get("EBEN_D_FWD_SPREAD", "1") != "0"
    #: feature-matching gradient inside the input-gradient epilogues (eben_conv1d_bwd_dx_fm) instead of a kernel per chain that
    #: writes it into buffers the epilogues then read
    fm_in_epilogue = os.environ.get("EBEN_FM_EPILOGUE", "1") != "0"
    spread_backward = os.environ.get("EBEN_D_BWD_SPREAD", "1") != "0"   # [MI355X] 19.6 -> 19.35 ms/step (the input-gradient phase shortens by 0.4 ms, the generator backward, which then shares the GPU with more weight-gradient work, lengthens by 0.15)

    def _launch_on_streams(self, fn, forward: bool = False, order=None):
        """fn(i) for each sub-discriminator on its own HIP stream, the longest chain (MelGAN, last) first so that it
        is never queued behind a short one (or in ``order``); results in chain order.  The caller joins with ``_join_streams``."""
        main = torch.cuda.current_stream()
        dev = main.device
        if self._streams is None or self._streams[0].device != dev:
            # MelGAN chain on one stream, the three (shorter, equal) PQMF-band chains in series on another: with the main
            # and the side stream that is four streams = four hardware queues, none shared (ops.aux_stream)
            pq, mel = ops.aux_stream(1, dev), ops.aux_stream(0, dev)
            self._streams = [pq] * (len(self.chains) - 1) + [mel]
        streams = list(self._streams)
        if (self.spread_forward if forward else self.spread_backward) and len(self.chains) >= 3:
            streams[len(self.chains) - 2] = ops.aux_stream(2, dev)
        self._used_streams = set(streams) | set(self._streams)
        results = [None] * len(self.chains)
        n = len(self.chains)
        ev = getattr(self, "_prepack_ev", None)
        for i in (order if order is not None else [n - 1] + list(range(n - 1))):
            st = streams[i]
            st.wait_stream(main)
            if ev is not None:
                st.wait_event(ev)   # weight images rebuilt ahead of time on the side stream (prepack)
            with torch.cuda.stream(st):
                results[i] = fn(i)
        return results

    def _join_streams(self):
        main = torch.cuda.current_stream()
        for st in getattr(self, "_used_streams", None) or set(self._streams):
            main.wait_stream(st)

    def _on_streams(self, fn):
        results = self._launch_on_streams(fn)
        self._join_streams()
        return results

    # ---- the two forwards: one batch-2B pass, or the reference half ahead of time ----------------------
    def _inputs(self, half, bands_like, audio_like):
        dev = audio_like.device
        sub = torch.empty((2 * half, self.q) + tuple(bands_like.shape[2:]), dtype=torch.float32, device=dev)
        wav = torch.empty((2 * half,) + tuple(audio_like.shape[1:]), dtype=torch.float32, device=dev)
        return sub, wav

    @torch.no_grad()
    def forward_reference(self, bands_ref: torch.Tensor, audio_ref: torch.Tensor):
        """Launches the discriminators on the REFERENCE half of the batch (rows [B, 2B) of every activation tensor) and
        returns at once: nothing in it depends on the generator, so the caller runs the generator forward on the main
        stream meanwhile; ``forward`` then only has the enhanced half left."""
        half = audio_ref.shape[0]
        sub, wav = self._inputs(half, bands_ref, audio_ref)
        sub[half:].copy_(bands_ref[:, -self.q:, :])
        wav[half:].copy_(audio_ref)
        inputs = [sub] * (len(self.chains) - 1) + [wav]
        bufs = self._launch_on_streams(lambda i: self.chains[i].forward_rows(inputs[i], None, half, 2 * half), forward=True)
        self._partial = dict(half=half, sub=sub, wav=wav, bufs=bufs)

    @torch.no_grad()
    def forward(self, bands: torch.Tensor, audio: torch.Tensor, bands_ref: torch.Tensor, audio_ref: torch.Tensor, join: bool = True):
        """join=False leaves the chains running (the caller overlaps independent work on the main stream and calls
        ``join()`` before anything reads the embeddings).  After ``forward_reference`` only the enhanced half is run."""
        half = bands.shape[0]
        part = getattr(self, "_partial", None)
        self._partial = None
        if part is not None and part["half"] == half:
            sub, wav = part["sub"], part["wav"]
            sub[:half].copy_(bands[:, -self.q:, :])
            wav[:half].copy_(audio)
            inputs = [sub] * (len(self.chains) - 1) + [wav]
            res = self._launch_on_streams(lambda i: self.chains[i].forward_rows(inputs[i], part["bufs"][i], 0, half), forward=True)
        else:
            sub = torch.cat((bands[:, -self.q:, :], bands_ref[:, -self.q:, :]), dim=0).contiguous()
            wav = torch.cat((audio, audio_ref), dim=0).contiguous()
            inputs = [sub] * (len(self.chains) - 1) + [wav]
            res = self._launch_on_streams(lambda i: self.chains[i].forward_rows(inputs[i], None, 0, 2 * half), forward=True)
        if join:
            self._join_streams()
        self._state = dict(half=half, emb=[r["emb"] for r in res], xp=[r["xp"] for r in res], bands_shape=tuple(bands.shape))
        return self._state["emb"]

    def join(self):
        self._join_streams()

    def prepack(self):
        """Rebuilds every packed weight image the last step used (forward at 2B rows, input gradients at 4B / 2B rows)
        on the side stream, right after the discriminator's optimiser step: the ~100 small launches then run under the
        next step's generator forward instead of at the head of the four chains."""
        if not self.chains or not any(lay.packs or (lay.keep_scale and lay.scale is not None) for ch in self.chains for lay in ch.layers):
            return
        dev = self.chains[0].layers[0].params()[0].device
        main = torch.cuda.current_stream(dev)
        side = ops._side_stream(dev)
        side.wait_stream(main)
        layers = [lay for ch in self.chains for lay in ch.layers]

        def body():
            jobs = []
            for lay in layers:   # weight-norm scales of all layers: one multi-tensor launch
                wkey = lay._weights_key()
                if (lay.packs or (lay.keep_scale and lay.scale is not None)) and lay.scale_key != wkey:
                    v, g, _ = lay.params()
                    rows = v.shape[0]
                    lay.scale = ops._buffer(lay.scale, rows, v, True)
                    lay.norm = ops._buffer(lay.norm, rows, v, True)
                    lay.scale_key = wkey
                    jobs.append((g.detach(), v.detach(), rows, v.numel() // rows, lay.scale, lay.norm))
            ops.wn_scale_multi(jobs)
            with ops.pack_batch():
                for lay in layers:
                    used, lay.used = lay.used, set()
                    lay.reuse = True
                    try:
                        for slot in list(lay.packs):
                            if slot in used:
                                lay.packed(*slot)     # what the last step launched: the next step most likely launches it again
                            else:
                                del lay.packs[slot]   # a shape of an earlier step: rebuilt on demand if it comes back
                    finally:
                        lay.reuse = False
                    lay.used = set()              # re-packing is not a use: the next step decides what survives the next prepack

        # the launch sequence as a function of everything but the weights' values (ops.ReplayedPrepack): layers, the slots the last
        # step used (= all the slots held, or the eager path prunes), parameter storage, the image / scale buffers written, and that
        # every image is stale
        sig = tuple((id(lay), tuple(sorted(lay.packs)), tuple(sorted(lay.used)) == tuple(sorted(lay.packs)), lay.params()[0].data_ptr(),
                     tuple(lay.packs[k][1].data_ptr() for k in sorted(lay.packs)), 0 if lay.scale is None else lay.scale.data_ptr(),
                     0 if lay.norm is None else lay.norm.data_ptr(),
                     lay.scale_key != lay._weights_key()) for lay in layers) + (ops._storage_epoch.get(-1, 0),)
        with torch.cuda.stream(side), torch.no_grad():
            if self._prepack_graph.run(sig, body, side):
                for lay in layers:   # replayed: images and scales are current, the cache keys are not
                    wkey = lay._weights_key()
                    lay.scale_key = wkey
                    for slot, (_, wp) in list(lay.packs.items()):
                        lay.packs[slot] = (wkey, wp)
                    lay.used = set()
            self._prepack_ev = torch.cuda.Event()
            self._prepack_ev.record()

    # ---- the four scalar losses (device tensors) ------------------------------------------------
    @torch.no_grad()
    def losses(self) -> Dict[str, torch.Tensor]:
        lib = load()
        s = self._state
        half, emb = s["half"], s["emb"]
        dev = emb[0][0].device
        a = [t[:half] for scale in emb for t in scale[1:-1]]
        b = [t[half:] for scale in emb for t in scale[1:-1]]
        n = len(a)
        inter = [None] * (2 * n)
        inter[0::2], inter[1::2] = a, b
        ptrs = (ctypes.c_void_p * (2 * n))(*[ptr(t) for t in inter])
        numel = (ctypes.c_int64 * n)(*[t.numel() for t in a])
        ws_bytes = lib.eben_fm_sums_workspace(n)
        ws = torch.empty(max(1, (ws_bytes + 3) // 4), dtype=torch.float32, device=dev)
        sums = torch.empty(2 * n, dtype=torch.float32, device=dev)
        check(lib.eben_fm_sums(ptrs, numel, n, ptr(ws), ws_bytes, ptr(sums), _stream()), "fm_sums")
        inv = 1.0 / (len(emb) * len(emb[-1][1:-1]))
        s.update(fm_a=a, fm_b=b, fm_sums=sums, fm_inv=inv, fm_numel=numel, fm_ptrs=ptrs)
        hinge = torch.empty(3 * len(emb), dtype=torch.float32, device=dev)
        terms = [(rows, target) for scale in emb for rows, target in ((scale[-1][:half], 1.0), (scale[-1][:half], -1.0), (scale[-1][half:], 1.0))]
        nt = len(terms)
        if nt <= 32 and os.environ.get("EBEN_HINGE_MULTI", "1") != "0":   # one launch for all of them (this phase runs alone on the GPU: every launch is on the step's critical path)
            check(lib.eben_hinge_fwd_multi((ctypes.c_void_p * nt)(*[ptr(r) for r, _ in terms]), (ctypes.c_int64 * nt)(*[r.numel() for r, _ in terms]),
                                           (ctypes.c_float * nt)(*[t for _, t in terms]), nt, ptr(hinge), _stream()), "hinge_fwd_multi")
        else:
            for k, (rows, target) in enumerate(terms):
                check(lib.eben_hinge_fwd(ptr(rows), rows.numel(), target, ptr(hinge[k:]), _stream()), "hinge_fwd")
        if not ops.FUSED_LOSS_GLUE:
            hv = hinge.reshape(len(emb), 3).sum(dim=0) / len(emb)
            return {"feature_matching_loss": (sums[0::2] / sums[1::2]).sum() * inv, "adv_loss_gen": hv[0], "fake_loss": hv[1], "real_loss": hv[2]}
        vals = torch.empty(4, dtype=torch.float32, device=dev)
        check(lib.eben_disc_losses(ptr(sums), n, inv, ptr(hinge), len(emb), ptr(vals), _stream()), "disc_losses")
        return {"feature_matching_loss": vals[0], "adv_loss_gen": vals[1], "fake_loss": vals[2], "real_loss": vals[3]}

    # ---- the four backwards as one stacked pass ------------------------------------------------------
    def backward(self, want_param_grads: bool = True, sink=None):
        self.backward_launch(want_param_grads, sink)
        return self.backward_finish()

    @torch.no_grad()
    def backward_launch(self, want_param_grads: bool = True, sink=None):
        """Launches the stacked input-gradient chains on the four streams and returns; ``backward_finish`` joins them
        (the caller may run independent main-stream work in between).  sink (``ddp.GradSync``): the weight gradients are
        written straight into its bucket views and reported by ``collect_param_grads`` (data-parallel runs)."""
        self._sink = sink
        lib = load()
        s = self._state
        half, emb = s["half"], s["emb"]
        dev = emb[0][0].device
        n = len(s["fm_a"])
        one = torch.ones(1, dtype=torch.float32, device=dev)
        # feature-matching gradient: formed in the input-gradient epilogues where every launch of a chain can (no buffers, no extra
        # pass over the embeddings), by one eben_fm_bwd launch per chain otherwise
        fused = [self.fm_in_epilogue and ch.fm_epilogue_ok(emb[i], s["xp"][i], half) for i, ch in enumerate(self.chains)]
        da, k = [], 0
        for i, scale in enumerate(emb):
            for t in s["fm_a"][k:k + len(scale) - 2]:
                da.append(None if fused[i] else torch.empty_like(t))
            k += len(scale) - 2
        # feature-matching gradients per chain, aligned with out_0 .. out_{L-2}; each chain forms its own on its own stream (one
        # launch over all 26 pairs in front of the chains was 0.4 ms during which nothing else could start)
        fm_per_chain, first, k = [], [], 0
        for scale in emb:
            cnt = len(scale) - 2
            fm_per_chain.append(da[k:k + cnt])
            first.append(k)
            k += cnt
        inv_scales = 1.0 / len(emb)
        fm_all, numel_all, sums_ptr = s["fm_ptrs"], s["fm_numel"], ptr(s["fm_sums"])

        def run(i):
            k0, cnt = first[i], len(fm_per_chain[i])
            if cnt and not fused[i]:
                pairs = (ctypes.c_void_p * (2 * cnt))(*fm_all[2 * k0:2 * (k0 + cnt)])
                outs = (ctypes.c_void_p * cnt)(*[ptr(t) for t in fm_per_chain[i]])
                numel = (ctypes.c_int64 * cnt)(*numel_all[k0:k0 + cnt])
                check(lib.eben_fm_bwd(pairs, outs, numel, cnt, sums_ptr + 8 * k0, ptr(one), s["fm_inv"], _stream()), "fm_bwd")
            scale = emb[i]
            lg = scale[-1]
            seeds = torch.zeros((4 * half,) + tuple(lg.shape[1:]), dtype=torch.float32, device=dev)
            per = lg[:half].numel()
            flat = seeds.reshape(-1)
            for k2, (rows, target) in enumerate(((lg[:half], 1.0), (lg[:half], -1.0), (lg[half:], 1.0))):
                check(lib.eben_hinge_bwd(ptr(rows), rows.numel(), target, ptr(one), inv_scales * self.seed_weights[k2],
                                         ptr(flat[(k2 + 1) * per:]), _stream()), "hinge_bwd")
            return self.chains[i].backward(scale, s["xp"][i], fm_per_chain[i], seeds, half, want_param_grads,
                                           (sums_ptr + 8 * k0, s["fm_inv"]) if fused[i] and cnt else None)

        # `da` / `one` live on the main stream's pool and are read by the chains: keep them referenced until the join
        self._bwd = (self._launch_on_streams(run), want_param_grads, (da, one, fm_per_chain))

    @torch.no_grad()
    def backward_finish(self):
        """Returns (d fm / d bands, d fm / d audio, d adv / d bands, d adv / d audio).  With ``want_param_grads`` the
        weight gradients of real_loss + fake_loss are launched behind the input-gradient chains and left running
        (``collect_param_grads`` joins them)."""
        res, want_param_grads, _keep = self._bwd
        self._join_streams()
        self._bwd = _keep = None
        s = self._state
        half = s["half"]
        dev = s["emb"][0][0].device
        main = torch.cuda.current_stream()
        for r in res:
            # allocated on a chain stream, read on the main stream from here on: without this the allocator may hand
            # the block to the chain's next allocation (the weight-gradient workspaces below) while it is still read
            r[0].record_stream(main)
        # input gradients: the PQMF-band chains share the `bands[:, -q:]` input, the MelGAN chain reads the waveform
        bshape = s["bands_shape"]
        gb = torch.zeros((2 * half,) + bshape[1:], dtype=torch.float32, device=dev)
        acc = res[0][0]
        for r in res[1:-1]:
            acc = acc + r[0]
        gb[:, -self.q:, :] = acc
        ga = res[-1][0]
        self._pending = None
        if want_param_grads:
            # phase B: weight gradients on the chains' streams, NOT joined here -- see collect_param_grads()
            pend = [None] * len(self.chains)
            n = len(self.chains)
            for i in [n - 1] + list(range(n - 1)):   # the longest chain first
                with torch.cuda.stream(self._streams[i]):
                    pend[i] = self.chains[i].weight_grads(res[i][1], half, self._sink)
                    if self._sink is not None:
                        # data-parallel run: this chain's gradients are in the buckets once its stream gets here -- report them
                        # now, from this stream, so that the buckets they complete (MelGAN's 75 MB first) are exchanged underneath
                        # the other chains' weight gradients and the generator backward instead of in front of Adam
                        self._sink.mark_ready([p for lay in self.chains[i].layers for p in lay.params() if p is not None and p.requires_grad])
            self._pending = (pend, s)   # keeps the saved activations alive until the kernels have run
        self._state = None
        return gb[:half], ga[:half], gb[half:], ga[half:]

    def collect_param_grads(self):
        """Joins the weight-gradient work launched by ``backward`` and returns the gradients of
        real_loss + fake_loss aligned with ``list(disc.parameters())`` (None if none were requested)."""
        if getattr(self, "_pending", None) is None:
            return None
        pend, *_keep = self._pending
        main = torch.cuda.current_stream()
        for st in set(self._streams) | set(getattr(self, "_used_streams", None) or ()):
            main.wait_stream(st)
        sink = getattr(self, "_sink", None)
        if sink is not None:   # already in the gradient buckets and reported chain by chain (backward_finish): nothing to inject
            self._pending = self._sink = None
            return None
        by_param = {}
        for ch, grads in zip(self.chains, pend):
            for lay, (dv, dg, dbias) in zip(ch.layers, grads):
                v, gain, bias = lay.params()
                by_param[id(v)], by_param[id(gain)] = dv, dg
                if bias is not None:
                    by_param[id(bias)] = dbias
                for t in (dv, dg, dbias):
                    if t is not None:
                        t.record_stream(main)
        self._pending = None
        return [by_param.get(id(p)) for p in ops.parameters_of(self.disc)]
