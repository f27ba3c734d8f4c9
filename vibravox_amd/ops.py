"""torch.autograd plumbing over the C ABI of libeben_hip.so.

PyTorch is used for device memory (caching allocator), streams and the autograd graph; every
arithmetic op of the EBEN hot path below is a hand-written HIP kernel (``vibravox_amd/csrc``).
Nothing here runs on CPU tensors: ``_lib.ptr`` raises for them.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import weakref
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import EbenConv1dDesc, EbenPackJob, EbenWnBwdItem, EbenWnScaleItem, check, load, ptr, stream

# Optimisers that write parameters behind autograd's back (FusedAdam) bump the epoch of exactly
# the storages they touched, so that only those layers' packed-weight caches are rebuilt.
_storage_epoch: Dict[int, int] = {}


def bump_weights_epoch(params=None) -> None:
    if params is None:
        for k in list(_storage_epoch):
            _storage_epoch[k] += 1
        _storage_epoch[-1] = _storage_epoch.get(-1, 0) + 1
        return
    for p in params:
        k = p.data_ptr()
        _storage_epoch[k] = _storage_epoch.get(k, 0) + 1


_param_lists: "weakref.WeakKeyDictionary" = None


def parameters_of(module) -> list:
    """``list(module.parameters())`` without walking the module tree each step (four walks of ~190 tensors were 0.7 ms of a step's host
    time): the list is kept per module until ``bump_weights_epoch()`` (no arguments) says the parameters were replaced -- the same call
    that invalidates the engines' packed-weight caches.  ``requires_grad`` is NOT cached: callers filter on it."""
    global _param_lists
    if _param_lists is None:
        import weakref
        _param_lists = weakref.WeakKeyDictionary()
    epoch = _storage_epoch.get(-1, 0)
    hit = _param_lists.get(module)
    if hit is None or hit[0] != epoch:
        hit = _param_lists[module] = (epoch, list(module.parameters()))
    return hit[1]


def conv_params(m):
    """(v, g) of a weight-normalised HipConv1d, (weight, None) of a plain one -- the Parameter objects, kept on the module until
    ``bump_weights_epoch()`` announces replaced parameters (the engines ask ~300 times a step)."""
    epoch = _storage_epoch.get(-1, 0)
    hit = m.__dict__.get("_vg")
    if hit is None or hit[0] != epoch:
        if m.weight_norm:
            prm = m.parametrizations["weight"]
            vg = (prm.original1, prm.original0)
        else:
            vg = (m.weight, None)
        hit = m.__dict__["_vg"] = (epoch, vg)
    return hit[1]


# Set (by the train step) while a backward pass only needs input gradients: a Function's
# ``needs_input_grad`` is fixed at forward time, so without this the weight-gradient GEMMs would
# run inside torch.autograd.grad(loss, bands) although their results are discarded.
_skip_weight_grads = [False]


# Set while a backward pass only needs WEIGHT gradients (the balancing norms: torch.autograd.grad(bands, last_conv.weight)): the
# Function cannot know that its input gradient is discarded either.
_skip_input_grads = [False]


class input_grads_disabled:
    def __enter__(self):
        self.prev = _skip_input_grads[0]
        _skip_input_grads[0] = True

    def __exit__(self, *exc):
        _skip_input_grads[0] = self.prev


class weight_grads_disabled:
    def __enter__(self):
        self.prev = _skip_weight_grads[0]
        _skip_weight_grads[0] = True

    def __exit__(self, *exc):
        _skip_weight_grads[0] = self.prev


# Weight-gradient work on a side HIP stream.  A layer's backward is a serial chain of launches: input
# gradient (needed by the next layer) and, independent of it, the weight-gradient GEMM + slab reduction +
# weight-norm chain rule.  Inside `weight_grads_on_side_stream()` the second group is issued on a side
# stream, so the input-gradient chain of the whole network is not held up by it (generator backward:
# ~2.9 ms of dX launches instead of ~8.6 ms of everything in series).  The caller must `join()` before
# anything on the main stream reads the parameter gradients (the optimiser step); only used while
# `.grad` is None for the parameters involved and no accumulation hook hangs on them: the deferred gradients never pass through
# autograd -- backward() returns None for them and `join()` assigns `p.grad` itself once the side stream is done.
# `keep` holds a reference to every upstream gradient a side-stream kernel reads until `join()`: autograd sums gradients IN PLACE
# into a buffer it holds the only reference to (a residual add hands the same tensor to both branches), which would rewrite the
# gradient on the main stream under the kernel reading it.
_side = {"enabled": False, "stream": None, "keep": [], "prepacked": None, "wn_jobs": [], "sink": None, "sunk": [], "assign": []}


def _no_grad_hooks(p) -> bool:
    return p is None or (not getattr(p, "_post_accumulate_grad_hooks", None) and not p._backward_hooks)


class weight_grads_on_side_stream:
    """sink: an object with ``grad_buffer(param) -> tensor | None`` and ``mark_ready(params)`` (``ddp.GradSync``): the
    weight gradients are then written straight into the sink's buffers (data-parallel gradient buckets) and reported at
    ``join()``, instead of being handed to autograd -- which would accumulate them into ``p.grad`` on the main stream
    the moment the layer's backward returns, i.e. force the weight-gradient work back onto the critical path."""

    def __init__(self, sink=None):
        self.sink = sink

    def __enter__(self):
        self.prev = (_side["enabled"], _side["sink"])
        _side["enabled"], _side["sink"] = True, self.sink
        return self

    def __exit__(self, exc_type, *exc):
        _side["enabled"], _side["sink"] = self.prev
        if exc_type is not None:
            # a backward that raised half-way leaves jobs pointing at tensors nobody will hand over: wait for what was
            # launched, then drop everything instead of letting the next join() write through stale pointers
            st = _side["stream"]
            if st is not None:
                for k in SIDE_STREAMS:
                    torch.cuda.current_stream(st.device).wait_stream(aux_stream(k, st.device))
            for key in ("wn_jobs", "keep", "sunk", "assign"):
                _side[key] = []

    def join(self):
        st = _side["stream"]
        if st is not None:
            for k in SIDE_STREAMS[1:]:
                st.wait_stream(aux_stream(k, st.device))
            if _side["wn_jobs"]:
                with torch.cuda.stream(st):
                    wn_bwd_multi(_side["wn_jobs"])   # slab sums + weight-norm chain rule of every layer: two launches
                _side["wn_jobs"] = []
            torch.cuda.current_stream(st.device).wait_stream(st)
        assign, _side["assign"] = _side["assign"], []
        for p, t in assign:   # complete on this stream from here on
            if p.grad is None:
                p.grad = t
            else:
                p.grad.add_(t)
        if _side["sunk"]:
            sunk, _side["sunk"] = _side["sunk"], []
            self.sink.mark_ready(sunk)
        _side["keep"].clear()   # the current stream has waited for the side stream: what its kernels read may be reused from here on


# The HIP runtime multiplexes streams onto FOUR hardware queues (GPU_MAX_HW_QUEUES; 8 measured 50 % slower): a fifth
# stream shares a queue with another one and its launches are executed in submission order with that one's -- a tiny
# Adam launch on the main stream then sits behind ~17 ms of weight-gradient kernels of a stream it has nothing to do with
# (seen as 28 / 33 / 36 ms steps from run to run, depending on which streams happened to collide).  So the step uses the
# main stream plus exactly THREE auxiliary streams, created once, in this order:
#   0: MelGAN discriminator chain   1: the three PQMF-band discriminator chains (in series)   2: generator weight
#   gradients, weight pre-packing
_aux = {"device": None, "streams": None}
#: HIP priorities of the three auxiliary streams (0 = default, -1 = high; torch's range on this device is (0, -1)).  The stream that
#: carries the generator's weight gradients, the weight pre-packing and (spread backward) one PQMF-band chain gets the high one: the
#: step ends on that work and every other queue's kernels can absorb a delay.  [MI355X, same box] ms/step: "0,0,0" 15.27 / 15.07,
#: "0,0,-1" 15.12 / 14.92 / 14.94, "-1,0,0" 15.42, "0,-1,0" 15.69, "-1,0,-1" 15.19, "0,-1,-1" 15.13, "-1,-1,-1" 15.27; the MelGAN
#: layer-4 forward launch inside the step 0.438 -> 0.36 ms.  Round 5 (9.3 ms step, persistent tile kernels that hold a CU for their whole
#: launch): the high priority no longer pays -- "0,0,-1" 9.282 / 9.287 / 9.269, "0,0,0" 9.199 / 9.223 / 9.229, "-1,0,0" 9.305 / 9.291 /
#: 9.322 (same box, 100 steps each) -- all default
AUX_PRIORITY = tuple(int(t) for t in os.environ.get("EBEN_AUX_PRIORITY", "0,0,0").split(","))


def aux_stream(i: int, device=None) -> "torch.cuda.Stream":
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    if _aux["streams"] is None or _aux["device"] != device:
        _aux["device"], _aux["streams"] = device, [torch.cuda.Stream(device=device, priority=AUX_PRIORITY[i]) for i in range(3)]
    return _aux["streams"][i]


#: auxiliary streams the deferred weight-gradient jobs are dealt over (round robin).  One stream is the measured best [MI355X]:
#: "2" 15.35 / 15.39 ms/step, "2,1" 15.62, "2,1,0" 15.54, "2,0" 16.76 -- the other queues still carry the discriminators'
#: weight gradients, and a generator job dealt behind them waits for work it does not depend on
SIDE_STREAMS = tuple(int(t) for t in os.environ.get("EBEN_SIDE_STREAMS", "2").split(",") if t.strip() != "")


def _side_stream(device, deal: bool = False) -> "torch.cuda.Stream":
    """The side stream (weight pre-packing; joins); ``deal``: the next one of SIDE_STREAMS, for a weight-gradient job."""
    st = _side["stream"] = aux_stream(SIDE_STREAMS[0], device)
    if not deal or len(SIDE_STREAMS) == 1:
        return st
    k = _side["rr"] = (_side.get("rr", -1) + 1) % len(SIDE_STREAMS)
    return aux_stream(SIDE_STREAMS[k], device)


def wn_bwd_multi(jobs) -> None:
    """jobs: (slabs, nslab, slab_stride, rows, cols, row_stride, g, v, norm, dg, dv, dbias[, col_perm_k]) per layer -- the arguments
    of ``eben_wn_bwd`` (col_perm_k: the slabs' columns are in the bundle-major order of ``eben_bl_conv1d_bwd_dw``) -- on the current
    stream, as one multi-tensor call."""
    if not jobs:
        return
    items = (EbenWnBwdItem * len(jobs))()
    p = lambda t: t if t is None or isinstance(t, int) else ptr(t)   # outputs may arrive as raw device pointers
    for it, job in zip(items, jobs):
        slabs, nslab, slab_stride, rows, cols, row_stride, g, v, norm, dg, dv, dbias = job[:12]
        it.col_perm_k = job[12] if len(job) > 12 else 0
        it.slabs, it.g, it.v, it.norm = ptr(slabs), ptr(g), ptr(v), ptr(norm)
        it.dg, it.dv, it.dbias = p(dg), p(dv), p(dbias)
        it.slab_stride, it.nslab, it.rows, it.cols, it.row_stride = slab_stride, nslab, rows, cols, row_stride
    check(load().eben_wn_bwd_multi(items, len(jobs), stream()), "wn_bwd_multi")


def wn_scale_multi(jobs) -> None:
    """jobs: (g, v, rows, cols, scale, norm) per layer, one multi-tensor call on the current stream."""
    if not jobs:
        return
    items = (EbenWnScaleItem * len(jobs))()
    for it, (g, v, rows, cols, scale, norm) in zip(items, jobs):
        it.g, it.v, it.scale, it.norm, it.rows, it.cols = ptr(g), ptr(v), ptr(scale), ptr(norm), rows, cols
    check(load().eben_wn_scale_multi(items, len(jobs), stream()), "wn_scale_multi")


class KernelTimer:
    """HIP-event timing of one layer's forward ("fwd") or input-gradient ("dx") launch on the stream it is launched on (bench.py)."""

    def __init__(self, spec: "ConvSpec", which: str = "fwd"):
        self.spec, self.which, self.enabled, self.events, self.batch = spec, which, False, [], None

    def start(self):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def stop(self, e0, batch: int) -> None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.events.append((e0, e1))
        self.batch = batch

    def mean_ms(self) -> Optional[float]:
        if not self.events:
            return None
        return sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)


_timers: List[KernelTimer] = []


def set_kernel_timers(timers: Sequence[KernelTimer]) -> None:
    _timers[:] = list(timers)


def set_kernel_timer(t: Optional[KernelTimer]) -> None:
    set_kernel_timers([] if t is None else [t])


def kernel_timer_for(spec: "ConvSpec", which: str) -> Optional[KernelTimer]:
    for t in _timers:
        if t.enabled and t.which == which and t.spec == spec:
            return t
    return None


def _empty(n_bytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(1, (n_bytes + 3) // 4), dtype=torch.float32, device=like.device)


# --------------------------------------------------------------------------------------------
# conv layers
# --------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ConvSpec:
    """Static description of one Conv1d / ConvTranspose1d layer (+ fused activations)."""

    c_in: int
    c_out: int
    ksize: int
    stride: int = 1
    dilation: int = 1
    groups: int = 1
    pad_l: int = 0
    pad_r: int = 0
    reflect: bool = False
    transposed: bool = False
    output_padding: int = 0
    in_slope: float = 1.0
    out_slope: float = 1.0

    def out_len(self, l_in: int) -> int:
        if self.transposed:
            return (l_in - 1) * self.stride - 2 * self.pad_l + self.dilation * (self.ksize - 1) + self.output_padding + 1
        return (l_in + self.pad_l + self.pad_r - self.dilation * (self.ksize - 1) - 1) // self.stride + 1

    def weight_shape(self) -> Tuple[int, int, int]:
        if self.transposed:
            return (self.c_in, self.c_out // self.groups, self.ksize)
        return (self.c_out, self.c_in // self.groups, self.ksize)


_desc_cache: Dict[Tuple[ConvSpec, int, int], EbenConv1dDesc] = {}


MATH_F32, MATH_BF16, MATH_BF16X2, MATH_BF16X3, MATH_BF16X6 = 0, 1, 2, 3, 4   # EBEN_MATH_* of include/eben_hip.h

# Arithmetic of the BACKWARD contractions (input and weight gradients) of `conv_layer`; the forward is always exact fp32.
# Read when the forward runs (the backward image of the weights is packed then).
_backward_math = [MATH_F32]


class backward_math:
    """with ops.backward_math(ops.MATH_BF16): forwards run inside record bf16 backward contractions."""

    def __init__(self, math: int):
        self.math = math

    def __enter__(self):
        self.prev = _backward_math[0]
        _backward_math[0] = self.math

    def __exit__(self, *exc):
        _backward_math[0] = self.prev


def conv_desc(spec: ConvSpec, batch: int, l_in: int, math: int = MATH_F32) -> EbenConv1dDesc:
    key = (spec, batch, l_in, math)
    d = _desc_cache.get(key)
    if d is None:
        if len(_desc_cache) >= 4096:   # variable clip lengths ("pad" collation): do not grow without bound
            _desc_cache.clear()
        d = EbenConv1dDesc(
            batch, spec.c_in, spec.c_out, l_in, spec.out_len(l_in), spec.ksize, spec.stride, spec.dilation, spec.groups,
            spec.pad_l, spec.pad_r, 1 if spec.reflect else 0, 1 if spec.transposed else 0, spec.in_slope, spec.out_slope, math,
        )
        _desc_cache[key] = d
    return d


class PackedWeights:
    """Weight-norm scale + MFMA-layout copies of one layer's weights, rebuilt when (v, g) change."""

    __slots__ = ("key", "scale", "norm", "wp_fwd", "wp_bwd", "last")

    def __init__(self):
        self.key = None
        self.scale = self.norm = self.wp_fwd = self.wp_bwd = None
        self.last = None   # (spec, descriptor) of the latest pack: what `prepack` rebuilds ahead of the next step


_pack_batch: List[Optional[list]] = [None]


def conv1d_pack(d: EbenConv1dDesc, v, scale, wp_fwd, wp_bwd) -> None:
    """``eben_conv1d_pack`` on the current stream -- or, inside ``pack_batch()``, one job of the ``eben_conv1d_pack_multi`` call
    issued when the context closes (the prepack sequences: ~150 images per step in ~15 launches)."""
    if _pack_batch[0] is not None:
        _pack_batch[0].append((d, v, scale, wp_fwd, wp_bwd))
        return
    check(load().eben_conv1d_pack(ctypes.byref(d), ptr(v), ptr(scale), ptr(wp_fwd), ptr(wp_bwd), stream()), "conv1d_pack")


@contextlib.contextmanager
def pack_batch():
    """Collects the ``conv1d_pack`` calls made inside and issues them as one ``eben_conv1d_pack_multi`` on the current stream at
    exit (everything the jobs read -- the weight-norm scales -- must have been launched on that stream before)."""
    if _pack_batch[0] is not None:   # nested: the outer context flushes
        yield
        return
    _pack_batch[0] = []
    try:
        yield
        jobs = _pack_batch[0]
    finally:
        _pack_batch[0] = None
    if jobs:
        table = (EbenPackJob * len(jobs))()
        for it, (d, v, scale, wp_fwd, wp_bwd) in zip(table, jobs):
            it.desc, it.v, it.scale, it.wp_fwd, it.wp_bwd = d, ptr(v), ptr(scale), ptr(wp_fwd), ptr(wp_bwd)
        check(load().eben_conv1d_pack_multi(table, len(jobs), stream()), "conv1d_pack_multi")


def _pack_key(v, g, d, d_bwd):
    return (v.data_ptr(), v._version, _storage_epoch.get(v.data_ptr(), 0), _storage_epoch.get(-1, 0),
            None if g is None else (g.data_ptr(), g._version, _storage_epoch.get(g.data_ptr(), 0)), d.batch, d.l_in, d.math, d_bwd.math)


def _buffer(old: Optional[torch.Tensor], numel: int, like: torch.Tensor, reuse: bool) -> torch.Tensor:
    """A float buffer of ``numel`` elements: ``old`` itself when ``reuse`` and it has that size (``prepack``: the step that read the
    old contents is complete, and a graph replay needs the buffers it captured to stay the ones in use), else a fresh one."""
    if reuse and old is not None and old.numel() == numel and old.device == like.device:
        return old
    return torch.empty(numel, dtype=torch.float32, device=like.device)


def pack_weights(spec: ConvSpec, d: EbenConv1dDesc, v: torch.Tensor, g: Optional[torch.Tensor],
                 cache: Optional[PackedWeights], need_bwd: bool, d_bwd: Optional[EbenConv1dDesc] = None, pre_scale=None,
                 reuse: bool = False) -> PackedWeights:
    """d_bwd: descriptor of the backward launches when it differs from the forward's (bf16 backward math);
    pre_scale: (scale, norm) already computed for the current weights (multi-tensor launch in `prepack`);
    reuse: rebuild into the buffers the cache already holds where their sizes fit (only `prepack` may: nothing reads them any more)."""
    lib = load()
    d_bwd = d if d_bwd is None else d_bwd
    key = _pack_key(v, g, d, d_bwd)
    pw = cache if cache is not None else PackedWeights()
    if _side["prepacked"] is not None and torch.cuda.current_stream() != _side["stream"]:
        join_prepack()   # images built ahead of time on the side stream: first consumer waits for them
    need_bwd = need_bwd or cache is not None  # a module-level cache serves every later pass
    if pw.key == key and (pw.wp_bwd is not None or not need_bwd):
        return pw
    st = stream()
    rows = v.shape[0]
    if g is not None and pre_scale is not None:
        pw.scale, pw.norm = pre_scale
    elif g is not None:
        pw.scale = _buffer(pw.scale, rows, v, reuse)
        pw.norm = _buffer(pw.norm, rows, v, reuse)
        check(lib.eben_wn_scale(ptr(g), ptr(v), rows, v.numel() // rows, ptr(pw.scale), ptr(pw.norm), st), "wn_scale")
    else:
        pw.scale = pw.norm = None
    pw.wp_fwd = _buffer(pw.wp_fwd, lib.eben_conv1d_packed_floats(ctypes.byref(d), 0), v, reuse)
    pw.wp_bwd = _buffer(pw.wp_bwd, lib.eben_conv1d_packed_floats(ctypes.byref(d_bwd), 1), v, reuse) if need_bwd else None
    if d_bwd is d or not need_bwd:
        conv1d_pack(d, v, pw.scale, pw.wp_fwd, pw.wp_bwd)
    else:
        conv1d_pack(d, v, pw.scale, pw.wp_fwd, None)
        conv1d_pack(d_bwd, v, pw.scale, None, pw.wp_bwd)
    pw.key = key
    pw.last = (spec, d, d_bwd)
    return pw


_replayed = weakref.WeakSet()   # every ReplayedPrepack / ReplayedChain alive: graphs_pending() asks them


def graphs_pending() -> int:
    """Launch sequences that are in use but still run eagerly on their way to a capture (a step that captures takes tens of ms: a
    measurement waits until this is 0)."""
    if capture_gate.settled and capture_gate.active():
        return 0   # multi-rank: the gate has closed for good, whatever still runs eagerly stays eager
    n = 0
    for r in list(_replayed):
        if type(r).enabled and r.sig is not None and r.graph is None and not ReplayedPrepack._multi_rank():
            n += 1
    return n


#: Multi-rank runs replay the collective-free launch sequences as graphs like a single-GPU run does (every graph holds the launches of ONE
#: stream between two joins; the all-reduces are issued outside them, from ``GradSync.mark_ready`` / ``finish``).  WHEN a sequence is
#: captured is agreed by all ranks (``CaptureGate``: a vote at the end of each of the first steps; captures only in steps every rank
#: opened, each behind a drain of the communication streams, with ``capture_error_mode="thread_local"`` so that the process group's
#: watchdog thread may go on querying its events).  Without the graphs a rank enqueues ~11 ms of host work per ~9 ms step.
#: ``EBEN_DDP_GRAPHS=0`` is the escape hatch (eager launches).  No N > 1 run on hardware has happened yet: the
#: protocol is covered by the world-size-2 gloo test of tests/test_ddp_gloo.py with a recorder in place of the HIP capture.
DDP_GRAPHS = os.environ.get("EBEN_DDP_GRAPHS", "1") != "0"


class CaptureGate:
    """Multi-rank agreement on WHEN launch sequences are captured into HIP graphs (one process per GPU, ``WORLD_SIZE > 1``).

    A capture is a host-side event of tens of milliseconds with a device-wide synchronisation at both ends.  On one rank its timing is
    nobody else's business; with several ranks exchanging gradient buckets it has to be the SAME step on every rank -- else one rank
    sits in a capture while the others wait for it inside a collective, step after step, and which step that is depends on each rank's
    allocator history (a signature contains addresses).  The gate makes the decision a function of the step index agreed by all ranks:

      * a ``ReplayedChain`` / ``ReplayedPrepack`` that reaches its capture threshold is HELD (it goes on running eagerly) and the rank
        votes "ready"; a sequence still counting eager rounds makes the rank vote "busy"; a rank with neither votes "idle";
      * ``step_end()`` -- called by the train step after its last collective was issued -- all-reduces the three flags (one tiny
        blocking collective per step, only until the gate has settled) and opens the gate for the NEXT step iff at least one rank is
        ready and every rank is ready or idle: all ranks see the same sums, so all open in the same step;
      * in an open step exactly the sequences held before it are captured, each after ``drain()``: no capture starts while a gradient
        bucket's collective is in flight (the communication streams registered by ``ddp.GradSync`` are synchronised first);
      * after three consecutive votes with nothing ready or busy anywhere the gate SETTLES: no more votes, no more captures (a
        signature that turns up later -- a validation shape -- runs eagerly for good).

    Single-rank runs (and ``EBEN_DDP_GRAPHS=0``, which keeps multi-rank runs eager altogether) never consult it."""

    QUIET_VOTES = 3

    def __init__(self):
        self._drains = []
        self.reset()

    def reset(self) -> None:
        self.open, self.settled = False, False
        self.ready = self.busy = self.quiet = self.step = 0
        #: (step index, what) of every capture under the gate and every vote result: identical on all ranks (tests/test_ddp_gloo.py)
        self.history = []

    @staticmethod
    def active() -> bool:
        d = torch.distributed
        return DDP_GRAPHS and d.is_available() and d.is_initialized() and d.get_world_size() > 1

    def register_drain(self, fn) -> None:
        """fn(): returns once none of the caller's collectives is in flight (``ddp.GradSync``: synchronise the communication stream)."""
        if fn not in self._drains:
            self._drains.append(fn)

    def drain(self) -> None:
        for fn in self._drains:
            fn()

    def counting(self) -> None:
        """A sequence ran one of its eager rounds below the capture threshold."""
        if not self.settled:
            self.busy += 1

    def may_capture(self, seq, what: str = "") -> bool:
        """Called by a sequence AT its capture threshold: True = capture now (gate open and the sequence was held before this step)."""
        if self.settled:
            return False
        if self.open and getattr(seq, "_held_since", None) is not None and seq._held_since < self.step:
            self.drain()
            self.history.append((self.step, "capture " + what))
            seq._held_since = None
            return True
        if getattr(seq, "_held_since", None) is None:
            seq._held_since = self.step
        self.ready += 1
        return False

    def step_end(self, group=None) -> None:
        if self.settled or not self.active():
            return
        d = torch.distributed
        world = d.get_world_size(group)
        ready, busy = int(self.ready > 0), int(self.busy > 0)
        dev = torch.device("cuda", torch.cuda.current_device()) if d.get_backend(group) == "nccl" else torch.device("cpu")
        vote = torch.tensor([ready, int(not ready and not busy), busy], dtype=torch.int32, device=dev)
        d.all_reduce(vote, op=d.ReduceOp.SUM, group=group)
        r, idle, b = (int(v) for v in vote.tolist())
        self.open = r >= 1 and r + idle == world
        self.quiet = self.quiet + 1 if (r == 0 and b == 0) else 0
        self.history.append((self.step, f"vote ready {r} idle {idle} busy {b} -> {'open' if self.open else 'closed'}"))
        if self.quiet >= self.QUIET_VOTES:
            self.settled, self.open = True, False
            self.history.append((self.step, "settled"))
        self.ready = self.busy = 0
        self.step += 1


capture_gate = CaptureGate()


def graphs_captured() -> int:
    """Launch sequences currently replayed as graphs."""
    return sum(1 for r in list(_replayed) if r.graph is not None)


class ReplayedPrepack:
    """Graph replay of a prepack sequence.  Rebuilding the packed weight images after an optimiser step is ~150 tiny launches per
    step (one or two per layer and direction) whose cost is entirely host-side: ~3 ms of Python / launch time per step during which
    the GPU has nothing else queued -- on a 19 ms step.  The sequence is static (same kernels, same parameter storage, same image
    buffers every step), so after two eager rounds it is captured into a HIP graph and replayed: one launch call.  ``sig`` names
    everything the launches depend on besides the weights' values (layers, shapes, parameter storage AND the addresses of the image /
    scale buffers the launches write: a forward at another shape between two train steps -- validation -- reallocates them, and a
    replay would go on filling the orphaned ones); a new signature falls back to eager rounds and a new capture.  The prepack bodies
    rebuild INTO the buffers the caches hold (``_buffer(reuse=True)``), so the signature settles after one eager round.  Tensors
    allocated by the body while capturing live in the graph's pool for as long as the graph."""

    enabled = os.environ.get("EBEN_PREPACK_GRAPH", "1") != "0"

    def __init__(self):
        self.graph, self.sig, self.rounds = None, None, 0
        _replayed.add(self)

    @staticmethod
    def _multi_rank() -> bool:
        """True when a multi-rank run asked for eager launches (``EBEN_DDP_GRAPHS=0``, see ``DDP_GRAPHS``)."""
        if DDP_GRAPHS:
            return False
        d = torch.distributed
        return d.is_available() and d.is_initialized() and d.get_world_size() > 1

    @staticmethod
    def _capture(body, stream_):
        """``body()`` recorded into a HIP graph on ``stream_`` (tests substitute a recorder)."""
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream_, capture_error_mode="thread_local"):
            body()
        return graph

    def run(self, sig, body, stream_) -> bool:
        """Runs (or replays) ``body`` on ``stream_`` (a torch side stream, current on entry); True when it was a replay, in which
        case the caller refreshes its own cache keys (the body's bookkeeping did not run)."""
        if not self.enabled or self._multi_rank():
            body()
            return False
        if sig != self.sig:
            self.graph, self.sig, self.rounds = None, sig, 0
        self.rounds += 1
        if self.graph is not None:
            self.graph.replay()
            return True
        gated = capture_gate.active()
        if self.rounds < 3:
            if gated:
                capture_gate.counting()
            body()
            return False
        if gated and not capture_gate.may_capture(self, "prepack"):
            body()
            return False
        try:
            graph = self._capture(body, stream_)
        except Exception as exc:   # capture is an optimisation: fall back to eager launches for good
            import warnings

            warnings.warn(f"prepack graph capture failed ({exc!r}): staying with eager launches")
            ReplayedPrepack.enabled = False
            body()
            return False
        self.graph = graph
        graph.replay()
        return False   # the body ran (under capture): its bookkeeping is current


class ReplayedChain:
    """Graph replay of one LINEAR launch sequence with device-side results (a sub-discriminator's forward body, stacked input
    gradients or weight gradients on its stream).  ``run(sig, fn, stream)`` returns what ``fn()`` returned -- tensors allocated by the
    body, which from the capture on live in the graph's pool and are REWRITTEN IN PLACE by every replay: valid until the next call.
    ``sig`` names everything the launches read besides values -- shapes, every input / weight-image / parameter address, arithmetic
    plan -- a new signature falls back to eager calls and, the second time in a row it is seen, a new capture.  Bodies chained
    through their results settle one after the other (the consumer's signature contains the producer's output addresses, which only
    stop changing once the producer replays).  One graph per chain and phase: no cross-stream edge inside a graph (the HIP runtime
    would serve parallel branches from extra hardware queues, see ``aux_stream``); events between chains stay outside."""

    enabled = os.environ.get("EBEN_CHAIN_GRAPHS", "1") != "0"
    #: graphs kept per chain, least recently used first out.  A run whose signature alternates -- variable clip lengths, a validation
    #: shape between train steps, ``update_discriminator_ratio < 1`` toggling ``want_param_grads`` -- replays each variant from its own
    #: graph instead of paying a capture (device-wide synchronisation, tens of ms) at every other change.  Each graph keeps its pool
    #: (the tensors its body allocates) for as long as it is cached: with KEEP variants alive a chain pins up to KEEP times its
    #: activation memory (~22 chains per step; config 2: ~1.5 GB per variant in total) -- lower it for runs over many clip lengths.
    KEEP = int(os.environ.get("EBEN_CHAIN_GRAPHS_KEEP", "4"))
    _generation = [0]

    def __init__(self):
        self.graph, self.sig, self.rounds, self.out = None, None, 0, None
        self.captures = 0   # names the generation of the tensors in ``out`` (unique per capture, restored on a cache hit)
        self._cache = {}    # signature -> (graph, out, generation), insertion order = recency
        self._seen = {}     # signature -> eager rounds so far (signatures that strictly alternate still reach their second round)
        _replayed.add(self)

    @staticmethod
    def _capture(fn, stream_):
        """(graph, what ``fn()`` returned while it was recorded) -- tests substitute a recorder.  stream_ None: torch's own capture
        stream (a sequence that normally runs on the default stream, which cannot capture); the replays are launched on whatever stream
        is current then."""
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream_, capture_error_mode="thread_local"):
            out = fn()
        return graph, out

    def run(self, sig, fn, stream_):
        if not self.enabled or ReplayedPrepack._multi_rank() or _timers_enabled():
            return fn()
        hit = self._cache.pop(sig, None)
        if hit is not None:
            self._cache[sig] = hit   # most recent
            self.graph, self.out, self.captures = hit
            self.sig = sig
            self.graph.replay()
            return self.out
        if sig != self.sig:
            self.graph, self.sig, self.out = None, sig, None
        self.rounds = self._seen.get(sig, 0) + 1
        self._seen[sig] = self.rounds
        while len(self._seen) > 4 * max(1, self.KEEP):   # bounded: forget the oldest signatures' counts
            self._seen.pop(next(iter(self._seen)))
        gated = capture_gate.active()
        if self.rounds < 2:
            if gated:
                capture_gate.counting()
            return fn()
        if gated and not capture_gate.may_capture(self, "chain"):
            self._seen[sig] = self.rounds = 1   # held at the threshold: the next sighting asks again
            return fn()
        try:
            graph, out = self._capture(fn, stream_)
        except Exception as exc:   # capture is an optimisation: fall back to eager launches for good
            import warnings

            warnings.warn(f"chain graph capture failed ({exc!r}): staying with eager launches")
            ReplayedChain.enabled = False
            return fn()
        ReplayedChain._generation[0] += 1
        self.graph, self.out, self.captures = graph, out, ReplayedChain._generation[0]
        self._cache[sig] = (graph, out, self.captures)
        self._seen.pop(sig, None)
        while len(self._cache) > max(1, self.KEEP):
            self._cache.pop(next(iter(self._cache)))
        graph.replay()   # the capture recorded the launches without running them
        return out


def _timers_enabled() -> bool:
    return any(t.enabled for t in _timers)


_conv_prepack_graph = ReplayedPrepack()


def prepack(layers) -> None:
    """Rebuilds the packed weights of ``layers`` (modules with ``_packed`` / ``spec`` / weight-norm parameters, i.e.
    ``torch_modules.utils.HipConv1d``) for the descriptors of their latest forward, on the side stream: called right
    after an optimiser step, the ~2 tiny launches per layer (weight-norm scale, pack) leave the next forward's critical
    path and run underneath whatever the main stream does next.  ``join_prepack`` must precede the next use."""
    todo = [m for m in layers if getattr(m, "_packed", None) is not None and m._packed.last is not None]
    if not todo:
        return
    dev = todo[0]._packed.wp_fwd.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    side.wait_stream(main)   # the step that used the old images (and the optimiser that changed the weights) is complete
    def params_of(m):
        v, g = conv_params(m)
        return v.detach(), None if g is None else g.detach()

    def body():
        scales, jobs = {}, []
        for m in todo:   # weight-norm scales of every layer: one multi-tensor launch
            if m.weight_norm:
                v, g = params_of(m)
                rows = v.shape[0]
                sc = _buffer(m._packed.scale, rows, v, True)
                nm = _buffer(m._packed.norm, rows, v, True)
                scales[id(m)] = (sc, nm)
                jobs.append((g, v, rows, v.numel() // rows, sc, nm))
        wn_scale_multi(jobs)
        with pack_batch():
            for m in todo:
                spec, d, d_bwd = m._packed.last
                v, g = params_of(m)
                pack_weights(spec, d, v, g, m._packed, True, d_bwd, scales.get(id(m)), reuse=True)

    # everything the launch sequence depends on besides the weights' values: the layers, their descriptors, the parameter storage,
    # the buffers the launches write, and WHICH layers are stale (a layer skipped while capturing would never be rebuilt by the replays)
    def _addr(t):
        return 0 if t is None else t.data_ptr()

    sig = tuple((id(m), m._packed.last[1].batch, m._packed.last[1].l_in, m._packed.last[1].math, m._packed.last[2].math, params_of(m)[0].data_ptr(),
                 _addr(m._packed.wp_fwd), _addr(m._packed.wp_bwd), _addr(m._packed.scale), _addr(m._packed.norm),
                 m._packed.key != _pack_key(*params_of(m), m._packed.last[1], m._packed.last[2])) for m in todo) + (_storage_epoch.get(-1, 0),)
    with torch.cuda.stream(side), torch.no_grad():
        if _conv_prepack_graph.run(sig, body, side):
            for m in todo:   # replayed: the images are current, the cache keys are not
                spec, d, d_bwd = m._packed.last
                v, g = params_of(m)
                m._packed.key = _pack_key(v, g, d, d_bwd)
        ev = torch.cuda.Event()
        ev.record()
    _side["prepacked"] = ev


def join_prepack() -> None:
    """The current stream waits for the images `prepack` built (an event: later side-stream work is not waited for)."""
    ev = _side.get("prepacked")
    if ev is not None:
        _side["prepacked"] = None
        torch.cuda.current_stream().wait_event(ev)


#: transposed convs with a fused output activation: mask the gradient by one element-wise launch so that the bf16 weight-gradient
#: kernel applies (see weight_grads)
PREMASK_TRANSPOSED_DW = os.environ.get("EBEN_PREMASK_DW", "1") != "0"


def weight_grads(d: EbenConv1dDesc, dy: torch.Tensor, y: Optional[torch.Tensor], x: torch.Tensor, v: torch.Tensor,
                 g: Optional[torch.Tensor], bias: Optional[torch.Tensor], norm: Optional[torch.Tensor]):
    """Weight (+ bias) gradient of one conv layer: ``eben_conv1d_bwd_dw`` into split-K slabs, then the slab sum and the
    weight-norm chain rule.  dy: gradient at the layer output (``y``: the saved output when a fused output activation has to
    be differentiated, else None); x: the layer input; v / g / bias: the PARAMETERS (g, bias may be None); norm: ||v|| rows.
    Returns (dv, dg, dbias) -- or Nones for gradients that were deferred: inside ``weight_grads_on_side_stream()`` the work is
    issued on the side stream, and the results are either written straight into the sink's buckets (reported at ``join()``)
    or assigned to ``.grad`` by ``join()`` (never handed to autograd: it may clone what it is given, and the deferred
    kernels would then fill a tensor nobody reads)."""
    lib = load()
    has_g, has_bias = g is not None, bias is not None
    use_side, sunk = _wg_route(v, g, bias)
    premask = None
    if PREMASK_TRANSPOSED_DW and d.transposed and d.out_slope != 1.0 and y is not None and d.math in (MATH_BF16, MATH_BF16X2):
        # ConvTranspose1d with a fused output activation: the bf16 weight-gradient kernel takes no mask on that operand (the layer
        # would fall to the exact-fp32 kernel, 3-4x the time for the decoder's transposed convs) -- the masked gradient is formed by
        # one element-wise launch on the stream the weight gradient runs on, and the layer handed over as one without activation
        plain = getattr(d, "_dw_plain", None)
        if plain is None:
            plain = d._dw_plain = type(d).from_buffer_copy(d)
            plain.out_slope = 1.0
        premask, d = (dy, y, d.out_slope), plain
    ws = getattr(d, "_dw_ws", None)   # (bytes, slabs, row stride) of this descriptor: asked once
    if ws is None:
        nslab, row_stride = ctypes.c_int(0), ctypes.c_int(0)
        ws = d._dw_ws = (lib.eben_conv1d_bwd_dw_workspace(ctypes.byref(d), ctypes.byref(nslab), ctypes.byref(row_stride)), nslab.value, row_stride.value)
    ws_bytes, nslab, row_stride = ws
    # Every buffer is allocated on the CURRENT stream's pool; on the deferred path the kernels are launched on the side stream through
    # its raw handle (no torch stream switch) and everything they touch stays referenced until join(), after which the current
    # stream -- which waits for the side stream there -- may reuse it.
    slabs = _empty(ws_bytes, x)
    job, outs = _wg_job(slabs, nslab, row_stride, v, g, bias, norm, sunk, x.device)
    raw = None
    if use_side:
        side = _side_stream(x.device, deal=True)
        side.wait_stream(torch.cuda.current_stream(x.device))   # dy (and everything saved by the forward) is complete on the main stream
        raw = side.cuda_stream
    if premask is not None:
        gm = torch.empty_like(dy)
        check(lib.eben_lrelu_bwd(ptr(dy), ptr(y), ptr(gm), dy.numel(), premask[2], raw if use_side else stream()), "lrelu_bwd")
        dy, y = gm, None
    if use_side:
        check(lib.eben_conv1d_bwd_dw(ctypes.byref(d), ptr(dy), ptr(y), ptr(x), 1 if has_bias else 0, ptr(slabs), ws_bytes, raw), "conv1d_bwd_dw")
        _side["keep"].append((dy, x, y, norm, slabs) + (premask[:2] if premask is not None else ()))
        _wg_defer(job, outs, v, g, bias, sunk)
        return None, None, None
    check(lib.eben_conv1d_bwd_dw(ctypes.byref(d), ptr(dy), ptr(y), ptr(x), 1 if has_bias else 0, ptr(slabs), ws_bytes, stream()), "conv1d_bwd_dw")
    if _wn_collect[0] is not None:
        _wn_collect[0].append(job)
    else:
        wn_bwd_multi([job])
    return outs


_wn_collect = [None, None]   # inside collect_wn_jobs(): the list the immediate path appends its slab-sum / weight-norm jobs to, the sink


class collect_wn_jobs:
    """Inside this context the immediate path of ``weight_grads`` / ``weight_grads_ru`` (no side-stream deferral: the caller has made
    the stream the work belongs on current) launches the weight-gradient kernels only and appends the slab-sum + weight-norm job of
    each layer to ``jobs``: the caller runs them as ONE ``wn_bwd_multi(jobs)`` (gen_engine's graph-captured weight-gradient body)."""

    def __init__(self, jobs: list, sink=None):
        self.jobs, self.sink = jobs, sink   # sink: results go straight into its gradient buffers (``grad_buffer(param)``)

    def __enter__(self):
        self.prev = (_wn_collect[0], _wn_collect[1], _side["enabled"])
        _wn_collect[0], _wn_collect[1], _side["enabled"] = self.jobs, self.sink, False
        return self

    def __exit__(self, *exc):
        _wn_collect[0], _wn_collect[1], _side["enabled"] = self.prev


def _wg_route(v, g, bias):
    """Where a layer's weight gradients go: (issue on the side stream?, bucket views of a data-parallel sink or None)."""
    has_g, has_bias = g is not None, bias is not None
    is_param = isinstance(v, torch.nn.Parameter)
    use_side = (_side["enabled"] and is_param and v.grad is None and (g is None or g.grad is None) and (bias is None or bias.grad is None)
                and _no_grad_hooks(v) and _no_grad_hooks(g) and _no_grad_hooks(bias))
    sunk = None
    sk = _side["sink"] if _side["enabled"] and not use_side else (_wn_collect[1] if _wn_collect[0] is not None else None)
    if sk is not None:
        # data-parallel run: p.grad is a view of a gradient bucket -- write the results there directly
        sunk = (sk.grad_buffer(v), sk.grad_buffer(g) if has_g else None, sk.grad_buffer(bias) if has_bias else None)
        if sunk[0] is None or (has_g and sunk[1] is None) or (has_bias and sunk[2] is None):
            sunk = None
        use_side = sunk is not None and _wn_collect[0] is None   # collecting: the caller has made the producing stream current
    return use_side, sunk


def _wg_job(slabs, nslab, row_stride, v, g, bias, norm, sunk, device):
    """The ``eben_wn_bwd`` job that sums a layer's split-K slabs and applies the weight-norm chain rule, and its outputs."""
    has_g, has_bias = g is not None, bias is not None
    rows = v.shape[0]
    cols = v.numel() // rows
    if sunk is not None:
        dv, dg, dbias = sunk
    else:
        dv = torch.empty_like(v)
        dg = torch.empty_like(g) if has_g else None
        dbias = torch.empty(rows, dtype=torch.float32, device=device) if has_bias else None
    job = (slabs, nslab, rows * row_stride, rows, cols, row_stride, g.detach() if has_g else None, v.detach(), norm if has_g else None, dg, dv, dbias)
    return job, (dv, dg, dbias)


def _wg_defer(job, outs, v, g, bias, sunk) -> None:
    """Deferred path: the slab sums and the weight-norm chain rule of ALL layers are one multi-tensor launch at join()."""
    _side["wn_jobs"].append(job)
    if sunk is None:
        _side["assign"].extend((p, t) for p, t in zip((v, g, bias), outs) if p is not None and t is not None)
    else:
        _side["sunk"].extend(p for p in (v, g, bias) if p is not None)


def weight_grads_ru(math: int, dilation: int, gy: torch.Tensor, u: torch.Tensor, out_slope: float, h: torch.Tensor, gh: torch.Tensor,
                    x: torch.Tensor, in_slope: float, pw_params, dil_params):
    """Both weight gradients of a fused ResidualUnit in one launch (``eben_ru_dw``: the reduction runs along time, no packing
    pass): ``*_params`` = (v, g, norm) of the pointwise / dilated conv (weight-normalised, no bias).  Routed like ``weight_grads``
    (side stream / gradient buckets / immediate); returns False when the two layers would be routed differently (the caller then
    takes the per-layer path), else a pair of (dv, dg, None) results -- Nones where deferred."""
    lib = load()
    (vp, gp, np_), (vd, gd, nd) = pw_params, dil_params
    route_p, route_d = _wg_route(vp, gp, None), _wg_route(vd, gd, None)
    if route_p[0] != route_d[0] or (route_p[1] is None) != (route_d[1] is None):
        return False
    use_side = route_p[0]
    b, c, l = gy.shape
    nslab = lib.eben_ru_dw_slabs(b, c, l)
    slabs_p = torch.empty(nslab * c * c, dtype=torch.float32, device=gy.device)
    slabs_d = torch.empty(nslab * c * 3 * c, dtype=torch.float32, device=gy.device)
    job_p, outs_p = _wg_job(slabs_p, nslab, c, vp, gp, None, np_, route_p[1], gy.device)
    job_d, outs_d = _wg_job(slabs_d, nslab, 3 * c, vd, gd, None, nd, route_d[1], gy.device)
    if use_side:
        side = _side_stream(gy.device, deal=True)
        side.wait_stream(torch.cuda.current_stream(gy.device))
        st = side.cuda_stream
    else:
        st = stream()
    check(lib.eben_ru_dw(math, b, c, l, dilation, ptr(gy), ptr(u), float(out_slope), ptr(h), ptr(gh), ptr(x), float(in_slope), ptr(slabs_p), ptr(slabs_d), st),
          "ru_dw")
    if use_side:
        _side["keep"].append((gy, u, h, gh, x, np_, nd, slabs_p, slabs_d))
        _wg_defer(job_p, outs_p, vp, gp, None, route_p[1])
        _wg_defer(job_d, outs_d, vd, gd, None, route_d[1])
        return (None, None, None), (None, None, None)
    if _wn_collect[0] is not None:
        _wn_collect[0].extend((job_p, job_d))
    else:
        wn_bwd_multi([job_p, job_d])
    return outs_p, outs_d


def weight_grads_ru_bl(dilation: int, gzb: torch.Tensor, hb: torch.Tensor, ghb: torch.Tensor, xb: torch.Tensor, pw_params, dil_params):
    """``weight_grads_ru`` on the four bf16 bundle planes of a unit (``eben_rubl_dw``, csrc/ru_bl.hip): g_z and g_h as written by
    ``eben_rubl_bwd``, h and xin as saved by ``eben_rubl_fwd`` -- planes of shape (batch, C / 8, L, 8)."""
    lib = load()
    (vp, gp, np_), (vd, gd, nd) = pw_params, dil_params
    route_p, route_d = _wg_route(vp, gp, None), _wg_route(vd, gd, None)
    if route_p[0] != route_d[0] or (route_p[1] is None) != (route_d[1] is None):
        return False
    use_side = route_p[0]
    b, cb, l, _ = gzb.shape
    c = 8 * cb
    nslab = lib.eben_rubl_dw_slabs(b, c, l)
    slabs_p = torch.empty(nslab * c * c, dtype=torch.float32, device=gzb.device)
    slabs_d = torch.empty(nslab * c * 3 * c, dtype=torch.float32, device=gzb.device)
    job_p, outs_p = _wg_job(slabs_p, nslab, c, vp, gp, None, np_, route_p[1], gzb.device)
    job_d, outs_d = _wg_job(slabs_d, nslab, 3 * c, vd, gd, None, nd, route_d[1], gzb.device)
    if use_side:
        side = _side_stream(gzb.device, deal=True)
        side.wait_stream(torch.cuda.current_stream(gzb.device))
        st = side.cuda_stream
    else:
        st = stream()
    check(lib.eben_rubl_dw(b, c, l, dilation, gzb.data_ptr(), hb.data_ptr(), ghb.data_ptr(), xb.data_ptr(), ptr(slabs_p), ptr(slabs_d), st), "rubl_dw")
    if use_side:
        _side["keep"].append((gzb, hb, ghb, xb, np_, nd, slabs_p, slabs_d))
        _wg_defer(job_p, outs_p, vp, gp, None, route_p[1])
        _wg_defer(job_d, outs_d, vd, gd, None, route_d[1])
        return (None, None, None), (None, None, None)
    if _wn_collect[0] is not None:
        _wn_collect[0].extend((job_p, job_d))
    else:
        wn_bwd_multi([job_p, job_d])
    return outs_p, outs_d


class _ConvLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, v, g, bias, spec: ConvSpec, cache):
        lib = load()
        x = x.contiguous()
        b, c, l_in = x.shape
        if c != spec.c_in:
            raise _lib.EbenError(f"conv expects {spec.c_in} input channels, got {c}")
        d = conv_desc(spec, b, l_in)
        d_bwd = conv_desc(spec, b, l_in, _backward_math[0]) if _backward_math[0] != MATH_F32 else d
        need_dx = ctx.needs_input_grad[0]
        pw = pack_weights(spec, d, v.detach(), None if g is None else g.detach(), cache, need_dx, d_bwd)
        y = torch.empty((b, spec.c_out, d.l_out), dtype=torch.float32, device=x.device)
        tm = kernel_timer_for(spec, "fwd")
        e0 = tm.start() if tm is not None else None
        check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(x), ptr(pw.wp_fwd), ptr(bias), None, ptr(y), stream()), "conv1d_fwd")
        if tm is not None:
            tm.stop(e0, b)
        ctx.spec, ctx.d = spec, d_bwd   # the descriptor the backward launches use
        ctx.wp_bwd, ctx.norm = pw.wp_bwd, pw.norm
        ctx.has_g, ctx.has_bias = g is not None, bias is not None
        # identities only: the gradient sink looks its bucket views up by parameter, join() assigns .grad to them
        ctx.bias_param = bias
        ctx.v_param = v if isinstance(v, torch.nn.Parameter) else None
        ctx.g_param = g if isinstance(g, torch.nn.Parameter) else None
        ctx.save_for_backward(x, v, g, y if spec.out_slope != 1.0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        x, v, g, y = ctx.saved_tensors
        spec, d = ctx.spec, ctx.d
        dy = dy.contiguous()
        st = stream()
        dx = dv = dg = dbias = None
        if ctx.needs_input_grad[0] and not _skip_input_grads[0]:
            ws_bytes = lib.eben_conv1d_bwd_dx_workspace(ctypes.byref(d))
            ws = _empty(ws_bytes, x) if ws_bytes else None
            dx = torch.empty_like(x)
            check(lib.eben_conv1d_bwd_dx(ctypes.byref(d), ptr(dy), ptr(y), ptr(ctx.wp_bwd), ptr(x), ptr(dx), 0, ptr(ws), ws_bytes, st),
                  "conv1d_bwd_dx")
        want_w = ctx.needs_input_grad[1] or (ctx.has_g and ctx.needs_input_grad[2]) or (ctx.has_bias and ctx.needs_input_grad[3])
        if want_w and not _skip_weight_grads[0]:
            dv, dg, dbias = weight_grads(d, dy, y, x, ctx.v_param if ctx.v_param is not None else v,
                                         (ctx.g_param if ctx.g_param is not None else g) if ctx.has_g else None,
                                         ctx.bias_param if ctx.has_bias else None, ctx.norm)
        return dx, dv, dg, dbias, None, None


def conv_layer(x: torch.Tensor, v: torch.Tensor, g: Optional[torch.Tensor], bias: Optional[torch.Tensor], spec: ConvSpec,
               cache: Optional[PackedWeights] = None) -> torch.Tensor:
    """lrelu_out(conv(lrelu_in(x); weight_norm(g, v)) + bias) with full autograd support."""
    return _ConvLayerFn.apply(x, v, g, bias, spec, cache)


# --------------------------------------------------------------------------------------------
# elementwise
# --------------------------------------------------------------------------------------------
class _LeakyReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope: float):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(load().eben_lrelu_fwd(ptr(x), ptr(y), x.numel(), slope, stream()), "lrelu_fwd")
        ctx.slope = slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(load().eben_lrelu_bwd(ptr(dy), ptr(y), ptr(dx), dy.numel(), ctx.slope, stream()), "lrelu_bwd")
        return dx, None


def leaky_relu(x: torch.Tensor, slope: float) -> torch.Tensor:
    return _LeakyReluFn.apply(x, slope)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        if a.shape != b.shape:
            raise _lib.EbenError(f"add: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty_like(a)
        check(load().eben_add(ptr(a), ptr(b), ptr(out), a.numel(), stream()), "add")
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout, dout


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _AddFn.apply(a, b)


class _TanhLiftFn(torch.autograd.Function):
    """tanh(x + cat(lift, 0)) -- eben_generator.py:203-208; ``lift`` carries no gradient."""

    @staticmethod
    def forward(ctx, x, lift):
        x, lift = x.contiguous(), lift.contiguous()
        b, c, l = x.shape
        out = torch.empty_like(x)
        check(load().eben_tanh_lift_fwd(ptr(x), ptr(lift), ptr(out), b, c, lift.shape[1], l, stream()), "tanh_lift_fwd")
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.empty_like(out)
        check(load().eben_tanh_bwd(ptr(dout), ptr(out), ptr(dx), out.numel(), stream()), "tanh_bwd")
        return dx, None


def tanh_lift(x: torch.Tensor, lift: torch.Tensor) -> torch.Tensor:
    return _TanhLiftFn.apply(x, lift.detach())


class _ReflectPadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad_l: int, pad_r: int):
        x = x.contiguous()
        b, c, l = x.shape
        y = torch.empty((b, c, l + pad_l + pad_r), dtype=torch.float32, device=x.device)
        check(load().eben_reflect_pad_fwd(ptr(x), ptr(y), b * c, l, pad_l, pad_r, stream()), "reflect_pad_fwd")
        ctx.geom = (b, c, l, pad_l, pad_r)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, c, l, pad_l, pad_r = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty((b, c, l), dtype=torch.float32, device=dy.device)
        check(load().eben_reflect_pad_bwd(ptr(dy), ptr(dx), b * c, l, pad_l, pad_r, stream()), "reflect_pad_bwd")
        return dx, None, None


def reflect_pad(x: torch.Tensor, pad_l: int, pad_r: int) -> torch.Tensor:
    return _ReflectPadFn.apply(x, pad_l, pad_r)


# --------------------------------------------------------------------------------------------
# FIR banks (PQMF, A-weighting)
# --------------------------------------------------------------------------------------------
def _fir_decimate(x, w, ly, bands, ntaps, stride, off0):
    b, _, lx = x.shape
    y = torch.empty((b, bands, ly), dtype=torch.float32, device=x.device)
    check(load().eben_fir_decimate(ptr(x), ptr(w), ptr(y), b, lx, ly, bands, ntaps, stride, off0, stream()), "fir_decimate")
    return y


def _fir_interp_sum(y, w, lx, bands, ntaps, stride, off0):
    b, _, ly = y.shape
    x = torch.empty((b, 1, lx), dtype=torch.float32, device=y.device)
    check(load().eben_fir_interp_sum(ptr(y), ptr(w), ptr(x), b, lx, ly, bands, ntaps, stride, off0, stream()), "fir_interp_sum")
    return x


class _FirDecimateFn(torch.autograd.Function):
    """y[b,k,t] = sum_j w[k,j] x[b,0,t*stride+off0+j]  (pqmf.py:194-202 with off0 = -(N-1))."""

    @staticmethod
    def forward(ctx, x, w, ly: int, stride: int, off0: int):
        x, w = x.contiguous(), w.contiguous()
        bands, ntaps = w.shape[0], w.shape[-1]
        ctx.geom = (x.shape[2], bands, ntaps, stride, off0)
        ctx.save_for_backward(w)
        return _fir_decimate(x, w, ly, bands, ntaps, stride, off0)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        lx, bands, ntaps, stride, off0 = ctx.geom
        dx = _fir_interp_sum(dy.contiguous(), w, lx, bands, ntaps, stride, off0) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None


class _FirInterpSumFn(torch.autograd.Function):
    """x[b,0,u] = sum_k sum_{t*stride+off0+j=u} w[k,j] y[b,k,t]  (pqmf.py:204-213 + band sum)."""

    @staticmethod
    def forward(ctx, y, w, lx: int, stride: int, off0: int):
        y, w = y.contiguous(), w.contiguous()
        bands, ntaps = w.shape[0], w.shape[-1]
        ctx.geom = (y.shape[2], bands, ntaps, stride, off0)
        ctx.save_for_backward(w)
        return _fir_interp_sum(y, w, lx, bands, ntaps, stride, off0)

    @staticmethod
    def backward(ctx, dx):
        (w,) = ctx.saved_tensors
        ly, bands, ntaps, stride, off0 = ctx.geom
        dy = _fir_decimate(dx.contiguous(), w, ly, bands, ntaps, stride, off0) if ctx.needs_input_grad[0] else None
        return dy, None, None, None, None


def fir_decimate(x, w, ly, stride, off0):
    return _FirDecimateFn.apply(x, w, ly, stride, off0)


def fir_interp_sum(y, w, lx, stride, off0):
    return _FirInterpSumFn.apply(y, w, lx, stride, off0)


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
    return arr


class _FeatureLossFn(torch.autograd.Function):
    """sum_i mean|a_i-b_i| / mean|a_i| * inv_count  (feature_loss.py:37-50)."""

    @staticmethod
    def forward(ctx, inv_count: float, n_pairs: int, *tensors):
        lib = load()
        a = [t.contiguous() for t in tensors[:n_pairs]]
        b = [t.contiguous() for t in tensors[n_pairs:]]
        for ta, tb in zip(a, b):
            if ta.shape != tb.shape:
                raise _lib.EbenError(f"feature loss: shape mismatch {tuple(ta.shape)} vs {tuple(tb.shape)}")
        inter = [None] * (2 * n_pairs)
        inter[0::2], inter[1::2] = a, b
        ptrs = _ptr_array(inter)
        numel = (ctypes.c_int64 * n_pairs)(*[t.numel() for t in a])
        dev = a[0].device
        ws_bytes = lib.eben_fm_sums_workspace(n_pairs)
        ws = _empty(ws_bytes, a[0])
        sums = torch.empty(2 * n_pairs, dtype=torch.float32, device=dev)
        check(lib.eben_fm_sums(ptrs, numel, n_pairs, ptr(ws), ws_bytes, ptr(sums), stream()), "fm_sums")
        ctx.inv_count, ctx.n_pairs = inv_count, n_pairs
        ctx.save_for_backward(sums, *a, *b)
        return (sums[0::2] / sums[1::2]).sum() * inv_count

    @staticmethod
    def backward(ctx, gout):
        lib = load()
        n = ctx.n_pairs
        sums = ctx.saved_tensors[0]
        a, b = ctx.saved_tensors[1 : 1 + n], ctx.saved_tensors[1 + n :]
        inter = [None] * (2 * n)
        inter[0::2], inter[1::2] = a, b
        da = [torch.empty_like(t) for t in a]
        gout = gout.contiguous().reshape(1)
        check(lib.eben_fm_bwd(_ptr_array(inter), _ptr_array(da), (ctypes.c_int64 * n)(*[t.numel() for t in a]), n, ptr(sums),
                              ptr(gout), ctx.inv_count, stream()), "fm_bwd")
        return (None, None, *da, *([None] * n))


def last_conv_grad_norms(seeds: Sequence[torch.Tensor], bands: torch.Tensor, pre: torch.Tensor, conv) -> Optional[List[torch.Tensor]]:
    """||d L_i / d conv.weight|| for the seeds s_i = d L_i / d bands of the balancing losses in one pass over ``pre`` (``eben_last_conv_norms``:
    bands = tanh(conv(pre) + lift), eben.py:222-229 / eben_generator.py:159-166, 203-208); None when ``conv`` is not the layer that kernel is
    built for (then the caller differentiates through autograd).  Returns 0-dim views of one device vector."""
    sp = getattr(conv, "spec", None)
    if (sp is None or getattr(conv, "weight_norm", True) or conv.bias is not None or not (1 <= len(seeds) <= 4)
            or (sp.c_in, sp.c_out, sp.ksize, sp.stride, sp.dilation, sp.groups, sp.pad_l, sp.pad_r) != (32, 4, 3, 1, 1, 1, 1, 1) or not sp.reflect
            or sp.in_slope != 1.0 or sp.out_slope != 1.0 or pre.dtype is not torch.float32 or not pre.is_cuda):
        return None
    b, _, l = pre.shape
    lib = load()
    ts = [t.contiguous() for t in seeds]
    bands_c, pre_c = bands.detach().contiguous(), pre.detach().contiguous()
    nbytes = lib.eben_last_conv_norms_workspace(b)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=pre.device)
    out = torch.empty(len(ts), dtype=torch.float32, device=pre.device)
    check(lib.eben_last_conv_norms(_ptr_array(ts), len(ts), ptr(bands_c), ptr(pre_c), b, sp.c_in, sp.c_out, l, sp.ksize, sp.pad_l, ptr(ws), nbytes, ptr(out),
                                   stream()), "last_conv_norms")
    return [out[i] for i in range(len(ts))]


def weighted_sum(tensors: Sequence[torch.Tensor], weights: torch.Tensor) -> torch.Tensor:
    """sum_i weights[i] * tensors[i] (weights: device vector), torch's order and roundings, one launch (``eben_weighted_sum``)."""
    ts = [t.contiguous() for t in tensors]
    out = torch.empty_like(ts[0])
    check(load().eben_weighted_sum(_ptr_array(ts), ptr(weights), len(ts), out.numel(), ptr(out), stream()), "weighted_sum")
    return out


def feature_loss(emb_a: List[List[torch.Tensor]], emb_b: List[List[torch.Tensor]]) -> torch.Tensor:
    a = [t for scale in emb_a for t in scale[1:-1]]
    b = [t.detach() for scale in emb_b for t in scale[1:-1]]
    inv = 1.0 / (len(emb_a) * len(emb_a[-1][1:-1]))
    return _FeatureLossFn.apply(inv, len(a), *a, *b)


class _HingeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target: float):
        x = x.contiguous()
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        check(load().eben_hinge_fwd(ptr(x), x.numel(), float(target), ptr(out), stream()), "hinge_fwd")
        ctx.target = float(target)
        ctx.save_for_backward(x)
        return out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        gout = gout.contiguous().reshape(1)
        check(load().eben_hinge_bwd(ptr(x), x.numel(), ctx.target, ptr(gout), 1.0, ptr(dx), stream()), "hinge_bwd")
        return dx, None


def hinge_mean(x: torch.Tensor, target: float) -> torch.Tensor:
    """mean(relu(1 - target*x)) -- hinge_loss.py:41."""
    return _HingeFn.apply(x, target)


@dataclass
class StftPlan:
    n_fft: int
    hop: int
    win: int
    bins: int
    pad: int
    spec_f: ConvSpec  # pointwise conv (win -> 2*bins): the windowed DFT of a frame matrix
    basis_f: torch.Tensor  # (2*bins, win, 1) windowed DFT rows: [cos ; -sin]
    spec_t: ConvSpec  # pointwise conv (2*bins -> win) used by the backward
    basis_t: torch.Tensor  # (win, 2*bins, 1) = basis transposed
    cache_fwd: PackedWeights
    cache_bwd: PackedWeights
    #: "dense": one (2*bins x win) GEMM per resolution; "folded": even / odd parts of the frames, two groups of win/2
    #: channels (half the products, exact fp32); "bf16x3": the folded form with hi / lo bf16 operand splits on the bf16 MFMA
    math: str = "folded"
    folded: Optional[dict] = None   # math -> (spec_f, basis_f, spec_t, basis_t, cache_fwd, cache_bwd), built on first use
    gemm: Optional[dict] = None     # (math, which) -> (basis address, packed image, basis) for the grouped GEMM kernel

    def gemm_image(self, math: str, which: int, basis: torch.Tensor) -> torch.Tensor:
        """Packed basis of the folded contraction for ``eben_gemm_fwd``: which = 0 the forward's (2 groups x bins x win/2), 1 its
        transpose's (2 groups x win/2 x bins); built once per (math, basis storage) -- the bases are constants."""
        if self.gemm is None:
            self.gemm = {}
        key = (math, which)
        hit = self.gemm.get(key)
        if hit is None or hit[0] != basis.data_ptr():
            lib = load()
            h, cmath = self.win // 2, _STFT_CONV_MATH[math]
            m, k = (self.bins, h) if which == 0 else (h, self.bins)
            wp = torch.empty(lib.eben_gemm_packed_floats(cmath, 2, m, k), dtype=torch.float32, device=basis.device)
            check(lib.eben_gemm_pack(cmath, 2, m, k, ptr(basis), ptr(wp), stream()), "gemm_pack")
            hit = self.gemm[key] = (basis.data_ptr(), wp, basis)
        return hit[1]

    def folded_parts(self, math: str):
        """Weights of the folded pointwise convs (eben_stft_frames_folded): forward, 2 groups x (nsub*h -> bins) with
        rows basis[:, h:] (x [W_hi, W_hi, W_lo] for bf16x3); backward, 2 groups x (nsub*bins -> h), its transpose."""
        if self.folded is None:
            self.folded = {}
        parts = self.folded.get(math)
        if parts is None:
            h, bins = self.win // 2, self.bins
            w = self.basis_f[:, h:, 0].contiguous()                       # (2*bins, h)
            wt = w.reshape(2, bins, h).transpose(1, 2).contiguous()       # (2, h, bins): group g, output row m, reduction k
            if math == "bf16x3":
                def split(a, dim):
                    hi = a.to(torch.bfloat16).to(torch.float32)
                    lo = (a - hi).to(torch.bfloat16).to(torch.float32)
                    return torch.cat((hi, hi, lo), dim=dim)
                w, wt, nsub = split(w, 1), split(wt, 2), 3
            else:
                nsub = 1
            parts = (ConvSpec(c_in=2 * nsub * h, c_out=2 * bins, ksize=1, groups=2), w.unsqueeze(-1).contiguous(),
                     ConvSpec(c_in=2 * nsub * bins, c_out=2 * h, ksize=1, groups=2), wt.reshape(2 * h, nsub * bins, 1).contiguous(),
                     PackedWeights(), PackedWeights())
            self.folded[math] = parts
        return parts


#: StftPlan.math -> EBEN_MATH_* of the folded windowed-DFT contractions ("bf16x3": single bf16 operands over the concatenated
#: [hi ; lo ; hi] x [W_hi ; W_hi ; W_lo] reduction; "folded_x3" / "folded_x6": the tap-conv's own split operands, tapconv3.hip)
_STFT_CONV_MATH = {"bf16x3": MATH_BF16, "folded_x3": MATH_BF16X3, "folded_x6": MATH_BF16X6}


#: loss values from their partial sums by one launch each (eben_stft_loss_total, eben_disc_losses) instead of one-element torch kernels
FUSED_LOSS_GLUE = os.environ.get("EBEN_FUSED_LOSS_GLUE", "1") != "0"
#: the folded windowed-DFT contractions on the grouped GEMM kernel (pw_gemm.hip) instead of the pointwise tap-conv
STFT_GEMM = os.environ.get("EBEN_STFT_GEMM", "1") != "0"


class _MRSTFTFn(torch.autograd.Function):
    """auraloss MultiResolutionSTFTLoss(x, y) as configured by multi_stft.yaml (see mrstft_loss.py).

    Per resolution: ``eben_stft_frames`` writes the reflect-padded frames of ALL signals (x and y rows) as one
    (win, R*frames) matrix, the windowed DFT is one dense GEMM over it (pointwise tap-conv, batch 1 -- a strided conv
    per item would pad every item's 134 / 267 frames to whole 128-column tiles), the loss sums read the flat
    (2*bins, R*frames) spectrum through strides; backward: d(spec) in the same flat form, d(frames) = basis^T . d(spec)
    as one GEMM, overlap-add back onto the waveform.  ``plan.math`` "folded" / "bf16x3" (even window lengths): the frames
    are written as their even and odd parts about the window centre and the GEMMs become 2-group contractions over
    win/2 channels (``StftPlan.folded_parts``)."""

    @staticmethod
    def forward(ctx, x, y, fir, plans: List[StftPlan], eps: float):
        lib = load()
        st = stream()
        x, y = x.contiguous(), y.contiguous()
        b, c, t = x.shape
        rows = b * c
        sig = torch.cat((x.reshape(rows, 1, t), y.reshape(rows, 1, t)), dim=0)
        if fir is not None:
            nt = fir.numel()
            sig = _fir_decimate(sig, fir, t, 1, nt, 1, -(nt // 2))
        total = None
        saved = []
        for p in plans:
            frames = (t + 2 * p.pad - p.win) // p.hop + 1
            cols = 2 * rows * frames
            math = p.math if p.win % 2 == 0 and p.pad == p.win // 2 else "dense"
            if math == "dense":
                fr = torch.empty((1, p.win, cols), dtype=torch.float32, device=x.device)
                check(lib.eben_stft_frames(ptr(sig), ptr(fr), 2 * rows, t, p.win, p.hop, p.pad, frames, st), "stft_frames")
                spec_f, basis_f, cache, cmath = p.spec_f, p.basis_f, p.cache_fwd, MATH_F32
            else:
                spec_f, basis_f, _, _, cache, _ = p.folded_parts(math)
                cmath = _STFT_CONV_MATH.get(math, MATH_F32)
                fr = torch.empty((1, spec_f.c_in, cols), dtype=torch.float32, device=x.device)
                check(lib.eben_stft_frames_folded(ptr(sig), ptr(fr), 2 * rows, t, p.win, p.hop, p.pad, frames, 1 if math == "bf16x3" else 0, st),
                      "stft_frames_folded")
            spec = torch.empty((1, 2 * p.bins, cols), dtype=torch.float32, device=x.device)
            if STFT_GEMM and math in ("folded_x3", "folded_x6"):
                # the folded contraction as what it is, a 2-group GEMM (pw_gemm.hip): no input tile, no staging pass per row tile
                check(lib.eben_gemm_fwd(cmath, 2, p.bins, p.win // 2, cols, ptr(fr), ptr(p.gemm_image(math, 0, basis_f)), ptr(spec), st), "stft_fwd")
            else:
                d2 = conv_desc(spec_f, 1, cols, cmath)
                pw = pack_weights(spec_f, d2, basis_f, None, cache, False)
                check(lib.eben_conv1d_fwd(ctypes.byref(d2), ptr(fr), ptr(pw.wp_fwd), None, None, ptr(spec), st), "stft_fwd")
            sums = torch.empty((rows, 3), dtype=torch.float32, device=x.device)
            # y rows: same strides, column offset rows*frames
            ws_bytes = lib.eben_stft_loss_sums_workspace(rows)
            check(lib.eben_stft_loss_sums_ex(ptr(spec), ptr(spec) + 4 * rows * frames, rows, p.bins, frames, frames, cols, p.bins * cols, eps,
                                             ptr(_empty(ws_bytes, x)), ws_bytes, ptr(sums), st), "stft_loss_sums")
            saved.append((spec, sums, frames, math))
        ctx.plans, ctx.eps, ctx.geom, ctx.saved, ctx.fir = plans, eps, (b, c, t, rows), saved, fir
        n = len(plans)
        if n <= 8 and FUSED_LOSS_GLUE:   # the value from the per-row sums of all resolutions: one launch instead of ~8 one-element torch kernels per resolution
            total = torch.empty((), dtype=torch.float32, device=x.device)
            check(lib.eben_stft_loss_total((ctypes.c_void_p * n)(*[ptr(sv[1]) for sv in saved]),
                                           (ctypes.c_float * n)(*[1.0 / float(rows * p.bins * sv[2]) for p, sv in zip(plans, saved)]), n, rows,
                                           ptr(total), st), "stft_loss_total")
            return total
        for p, (_, sums, frames, _) in zip(plans, saved):
            term = torch.sqrt(sums[:, 0] / sums[:, 1]).mean() + sums[:, 2].sum() / float(rows * p.bins * frames)
            total = term if total is None else total + term
        return total / n

    @staticmethod
    def backward(ctx, gout):
        lib = load()
        st = stream()
        b, c, t, rows = ctx.geom
        gout = gout.contiguous().reshape(1)
        dsig = torch.empty((rows, 1, t), dtype=torch.float32, device=gout.device)
        for i, (p, (spec, sums, frames, math)) in enumerate(zip(ctx.plans, ctx.saved)):
            cols, xcols = 2 * rows * frames, rows * frames
            dspec = torch.empty((1, 2 * p.bins, xcols), dtype=torch.float32, device=gout.device)
            check(lib.eben_stft_loss_bwd_ex(ptr(spec), ptr(spec) + 4 * xcols, rows, p.bins, frames, frames, cols, p.bins * cols, ctx.eps,
                                            ptr(sums), ptr(gout), 1.0 / len(ctx.plans), ptr(dspec), frames, xcols, p.bins * xcols, st),
                  "stft_loss_bwd")
            # d(frames)[j, (r, f)] = sum_m basis[m, j] dspec[m, (r, f)]: one dense GEMM, then overlap-add
            if math == "dense":
                spec_t, basis_t, cache, cmath = p.spec_t, p.basis_t, p.cache_bwd, MATH_F32
            else:
                _, _, spec_t, basis_t, _, cache = p.folded_parts(math)
                cmath = _STFT_CONV_MATH.get(math, MATH_F32)
                if math == "bf16x3":
                    dsplit = torch.empty((1, spec_t.c_in, xcols), dtype=torch.float32, device=gout.device)
                    check(lib.eben_split3(ptr(dspec), ptr(dsplit), 2, p.bins, xcols, st), "split3")
                    dspec = dsplit
            dfr = torch.empty((1, spec_t.c_out, xcols), dtype=torch.float32, device=gout.device)
            if STFT_GEMM and math in ("folded_x3", "folded_x6"):
                check(lib.eben_gemm_fwd(cmath, 2, p.win // 2, p.bins, xcols, ptr(dspec), ptr(p.gemm_image(math, 1, basis_t)), ptr(dfr), st), "stft_bwd_gemm")
            else:
                d1 = conv_desc(spec_t, 1, xcols, cmath)
                pw = pack_weights(spec_t, d1, basis_t, None, cache, False)
                check(lib.eben_conv1d_fwd(ctypes.byref(d1), ptr(dspec), ptr(pw.wp_fwd), None, None, ptr(dfr), st), "stft_bwd_gemm")
            if math == "dense":
                check(lib.eben_overlap_add_ex(ptr(dfr), ptr(dsig), rows, t, p.win, frames, p.hop, p.pad, 1, 1 if i else 0, frames, xcols, st),
                      "overlap_add")
            else:
                check(lib.eben_overlap_add_folded(ptr(dfr), ptr(dsig), rows, t, p.win, frames, p.hop, p.pad, 1 if i else 0, frames, xcols, st),
                      "overlap_add_folded")
        if ctx.fir is not None:
            nt = ctx.fir.numel()
            dsig = _fir_interp_sum(dsig, ctx.fir, t, 1, nt, 1, -(nt // 2))
        return dsig.reshape(b, c, t), None, None, None, None


def mrstft(x: torch.Tensor, y: torch.Tensor, fir: Optional[torch.Tensor], plans: List[StftPlan], eps: float = 1e-8) -> torch.Tensor:
    return _MRSTFTFn.apply(x, y.detach(), fir, plans, eps)
