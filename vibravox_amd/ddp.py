"""Data-parallel gradient exchange for the EBEN step: one process per GPU, RCCL over xGMI.

Replaces what Lightning's ``ddp_find_unused_parameters_true`` strategy does implicitly for the
reference (configs/trainer/ddp.yaml:7; gradients averaged inside ``manual_backward``,
eben.py:108,125).  Design for MI355X:
  * one ``GradSync`` per network (generator 7.8 MB, discriminator 92.6 MB of fp32 grads) -- the
    frozen network of a phase simply has no sync, so "unused parameter" detection is moot;
  * gradients live *in* flat bucket buffers (``p.grad`` is a view), buckets are filled in reverse
    parameter order (the order backward produces them) and each bucket's all-reduce is issued on a
    side HIP stream the moment its last gradient has been accumulated
    (``register_post_accumulate_grad_hook``), overlapping RCCL with the rest of backward;
  * the 1/world_size average is not a separate pass: ``finish()`` returns the factor and the fused
    Adam kernel applies it while reading the gradient;
  * balancing norms / EMA state stay rank-local, as in the reference (``torch.autograd.grad`` does
    not run DDP hooks).
Works on CPU tensors with the gloo backend (used by the world_size-2 tests).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


_comm_streams = {}   # device -> the ordering stream of the gradient exchanges


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], device, dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.buf = torch.zeros(self.numel, device=device, dtype=dtype)
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.buf[off : off + p.numel()].view_as(p))
            off += p.numel()
        self.pending = len(params)
        self.work = None
        self.events = []   # producer-stream events the exchange has to wait for (gradients written outside autograd)
        self.index = {id(p): i for i, p in enumerate(params)}
        self.idx = -1      # position in GradSync.buckets


class GradSync:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 32 << 20, group=None,
                 overlap: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameter to synchronise"
        dev, dtype = self.params[0].device, self.params[0].dtype
        self.on_gpu = dev.type == "cuda"
        if self.on_gpu:
            # The stream the exchanges are ORDERED on (torch's process group runs the collective on a stream of its own, behind whatever
            # stream is current at the call): a stream that carries nothing else, shared by the GradSyncs of the process.  With the six
            # hardware queues of a data-parallel rank (_env.py) it no longer has to ride on the side stream (generator weight
            # gradients, pre-packing), whose queued work an exchange then waited for; EBEN_COMM_STREAM=side restores that.
            import os
            from . import ops
            if os.environ.get("EBEN_COMM_STREAM", "own") == "side":
                self.comm_stream = ops.aux_stream(2, dev)
            else:
                if _comm_streams.get(dev) is None:
                    _comm_streams[dev] = torch.cuda.Stream(device=dev)
                self.comm_stream = _comm_streams[dev]
            # no graph capture starts while a bucket's exchange is in flight (ops.CaptureGate)
            ops.capture_gate.register_drain(self.comm_stream.synchronize)
        else:
            self.comm_stream = None
        self.overlap = overlap
        # reverse parameter order ~ the order in which backward finishes gradients
        self.buckets: List[_Bucket] = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self.buckets.append(_Bucket(cur, dev, dtype))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(_Bucket(cur, dev, dtype))
        self._owner = {}
        for i, b in enumerate(self.buckets):
            b.idx = i
            for p, v in zip(b.params, b.views):
                self._owner[id(p)] = b
                p.grad = v  # gradient-as-bucket-view: autograd accumulates in place
                p.register_post_accumulate_grad_hook(self._on_grad)
        self.launched = 0
        #: bucket indices in the order their exchanges were issued (identical on every rank); the last few steps' worth is kept
        self.order: List[int] = []
        self.profile = False
        self._pending_timing = []

    # -- hooks -------------------------------------------------------------------------------
    def grad_buffer(self, p: torch.nn.Parameter) -> Optional[torch.Tensor]:
        """The bucket view that holds p's gradient (contiguous, p's shape) -- a kernel may write the finished gradient
        there directly (then report it with ``mark_ready``) instead of handing a tensor to autograd."""
        b = self._owner.get(id(p))
        return None if b is None else b.views[b.index[id(p)]]

    def mark_ready(self, params: Iterable[torch.nn.Parameter]) -> None:
        """Gradients written straight into ``grad_buffer(p)`` by kernels enqueued on the CURRENT stream: the same accounting
        as the post-accumulate hook -- a bucket's all-reduce is issued when its last gradient is there.  May be called once
        per producer stream (the discriminator engine reports each sub-discriminator chain from that chain's stream, the
        largest layers first): a bucket filled from several streams waits for an event of each."""
        ev = None
        touched = []
        for p in params:
            b = self._owner.get(id(p))
            if b is None:
                continue
            p.grad = b.views[b.index[id(p)]]
            b.pending -= 1
            if b not in touched:
                touched.append(b)
        if self.on_gpu and touched:
            ev = torch.cuda.Event()
            ev.record()
        for b in touched:
            if ev is not None:
                b.events.append(ev)
            if b.pending == 0 and self.overlap:
                self._launch(b)

    def _on_grad(self, p: torch.nn.Parameter):
        b = self._owner[id(p)]
        view = b.views[b.index[id(p)]]
        if p.grad is not view:  # autograd replaced the tensor (first accumulation after set_to_none)
            view.copy_(p.grad)
            p.grad = view
        b.pending -= 1
        if b.pending == 0 and self.overlap:
            self._launch(b)

    def _launch(self, b: _Bucket):
        if self.world == 1 and not dist.is_initialized():
            b.events.clear()
            return
        self.order.append(b.idx)
        if len(self.order) > 16 * len(self.buckets):   # a training run issues these for ever: keep the recent ones
            del self.order[: len(self.order) - 8 * len(self.buckets)]
        if self.on_gpu:
            # the collective is ordered behind what produced the bucket: the current stream (autograd hooks, the last
            # mark_ready) and the recorded events of the other producer streams -- not behind unrelated work queued on the
            # side stream.  (torch's NCCL process group runs the collective on a stream of its own and orders it after the
            # stream that is current at the call.)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for ev in b.events:
                self.comm_stream.wait_event(ev)
            b.events.clear()
            with torch.cuda.stream(self.comm_stream):
                b.work = dist.all_reduce(b.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.launched += 1

    # -- step API ----------------------------------------------------------------------------
    def finish(self) -> float:
        """Wait for every bucket's all-reduce; returns the factor that turns the sums into means."""
        timing = self.on_gpu and self.profile
        if timing:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for b in self.buckets:
            if b.work is None and b.pending < len(b.params):
                self._launch(b)  # not launched from a hook (overlap disabled or partial bucket)
            if b.work is not None:
                b.work.wait()
                b.work = None
            b.pending = len(b.params)
            b.events.clear()
        if self.on_gpu and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if timing:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._pending_timing.append((e0, e1))
        return 1.0 / self.world

    def exposed_comm_ms(self) -> Optional[float]:
        """Mean main-stream wait per ``finish()`` since the last call (``profile = True``; synchronises)."""
        if not self._pending_timing:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._pending_timing]
        self._pending_timing = []
        return sum(ms) / len(ms)

    def zero(self):
        for b in self.buckets:
            b.buf.zero_()
            for p, v in zip(b.params, b.views):
                p.grad = v


class BucketedZeroGrad:
    """Wraps an optimizer so that ``zero_grad()`` keeps ``p.grad`` pointing into the buckets."""

    def __init__(self, optimizer: torch.optim.Optimizer, sync: GradSync):
        self.optimizer, self.sync = optimizer, sync

    def __getattr__(self, name):
        return getattr(self.optimizer, name)

    def step(self, *a, **k):
        return self.optimizer.step(*a, **k)

    def zero_grad(self, set_to_none: bool = True):
        self.sync.zero()


def all_reduce_scalars(values: List[torch.Tensor], group=None) -> List[torch.Tensor]:
    """The 7 ``self.log(..., sync_dist=True)`` reductions of one step (eben.py:103-124) packed into
    ONE all-reduce(mean) instead of seven."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return values
    packed = torch.stack([v.detach().reshape(()) for v in values])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    packed /= dist.get_world_size(group)
    return list(packed.unbind(0))
