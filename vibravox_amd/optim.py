"""Multi-tensor Adam on one HIP kernel launch per 48 tensors.

Numerically follows ``torch.optim.Adam`` (amsgrad=False, maximize=False) as configured by
``configs/lightning_module/optimizer/adam.yaml:1-9`` (lr 3e-4, betas (0.5, 0.9), eps 1e-8, wd 0);
keeps torch's ``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) so
optimizer checkpoints interchange with the reference's.
"""
from __future__ import annotations

import ctypes

import torch

from . import ops
from ._lib import EbenAdamTensor, check, load, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the EBEN configuration")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables = {}   # group index -> (parameter identities, largest numel, EbenAdamTensor table with the static columns filled)

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}   # Optimizer.__getstate__ keeps defaults / state / param_groups only (pickle, deepcopy)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}   # the restored moments are new tensors: the cached tables point at the old ones

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = load()
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            steps = set()
            for p in live:
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                steps.add(float(st["step"]))
            beta1, beta2 = group["betas"]
            if len(steps) == 1:
                # the usual case: every parameter of the group at the same step.  The table's static columns (parameter and
                # moment pointers, sizes) are kept from step to step; only the gradient pointers are refreshed (this loop runs
                # at the end of every train step with the GPU idle behind it)
                tables = self.__dict__.setdefault("_tables", {})
                cache = tables.get(gi)
                ident = tuple(id(p) for p in live)
                # (the moment tensors are replaced only through load_state_dict / __setstate__, which drop the tables)
                if cache is None or cache[0] != ident or any(p.data_ptr() != cache[2][i].param for i, p in enumerate(live)):
                    table = (EbenAdamTensor * len(live))()
                    for i, p in enumerate(live):
                        st = self.state[p]
                        table[i] = EbenAdamTensor(ptr(p.data), 0, ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), p.numel())
                    cache = tables[gi] = (ident, max(p.numel() for p in live), table)
                table = cache[2]
                keep = []
                for i, p in enumerate(live):
                    g = p.grad
                    if not g.is_contiguous() or g.dtype is not torch.float32 or not g.is_cuda:
                        g = g.contiguous()
                        keep.append(g)
                        table[i].grad = ptr(g)
                    else:
                        table[i].grad = g.data_ptr()
                check(lib.eben_adam_step(table, len(live), cache[1], group["lr"], beta1, beta2, group["eps"], group["weight_decay"],
                                         int(next(iter(steps))), grad_scale, stream()), "adam_step")
                continue
            by_step = {}
            for p in live:
                by_step.setdefault(int(self.state[p]["step"].item()), []).append(p)
            for step, plist in by_step.items():
                table = (EbenAdamTensor * len(plist))()
                grads = []
                for i, p in enumerate(plist):
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    grads.append(g)
                    st = self.state[p]
                    table[i] = EbenAdamTensor(ptr(p.data), ptr(g), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), p.numel())
                check(lib.eben_adam_step(table, len(plist), max(p.numel() for p in plist), group["lr"], beta1, beta2, group["eps"],
                                         group["weight_decay"], step, grad_scale, stream()), "adam_step")
        ops.bump_weights_epoch([p for group in self.param_groups for p in group["params"] if p.grad is not None])
        return loss
