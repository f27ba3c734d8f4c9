"""Hinge loss on HIP reductions -- drop-in for ``vibravox/torch_modules/losses/hinge_loss.py:6-43``."""
from __future__ import annotations

from typing import List

import torch

from ... import ops


class HingeLossForDiscriminatorMelganMultiScales(torch.nn.Module):
    def forward(self, embeddings: List[List[torch.Tensor]], target: float):
        total = None
        for scale in embeddings:
            term = ops.hinge_mean(scale[-1], target)
            total = term if total is None else total + term
        return total / len(embeddings)
