"""Feature-matching loss on HIP reductions -- drop-in for
``vibravox/torch_modules/losses/feature_loss.py:7-50`` (same class name, same forward signature,
same quirks: normalised by mean|a| of the FIRST argument and divided by
``len(embeddings_a) * len(last_scale[1:-1])``)."""
from __future__ import annotations

from typing import List

import torch

from ... import ops


class FeatureLossForDiscriminatorMelganMultiScales(torch.nn.Module):
    def forward(self, embeddings_a: List[List[torch.Tensor]], embeddings_b: List[List[torch.Tensor]]) -> torch.Tensor:
        return ops.feature_loss(embeddings_a, embeddings_b)
