"""Synthetic stand-in for ``BWELightningDataModule`` (vibravox/lightning_datamodules/bwe.py:14-293):
yields the same batch dict -- ``{"audio_body_conducted", "audio_airborne"}`` of float32 ``(B, 1, T)``
tensors (bwe.py:290-293) -- without the HF-hub dataset (no network here).  Out-of-scope rows of
SURVEY.md section 2 (#12) stay out of scope; this only feeds ``run.py`` / ``bench.py``."""
from __future__ import annotations

from typing import Dict, Iterator

import torch


class SyntheticBWEDataModule:
    def __init__(self, sample_rate: int = 16000, batch_size: int = 32, constant_length_ms: int = 2000, seed: int = 1234,
                 device: str = "cuda"):
        self.sample_rate, self.batch_size, self.seed, self.device = sample_rate, batch_size, seed, device
        self.length = sample_rate * constant_length_ms // 1000

    def train_dataloader(self, rank: int = 0) -> Iterator[Dict[str, torch.Tensor]]:
        g = torch.Generator().manual_seed(self.seed + rank)
        while True:
            yield {
                "audio_body_conducted": (0.1 * torch.randn(self.batch_size, 1, self.length, generator=g)).to(self.device),
                "audio_airborne": (0.1 * torch.randn(self.batch_size, 1, self.length, generator=g)).to(self.device),
            }
