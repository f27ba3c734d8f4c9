"""Pseudo-QMF analysis / synthesis filterbank on HIP FIR kernels.

Counterpart of ``vibravox/torch_modules/dsp/pqmf.py:17-232`` (``PseudoQMFBanks``): same
constructor, ``forward(signal, stage, bands)``, ``kernel_size`` / ``decimation`` properties and
frozen ``analysis_weights`` / ``synthesis_weights`` parameters of shape (M, 1, N).
The bank design (Kaiser prototype, cutoff by 5 LBFGS outer steps) runs once on the CPU at
construction; the filtering itself is ``eben_fir_decimate`` / ``eben_fir_interp_sum``.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from ... import ops


def _kaiser_sinc(cutoff, taps: int, beta: float) -> torch.Tensor:
    # window evaluated in float32 then held as float64, product stored as float32 (pqmf.py:77-89)
    window = torch.kaiser_window(taps, periodic=False, beta=beta).to(torch.float64)
    centre = torch.arange(taps) - (taps - 1) / 2
    proto = torch.ones(1, 1, taps)
    proto[0, 0, :] = (cutoff * torch.special.sinc(cutoff * centre)) * window
    return proto


def design_cutoff(bands: int, taps: int, beta: float) -> float:
    """Lin-Vaidyanathan objective minimised exactly as pqmf.py:93-140 does it."""

    def cost(c):
        h = _kaiser_sinc(c, taps, beta)
        corr = nn.functional.conv1d(nn.functional.pad(h, (taps // 2, taps // 2)), h)
        corr[..., taps // 2] = 0
        peak = torch.max(torch.abs(corr[..., :: 2 * bands]))
        outside = abs(c - 1 / (2 * bands)) > 1 / (4 * bands)
        return peak + (1 / (4 * bands) if outside else 0)

    c = (torch.ones(1) / (2 * bands)).requires_grad_(True)
    lbfgs = torch.optim.LBFGS([c], line_search_fn="strong_wolfe")
    for _ in range(5):
        lbfgs.zero_grad()
        cost(c).backward()
        lbfgs.step(lambda: cost(c))
    return c.item()


def design_bank(bands: int, taps: int, beta: float, cutoff: float):
    """Cosine modulation of the prototype (pqmf.py:142-180)."""
    h = _kaiser_sinc(cutoff, taps, beta).squeeze()
    centre = torch.arange(taps) - (taps - 1) / 2
    analysis = torch.zeros(bands, 1, taps)
    synthesis = torch.zeros(bands, 1, taps)
    for k in range(bands):
        phase = (2 * k + 1) * math.pi / 2 / bands * centre
        shift = (-1) ** k * math.pi / 4
        analysis[k, 0] = 2 * torch.flip(h * torch.cos(phase + shift), [0])
        synthesis[k, 0] = bands * 2 * h * torch.cos(phase - shift)
    return analysis, synthesis


class PseudoQMFBanks(nn.Module):
    def __init__(self, decimation: int = 32, kernel_size: int = 1024, beta: int = 9):
        super().__init__()
        assert kernel_size % (4 * decimation) == 0
        self._decimation, self._kernel_size, self._beta = decimation, kernel_size, beta
        self._cutoff_ratio = design_cutoff(decimation, kernel_size, beta)
        analysis, synthesis = design_bank(decimation, kernel_size, beta, self._cutoff_ratio)
        self.analysis_weights = nn.Parameter(analysis, requires_grad=False)
        self.synthesis_weights = nn.Parameter(synthesis, requires_grad=False)

    @property
    def kernel_size(self) -> int:
        return self._kernel_size

    @property
    def decimation(self) -> int:
        return self._decimation

    def forward(self, signal: torch.Tensor, stage: str, bands: int = -1) -> torch.Tensor:
        m, n = self._decimation, self._kernel_size
        if stage == "analysis":
            w = self.analysis_weights if bands == -1 else self.analysis_weights[:bands]
            l_out = (signal.shape[2] + n - 2) // m + 1
            return ops.fir_decimate(signal, w.reshape(w.shape[0], n), l_out, m, -(n - 1))
        if stage == "synthesis":
            # returns the per-band signals like the reference; prefer synthesis_sum on the hot path
            outs = [self.synthesis_sum(signal, only_band=k) for k in range(m)]
            return torch.cat(outs, dim=1)
        raise ValueError(f"Invalid stage '{stage}'. Expected 'analysis' or 'synthesis'.")

    def synthesis_sum(self, bands: torch.Tensor, only_band: int = -1) -> torch.Tensor:
        """sum over bands of conv_transpose1d(bands, g) (pqmf.py:204-213 + eben_generator.py:209-211)."""
        m, n = self._decimation, self._kernel_size
        l_out = m * bands.shape[2] - n
        w = self.synthesis_weights.reshape(m, n)
        if only_band >= 0:
            return ops.fir_interp_sum(bands[:, only_band : only_band + 1].contiguous(), w[only_band : only_band + 1], l_out, m, -(n - 1))
        return ops.fir_interp_sum(bands, w, l_out, m, -(n - 1))

    def cut_tensor(self, tensor: torch.Tensor) -> torch.Tensor:
        old = tensor.shape[2]
        return torch.narrow(tensor, 2, 0, old - (old + self._kernel_size) % self._decimation)
