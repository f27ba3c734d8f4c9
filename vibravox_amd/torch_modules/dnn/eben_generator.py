"""EBEN generator on the MI355X HIP kernels.

Drop-in for ``vibravox/torch_modules/dnn/eben_generator.py:89-316``: ``EBENGenerator(m, n, p)``,
``forward(cut_audio) -> (enhanced, bands)``, ``cut_to_valid_length``, the ``pqmf`` / ``first_conv`` /
``encoder_blocks`` / ``latent_conv`` / ``decoder_blocks`` / ``last_conv`` attribute tree and therefore
the 92-entry ``state_dict`` (``...parametrizations.weight.original0/1``) are the reference's.
Layers are described by small tables instead of being spelled out; module construction order (and
so the default-init RNG stream under ``torch.manual_seed``) matches the reference.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from ... import ops
from ..dsp.pqmf import PseudoQMFBanks
from ..utils import HipConv1d, normalized_conv1d, normalized_conv_trans1d

try:  # hub interop is optional plumbing (eben_generator.py:9, 89-92)
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover

    class PyTorchModelHubMixin:  # type: ignore
        pass


SLOPE = 0.01                       # the single shared nn.LeakyReLU of eben_generator.py:110
ENCODER = ((64, 2), (128, 4), (256, 8))   # (out_channels, stride)    eben_generator.py:121-127
DECODER = ((128, 8), (64, 4), (32, 2))    #                           eben_generator.py:151-157
DILATIONS = (1, 3, 9)


class LeakyReLU(nn.Module):
    def __init__(self, negative_slope: float = SLOPE):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        return ops.leaky_relu(x, self.negative_slope)


class _Fused(nn.Module):
    """Placeholder keeping nn.Sequential indices where an activation was fused into a conv epilogue."""

    def forward(self, x):
        return x


class ResidualUnit(nn.Module):
    """x + lrelu(pointwise(dilated(x))) -- eben_generator.py:287-316 (no activation between the convs)."""

    def __init__(self, channels, nl, dilation, bias=False):
        super().__init__()
        self.dilated_conv = normalized_conv1d(channels, channels, kernel_size=3, dilation=dilation, padding="same", bias=bias,
                                              padding_mode="reflect")
        self.pointwise_conv = normalized_conv1d(channels, channels, kernel_size=1, padding="same", bias=bias,
                                                padding_mode="reflect", out_slope=nl.negative_slope)
        self.nl = nl

    def forward(self, x):
        return ops.add(x, self.pointwise_conv(self.dilated_conv(x)))


def _residual_stack(channels, nl):
    return nn.Sequential(*[ResidualUnit(channels=channels, nl=nl, dilation=d) for d in DILATIONS])


class EncBlock(nn.Module):
    """eben_generator.py:257-284."""

    def __init__(self, out_channels, stride, nl, bias=False):
        super().__init__()
        self.nl = nl
        self.residuals = _residual_stack(out_channels // 2, nl)
        self.conv = normalized_conv1d(out_channels // 2, out_channels, kernel_size=2 * stride, stride=stride, padding=stride - 1,
                                      bias=bias, padding_mode="reflect")

    def forward(self, x):
        return self.conv(self.residuals(x))


class DecBlock(nn.Module):
    """eben_generator.py:225-254: residuals(lrelu(conv_trans(x + skip))); the lrelu rides in the convT epilogue."""

    def __init__(self, out_channels, stride, nl, bias=False):
        super().__init__()
        self.nl = nl
        self.residuals = _residual_stack(out_channels, nl)
        self.conv_trans = normalized_conv_trans1d(2 * out_channels, out_channels, kernel_size=2 * stride, stride=stride,
                                                  padding=stride // 2, output_padding=0, bias=bias, out_slope=nl.negative_slope)

    def forward(self, x, encoder_output):
        return self.residuals(self.conv_trans(ops.add(x, encoder_output)))


class EBENGenerator(nn.Module, PyTorchModelHubMixin):
    def __init__(self, m: int, n: int, p: int):
        super().__init__()
        self.p = p
        self.pqmf = PseudoQMFBanks(decimation=m, kernel_size=n)
        self.multiple = 2 * 4 * 8 * m
        self.nl = LeakyReLU(SLOPE)
        self.first_conv = HipConv1d(self.p, 32, 3, padding="same", bias=False, padding_mode="reflect", weight_norm=False)
        self.encoder_blocks = nn.ModuleList([EncBlock(out_channels=c, stride=s, nl=self.nl) for c, s in ENCODER])
        self.latent_conv = nn.Sequential(
            self.nl,
            normalized_conv1d(256, 64, kernel_size=7, padding="same", bias=False, padding_mode="reflect", out_slope=SLOPE),
            _Fused(),
            normalized_conv1d(64, 256, kernel_size=7, padding="same", bias=False, padding_mode="reflect", out_slope=SLOPE),
            _Fused(),
        )
        self.decoder_blocks = nn.ModuleList([DecBlock(out_channels=c, stride=s, nl=self.nl) for c, s in DECODER])
        self.last_conv = HipConv1d(32, 4, 3, padding="same", bias=False, padding_mode="reflect", weight_norm=False)

    #: run everything up to ``last_conv`` through ``vibravox_amd.gen_engine`` (fused ResidualUnit launches, explicit backward
    #: chain) instead of module by module through autograd; same parameters, same values
    use_engine: bool = os.environ.get("EBEN_GEN_ENGINE", "1") != "0"

    def forward(self, cut_audio):
        if self.use_engine and cut_audio.is_cuda:
            from ... import gen_engine

            x, first_bands = gen_engine.core(self, cut_audio)
            # the input of last_conv of the latest forward: the train step's balancing takes the three loss-gradient norms at
            # last_conv.weight in one pass over it (ops.last_conv_grad_norms) instead of three autograd passes
            object.__setattr__(self, "_last_pre", x.detach() if torch.is_grad_enabled() else None)
            x = self.last_conv(x)
            enhanced_speech_decomposed = ops.tanh_lift(x, first_bands)
            return self.pqmf.synthesis_sum(enhanced_speech_decomposed), enhanced_speech_decomposed
        first_bands = self.pqmf(cut_audio, "analysis", bands=self.p)
        x = self.first_conv(first_bands)
        skips = []
        for block in self.encoder_blocks:
            x = block(self.nl(x))
            skips.append(x)
        x = self.latent_conv(x)
        for block, skip in zip(self.decoder_blocks, reversed(skips)):
            x = block(x, skip)
        x = self.last_conv(x)
        enhanced_speech_decomposed = ops.tanh_lift(x, first_bands)
        enhanced_speech = self.pqmf.synthesis_sum(enhanced_speech_decomposed)
        return enhanced_speech, enhanced_speech_decomposed

    def cut_to_valid_length(self, tensor):
        old_len = tensor.shape[2]
        new_len = old_len - (old_len + self.pqmf.kernel_size) % self.multiple
        return torch.narrow(tensor, 2, 0, new_len)
