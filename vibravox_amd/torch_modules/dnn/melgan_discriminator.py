"""MelGAN waveform discriminator on the HIP grouped tap-conv kernels.

Drop-in for ``DiscriminatorMelGAN`` of ``vibravox/torch_modules/dnn/melgan_discriminator.py:76-169``
(the multi-scale wrapper of :17-73 is not on the EBEN path and is not provided).  Same
``discriminator`` ModuleList indices -> same ``state_dict`` keys (``discriminator.0.1...``,
``discriminator.1.0...``, ..., ``discriminator.6...``); LeakyReLU(alpha) is fused in the conv epilogue.
"""
from __future__ import annotations

from torch import nn

from ... import ops
from ..utils import normalized_conv1d

# (c_in, c_out, kernel, stride, padding, groups) -- melgan_discriminator.py:89-156
MELGAN_LAYERS = (
    (1, 16, 15, 1, 0, 1),
    (16, 64, 41, 4, 20, 4),
    (64, 256, 41, 4, 20, 4),
    (256, 1024, 41, 4, 20, 4),
    (1024, 1024, 41, 4, 20, 4),
    (1024, 1024, 5, 1, 2, 1),
    (1024, 1, 3, 1, 1, 1),
)


class ReflectionPad1d(nn.Module):
    def __init__(self, padding: int):
        super().__init__()
        self.padding = padding

    def forward(self, x):
        return ops.reflect_pad(x, self.padding, self.padding)


class DiscriminatorMelGAN(nn.Module):
    def __init__(self, alpha_leaky_relu: float):
        super().__init__()
        layers = []
        last = len(MELGAN_LAYERS) - 1
        for i, (ci, co, k, s, p, g) in enumerate(MELGAN_LAYERS):
            conv = normalized_conv1d(in_channels=ci, out_channels=co, kernel_size=k, stride=s, padding=p, groups=g,
                                     out_slope=1.0 if i == last else alpha_leaky_relu)
            if i == 0:
                layers.append(nn.Sequential(ReflectionPad1d(7), conv))
            elif i < last:
                layers.append(nn.Sequential(conv))
            else:
                layers.append(conv)
        self.discriminator = nn.ModuleList(layers)

    def forward(self, audio):
        embeddings = [audio]
        for module in self.discriminator:
            embeddings.append(module(embeddings[-1]))
        return embeddings
