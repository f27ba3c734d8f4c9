"""EBEN multi-scale discriminators on the HIP grouped tap-conv kernels.

Drop-in for ``vibravox/torch_modules/dnn/eben_discriminator.py:10-163``:
``DiscriminatorEBENMultiScales(q, min_channels)`` = three dilated PQMF-band discriminators
(dilation 1, 2, 3) + one MelGAN discriminator; ``forward(bands, audio) -> List[List[Tensor]]`` with
element 0 the input and the last element the logits.  Attribute tree / ``state_dict`` keys
(``pqmf_discriminators.{i}.discriminator.{0.1,1.0,...,7}``, ``melgan_discriminator...``) are kept.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from ..utils import normalized_conv1d
from .melgan_discriminator import DiscriminatorMelGAN, ReflectionPad1d

try:
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover

    class PyTorchModelHubMixin:  # type: ignore
        pass


class DiscriminatorEBEN(nn.Module):
    """eben_discriminator.py:54-163: grouped dilated conv stack on the q last PQMF bands."""

    def __init__(self, dilation=1, q: int = 3, min_channels: int = 24):
        super().__init__()
        self.dilation = dilation
        assert min_channels % q == 0, "min_channels must be a multiple of q"
        c = min_channels
        # (c_in, c_out, kernel, stride, padding, groups, dilation, activated)
        table = [(q, c, 3, 1, 1, q, dilation, True)]
        table += [(c * 2 ** (i - 1), c * 2 ** i, 7, 2, 3, q, dilation, True) for i in range(1, 6)]
        table += [(c * 32, c * 32, 5, 1, 2, q, dilation, True), (c * 32, 1, 3, 1, 1, 1, 1, False)]
        layers = []
        for i, (ci, co, k, s, p, g, d, act) in enumerate(table):
            conv = normalized_conv1d(ci, co, kernel_size=(k,), stride=(s,), padding=(p,), dilation=d, groups=g,
                                     out_slope=0.2 if act else 1.0)
            if i == 0:
                layers.append(nn.Sequential(ReflectionPad1d(1), conv))
            elif act:
                layers.append(nn.Sequential(conv))
            else:
                layers.append(conv)
        self.discriminator = nn.ModuleList(layers)

    def forward(self, bands):
        embeddings = [bands]
        for module in self.discriminator:
            embeddings.append(module(embeddings[-1]))
        return embeddings


class DiscriminatorEBENMultiScales(nn.Module, PyTorchModelHubMixin):
    def __init__(self, q: int = 3, min_channels: int = 24):
        super().__init__()
        self.q = q
        self.pqmf_discriminators = torch.nn.ModuleList(
            [DiscriminatorEBEN(dilation=d, q=q, min_channels=min_channels) for d in (1, 2, 3)]
        )
        self.melgan_discriminator = DiscriminatorMelGAN(alpha_leaky_relu=0.2)

    #: run the four independent sub-discriminators on separate HIP streams (forward here, backward by
    #: autograd on the same streams): their kernels are latency-bound with small grids on the deep
    #: layers, so overlapping them fills CUs that a single stream leaves idle.
    concurrent_streams: bool = os.environ.get("EBEN_D_STREAMS", "1") != "0"

    def forward(self, bands, audio):
        sub = bands[:, -self.q :, :]
        if not (self.concurrent_streams and bands.is_cuda):
            embeddings = [dis(sub) for dis in self.pqmf_discriminators]
            embeddings.append(self.melgan_discriminator(audio))
            return embeddings
        if sub.data_ptr() != bands.data_ptr() or not sub.is_contiguous():
            sub = sub.contiguous()
        main = torch.cuda.current_stream()
        if getattr(self, "_streams", None) is None or self._streams[0].device != bands.device:
            self._streams = [torch.cuda.Stream(device=bands.device) for _ in range(4)]
        nets = list(self.pqmf_discriminators) + [self.melgan_discriminator]
        inputs = [sub, sub, sub, audio]
        embeddings = []
        for net, x, st in zip(nets, inputs, self._streams):
            st.wait_stream(main)
            x.record_stream(st)
            with torch.cuda.stream(st):
                embeddings.append(net(x))
        for st, scale in zip(self._streams, embeddings):
            main.wait_stream(st)
            for t in scale[1:]:
                t.record_stream(main)
        return embeddings
