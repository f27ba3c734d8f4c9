"""Weight-normalised conv layers on the HIP tap-conv kernels.

Counterpart of ``vibravox/torch_modules/utils.py:4-9`` (``weight_norm(nn.Conv1d)`` /
``weight_norm(nn.ConvTranspose1d)`` via ``torch.nn.utils.parametrizations``).  The parameter
layout is kept -- ``<layer>.parametrizations.weight.original0`` (gain g, shape (dim0,1,1)) and
``.original1`` (direction v) plus ``<layer>.bias`` -- so reference checkpoints load unchanged;
``w = g * v / ||v||`` itself is never materialised: the HIP pack kernel folds ``g/||v||`` into
the MFMA-layout copy of the weights.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


def _one(v) -> int:
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


class _WeightNormParams(nn.Module):
    def __init__(self, weight: torch.Tensor):
        super().__init__()
        norm = weight.detach().reshape(weight.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (weight.dim() - 1)))
        self.original0 = nn.Parameter(norm)
        self.original1 = nn.Parameter(weight.detach().clone())


class HipConv1d(nn.Module):
    """Conv1d / ConvTranspose1d (optionally weight-normalised) with fused LeakyReLUs.

    ``in_slope`` / ``out_slope`` fuse ``LeakyReLU`` on the input load / output store of the
    kernel (1.0 = none); they are build-side fusions, not reference constructor arguments.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", output_padding=0, transposed=False, weight_norm=True, in_slope=1.0, out_slope=1.0):
        super().__init__()
        k, s, d = _one(kernel_size), _one(stride), _one(dilation)
        if padding == "same":
            if s != 1:
                raise ValueError("padding='same' requires stride 1")
            total = d * (k - 1)
            pad_l, pad_r = total // 2, total - total // 2
        else:
            pad_l = pad_r = _one(padding)
        if padding_mode not in ("zeros", "reflect"):
            raise ValueError(f"unsupported padding_mode {padding_mode}")
        # same RNG consumption / initial values as the reference's nn.Conv1d + weight_norm
        if transposed:
            tmp = nn.ConvTranspose1d(in_channels, out_channels, k, stride=s, padding=pad_l, output_padding=output_padding,
                                     groups=groups, bias=bias, dilation=d)
        else:
            tmp = nn.Conv1d(in_channels, out_channels, k, stride=s, padding=0, dilation=d, groups=groups, bias=bias)
        self.spec = ops.ConvSpec(
            c_in=in_channels, c_out=out_channels, ksize=k, stride=s, dilation=d, groups=groups, pad_l=pad_l, pad_r=pad_r,
            reflect=(padding_mode == "reflect") and (pad_l > 0 or pad_r > 0), transposed=transposed, output_padding=output_padding,
            in_slope=float(in_slope), out_slope=float(out_slope),
        )
        self.weight_norm = weight_norm
        if weight_norm:
            self.parametrizations = nn.ModuleDict({"weight": _WeightNormParams(tmp.weight)})
        else:
            self.weight = nn.Parameter(tmp.weight.detach().clone())
        if bias:
            self.bias = nn.Parameter(tmp.bias.detach().clone())
        else:
            self.register_parameter("bias", None)
        self._packed = ops.PackedWeights()

    def weight_tensor(self) -> torch.Tensor:
        """The effective weight (torch ops; for inspection / export only, not on the hot path)."""
        if not self.weight_norm:
            return self.weight
        g, v = self.parametrizations["weight"].original0, self.parametrizations["weight"].original1
        return v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.weight_norm:
            prm = self.parametrizations["weight"]
            return ops.conv_layer(x, prm.original1, prm.original0, self.bias, self.spec, self._packed)
        return ops.conv_layer(x, self.weight, None, self.bias, self.spec, self._packed)

    def extra_repr(self) -> str:
        return f"{self.spec}"


def normalized_conv1d(*args, **kwargs) -> HipConv1d:
    """vibravox/torch_modules/utils.py:4-5."""
    return HipConv1d(*args, weight_norm=True, **kwargs)


def normalized_conv_trans1d(*args, **kwargs) -> HipConv1d:
    """vibravox/torch_modules/utils.py:8-9."""
    return HipConv1d(*args, weight_norm=True, transposed=True, **kwargs)
