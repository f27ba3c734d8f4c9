"""Closed-form deterministic tensors for parity fixtures.

Every weight / input used by the golden fixtures is a pure function of
(tensor name, shape): an integer hash of the flat index seeded by crc32(name).
The same function fills the reference modules (when the goldens are generated,
in the build container), the CPU oracle and the HIP product (in the tests), so
no multi-MB weight file has to be committed.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def hash_uniform(name: str, numel: int) -> np.ndarray:
    """float64 values in [-1, 1): 32-bit integer mix of (crc32(name), index)."""
    seed = np.uint64(zlib.crc32(name.encode()))
    i = np.arange(numel, dtype=np.uint64)
    x = (i * np.uint64(2654435761) + seed * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.float64) / 2147483648.0 - 1.0


def formula_tensor(name: str, shape: Tuple[int, ...], amp: float = 1.0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy((amp * hash_uniform(name, n)).astype(np.float32)).reshape(shape)


def formula_state_dict(shapes: Dict[str, Tuple[int, ...]], tag: str) -> Dict[str, torch.Tensor]:
    """Fill a network's trainable tensors from its {key: shape} map.

    * ``...original1`` (weight-norm direction v) and plain ``weight``: U(+-1/sqrt(fan_in))
    * ``...original0`` (weight-norm gain g): ||v|| * (1 + 0.25 u) so that g != ||v||
    * ``bias``: U(+-1/sqrt(fan_in)) of the matching weight
    Keys starting with ``pqmf.`` are left to the caller (they are design outputs).
    """
    out: Dict[str, torch.Tensor] = {}
    for key, shape in shapes.items():
        if key.startswith("pqmf."):
            continue
        if key.endswith("original1") or key.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            out[key] = formula_tensor(f"{tag}/{key}", shape, 1.0 / np.sqrt(fan_in))
    for key, shape in shapes.items():
        if key.endswith("original0"):
            v = out[key[:-1] + "1"]
            nrm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shape)
            out[key] = nrm * (1.0 + 0.25 * formula_tensor(f"{tag}/{key}", shape))
        elif key.endswith(".bias"):
            stem = key[: -len(".bias")]
            v = out[stem + ".parametrizations.weight.original1"]
            fan_in = int(np.prod(v.shape[1:]))
            out[key] = formula_tensor(f"{tag}/{key}", shape, 1.0 / np.sqrt(fan_in))
    return out


def formula_audio(name: str, batch: int, length: int, amp: float = 0.1) -> torch.Tensor:
    """(B,1,T) pseudo-speech: two slow chirps plus hash noise, |x| <~ amp."""
    t = np.arange(length, dtype=np.float64)
    rows = []
    for b in range(batch):
        f0 = 90.0 + 35.0 * b
        tone = 0.5 * np.sin(2 * np.pi * (f0 + 0.004 * t) * t / 16000.0) + 0.25 * np.sin(2 * np.pi * 3.1 * f0 * t / 16000.0 + b)
        noise = 0.25 * hash_uniform(f"{name}/{b}", length)
        rows.append(amp * (tone + noise))
    return torch.from_numpy(np.stack(rows).astype(np.float32)).unsqueeze(1)


def summarize(t: torch.Tensor, n_probe: int = 16) -> Dict[str, np.ndarray]:
    """Size-independent summary of a tensor: shape, sum, L2, and n_probe strided samples."""
    flat = t.detach().double().reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, min(n_probe, flat.numel())).long()
    return {
        "shape": np.array(t.shape, dtype=np.int64),
        "sum": np.array(flat.sum().item()),
        "l2": np.array(flat.norm().item()),
        "probe": flat[idx].numpy(),
    }


def flatten_summary(prefix: str, t: torch.Tensor, out: Dict[str, np.ndarray]) -> None:
    for k, v in summarize(t).items():
        out[f"{prefix}:{k}"] = v


def iter_embeddings(embs: Iterable[Iterable[torch.Tensor]]):
    for si, scale in enumerate(embs):
        for li, t in enumerate(scale):
            yield f"s{si}l{li}", t
