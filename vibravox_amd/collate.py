"""Noisy-BWE batch assembly on the device (BASELINE config 4).

Counterpart of ``NoisyBWELightningDataModule.data_collator`` (vibravox/lightning_datamodules/noisybwe.py:219-291)
for clips that already live in HBM: the random noise slice (``mix_speech_and_noise_without_rescaling``,
vibravox/utils.py:195-254), the addition and the crop / pad to a constant length (``set_audio_duration`` /
``pad_audio``, utils.py:7-81) are ONE gather kernel (``eben_noisy_collate``) instead of a per-item Python loop on
the host.  The random draws are taken from the CPU generator with the reference's calls, in the reference's
order (all noise offsets first, then the crop offsets), so the same ``torch.manual_seed`` selects the same
samples as the reference collator.  The default augmentation of noisybwe.yaml:17 is the identity.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from ._lib import EbenCollateItem, EbenError, check, load, ptr, stream


def plan_noisy_bwe(lengths: List[int], noise_lengths: List[int], samples_or_none, deterministic: bool) -> Tuple[int, List[Tuple[int, int, int]]]:
    """Host-side part of the collator: (output length T, [(length, noise_start, shift)] per item), drawing the
    random numbers exactly as the reference does."""
    starts = []
    for ls, ln in zip(lengths, noise_lengths):
        if ln < ls:
            raise ValueError(f"noise_sample length ({ln}) must be >= speech_sample length ({ls})")
        starts.append(int(torch.randint(0, ln - ls, (1,))))            # utils.py:245
    if samples_or_none is None:                                          # collate_strategy == "pad"
        return max(lengths), [(ls, st, 0) for ls, st in zip(lengths, starts)]
    t = int(samples_or_none)
    plan = []
    for ls, st in zip(lengths, starts):
        if ls >= t:
            off = (ls - t) // 2 if deterministic else int(torch.randint(low=0, high=ls - t + 1, size=(1,)))   # utils.py:71-73
            plan.append((ls, st, off))
        else:
            plan.append((ls, st, -(t - ls // 2)))                        # pad_audio's left run, utils.py:23 (sic)
    return t, plan


def noisy_bwe_collate(batch: List[Dict[str, torch.Tensor]], sample_rate: int, collate_strategy: str = "pad",
                      deterministic: bool = False) -> Dict[str, torch.Tensor]:
    """``batch``: dicts of 1-D float32 DEVICE tensors ``audio_body_conducted`` [, ``audio_airborne``,
    ``audio_body_conducted_speechless_noisy``].  Returns the (B, 1, T) tensors of the reference collator."""
    lib = load()
    body = [item["audio_body_conducted"] for item in batch]
    for tns in body:
        if tns.dim() != 1:
            raise ValueError(f"Each speech sample must be a 1D tensor, but got shape {tuple(tns.shape)}")
    dev = body[0].device
    n = len(batch)
    table = (EbenCollateItem * n)()
    if "audio_airborne" not in batch[0]:
        t = max(x.shape[0] for x in body)
        for i, x in enumerate(body):
            table[i] = EbenCollateItem(ptr(x), None, None, x.shape[0], 0, 0)
        out = torch.empty((n, 1, t), dtype=torch.float32, device=dev)
        check(lib.eben_noisy_collate(table, n, t, ptr(out), None, stream()), "noisy_collate")
        return {"audio_body_conducted": out}
    air = [item["audio_airborne"] for item in batch]
    noise = [item["audio_body_conducted_speechless_noisy"] for item in batch]
    for a, b in zip(air, body):
        if a.shape != b.shape:
            raise EbenError("audio_airborne and audio_body_conducted must have the same length")
    samples = None if collate_strategy == "pad" else int(sample_rate * int(collate_strategy.split("-")[1]) / 1000)
    t, plan = plan_noisy_bwe([x.shape[0] for x in body], [x.shape[0] for x in noise], samples, deterministic)
    for i, (ls, st, sh) in enumerate(plan):
        table[i] = EbenCollateItem(ptr(body[i]), ptr(air[i]), ptr(noise[i]), ls, st, sh)
    bc = torch.empty((n, 1, t), dtype=torch.float32, device=dev)
    ab = torch.empty((n, 1, t), dtype=torch.float32, device=dev)
    check(lib.eben_noisy_collate(table, n, t, ptr(bc), ptr(ab), stream()), "noisy_collate")
    return {"audio_body_conducted": bc, "audio_airborne": ab}
