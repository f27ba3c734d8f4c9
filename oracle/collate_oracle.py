"""CPU restatement of the reference's noisy-BWE batch assembly (TEST INFRASTRUCTURE: only tests/ import it).

Follows, function by function:
  * ``pad_audio``            vibravox/utils.py:7-31   (incl. its operator-precedence quirk: the left pad is
                                                       ``desired - initial // 2`` and the right pad negative,
                                                       i.e. zeros first, then the HEAD of the clip)
  * ``slice_audio``          vibravox/utils.py:33-48
  * ``set_audio_duration``   vibravox/utils.py:50-81  (random offset: ``torch.randint(0, initial - desired + 1)``)
  * ``mix_speech_and_noise_without_rescaling``  vibravox/utils.py:195-254 (``torch.randint(0, len_noise - len_speech)``)
  * the glue of ``NoisyBWELightningDataModule.data_collator``  vibravox/lightning_datamodules/noisybwe.py:219-291
    (default augmentation = identity, noisybwe.yaml:17).
Pinned against the reference functions by tests/golden/collate_golden.npz (tests/golden/make_collate_golden.py
imports vibravox.utils in the build container); the collator glue is a restatement (its module needs Lightning).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor


def pad_audio(audio: Tensor, desired_samples: int) -> Tensor:
    assert audio.shape[-1] <= desired_samples
    initial = audio.shape[-1]
    left = desired_samples - initial // 2            # sic: utils.py:23
    return torch.nn.functional.pad(audio, (left, desired_samples - initial - left), mode="constant", value=0)


def slice_audio(audio: Tensor, desired_samples: int, offset_samples: int) -> Tensor:
    assert audio.shape[-1] >= desired_samples
    return audio[..., offset_samples: offset_samples + desired_samples]


def set_audio_duration(audio: Tensor, desired_samples: int, audio_bis: Optional[Tensor] = None, deterministic: bool = False):
    initial = audio.shape[-1]
    assert audio_bis is None or audio.shape == audio_bis.shape
    if initial >= desired_samples:
        off = (initial - desired_samples) // 2 if deterministic else int(torch.randint(low=0, high=initial - desired_samples + 1, size=(1,)))
        audio = slice_audio(audio, desired_samples, off)
        if audio_bis is not None:
            audio_bis = slice_audio(audio_bis, desired_samples, off)
    else:
        audio = pad_audio(audio, desired_samples)
        if audio_bis is not None:
            audio_bis = pad_audio(audio_bis, desired_samples)
    return (audio, audio_bis) if audio_bis is not None else audio


def mix_speech_and_noise_without_rescaling(speech_batch: List[Tensor], noise_batch: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
    if len(speech_batch) != len(noise_batch):
        raise ValueError("speech_batch and noise_batch must have the same length")
    corrupted, sliced = [], []
    for speech, noise in zip(speech_batch, noise_batch):
        if speech.dim() != 1 or noise.dim() != 1:
            raise ValueError("samples must be 1D tensors")
        if noise.size(0) < speech.size(0):
            raise ValueError("noise must be at least as long as the speech")
        start = int(torch.randint(0, noise.size(0) - speech.size(0), (1,)))
        ns = noise[start: start + speech.size(0)]
        corrupted.append(speech + ns)
        sliced.append(ns)
    return corrupted, sliced


def noisy_bwe_collate(batch: List[Dict[str, Tensor]], sample_rate: int, collate_strategy: str, deterministic: bool) -> Dict[str, Tensor]:
    """noisybwe.py:219-291 with items already reduced to their arrays:
    {"audio_body_conducted", ["audio_airborne", "audio_body_conducted_speechless_noisy"]} -> (B, 1, T) tensors."""
    body = [item["audio_body_conducted"] for item in batch]
    if "audio_airborne" not in batch[0]:
        return {"audio_body_conducted": torch.nn.utils.rnn.pad_sequence(body, batch_first=True, padding_value=0.0).unsqueeze(1)}
    air = [item["audio_airborne"] for item in batch]
    noise = [item["audio_body_conducted_speechless_noisy"] for item in batch]
    noisy, _ = mix_speech_and_noise_without_rescaling(body, noise)
    if collate_strategy == "pad":
        bc = torch.nn.utils.rnn.pad_sequence(noisy, batch_first=True, padding_value=0.0).unsqueeze(1)
        ab = torch.nn.utils.rnn.pad_sequence(air, batch_first=True, padding_value=0.0).unsqueeze(1)
    else:
        samples = int(sample_rate * int(collate_strategy.split("-")[1]) / 1000)
        bcs, abs_ = [], []
        for b, a in zip(noisy, air):
            bp, ap = set_audio_duration(audio=b, desired_samples=samples, audio_bis=a, deterministic=deterministic)
            bcs.append(bp.unsqueeze(0))
            abs_.append(ap.unsqueeze(0))
        bc, ab = torch.stack(bcs, dim=0), torch.stack(abs_, dim=0)
    return {"audio_body_conducted": bc, "audio_airborne": ab}
