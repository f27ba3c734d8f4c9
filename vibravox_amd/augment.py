"""Waveform augmentation on the device.

Counterpart of ``vibravox/torch_modules/dsp/data_augmentation.py:8-71`` (``WaveformDataAugmentation``, applied by the BWE
collator to the padded batch, bwe.py:286) for batches that already live in HBM: the same constructor, the same sequence
of ``torch.rand(1)`` / ``torch.randint`` draws on the CPU generator (so a seed selects the same augmentation), the work
itself as HIP kernels:

  * time masking (``dsp/time_masking_waveform.py:18-36``): one in-place zero-fill launch per waveform;
  * speed perturbation (``T.SpeedPerturbation`` = ``torchaudio.functional.speed`` -> ``resample``): windowed-sinc polyphase
    interpolation, ``eben_resample``; the kernel table follows torchaudio's ``_get_sinc_resample_kernel`` (hann window,
    lowpass width 6, rolloff 0.99) -- torchaudio is not installed here, so this part is a restatement (parity unpinned);
  * pitch shift (``T.PitchShift`` = ``torchaudio.functional.pitch_shift``): STFT (n_fft 512, hop 128, periodic hann) as
    framing + one GEMM, ``eben_phase_vocoder`` (time stretch by 2^(-steps/12)), inverse DFT as one GEMM + overlap-add +
    window-envelope normalisation, then ``eben_resample`` back to the original duration, cropped / padded to the input
    length.  Restated like the resampling (parity unpinned).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import ops
from ._lib import check, load, ptr, stream


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99, device=None):
    """(kernels (new, 2*width+orig) float32, width, orig, new) -- torchaudio.functional._get_sinc_resample_kernel, hann."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, :] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    return k.to(torch.float32).contiguous().to(device), width, orig, new


_kernel_cache = {}


def resample(x: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """torchaudio.functional.resample(x, orig_freq, new_freq) on the device; x (..., time) float32."""
    key = (int(orig_freq), int(new_freq), x.device)
    if key not in _kernel_cache:
        _kernel_cache[key] = sinc_resample_kernel(orig_freq, new_freq, device=x.device)
    k, width, orig, new = _kernel_cache[key]
    if orig == new:
        return x.clone()
    lead, t = x.shape[:-1], x.shape[-1]
    rows = max(1, math.prod(lead))
    x2 = x.contiguous().reshape(rows, t)
    t_out = int(math.ceil(new * t / orig))
    out = torch.empty((rows, t_out), dtype=torch.float32, device=x.device)
    check(load().eben_resample(ptr(x2), ptr(k), ptr(out), rows, t, t_out, orig, new, width, stream()), "resample")
    return out.reshape(*lead, t_out)


def speed(x: torch.Tensor, sample_rate: int, factor: float) -> torch.Tensor:
    """torchaudio.functional.speed: play `factor` times faster = resample from int(factor * rate) to rate."""
    return resample(x, int(factor * sample_rate), int(sample_rate))


_ps_cache = {}


def _pitch_plan(n_fft: int, device):
    key = (n_fft, device)
    if key not in _ps_cache:
        bins = n_fft // 2 + 1
        w = torch.hann_window(n_fft, periodic=True, dtype=torch.float64)
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(bins, dtype=torch.float64)
        ang = 2 * math.pi * k[:, None] * n[None, :] / n_fft
        fwd = torch.cat((torch.cos(ang) * w, -torch.sin(ang) * w), dim=0)                        # (2*bins, n_fft): windowed rDFT
        c = torch.full((bins,), 2.0, dtype=torch.float64)
        c[0] = c[-1] = 1.0
        inv = torch.cat((torch.cos(ang) * c[:, None], -torch.sin(ang) * c[:, None]), dim=0).t() * (w / n_fft)[:, None]   # (n_fft, 2*bins)
        _ps_cache[key] = dict(
            bins=bins, window=w,
            fwd=fwd.to(torch.float32).unsqueeze(-1).contiguous().to(device), inv=inv.to(torch.float32).unsqueeze(-1).contiguous().to(device),
            spec_f=ops.ConvSpec(c_in=n_fft, c_out=2 * bins, ksize=1), spec_i=ops.ConvSpec(c_in=2 * bins, c_out=n_fft, ksize=1),
            cache_f=ops.PackedWeights(), cache_i=ops.PackedWeights(), env={})
    return _ps_cache[key]


def pitch_shift(x: torch.Tensor, sample_rate: int, n_steps: int, bins_per_octave: int = 12, n_fft: int = 512) -> torch.Tensor:
    """torchaudio.functional.pitch_shift(x, sample_rate, n_steps) (defaults: n_fft 512, win 512, hop 128, hann) on the device."""
    import ctypes

    lib = load()
    hop = n_fft // 4
    rate = 2.0 ** (-float(n_steps) / bins_per_octave)
    lead, t = x.shape[:-1], x.shape[-1]
    rows = max(1, math.prod(lead))
    dev = x.device
    pl = _pitch_plan(n_fft, dev)
    bins = pl["bins"]
    sig = x.contiguous().reshape(rows, t)
    st = stream()
    # STFT (center=True, reflect): frames of all rows as one (n_fft, rows*F) matrix, one GEMM
    nf = t // hop + 1
    fr = torch.empty((1, n_fft, rows * nf), dtype=torch.float32, device=dev)
    check(lib.eben_stft_frames(ptr(sig), ptr(fr), rows, t, n_fft, hop, n_fft // 2, nf, st), "stft_frames")
    d = ops.conv_desc(pl["spec_f"], 1, rows * nf)
    pw = ops.pack_weights(pl["spec_f"], d, pl["fwd"], None, pl["cache_f"], False)
    spec = torch.empty((1, 2 * bins, rows * nf), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(fr), ptr(pw.wp_fwd), None, None, ptr(spec), st), "stft_gemm")
    # time stretch
    nf_out = int(math.ceil(nf / rate))
    if (nf_out - 1) * rate >= nf:   # torch.arange(0, nf, rate) excludes nf itself
        nf_out -= 1
    stretched = torch.empty((1, 2 * bins, rows * nf_out), dtype=torch.float32, device=dev)
    check(lib.eben_phase_vocoder(ptr(spec), ptr(stretched), rows, bins, nf, nf_out, rate, float(hop), st), "phase_vocoder")
    # inverse STFT: one GEMM to windowed frames, overlap-add, window-envelope normalisation
    d = ops.conv_desc(pl["spec_i"], 1, rows * nf_out)
    pw = ops.pack_weights(pl["spec_i"], d, pl["inv"], None, pl["cache_i"], False)
    frames = torch.empty((1, n_fft, rows * nf_out), dtype=torch.float32, device=dev)
    check(lib.eben_conv1d_fwd(ctypes.byref(d), ptr(stretched), ptr(pw.wp_fwd), None, None, ptr(frames), st), "istft_gemm")
    len_stretch = int(round(t / rate))
    full = n_fft + hop * (nf_out - 1)
    valid = min(len_stretch, full - n_fft // 2)
    y = torch.zeros((rows, len_stretch), dtype=torch.float32, device=dev)
    ola = torch.empty((rows, valid), dtype=torch.float32, device=dev)
    check(lib.eben_overlap_add_ex(ptr(frames), ptr(ola), rows, valid, n_fft, nf_out, hop, n_fft // 2, 0, 0, nf_out, rows * nf_out, st),
          "overlap_add")
    ek = (nf_out, valid)
    if ek not in pl["env"]:
        w2 = pl["window"] ** 2
        env = torch.zeros(full, dtype=torch.float64)
        for f in range(nf_out):
            env[f * hop:f * hop + n_fft] += w2
        pl["env"][ek] = (1.0 / env[n_fft // 2:n_fft // 2 + valid]).to(torch.float32).to(dev)
    y[:, :valid] = ola * pl["env"][ek]
    out = resample(y, int(sample_rate / rate), int(sample_rate))
    if out.shape[-1] >= t:
        out = out[:, :t]
    else:
        out = torch.nn.functional.pad(out, (0, t - out.shape[-1]))
    return out.contiguous().reshape(*lead, t)


def time_masking_(x: torch.Tensor, masking_percentage: float) -> torch.Tensor:
    """TimeMaskingBlockWaveform.forward (time_masking_waveform.py:28-36): in place, one torch.randint draw."""
    t = x.shape[-1]
    masked = int(t * masking_percentage / 100)
    first = torch.randint(0, t - masked, (1,)).item()
    if not x.is_contiguous():
        raise ValueError("time masking works in place on a contiguous (..., time) tensor")
    check(load().eben_time_mask(ptr(x), x.numel() // t, t, first, masked, stream()), "time_mask")
    return x


class WaveformDataAugmentation(torch.nn.Module):
    def __init__(self, sample_rate, p_data_augmentation=0, p_speed_perturbation=0.3, p_pitch_shift=0.3, p_time_masking=0.3,
                 speed_perturbation_factors=(0.7, 0.8, 0.85, 0.9, 0.95, 1.05, 1.1, 1.15, 1.2, 1.3),
                 pitch_shift_steps=(-4, -3, -2, -1, 1, 2, 3, 4, 5, 6), time_masking_percentage=(1, 2, 3, 4, 5, 6, 7, 8)):
        super().__init__()
        self.sample_rate = sample_rate
        assert 0 <= p_data_augmentation <= 1, "p_data_augmentation must be in [0, 1]"
        assert 0 <= p_speed_perturbation <= 1, "p_speed_perturbation must be in [0, 1]"
        assert 0 <= p_pitch_shift <= 1, "p_pitch_shift must be in [0, 1]"
        assert 0 <= p_time_masking <= 1, "p_time_masking must be in [0, 1]"
        self.apply_data_augmentation = p_data_augmentation
        self.p_speed_perturbation = p_speed_perturbation
        self.p_pitch_shift = p_pitch_shift
        self.p_time_masking = p_time_masking
        self.speed_perturbation_factors = speed_perturbation_factors
        self.pitch_shift_steps = pitch_shift_steps
        self.time_masking_percentage = time_masking_percentage

    def forward(self, waveform_1: torch.Tensor, waveform_2: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        if torch.rand(1) < self.apply_data_augmentation:
            if torch.rand(1) < self.p_speed_perturbation:
                f = self.speed_perturbation_factors[torch.randint(len(self.speed_perturbation_factors), size=(1,)).item()]
                # torchaudio's T.SpeedPerturbation.forward picks its speeder with torch.randint(len(factors), ()) on EVERY call --
                # one draw per waveform even with the single factor the reference constructs it with (data_augmentation.py:56-60)
                torch.randint(1, ())
                waveform_1 = speed(waveform_1, self.sample_rate, f)
                if waveform_2 is not None:
                    torch.randint(1, ())
                    waveform_2 = speed(waveform_2, self.sample_rate, f)
            if torch.rand(1) < self.p_pitch_shift:
                steps = self.pitch_shift_steps[torch.randint(len(self.pitch_shift_steps), size=(1,)).item()]
                waveform_1 = pitch_shift(waveform_1, self.sample_rate, steps)
                if waveform_2 is not None:
                    waveform_2 = pitch_shift(waveform_2, self.sample_rate, steps)
            if torch.rand(1) < self.p_time_masking:
                pct = self.time_masking_percentage[torch.randint(len(self.time_masking_percentage), size=(1,)).item()]
                waveform_1 = time_masking_(waveform_1, pct)
                if waveform_2 is not None:
                    waveform_2 = time_masking_(waveform_2, pct)
        return waveform_1, waveform_2
