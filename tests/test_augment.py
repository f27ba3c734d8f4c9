"""Waveform augmentation: the CPU oracle against golden vectors produced by the reference's own
classes (draw order + time masking), the sinc resampling restatement against analytic signals (torchaudio is not
installed: unpinned), and (GPU) the device module against both."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import augment_oracle as A

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from formula import formula_tensor  # noqa: E402
from make_augment_golden import CASES, inputs  # noqa: E402  (only the case table and the inputs; the reference is not imported)


@pytest.fixture(scope="module")
def agold():
    return np.load(os.path.join(HERE, "golden", "augment_golden.npz"))


def expected(x, rec):
    first, count = int(rec[0]), int(rec[1])
    y = x.clone()
    y[..., first:first + count] = 0
    assert abs(float(y.double().sum()) - rec[2]) <= 1e-9 * max(1.0, abs(rec[2]))
    return y


def run_cases(agold, make, to_dev=lambda t: t):
    for i, (seed, p_aug, p_mask, shape) in enumerate(CASES):
        aug = make(16000, p_data_augmentation=p_aug, p_speed_perturbation=0.0, p_pitch_shift=0.0, p_time_masking=p_mask)
        a, b = inputs(i, shape)
        torch.manual_seed(seed)
        oa, ob = aug(to_dev(a.clone()), to_dev(b.clone()))
        assert torch.equal(oa.cpu(), expected(a, agold[f"aug/{i}/a"])), i
        assert torch.equal(ob.cpu(), expected(b, agold[f"aug/{i}/b"])), i
        assert float(torch.rand(1)) == float(agold[f"aug/{i}/next_draw"][0]), i   # same number of draws as the reference


def test_oracle_matches_reference_draws_and_time_masking(agold):
    run_cases(agold, A.WaveformDataAugmentation)
    for pct in (1, 3, 8, 50):
        x = formula_tensor(f"tm/{pct}", (2, 2, 1000))
        torch.manual_seed(10 + pct)
        assert torch.equal(A.time_masking(x.clone(), pct), expected(x, agold[f"tm/{pct}"]))


@pytest.mark.parametrize("factor", [0.85, 0.9, 0.95, 1.05, 1.1, 1.15, 0.7, 1.3])
def test_oracle_speed_on_a_sine(factor):
    """speed(x, factor) plays x `factor` times faster: a 200 Hz sine comes out at 200 * factor Hz, ceil(new*T/orig) samples."""
    sr, t = 16000, 8000
    src = int(factor * sr)
    g = math.gcd(src, sr)
    n = np.arange(t)
    y = A.speed(np.sin(2 * np.pi * 200.0 * n / sr)[None, :], sr, factor)
    assert y.shape == (1, math.ceil((sr // g) * t / (src // g)))
    m = np.arange(y.shape[1])
    want = np.sin(2 * np.pi * 200.0 * factor * m / sr)
    assert np.abs(y[0, 200:-200] - want[200:-200]).max() < 2e-3   # rolloff 0.99 * width-6 hann sinc: ~1e-3 passband ripple
    assert np.array_equal(A.resample(y, 16000, 16000), y)


def test_device_module_surface_on_cpu():
    from vibravox_amd.augment import WaveformDataAugmentation

    with pytest.raises(AssertionError):
        WaveformDataAugmentation(16000, p_data_augmentation=1.5)
    ident = WaveformDataAugmentation(16000)                                 # identity.yaml: p_data_augmentation = 0
    x = torch.randn(2, 1, 100)
    a, b = ident(x, None)
    assert a is x and b is None


@pytest.mark.parametrize("two", [False, True])
def test_draw_sequence_with_every_transform_enabled(monkeypatch, two):
    """CPU: the module consumes the CPU generator exactly as the reference does when every branch fires -- including the
    per-call ``torch.randint(len(factors), ())`` inside torchaudio's ``T.SpeedPerturbation.forward`` (one per waveform) that
    sits between the speed factor draw and the pitch-shift decision (data_augmentation.py:52-69)."""
    import vibravox_amd.augment as DA

    monkeypatch.setattr(DA, "speed", lambda w, sr, f: w)
    monkeypatch.setattr(DA, "pitch_shift", lambda w, sr, s: w)

    def masking(x, pct):   # the draw of time_masking_waveform.py:30 without the kernel
        t = x.shape[-1]
        torch.randint(0, t - int(t * pct / 100), (1,))
        return x

    monkeypatch.setattr(DA, "time_masking_", masking)
    mod = DA.WaveformDataAugmentation(16000, p_data_augmentation=1, p_speed_perturbation=1, p_pitch_shift=1, p_time_masking=1)
    x = torch.zeros(1, 1, 1000)
    n = 2 if two else 1
    torch.manual_seed(7)
    mod(x, x.clone() if two else None)
    after = float(torch.rand(1))
    torch.manual_seed(7)
    torch.rand(1); torch.rand(1)                                   # apply at all?  speed?
    torch.randint(len(mod.speed_perturbation_factors), size=(1,))
    for _ in range(n):
        torch.randint(1, ())                                       # SpeedPerturbation.forward, once per waveform
    torch.rand(1)
    torch.randint(len(mod.pitch_shift_steps), size=(1,))
    torch.rand(1)
    pct = mod.time_masking_percentage[torch.randint(len(mod.time_masking_percentage), size=(1,)).item()]
    for _ in range(n):
        torch.randint(0, 1000 - int(1000 * pct / 100), (1,))
    assert float(torch.rand(1)) == after


@pytest.mark.gpu
def test_device_time_masking_matches_reference_golden(agold):
    from vibravox_amd.augment import WaveformDataAugmentation, time_masking_

    dev = torch.device("cuda")
    run_cases(agold, WaveformDataAugmentation, lambda t: t.to(dev))
    for pct in (1, 3, 8, 50):
        x = formula_tensor(f"tm/{pct}", (2, 2, 1000))
        torch.manual_seed(10 + pct)
        assert torch.equal(time_masking_(x.clone().to(dev), pct).cpu(), expected(x, agold[f"tm/{pct}"]))


@pytest.mark.gpu
@pytest.mark.parametrize("factor,shape", [(0.85, (3, 1, 4000)), (0.9, (2, 1, 7777)), (1.15, (4, 1, 1234)), (1.3, (1, 2, 3, 501)), (0.7, (2, 16000))])
def test_device_speed_matches_oracle(factor, shape):
    from vibravox_amd.augment import speed

    x = formula_tensor(f"sp/{factor}", shape)
    got = speed(x.to(torch.device("cuda")), 16000, factor).cpu()
    ref = A.speed(x.numpy(), 16000, factor)
    assert tuple(got.shape) == ref.shape
    assert float(np.abs(got.double().numpy() - ref).max()) < 2e-6 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("steps", [-2, 1, 5])   # (the float64 oracle is a per-bin Python walk: ~1 min of CPU per case; five cases were 4 of the suite's 6.5 minutes)
def test_oracle_pitch_shift_on_a_sine(steps):
    """pitch_shift keeps the duration and moves a 440 Hz sine to 440 * 2^(steps/12) Hz."""
    sr, t = 16000, 8000
    y = A.pitch_shift(np.sin(2 * np.pi * 440.0 * np.arange(t) / sr)[None, :], sr, steps)
    assert y.shape == (1, t)
    seg = y[0, 1000:7000] * np.hanning(6000)
    peak = np.argmax(np.abs(np.fft.rfft(seg))) * sr / 6000
    assert abs(peak - 440.0 * 2 ** (steps / 12)) < 3.0
    assert 0.4 < np.abs(y[0, 1000:7000]).max() < 1.2   # bin-wise phase propagation is not amplitude preserving between bins


@pytest.mark.gpu
@pytest.mark.parametrize("steps,shape", [(-2, (3, 1, 4000)), (1, (2, 1, 7777)), (4, (2, 2, 3001)), (-4, (1, 16000)), (6, (1, 32000)),
                                         (-3, (4, 32000))])
def test_device_pitch_shift_matches_oracle(steps, shape):
    from vibravox_amd.augment import pitch_shift

    x = formula_tensor(f"ps/{steps}", shape)
    got = pitch_shift(x.to(torch.device("cuda")), 16000, steps).cpu()
    ref = A.pitch_shift(x.numpy(), 16000, steps)
    assert tuple(got.shape) == ref.shape == tuple(shape)
    err = np.abs(got.double().numpy() - ref)
    scale = float(np.abs(ref).max())
    # fp32 STFT / inverse STFT / resampling around a float64 phase walk (angles, wrap and running sum) against the float64 oracle:
    # measured 2e-7 .. 1.2e-6 RMS and <= 8e-6 peak of the signal's scale (the fp32 walk of round 1 sat at 1e-3 / 2e-2)
    assert float(np.sqrt((err ** 2).mean())) < 5e-6 * scale and float(err.max()) < 4e-5 * scale


@pytest.mark.gpu
def test_device_module_follows_the_oracle_with_speed_and_masking():
    from vibravox_amd.augment import WaveformDataAugmentation

    dev = torch.device("cuda")
    kw = dict(p_data_augmentation=0.8, p_speed_perturbation=0.6, p_pitch_shift=0.5, p_time_masking=0.6,
              speed_perturbation_factors=(0.85, 0.9, 0.95, 1.05, 1.1, 1.15), pitch_shift_steps=(-2, -1, 1, 2), time_masking_percentage=(1, 2, 3))
    d, o = WaveformDataAugmentation(16000, **kw), A.WaveformDataAugmentation(16000, **kw)
    changed = 0
    for seed in (0, 1, 2, 3, 5, 8, 9, 11):   # pitch shift (0, 3), speed only, masking only, untouched (5); the oracle's pitch shift is the slow part
        a, b = formula_tensor(f"augm/{seed}/a", (3, 1, 3000)), formula_tensor(f"augm/{seed}/b", (3, 1, 3000))
        torch.manual_seed(seed)
        ra, rb = o(a.clone(), b.clone())
        state = float(torch.rand(1))
        torch.manual_seed(seed)
        ga, gb = d(a.clone().to(dev), b.clone().to(dev))
        assert float(torch.rand(1)) == state
        assert ga.shape == ra.shape and gb.shape == rb.shape
        assert float((ga.cpu() - ra).abs().max()) < 2e-2 and float((gb.cpu() - rb).abs().max()) < 2e-2
        changed += int(ra.shape != a.shape or not torch.equal(ra, a))
    assert changed >= 6   # the seeds exercise both transforms
