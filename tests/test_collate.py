"""Noisy-BWE batch assembly: the CPU oracle against golden vectors produced by the
reference's own functions, the host-side planner against the oracle (same seed -> same samples), and (GPU)
the gather kernel against the oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import collate_oracle as C

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_collate_golden import LENGTHS, items  # noqa: E402  (only the input definitions; the reference is not imported)


@pytest.fixture(scope="module")
def cgold():
    return np.load(os.path.join(HERE, "golden", "collate_golden.npz"))


def test_oracle_matches_reference_functions(cgold):
    batch = items()
    for seed in (0, 1):
        torch.manual_seed(seed)
        noisy, sliced = C.mix_speech_and_noise_without_rescaling([b["audio_body_conducted"] for b in batch],
                                                                 [b["audio_body_conducted_speechless_noisy"] for b in batch])
        for i, (n, s) in enumerate(zip(noisy, sliced)):
            np.testing.assert_array_equal(n.numpy(), cgold[f"mix/seed{seed}/noisy{i}"])
            np.testing.assert_array_equal(s.numpy(), cgold[f"mix/seed{seed}/slice{i}"])
        for det in (False, True):
            for i, (n, b) in enumerate(zip(noisy, batch)):
                a, ab = C.set_audio_duration(audio=n, desired_samples=800, audio_bis=b["audio_airborne"], deterministic=det)
                np.testing.assert_array_equal(a.numpy(), cgold[f"dur/seed{seed}/det{int(det)}/bc{i}"])
                np.testing.assert_array_equal(ab.numpy(), cgold[f"dur/seed{seed}/det{int(det)}/air{i}"])
    np.testing.assert_array_equal(C.pad_audio(torch.arange(10.0), 16).numpy(), cgold["pad/10_16"])
    np.testing.assert_array_equal(C.pad_audio(torch.arange(7.0), 7).numpy(), cgold["pad/7_7"])
    np.testing.assert_array_equal(C.pad_audio(torch.arange(10.0).reshape(2, 5), 9).numpy(), cgold["pad/2x5_9"])


def _apply_plan(batch, t, plan):
    """numpy statement of what eben_noisy_collate computes from a plan."""
    bc = np.zeros((len(batch), 1, t), np.float32)
    ab = np.zeros((len(batch), 1, t), np.float32)
    for i, (item, (ls, st, sh)) in enumerate(zip(batch, plan)):
        sp, ai, no = (item[k].numpy() for k in ("audio_body_conducted", "audio_airborne", "audio_body_conducted_speechless_noisy"))
        for tt in range(t):
            u = tt + sh
            if 0 <= u < ls:
                bc[i, 0, tt] = sp[u] + no[st + u]
                ab[i, 0, tt] = ai[u]
    return bc, ab


@pytest.mark.parametrize("strategy,deterministic", [("pad", False), ("constant_length-50-ms", False), ("constant_length-50-ms", True),
                                                    ("constant_length-100-ms", False)])
def test_planner_selects_the_reference_samples(strategy, deterministic):
    """Same seed, same draws, same order as the reference collator (noisybwe.py:219-291)."""
    from vibravox_amd.collate import plan_noisy_bwe

    batch = items()
    torch.manual_seed(3)
    want = C.noisy_bwe_collate(batch, 16000, strategy, deterministic)
    torch.manual_seed(3)
    samples = None if strategy == "pad" else int(16000 * int(strategy.split("-")[1]) / 1000)
    t, plan = plan_noisy_bwe([ls for ls, _, _ in LENGTHS], [ln for _, _, ln in LENGTHS], samples, deterministic)
    bc, ab = _apply_plan(batch, t, plan)
    np.testing.assert_array_equal(bc, want["audio_body_conducted"].numpy())
    np.testing.assert_array_equal(ab, want["audio_airborne"].numpy())
    assert t == (max(ls for ls, _, _ in LENGTHS) if strategy == "pad" else samples)


def test_planner_error_behaviour():
    from vibravox_amd.collate import plan_noisy_bwe

    with pytest.raises(ValueError):
        plan_noisy_bwe([100], [99], None, False)          # utils.py:240-241
    with pytest.raises(RuntimeError):
        plan_noisy_bwe([100], [100], None, False)         # torch.randint(0, 0): the reference fails the same way


@pytest.mark.gpu
@pytest.mark.parametrize("strategy,deterministic", [("pad", False), ("constant_length-50-ms", False), ("constant_length-50-ms", True),
                                                    ("constant_length-100-ms", False)])
def test_device_collate_matches_oracle_bit_exact(hip, strategy, deterministic):
    from vibravox_amd.collate import noisy_bwe_collate

    batch = items()
    torch.manual_seed(11)
    want = C.noisy_bwe_collate(batch, 16000, strategy, deterministic)
    dev_batch = [{k: v.cuda() for k, v in it.items()} for it in batch]
    torch.manual_seed(11)
    got = noisy_bwe_collate(dev_batch, 16000, strategy, deterministic)
    assert set(got) == set(want)
    for k in want:
        assert got[k].shape == want[k].shape
        assert torch.equal(got[k].cpu(), want[k]), k


@pytest.mark.gpu
def test_device_collate_real_noisy_items_and_large_batch(hip):
    """Items without a reference clip are only padded (noisybwe.py:243-248); more than 48 items take two launches."""
    from vibravox_amd.collate import noisy_bwe_collate

    g = torch.Generator().manual_seed(5)
    clips = [torch.randn(int(n), generator=g) for n in torch.randint(50, 900, (60,), generator=g)]
    want = C.noisy_bwe_collate([{"audio_body_conducted": c} for c in clips], 16000, "pad", False)
    got = noisy_bwe_collate([{"audio_body_conducted": c.cuda()} for c in clips], 16000, "pad", False)
    assert torch.equal(got["audio_body_conducted"].cpu(), want["audio_body_conducted"])
