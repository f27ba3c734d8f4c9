"""Generate tests/golden/augment_golden.npz from the REFERENCE classes
(vibravox/torch_modules/dsp/data_augmentation.py, dsp/time_masking_waveform.py).  Build container only (needs
/root/reference).  ``torchaudio.transforms`` (imported at data_augmentation.py:4) is not installed: it is stubbed in a temp
dir, so only draw sequences that never reach SpeedPerturbation / PitchShift are frozen -- the order of the random draws and
the time masking are the reference's own; the sinc resampling has no reference to be pinned to here.

Usage:  python tests/golden/make_augment_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from formula import formula_tensor  # noqa: E402

CASES = [  # (seed, p_data_augmentation, p_time_masking, shape)
    (0, 1.0, 1.0, (3, 1, 4000)), (1, 1.0, 1.0, (2, 1, 777)), (2, 0.5, 1.0, (3, 1, 4000)), (3, 1.0, 0.5, (1, 1, 16000)),
    (4, 1.0, 1.0, (4, 1, 1234)), (5, 0.5, 0.5, (3, 1, 4000)), (6, 0.5, 0.5, (3, 1, 4000)), (7, 0.5, 0.5, (3, 1, 4000)),
]


def inputs(i, shape):
    return formula_tensor(f"aug/{i}/a", shape), formula_tensor(f"aug/{i}/b", shape)


def zero_run(x, y):
    """(first, count, float64 sum of y): y must equal x except for ONE run of zeros along time, the same in every row."""
    diff = (x != y).reshape(-1, x.shape[-1])
    cols = diff.any(dim=0).nonzero().flatten()
    if cols.numel() == 0:
        assert torch.equal(x, y)
        return np.array([0, 0, float(y.double().sum())])
    first, last = int(cols[0]), int(cols[-1])
    ref = x.clone()
    ref[..., first:last + 1] = 0
    assert torch.equal(ref, y), "not a single zero run"
    return np.array([first, last + 1 - first, float(y.double().sum())])


def main():
    stub = tempfile.mkdtemp(prefix="ta_stub_")
    os.makedirs(os.path.join(stub, "torchaudio"))
    open(os.path.join(stub, "torchaudio", "__init__.py"), "w").close()
    with open(os.path.join(stub, "torchaudio", "transforms.py"), "w") as f:
        f.write("class SpeedPerturbation:\n    def __init__(self, *a, **k):\n        raise NotImplementedError\n"
                "class PitchShift:\n    def __init__(self, *a, **k):\n        raise NotImplementedError\n")
    sys.path.insert(0, stub)
    sys.path.insert(0, "/root/reference")
    from vibravox.torch_modules.dsp.data_augmentation import WaveformDataAugmentation
    from vibravox.torch_modules.dsp.time_masking_waveform import TimeMaskingBlockWaveform

    g = {}
    for i, (seed, p_aug, p_mask, shape) in enumerate(CASES):
        aug = WaveformDataAugmentation(16000, p_data_augmentation=p_aug, p_speed_perturbation=0.0, p_pitch_shift=0.0, p_time_masking=p_mask)
        a, b = inputs(i, shape)
        torch.manual_seed(seed)
        oa, ob = aug(a.clone(), b.clone())
        g[f"aug/{i}/a"], g[f"aug/{i}/b"] = zero_run(a, oa), zero_run(b, ob)   # the outputs are the inputs with one run zeroed
        g[f"aug/{i}/next_draw"] = torch.rand(1).numpy()   # the generator state after the call: the number of draws taken
    for pct in (1, 3, 8, 50):
        x = formula_tensor(f"tm/{pct}", (2, 2, 1000))
        torch.manual_seed(10 + pct)
        g[f"tm/{pct}"] = zero_run(x, TimeMaskingBlockWaveform(masking_percentage=pct)(x.clone()))
    np.savez_compressed(os.path.join(HERE, "augment_golden.npz"), **g)
    print("wrote", len(g), "arrays")


if __name__ == "__main__":
    main()
