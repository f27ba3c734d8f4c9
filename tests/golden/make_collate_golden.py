"""Generate tests/golden/collate_golden.npz from the REFERENCE functions of vibravox/utils.py
(pad_audio, set_audio_duration, mix_speech_and_noise_without_rescaling).  Build container only
(needs /root/reference); ``torchaudio.functional.lowpass_biquad`` (imported at utils.py:4, unused here)
is stubbed in a temp dir.  Inputs are the closed-form clips of formula.py with ragged lengths.

Usage:  python tests/golden/make_collate_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from formula import formula_tensor  # noqa: E402

LENGTHS = [(1250, 1250, 2250), (775, 775, 2000), (1025, 1025, 1125), (250, 250, 251), (600, 600, 1750)]   # speech, air, noise


def items():
    out = []
    for i, (ls, la, ln) in enumerate(LENGTHS):
        out.append({"audio_body_conducted": formula_tensor(f"col/bc{i}", (ls,)), "audio_airborne": formula_tensor(f"col/air{i}", (la,)),
                    "audio_body_conducted_speechless_noisy": formula_tensor(f"col/noise{i}", (ln,))})
    return out


def main():
    stub = tempfile.mkdtemp(prefix="ta_stub_")
    os.makedirs(os.path.join(stub, "torchaudio"))
    open(os.path.join(stub, "torchaudio", "__init__.py"), "w").close()
    with open(os.path.join(stub, "torchaudio", "functional.py"), "w") as f:
        f.write("def lowpass_biquad(*a, **k):\n    raise NotImplementedError\n")
    sys.path.insert(0, stub)
    sys.path.insert(0, "/root/reference")
    from vibravox.utils import mix_speech_and_noise_without_rescaling, pad_audio, set_audio_duration

    g = {}
    batch = items()
    for seed in (0, 1):
        torch.manual_seed(seed)
        noisy, sliced = mix_speech_and_noise_without_rescaling([b["audio_body_conducted"] for b in batch],
                                                               [b["audio_body_conducted_speechless_noisy"] for b in batch])
        for i, (n, s) in enumerate(zip(noisy, sliced)):
            g[f"mix/seed{seed}/noisy{i}"] = n.numpy()
            g[f"mix/seed{seed}/slice{i}"] = s.numpy()
        for det in (False, True):
            for i, (n, b) in enumerate(zip(noisy, batch)):
                a, ab = set_audio_duration(audio=n, desired_samples=800, audio_bis=b["audio_airborne"], deterministic=det)
                g[f"dur/seed{seed}/det{int(det)}/bc{i}"] = a.numpy()
                g[f"dur/seed{seed}/det{int(det)}/air{i}"] = ab.numpy()
    g["pad/10_16"] = pad_audio(torch.arange(10.0), 16).numpy()
    g["pad/7_7"] = pad_audio(torch.arange(7.0), 7).numpy()
    g["pad/2x5_9"] = pad_audio(torch.arange(10.0).reshape(2, 5), 9).numpy()
    np.savez_compressed(os.path.join(HERE, "collate_golden.npz"), **g)
    print("wrote", len(g), "arrays")


if __name__ == "__main__":
    main()
