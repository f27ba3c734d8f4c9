import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np, torch
from oracle import augment_oracle as A
from vibravox_amd.augment import pitch_shift
from formula import formula_tensor
for steps, shape in [(-2, (3, 1, 4000)), (1, (2, 1, 7777)), (4, (2, 2, 3001)), (-4, (1, 16000)), (6, (4, 32000)), (-3, (4, 32000))]:
    x = formula_tensor(f"ps/{steps}", shape)
    got = pitch_shift(x.cuda(), 16000, steps).cpu()
    ref = A.pitch_shift(x.numpy(), 16000, steps)
    err = np.abs(got.double().numpy() - ref); sc = float(np.abs(ref).max())
    print(steps, shape, "rms %.3e max %.3e" % (np.sqrt((err**2).mean())/sc, err.max()/sc))
