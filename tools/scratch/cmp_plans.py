import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import bench
from test_gpu_models import _one_step, _rel
DEV = torch.device("cuda")
def make():
    return bench.build_module(DEV, 1234), bench.synthetic_batch(32, 32000, 1234, DEV)
f = _one_step(make, "f32")
a = _one_step(make, "bf16", "bf16", stft_math="folded_x3")
b = _one_step(make, "bf16_bl", "bf16", stft_math="folded_x3")
def rep(tag, x, y):
    worst = max(abs(y[1][k] - v) / abs(v) for k, v in x[1].items())
    wk = max(x[1], key=lambda k: abs(y[1][k] - x[1][k]) / abs(x[1][k]))
    norms = float(((x[2] - y[2]).abs() / x[2].abs()).max())
    print(f"{tag}: worst logged {worst:.2e} ({wk}), norms {norms:.2e}, G grad {_rel(x[3][0], y[3][0]):.2e}, D grad {_rel(x[3][1], y[3][1]):.2e}")
rep("bf16 vs f32", f, a); rep("bf16_bl vs f32", f, b); rep("bf16_bl vs bf16", a, b)
