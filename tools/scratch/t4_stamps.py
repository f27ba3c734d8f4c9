"""Cycle stamps of tap4_kernel (scratch library built with -DEBEN_T4_DBG=32): per-segment cycles of the chunk loop of one layer.
Usage: EBEN_HIP_LIB=vibravox_amd/lib/var/libeben_t4st.so python tools/scratch/t4_stamps.py melgan.4"""
import ctypes, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
layer = sys.argv[1] if len(sys.argv) > 1 else "melgan.4"
import torch
from vibravox_amd._lib import load
sys.argv = ["layer_bench_bl.py", "--only", layer, "--iters", "2"]
import runpy
lib = load()
# run fwd only by monkeypatching is overkill: the bench runs fwd, dx, dw in that order; the stamps hold the LAST tap4 launch (dx) -- so
# run twice with EBEN_BIG_MIN_KS_DX to keep dx off the big kernel when asked
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "layer_bench_bl.py"), run_name="__main__")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8))()
lib.eben_debug_t4_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.eben_debug_t4_stamps(buf, 256) == 0
a = np.array(buf, dtype=np.float64).reshape(256, 8)
a = a[a[:, 5] > 0]
n = a[:, 5]
print(f"blocks {len(a)}  tiles/block {a[0,6]:.0f}  chunks/tile {a[0,7]:.0f}")
print(f"per chunk: A {np.mean(a[:,0]/n):.0f}  wait+barrier {np.mean(a[:,1]/n):.0f}  B {np.mean(a[:,2]/n):.0f}  sum {np.mean((a[:,0]+a[:,1]+a[:,2])/n):.0f} cycles")
print(f"epilogue per tile {np.mean(a[:,3]/a[:,6]):.0f}  kernel {np.mean(a[:,4]):.0f} cycles (s_memtime ticks)")
