"""Where a bl_dw block's chunk goes, in cycles (s_memtime of wave 0): waiting for its LDS-DMA pieces, at the barrier, issuing the next
chunk's pieces, fragment reads + MFMAs.  Needs the scratch build with the stamps:
  tools/build_variant.sh stamp bl_dw -DEBEN_BLDW_STAMP=1;  EBEN_HIP_LIB=vibravox_amd/lib/var/libeben_stamp.so python tools/bldw_stamps.py melgan.4"""
import ctypes, os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["layer_bench_bl.py", "--only", sys.argv[1] if len(sys.argv) > 1 else "melgan.4", "--iters", "3"]
import runpy
try:
    runpy.run_path(os.path.join(os.path.dirname(__file__), "layer_bench_bl.py"), run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
from vibravox_amd._lib import load
lib = load()
buf = (ctypes.c_ulonglong * (8192 * 8))()
lib.eben_debug_bldw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
print("rc", lib.eben_debug_bldw_stamps(buf, 8192 * 8))
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
a = a[a[:, 4] > 0]
n = a[:, 4]
print("blocks", len(a), "chunks per block", n.mean())
for i, name in enumerate(("wait for pieces", "barrier", "issue next", "fragment reads + MFMAs")):
    print(f"{name:26s} {np.mean(a[:, i] / n):8.0f} cycles per chunk (min block {np.min(a[:, i] / n):.0f}, max {np.max(a[:, i] / n):.0f})")
print(f"{'loop total per chunk':26s} {np.mean(a[:, 5] / n):8.0f}")

