"""Where a block of the bundle-layout tap-conv forward goes, in shader cycles (s_memtime stamps of wave 0, tapconv3.hip EBEN_T3_DBG & 512):
set-up, prologue issue, first landing, k-step loop, epilogue arithmetic, stores gone.  Needs the scratch build:
  tools/build_variant.sh t3stamp tapconv3 -DEBEN_T3_DBG=512
  EBEN_HIP_LIB=vibravox_amd/lib/var/libeben_t3stamp.so python tools/t3_stamps.py --cin 24 --cout 48 --k 7 --stride 2 --groups 4 --length 7994 --math bf16x3"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vibravox_amd import ops
from vibravox_amd._lib import check, load

ap = argparse.ArgumentParser()
for k, v in (("cin", 24), ("cout", 48), ("k", 7), ("stride", 2), ("dil", 1), ("groups", 4), ("length", 7994), ("rows", 64)):
    ap.add_argument("--" + k, type=int, default=v)
ap.add_argument("--math", default="bf16x3")
a = ap.parse_args()
lib = load(); dev = torch.device("cuda")
pad = (a.k - 1) * a.dil // 2
spec = ops.ConvSpec(c_in=a.cin, c_out=a.cout, ksize=a.k, stride=a.stride, dilation=a.dil, groups=a.groups, pad_l=pad, pad_r=pad, out_slope=0.2)
math = {"bf16": ops.MATH_BF16, "bf16x3": ops.MATH_BF16X3}[a.math] | 0x100
d = ops.conv_desc(spec, a.rows, a.length, math)
g = torch.Generator().manual_seed(1)
xh = torch.randn(a.rows, a.cin // 8, a.length, 8, generator=g).bfloat16().to(dev)
xl = (torch.randn(a.rows, a.cin // 8, a.length, 8, generator=g) * 2.0 ** -9).bfloat16().to(dev)
v = torch.randn(spec.weight_shape(), generator=g).to(dev) * 0.05
scale = torch.ones(a.cout, device=dev)
bias = torch.zeros(a.cout, device=dev)
wp = torch.empty(lib.eben_conv1d_packed_floats(ctypes.byref(d), 0), dtype=torch.float32, device=dev)
ops.conv1d_pack(d, v, scale, wp, None)
yh = torch.empty(a.rows, a.cout // 8, d.l_out, 8, dtype=torch.bfloat16, device=dev)
yl = torch.empty_like(yh)
st = torch.cuda.current_stream().cuda_stream
lib.eben_debug_t3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    lib.eben_debug_t3_stamps(None, 0)
    e0.record()
    check(lib.eben_bl_conv1d_fwd(ctypes.byref(d), xh.data_ptr(), xl.data_ptr() if a.math == "bf16x3" else None, wp.data_ptr(), bias.data_ptr(), yh.data_ptr(), yl.data_ptr(), st), "fwd")
    e1.record()
torch.cuda.synchronize()
rows = 65536
buf = (ctypes.c_ulonglong * (rows * 16))()
print("rc", lib.eben_debug_t3_stamps(buf, rows), "launch", round(e0.elapsed_time(e1) * 1e3, 1), "us (with the zeroing memset in front)")
s = np.frombuffer(buf, dtype=np.uint64).reshape(rows, 16).astype(np.float64)
s = s[s[:, 7] > 0]
print("blocks", len(s))
names = ("set-up (addresses, k-step table)", "prologue issue (tile + first weight chunk)", "to the first barrier", "first barrier (pieces landed)", "k-step loop", "epilogue arithmetic", "stores gone")
for i, nme in enumerate(names):
    dlt = s[:, i + 1] - s[:, i]
    print(f"{nme:44s} {dlt.mean():8.0f} cycles (median {np.median(dlt):.0f})")
tot = s[:, 7] - s[:, 0]
print(f"{'block total':44s} {tot.mean():8.0f}")
