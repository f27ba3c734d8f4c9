"""Fused ResidualUnit launches alone on the device, per channel count / dilation / math mode (batch 32, the generator's shapes at
BASELINE config 2): us per launch and the rate of the 4 tensor passes each launch makes (forward: read x, write y, h, u; backward:
read g_y, u, write g_x, g_h).  Usage: python tools/ru_bench.py [--iters 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vibravox_amd._lib import check, load, ptr, stream

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--batch", type=int, default=32)
args = ap.parse_args()
lib, dev = load(), torch.device("cuda")
names = {0: "f32", 4: "bf16x6", 3: "bf16x3", 1: "bf16"}
print(f"{'shape':22s} " + " ".join(f"{'fwd ' + n:>12s}" for n in names.values()) + "   " + " ".join(f"{'bwd ' + n:>12s}" for n in names.values()) + "   (us; 4 passes at 6.3 TB/s)")
for c, l in ((32, 7992), (64, 3996), (128, 999)):
    for d in (1, 3, 9):
        b = args.batch
        x = torch.randn(b, c, l, device=dev); gy = torch.randn(b, c, l, device=dev)
        vd = torch.randn(c, c, 3, device=dev) / (3 * c) ** 0.5; vp = torch.randn(c, c, 1, device=dev) / c ** 0.5
        y, h, u, gx, gh = (torch.empty_like(x) for _ in range(5))
        row = {}
        for which in ("fwd", "bwd"):
            for mm in names:
                img = torch.empty(lib.eben_ru_packed_floats_ex(c, mm), dtype=torch.float32, device=dev)
                check(lib.eben_ru_pack_ex(c, mm, 0 if which == "fwd" else 1, ptr(vd), None, ptr(vp), None, ptr(img), stream()))
                def run():
                    if which == "fwd":
                        check(lib.eben_ru_fwd_ex(mm, b, c, l, d, ptr(x), 1.0, 0.01, ptr(img), ptr(y), ptr(h), ptr(u), stream()))
                    else:
                        check(lib.eben_ru_bwd_ex(mm, b, c, l, d, ptr(gy), ptr(u), 0.01, None, 1.0, None, ptr(img), ptr(gx), ptr(gh), stream()))
                for _ in range(3): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters): run()
                e1.record(); torch.cuda.synchronize()
                row[(which, mm)] = e0.elapsed_time(e1) / args.iters * 1e3
        nslab = lib.eben_ru_dw_slabs(b, c, l)
        sp = torch.empty(nslab * c * c, device=dev); sd = torch.empty(nslab * 3 * c * c, device=dev)
        for mm in (1, 4):
            def run_dw():
                check(lib.eben_ru_dw(mm, b, c, l, d, ptr(gy), ptr(u), 0.01, ptr(h), ptr(gh), ptr(x), 1.0, ptr(sp), ptr(sd), stream()))
            for _ in range(3): run_dw()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters): run_dw()
            e1.record(); torch.cuda.synchronize()
            row[("dw", mm)] = e0.elapsed_time(e1) / args.iters * 1e3
        ideal = 4 * x.numel() * 4 / 6.3e12 * 1e6
        print(f"C {c:3d} L {l:5d} d {d}      " + " ".join(f"{row[('fwd', m)]:12.1f}" for m in names) + "   " + " ".join(f"{row[('bwd', m)]:12.1f}" for m in names) + f"   ({ideal:.1f})   dw bf16 {row[('dw', 1)]:.1f} x6 {row[('dw', 4)]:.1f} (5 reads: {1.25 * ideal:.1f})")
