"""Fused ResidualUnit launches alone on the device, per channel count / dilation (batch 32, the generator's shapes at BASELINE config 2):
us per launch for the fp32-at-rest kernels (ru_split.hip / ru_dw.hip, per math mode) and for the bundle-layout kernels the bf16 generator
backward uses (ru_bl.hip: forward that saves bf16 bundles, input gradients, weight gradients), with the HBM time of each launch's
algorithmic bytes at 6.3 TB/s (the measured streaming rate; 8 TB/s is the datasheet peak) and the ratio to it.
Bytes per element: fp32-at-rest forward / backward 16, weight gradients 20; bundle layout 12.125 / 12.125 / 8.
Usage: python tools/ru_bench.py [--iters 20] [--batch 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vibravox_amd._lib import check, load, ptr, stream

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--bl-only", action="store_true")
args = ap.parse_args()
lib, dev = load(), torch.device("cuda")
names = {0: "f32", 4: "bf16x6", 3: "bf16x3", 1: "bf16"}
RATE = 6.3e12


def timed(fn):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


print(f"{'shape':20s} " + ("" if args.bl_only else " ".join(f"{'fwd ' + n:>10s}" for n in names.values()) + "  " + " ".join(f"{'bwd ' + n:>10s}" for n in names.values()) + "  dw bf16 / x6 |")
      + "   bundle layout: fwd (x HBM)   bwd (x HBM)    dw (x HBM)      [us; HBM time at 6.3 TB/s]")
tot = {"fwd": 0.0, "bwd": 0.0, "dw": 0.0, "fwd0": 0.0, "bwd0": 0.0, "dw0": 0.0}
for c, l in ((32, 7992), (64, 3996), (128, 999)):
    for d in (1, 3, 9):
        b = args.batch
        x = torch.randn(b, c, l, device=dev); gy = torch.randn(b, c, l, device=dev)
        vd = torch.randn(c, c, 3, device=dev) / (3 * c) ** 0.5; vp = torch.randn(c, c, 1, device=dev) / c ** 0.5
        y, h, u, gx, gh = (torch.empty_like(x) for _ in range(5))
        row = {}
        imgs = {}
        for which in ("fwd", "bwd"):
            for mm in names:
                img = torch.empty(lib.eben_ru_packed_floats_ex(c, mm), dtype=torch.float32, device=dev)
                check(lib.eben_ru_pack_ex(c, mm, 0 if which == "fwd" else 1, ptr(vd), None, ptr(vp), None, ptr(img), stream()))
                imgs[(which, mm)] = img
                if args.bl_only and not (which == "fwd" and mm == 4):
                    continue
                if which == "fwd":
                    row[(which, mm)] = timed(lambda: check(lib.eben_ru_fwd_ex(mm, b, c, l, d, ptr(x), 1.0, 0.01, ptr(img), ptr(y), ptr(h), ptr(u), stream())))
                else:
                    row[(which, mm)] = timed(lambda: check(lib.eben_ru_bwd_ex(mm, b, c, l, d, ptr(gy), ptr(u), 0.01, None, 1.0, None, ptr(img), ptr(gx), ptr(gh), stream())))
        if not args.bl_only:
            nslab = lib.eben_ru_dw_slabs(b, c, l)
            sp = torch.empty(nslab * c * c, device=dev); sd = torch.empty(nslab * 3 * c * c, device=dev)
            for mm in (1, 4):
                row[("dw", mm)] = timed(lambda: check(lib.eben_ru_dw(mm, b, c, l, d, ptr(gy), ptr(u), 0.01, ptr(h), ptr(gh), ptr(x), 1.0, ptr(sp), ptr(sd), stream())))
        # bundle layout
        xb = torch.empty(b, c // 8, l, 8, dtype=torch.bfloat16, device=dev); hb, gzb, ghb = (torch.empty_like(xb) for _ in range(3))
        um = torch.empty(b, c // 8, l, dtype=torch.uint8, device=dev)
        f_bl = timed(lambda: check(lib.eben_rubl_fwd(4, b, c, l, d, ptr(x), 1.0, 0.01, ptr(imgs[("fwd", 4)]), ptr(y), xb.data_ptr(), hb.data_ptr(), um.data_ptr(), stream())))
        b_bl = timed(lambda: check(lib.eben_rubl_bwd(b, c, l, d, ptr(gy), um.data_ptr(), 0.01, None, 1.0, None, ptr(imgs[("bwd", 1)]), ptr(gx), gzb.data_ptr(), ghb.data_ptr(), stream())))
        ns = lib.eben_rubl_dw_slabs(b, c, l)
        sp2 = torch.empty(ns * c * c, device=dev); sd2 = torch.empty(ns * 3 * c * c, device=dev)
        w_bl = timed(lambda: check(lib.eben_rubl_dw(b, c, l, d, gzb.data_ptr(), hb.data_ptr(), ghb.data_ptr(), xb.data_ptr(), ptr(sp2), ptr(sd2), stream())))
        n = x.numel()
        t_f, t_b, t_w = 12.125 * n / RATE * 1e6, 12.125 * n / RATE * 1e6, 8 * n / RATE * 1e6
        tot["fwd"] += f_bl; tot["bwd"] += b_bl; tot["dw"] += w_bl
        if not args.bl_only:
            tot["fwd0"] += row[("fwd", 4)]; tot["bwd0"] += row[("bwd", 1)]; tot["dw0"] += row[("dw", 1)]
        left = "" if args.bl_only else (" ".join(f"{row[('fwd', m)]:10.1f}" for m in names) + "  " + " ".join(f"{row[('bwd', m)]:10.1f}" for m in names)
                                        + f"  {row[('dw', 1)]:6.1f} / {row[('dw', 4)]:5.1f} |")
        print(f"C {c:3d} L {l:5d} d {d}   " + left + f"   {f_bl:6.1f} ({f_bl / t_f:4.2f}x {t_f:4.1f})  {b_bl:6.1f} ({b_bl / t_b:4.2f}x {t_b:4.1f})  {w_bl:6.1f} ({w_bl / t_w:4.2f}x {t_w:4.1f})   slabs {ns}")
print(f"sum over the 9 shapes x 2 (the generator's 18 units), us: bundle layout fwd {2 * tot['fwd']:.0f} bwd {2 * tot['bwd']:.0f} dw {2 * tot['dw']:.0f}"
      + ("" if args.bl_only else f"   fp32 at rest: fwd x6 {2 * tot['fwd0']:.0f} bwd bf16 {2 * tot['bwd0']:.0f} dw bf16 {2 * tot['dw0']:.0f}"))
