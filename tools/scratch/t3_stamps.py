"""Scratch: block-lifetime stamps of tap3 launches (variant build with -DEBEN_T3_DBG=512, loaded through EBEN_HIP_LIB)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vibravox_amd import ops
from vibravox_amd._lib import check, load, ptr, stream
from vibravox_amd.disc_engine_bl import Planes, _ChainBL
from vibravox_amd.lightning_modules.eben import DISC_MATH_PLANS
from vibravox_amd.torch_modules.dnn.eben_discriminator import DiscriminatorEBENMultiScales

lib = load()
lib.eben_debug_t3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda")
torch.manual_seed(0)
disc = DiscriminatorEBENMultiScales(q=4, min_channels=24).to(dev)
plan = DISC_MATH_PLANS["bf16_bl"]
half = 32
r2, r4 = 64, 128
st = stream()
seg_map = (ctypes.c_int * 4)(0, 0, 0, 1)
sums = torch.tensor([3.0, 7.0], device=dev)
import numpy as np
ROWS = 65536
buf = np.zeros((ROWS, 16), dtype=np.uint64)
NAMES = ["args", "issued", "staged", "barrier", "loop", "epilogue", "stores"]


def report(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.eben_debug_t3_stamps(None, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    lib.eben_debug_t3_stamps(buf.ctypes.data, ROWS)
    t = buf.astype(np.int64)
    t = t[t[:, 7] > 0]
    n = len(t)
    life = t[:, 7] - t[:, 0]
    d = [np.median(t[:, k + 1] - t[:, k]) for k in range(7)]
    # per-CU concurrency: CU key = (xcc, se, sh, cu) from HW_ID; per-XCC clocks may differ, so spans are taken per CU
    key = (t[:, 9] & 0xf) * 65536 + ((t[:, 8] >> 8) & 0xff)
    conc, spans, per_cu = [], [], []
    for k in np.unique(key):
        r = t[key == k]
        span = r[:, 7].max() - r[:, 0].min()
        conc.append((r[:, 7] - r[:, 0]).sum() / span)
        spans.append(span)
        per_cu.append(len(r))
    print(f"{name:14s} {us:7.1f} us blk {n:6d} CUs {len(conc):3d} blk/CU {np.mean(per_cu):5.1f} CU span med {np.median(spans):8.0f} max {np.max(spans):8.0f} | "
          + " ".join(f"{nm} {v:5.0f}" for nm, v in zip(NAMES, d)) + f" | life med {np.median(life):6.0f} p90 {np.percentile(life, 90):6.0f} | blocks in flight per CU {np.mean(conc):4.1f}")


chains = [("pqmf0", _ChainBL(disc.pqmf_discriminators[0].discriminator, plan["pqmf"]), 4, 7992), ("melgan", _ChainBL(disc.melgan_discriminator.discriminator, plan["melgan"]), 1, 31968)]
for cname, ch, c_in, l_in in chains:
    cur_len = ch.head_out_len(l_in)
    for i in range(1, len(ch.layers) - 1):
        lay = ch.layers[i]; sp = lay.spec
        d = ops.conv_desc(sp, r2, cur_len, lay.math_fwd)
        y = Planes(r2, sp.c_out, d.l_out, dev)
        xin = Planes.from_f32(torch.randn(r2, sp.c_in, cur_len, device=dev))
        split = (lay.math_fwd & 0xff) == ops.MATH_BF16X3
        _, _, bias = lay.params()
        wp = lay.packed(0, r2, cur_len)
        report(f"{cname}.{i} fwd", lambda: check(lib.eben_bl_conv1d_fwd(ctypes.byref(d), xin.hi.data_ptr(), xin.lo.data_ptr() if split else None, ptr(wp), ptr(bias), y.hi.data_ptr(), y.lo.data_ptr(), st)))
        d4 = ops.conv_desc(lay.spec_lin, r4, cur_len, lay.math_dx)
        g = Planes.from_f32(torch.randn(r4, sp.c_out, d.l_out, device=dev), lo=False)
        gp = Planes(r4, sp.c_in, cur_len, dev, lo=False)
        wpb = lay.packed(1, r4, cur_len)
        report(f"{cname}.{i} dx", lambda: check(lib.eben_bl_conv1d_bwd_dx(ctypes.byref(d4), g.hi.data_ptr(), ptr(wpb), xin.hi.data_ptr(), xin.lo.data_ptr(), 0.2, half, seg_map, half, half, ptr(sums), 0.1, gp.hi.data_ptr(), None, st)))
        cur_len = d.l_out
