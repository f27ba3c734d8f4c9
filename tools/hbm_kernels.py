"""Achieved HBM rate of the kernels whose roof IS HBM, from a rocprofv3 --kernel-trace --stats summary of the bench command
(profiles/<tag>_rocprofv3_kernel_stats.csv: calls and average duration per kernel, IN the step -- the launches share the GPU with the other
streams' kernels) and the algorithmic bytes of ONE launch at BASELINE config 2 (batch 32 x 31968 samples; SURVEY 8d figures per unit, stated
per row below).  Rate = bytes per launch / average duration, against 8 TB/s (MI355X_MICROARCH.md; ~6.3 TB/s is what a plain copy reaches).
A row above 1.0 of the peak is a byte-count error, not a measurement: the script fails on it.
Usage: python tools/hbm_kernels.py profiles/r04_rocprofv3_kernel_stats.csv"""
import csv, sys

rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
B, T = 32, 31968
R2 = 2 * B   # discriminator rows (enhanced + reference)
# embedding elements per row that enter the feature-matching loss, per chain (the feature-matching sums run as ONE launch per chain:
# disc_engine_bl.forward, body(i)) -- layers 0 .. n - 2 of each sub-discriminator
mel = [16 * 31968, 64 * 7992, 256 * 1998, 1024 * 500, 1024 * 125, 1024 * 125]
pq = [[24 * 7992, 48 * 3997, 96 * 1999, 192 * 1000, 384 * 500, 768 * 250, 768 * 250],
      [24 * 7992, 48 * 3993, 96 * 1994, 192 * 994, 384 * 494, 768 * 244, 768 * 244],
      [24 * 7992, 48 * 3989, 96 * 1989, 192 * 989, 384 * 489, 768 * 239, 768 * 239]]
fm_chain = [sum(mel)] + [sum(p) for p in pq]
fm_avg = sum(fm_chain) / 4.0           # the stats file averages the four per-chain launches of a step
n32, n64, n128 = B * 32 * 7992, B * 64 * 3996, B * 128 * 999
spec = [
    # kernel name (prefix match), bytes per launch, what
    ("eben::bl_fm_partial_kernel", R2 * fm_avg * 4, "feature-matching sums, one launch per chain (average of the four): its embeddings of 64 rows, hi + lo planes (4 B per element), read once"),
    ("eben::fir_decimate_kernel", None, None),   # several shapes share the name (PQMF analysis at 2 / 4 bands, A-weighting FIR, adjoints): see ru / layer tables
    ("void eben::ru3_fwd_kernel<1, 4, 3, 2, true>", n32 * 12.125, "ResidualUnit forward C = 32, L = 7992, bundle-saving form: x read 4, y written 4, xin / h planes 2 + 2, sign bytes 1/8 (B per element)"),
    ("void eben::ru3_fwd_kernel<2, 4, 3, 2, true>", n64 * 12.125, "ResidualUnit forward C = 64, L = 3996"),
    ("void eben::ru3_fwd_kernel<4, 4, 3, 2, true>", n128 * 12.125, "ResidualUnit forward C = 128, L = 999"),
    ("void eben::rubl_bwd_kernel<1,", n32 * 12.125, "ResidualUnit input gradients C = 32: g_y read 4, sign bytes 1/8, g_x written 4, g_z / g_h planes 2 + 2"),
    ("void eben::rubl_bwd_kernel<2,", n64 * 12.125, "ResidualUnit input gradients C = 64"),
    ("void eben::rubl_bwd_kernel<4,", n128 * 12.125, "ResidualUnit input gradients C = 128"),
    ("void eben::rubl_dw_kernel<1,", n32 * 8, "ResidualUnit weight gradients C = 32: the four bf16 planes g_z, h, g_h, xin read once (8 B per element) + slabs"),
    ("void eben::rubl_dw_kernel<2,", n64 * 8, "ResidualUnit weight gradients C = 64"),
    ("void eben::rubl_dw_kernel<4,", n128 * 8, "ResidualUnit weight gradients C = 128"),
    ("void eben::ru3_fwd_kernel<1, 4, 3, 2, false>", n32 * 16, "ResidualUnit forward C = 32, fp32-at-rest form (fp32 backward plans / inference without h, u): 4 passes x 4 B"),
    ("void eben::ru3_bwd_kernel<1,", n32 * 20, "ResidualUnit input gradients C = 32, fp32 at rest: 5 passes"),
    ("void eben::ru_dw_kernel<1,", n32 * 20, "ResidualUnit weight gradients C = 32, fp32 at rest: 5 reads"),
    ("eben::fir_interp_sum_kernel", None, None),
]
print(f"{'kernel':58s} {'calls':>6s} {'avg us':>8s} {'MB/launch':>10s} {'TB/s':>6s} {'of 8':>5s}  what")
bad = []
for name, nbytes, what in spec:
    if nbytes is None:
        continue
    hit = [r for n, r in rows.items() if n.startswith(name)]
    if not hit:
        continue
    r = hit[0]
    us = float(r["AverageNs"]) / 1e3
    frac = nbytes / us / 1e6 / 8
    if frac > 1.0:
        bad.append((r["Name"], frac))
    print(f"{r['Name'][:58]:58s} {r['Calls']:>6s} {us:8.1f} {nbytes / 1e6:10.1f} {nbytes / us / 1e6:6.2f} {frac:5.2f}  {what}")
# the PQMF filter-bank kernels (north_star names them): one name, several shapes -- the per-shape rates come from tools/pqmf_bench.py
# (stand-alone launches); here the step's launches together against the bytes they move together:
#   fir_decimate: analysis of the corrupted clip at 2 bands (4 + 2 B per input sample... (32, 1, 31968) in, (32, 2, 7992) out) and of the
#   reference at 4 bands ((32, 4, 7992) out), the A-weighting FIR of the MRSTFT loss on both signals, and the synthesis adjoints of the
#   balancing seeds; fir_interp_sum: PQMF synthesis (+ band sum) of the enhanced bands and the analysis adjoint.
fd = [r for n, r in rows.items() if n.startswith("eben::fir_decimate_kernel")]
if fd:
    print(f"{'eben::fir_decimate_kernel (all shapes)':58s} {fd[0]['Calls']:>6s} {float(fd[0]['AverageNs']) / 1e3:8.1f}       (see profiles/*_pqmf.txt for the per-shape rates)")
fi = [r for n, r in rows.items() if n.startswith("eben::fir_interp_sum_kernel")]
if fi:
    print(f"{'eben::fir_interp_sum_kernel (all shapes)':58s} {fi[0]['Calls']:>6s} {float(fi[0]['AverageNs']) / 1e3:8.1f}       (see profiles/*_pqmf.txt for the per-shape rates)")
# Adam: all launches of a step together (48 tensors per launch): 28 B per parameter (p, g, m, v read; p, m, v written)
ad = [r for n, r in rows.items() if n.startswith("eben::adam_kernel")]
if ad:
    tot_ns, calls = float(ad[0]["TotalDurationNs"]), int(ad[0]["Calls"])
    params = 1945984 + 23161344
    steps = calls / 4.0   # four launches per step at this size (1 generator + 3 discriminator)
    frac = params * 28 / (tot_ns / steps) / 1e3 / 8
    if frac > 1.0:
        bad.append(("adam", frac))
    print(f"{'eben::adam_kernel (all launches of a step)':58s} {calls:>6d} {tot_ns / steps / 1e3:8.1f} {params * 28 / 1e6:10.1f} {params * 28 / (tot_ns / steps) / 1e3:6.2f} "
          f"{frac:5.2f}  multi-tensor Adam over 25.1 M parameters, 28 B each")
if bad:
    sys.exit(f"rows above the HBM peak (byte count wrong): {bad}")
