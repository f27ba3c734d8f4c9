"""Achieved HBM rate of the kernels whose roof IS HBM, from a rocprofv3 --kernel-trace --stats summary of the bench command
(profiles/<tag>_rocprofv3_kernel_stats.csv: calls and average duration per kernel) and the algorithmic bytes of each launch at BASELINE
config 2 (batch 32 x 31968 samples; SURVEY 8d figures per unit, stated per row below).  Rate = bytes per launch / average duration,
against 8 TB/s (MI355X_MICROARCH.md; ~6.3 TB/s is what a plain copy reaches).
Usage: python tools/hbm_kernels.py profiles/r03_rocprofv3_kernel_stats.csv"""
import csv, sys

rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
B, T = 32, 31968
R2 = 2 * B   # discriminator rows (enhanced + reference)
mel = [16 * 31968, 64 * 7992, 256 * 1998, 1024 * 500, 1024 * 125, 1024 * 125]
pq = [24 * 7992, 48 * 3997, 96 * 1999, 192 * 1000, 384 * 500, 768 * 250, 768 * 250]
emb = sum(mel) + 3 * sum(pq)           # embedding elements per row that enter the feature-matching loss
spec = [
    # kernel name (prefix match), bytes per launch, what
    ("eben::bl_fm_partial_kernel", R2 * emb * 4, "feature-matching sums: every embedding of 64 rows, hi + lo planes (4 B per element), read once"),
    ("void eben::ru3_fwd_kernel<1,", B * 32 * 7992 * 4 * 4, "ResidualUnit forward C = 32, L = 7992: x read, y / h / u written (4 passes x 4 B)"),
    ("void eben::ru3_fwd_kernel<2,", B * 64 * 3996 * 4 * 4, "ResidualUnit forward C = 64, L = 3996"),
    ("void eben::ru3_fwd_kernel<4,", B * 128 * 999 * 4 * 4, "ResidualUnit forward C = 128, L = 999"),
    ("void eben::ru3_bwd_kernel<1,", B * 32 * 7992 * 4 * 5, "ResidualUnit input gradients C = 32: g_y, u, x read, g_x, g_h written (5 passes)"),
    ("void eben::ru3_bwd_kernel<2,", B * 64 * 3996 * 4 * 5, "ResidualUnit input gradients C = 64"),
    ("void eben::ru3_bwd_kernel<4,", B * 128 * 999 * 4 * 5, "ResidualUnit input gradients C = 128"),
    ("void eben::ru_dw_kernel<1,", B * 32 * 7992 * 4 * 5, "ResidualUnit weight gradients C = 32: g_y, u, h, g_h, x read (5 passes)"),
    ("void eben::ru_dw_kernel<2,", B * 64 * 3996 * 4 * 5, "ResidualUnit weight gradients C = 64"),
    ("void eben::ru_dw_kernel<4,", B * 128 * 999 * 4 * 5, "ResidualUnit weight gradients C = 128"),
    ("eben::bl_tail_fwd_kernel", None, None),
]
print(f"{'kernel':58s} {'calls':>6s} {'avg us':>8s} {'MB/launch':>10s} {'TB/s':>6s} {'of 8':>5s}  what")
for name, nbytes, what in spec:
    if nbytes is None:
        continue
    hit = [r for n, r in rows.items() if n.startswith(name)]
    if not hit:
        continue
    r = hit[0]
    us = float(r["AverageNs"]) / 1e3
    print(f"{r['Name'][:58]:58s} {r['Calls']:>6s} {us:8.1f} {nbytes / 1e6:10.1f} {nbytes / us / 1e6:6.2f} {nbytes / us / 1e6 / 8:5.2f}  {what}")
# Adam: all launches of a step together (48 tensors per launch): 28 B per parameter (p, g, m, v read; p, m, v written)
ad = [r for n, r in rows.items() if n.startswith("eben::adam_kernel")]
if ad:
    tot_ns, calls = float(ad[0]["TotalDurationNs"]), int(ad[0]["Calls"])
    params = 1945984 + 23161344
    steps = calls / 4.0   # four launches per step at this size (1 generator + 3 discriminator)
    print(f"{'eben::adam_kernel (all launches of a step)':58s} {calls:>6d} {tot_ns / steps / 1e3:8.1f} {params * 28 / 1e6:10.1f} {params * 28 / (tot_ns / steps) / 1e3:6.2f} "
          f"{params * 28 / (tot_ns / steps) / 1e3 / 8:5.2f}  multi-tensor Adam over 25.1 M parameters, 28 B each")
