#!/usr/bin/env python
"""Where the GPU sits idle inside a step (rocprofv3 rocpd database): the last step of the trace (between two discriminator-Adam
launches), the intervals during which NO kernel runs, grouped by the kernel that ends before the gap: count, total, and the
longest ones with the kernel that follows.  Usage: rocpd_gaps.py results.db [--min-us 5]"""
import argparse, re, sqlite3
ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--marker", default="adam_kernel"); ap.add_argument("--min-us", type=float, default=5.0)
a = ap.parse_args()
con = sqlite3.connect(a.db)
suf = [r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
rows = con.execute(f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id order by d.start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("eben::", "")[:60]
marks = [e for n, s, e in rows if a.marker in n]
t1, t0 = marks[-1], marks[-1 - 4 * 3]   # last three steps
ks = [(s, e, short(n)) for n, s, e in rows if e > t0 and s < t1]
ks.sort()
gaps = []
cur_end, cur_name = ks[0][1], ks[0][2]
for s, e, n in ks[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n, cur_end - t0))
    if e > cur_end:
        cur_end, cur_name = e, n
tot = sum(g[0] for g in gaps)
print(f"3 steps, {(t1 - t0) / 3e6:.2f} ms per step; idle {tot / 3e6:.2f} ms per step in {len(gaps) / 3:.0f} gaps; gaps >= {a.min_us} us: "
      f"{sum(g[0] for g in gaps if g[0] >= a.min_us * 1e3) / 3e6:.2f} ms in {sum(1 for g in gaps if g[0] >= a.min_us * 1e3) / 3:.0f}")
hist = {}
for g, before, after, _ in gaps:
    h = hist.setdefault(before, [0, 0]); h[0] += 1; h[1] += g
print("idle time by the kernel that ENDS before the gap (ms per step, count per step):")
for k, (c, t) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 3e6:6.3f} ms {c / 3:6.1f}x  {k}")
print("longest gaps of the last step (us, at ms, before -> after):")
last = [g for g in gaps if g[3] > 2 * (t1 - t0) / 3]
for g, before, after, at in sorted(last, key=lambda x: -x[0])[:30]:
    print(f"  {g / 1e3:7.1f} us at {(at - 2 * (t1 - t0) / 3) / 1e6:6.2f} ms  {before}  ->  {after}")
