#!/usr/bin/env python
"""Timeline summary of a rocprofv3 rocpd database: for the last N steps (windows delimited by the adam
kernel), how long the GPU ran 0 / 1 / 2+ kernels at once, and which kernels ran ALONE the longest (those
are the serial critical path; everything else overlaps on the side streams)."""
import argparse, re, sqlite3
ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--marker", default="adam_kernel"); ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()
con = sqlite3.connect(a.db)
suf = [r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
rows = con.execute(f"select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id order by d.start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:70]
marks = [e for n, s, e in rows if a.marker in n]
# a step ends with the last adam launch of the discriminator: take windows between every 4th marker from the end
if len(marks) < 9:
    raise SystemExit("not enough steps in trace")
t1 = marks[-1]; t0 = marks[-1 - 4 * 5]   # last five steps (4 adam launches per step)
ev = []
for n, s, e in rows:
    if e <= t0 or s >= t1: continue
    s = max(s, t0); e = min(e, t1)
    ev.append((s, 1, short(n))); ev.append((e, -1, short(n)))
ev.sort()
active = {}
busy = [0, 0, 0]; alone = {}
last = t0; cnt = 0
for t, d, n in ev:
    dt = t - last
    if dt > 0:
        busy[min(cnt, 2)] += dt
        if cnt == 1:
            k = next(iter(k for k, v in active.items() if v > 0))
            alone[k] = alone.get(k, 0) + dt
    active[n] = active.get(n, 0) + d
    cnt += d; last = t
span = t1 - t0
print(f"5 steps, {span/5e6:.2f} ms per step: idle {busy[0]/5e6:.2f} ms, one kernel {busy[1]/5e6:.2f} ms, two or more {busy[2]/5e6:.2f} ms")
print("kernels running ALONE (ms per step):")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[: a.top]:
    print(f"  {v/5e6:6.2f}  {k}")
