#!/usr/bin/env python
"""Text Gantt of the last step in a rocprofv3 rocpd database: every dispatch longer than --min-us with its
start time (ms from the step start), duration and queue, plus per-queue busy time."""
import argparse, re, sqlite3
ap = argparse.ArgumentParser(); ap.add_argument("db"); ap.add_argument("--marker", default="adam_kernel"); ap.add_argument("--min-us", type=float, default=40.0)
a = ap.parse_args()
con = sqlite3.connect(a.db)
suf = [r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
cols = [r[1] for r in con.execute(f"pragma table_info(rocpd_kernel_dispatch{suf})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = con.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id order by d.start").fetchall()
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    n = re.sub(r"_ZN4eben\d+", "", n).replace(".kd", "")
    return n[:60]
marks = [e for n, s, e, q in rows if a.marker in n]
t1 = marks[-1]; t0 = marks[-5]
qs = {}
busy = {}
for n, s, e, q in rows:
    if s < t0 or s >= t1: continue
    qs.setdefault(q, len(qs))
    busy[q] = busy.get(q, 0) + (e - s)
    if (e - s) / 1e3 >= a.min_us:
        print(f"{(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  q{qs[q]}  {short(n)}")
print("step", (t1 - t0) / 1e6, "ms; busy per queue (ms):", {f"q{qs[q]}": round(v / 1e6, 2) for q, v in busy.items()})
