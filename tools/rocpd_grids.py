#!/usr/bin/env python
"""Launch geometry of a rocprofv3 rocpd database: per kernel (name, grid, workgroup) calls, mean duration, blocks -- to find launches that
cannot fill 256 CUs (few blocks) yet take time."""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
suf = [r[0] for r in con.execute("select name from sqlite_master where type='table'") if r[0].startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
cols = [r[1] for r in con.execute(f"pragma table_info(rocpd_kernel_dispatch{suf})")]
print("columns:", cols, file=sys.stderr)
g = [c for c in cols if c.startswith("grid_size")] or [c for c in cols if "grid" in c]
w = [c for c in cols if c.startswith("workgroup_size")] or [c for c in cols if "workgroup" in c]
sel = ", ".join(f"d.{c}" for c in g + w)
rows = con.execute(f"select s.kernel_name, d.start, d.end, {sel} from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id").fetchall()
agg = {}
for r in rows:
    n, s, e = r[0], r[1], r[2]
    gs, ws = r[3:3 + len(g)], r[3 + len(g):]
    blocks = 1
    for a, b in zip(gs, ws):
        blocks *= max(1, (a or 1) // max(1, (b or 1)))
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    n = re.sub(r"_ZN4eben\d+", "", n).replace(".kd", "")[:70]
    k = (n, blocks, tuple(ws))
    a_ = agg.setdefault(k, [0, 0.0])
    a_[0] += 1; a_[1] += (e - s) / 1e3
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (n, blocks, ws), (c, tot) in out:
    if blocks < 512 and tot / c > 8:
        print(f"{c:5d} calls  {tot / c:8.1f} us  {blocks:6d} blocks x {ws}  {n}")
