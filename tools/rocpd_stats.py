#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output) into a per-kernel table:
calls, total / average / min / max duration, share of GPU time.  Optionally restrict to the last
N fraction of the trace (to skip warm-up).  Usage: rocpd_stats.py results.db [--grid] [--csv out.csv]"""
import argparse
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--grid", action="store_true", help="split rows by grid size")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    gcol = ", grid_x" if a.grid and "grid_x" in cols else ""
    rows = con.execute(f"select {name_col}, start, end{gcol} from kernels").fetchall()
    if not rows:
        print("no kernel dispatches in", a.db)
        return
    agg = {}
    for r in rows:
        key = short(r[0]) + (f" [grid {r[3]}]" if gcol else "")
        d = (r[2] - r[1]) / 1e3
        e = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
    total = sum(e[1] for e in agg.values())
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e3
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
    for k, e in out:
        lines.append(f"\"{k}\",{e[0]},{e[1]:.1f},{e[1]/e[0]:.2f},{e[2]:.2f},{e[3]:.2f},{100*e[1]/total:.2f}")
    if a.csv:
        open(a.csv, "w").write("\n".join(lines) + "\n")
    print(f"# {len(rows)} dispatches, {total/1e3:.2f} ms of kernel time over a {span/1e3:.2f} ms span")
    print(f"{'calls':>7} {'total ms':>10} {'avg us':>10} {'pct':>6}  kernel")
    for k, e in out[: a.top]:
        print(f"{e[0]:7d} {e[1]/1e3:10.3f} {e[1]/e[0]:10.2f} {100*e[1]/total:6.2f}  {k}")


if __name__ == "__main__":
    main()
