#!/usr/bin/env python
"""Per-dispatch PMC table from a rocprofv3 rocpd SQLite database (ROCm 7.2 `--pmc ... --kernel-trace`).

One row per kernel dispatch: duration, grid, LDS bytes, VGPRs and every collected counter (summed
over instances).  Usage: rocpd_pmc.py results.db [--match tapconv] [--min-us 100] [--agg]
"""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="")
    ap.add_argument("--min-us", type=float, default=0.0)
    ap.add_argument("--agg", action="store_true", help="average rows with the same kernel name and grid")
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    q = f"""select d.id, s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size,
                   s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, d.event_id
            from rocpd_kernel_dispatch{suf} d join rocpd_info_kernel_symbol{suf} s on d.kernel_id = s.id order by d.start"""
    disp = con.execute(q).fetchall()
    pmc = {}
    names = []
    for ev, name, val in con.execute(
            f"select e.event_id, p.name, sum(e.value) from rocpd_pmc_event{suf} e join rocpd_info_pmc{suf} p on e.pmc_id = p.id "
            f"group by e.event_id, p.name"):
        pmc.setdefault(ev, {})[name] = val
        if name not in names:
            names.append(name)
    rows = []
    for (_id, kname, st, en, gx, wx, lds, vg, ag, sg, ev) in disp:
        short = re.sub(r"\(.*", "", kname).replace("void ", "")
        us = (en - st) / 1e3
        if a.match and a.match not in short:
            continue
        if us < a.min_us:
            continue
        rows.append((short[:60], gx // max(wx, 1), wx, lds, vg, ag, sg, us, pmc.get(ev, {})))
    if a.agg:
        agg = {}
        for r in rows:
            k = (r[0], r[1], r[2], r[3], r[4], r[5], r[6])
            e = agg.setdefault(k, [0, 0.0, {}])
            e[0] += 1
            e[1] += r[7]
            for n, v in r[8].items():
                e[2][n] = e[2].get(n, 0.0) + v
        rows = [(k[0], k[1], k[2], k[3], k[4], k[5], k[6], e[1] / e[0], {n: v / e[0] for n, v in e[2].items()}) for k, e in agg.items()]
    print("kernel,blocks,threads,lds,vgpr,agpr,sgpr,us," + ",".join(names))
    for r in rows:
        print(",".join([r[0]] + [str(x) for x in r[1:7]] + [f"{r[7]:.1f}"] + [f"{r[8].get(n, 0):.4g}" for n in names]))


if __name__ == "__main__":
    main()
