#!/usr/bin/env python
"""Which kernels of libeben_hip.so were launched: rocprofv3 --kernel-trace --stats summaries (CSV) of the GPU test suite and of
bench.py against the kernel list of the library (tools/kernel_resources.py).
Usage: python tools/kernel_coverage.py <kernel_stats.csv> [...]   (prints per source file: kernels, launched, and the ones never launched)"""
import csv
import os
import re
import subprocess
import sys


def norm(name):
    name = re.sub(r"^void ", "", name.strip())
    depth, out = 0, []
    for ch in name:          # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).replace("eben::", "").replace(" ", "")


def main():
    used = {}
    for f in sys.argv[1:]:
        for r in csv.DictReader(open(f)):
            used[norm(r["Name"])] = used.get(norm(r["Name"]), 0) + int(r["Calls"])
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "kernel_resources.py")], capture_output=True, text=True).stdout
    per = {}
    for line in out.splitlines():
        m = re.match(r"(\S+)\s+(.*?)\s+vgpr", line)
        if m:
            per.setdefault(m.group(1), []).append(m.group(2).strip())
    for src in sorted(per):
        dead = [n for n in per[src] if norm(n) not in used]
        print(f"{src:12s} {len(per[src]):3d} kernels, {len(per[src]) - len(dead):3d} launched")
        for n in dead:
            print(f"      never launched: {n}")


if __name__ == "__main__":
    main()
