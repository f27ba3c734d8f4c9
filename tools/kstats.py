"""Rows of a rocprofv3 --kernel-trace --stats CSV whose kernel name contains one of the given substrings: calls, average and minimum us.
Usage: kstats.py kernel_stats.csv substr [substr ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if len(sys.argv) == 2 or any(t in n for t in sys.argv[2:]):
        print(f"{n[:86]:86s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us min {float(r['MinNs']) / 1e3:8.1f}")
