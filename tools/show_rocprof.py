#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: per-kernel calls, avg/min/max us and share; optional calls-per-forward divisor."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    n, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    per = n * avg / div; tot += per
    print(f"{r['Name'][:86]:86s} calls {n:6d} avg {avg:8.2f} min {int(r['MinNs'])/1e3:7.2f} max {int(r['MaxNs'])/1e3:7.2f} us  per-fwd {per:7.1f} us {r['Percentage']}%")
print(f"total kernel time per forward: {tot:.1f} us")
