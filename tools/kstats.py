#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv (per-kernel totals) -- helper for profiles/."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    name = r['Name'].split('(')[0].replace('void ', '').replace('chem::', '')[:44]
    print(f"{name:46s} calls {r['Calls']:>6s} total {int(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}% min {int(r['MinNs'])/1e3:8.1f} max {int(r['MaxNs'])/1e3:8.1f}")
