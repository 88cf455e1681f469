#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel (counter_collection.csv)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('chem::', '')[:40]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if not any(x in k for x in ('pair', 'nlist', 'integrate', 'bin')):
        continue
    print(k, {c: round(sorted(v)[len(v)//2], 1) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
