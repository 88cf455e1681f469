#!/usr/bin/env python3
"""Per-kernel medians of rocprofv3 --pmc counters (counter_collection.csv files of one or more passes)
with the derived figures used in DESIGN.md: LDS bank-conflict share, VALU issue share, wait share.
usage: pmc_kernels.py <csv> [<csv> ...] [--match substr,substr]"""
import csv, sys, collections
match = ("pair_tiles", "rebuild_fused", "integrate", "bonded_work")
paths = []
for a in sys.argv[1:]:
    if a.startswith("--match="):
        match = tuple(a.split("=", 1)[1].split(","))
    else:
        paths.append(a)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in paths:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('chem::', '')[:48]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(acc.items()):
    if not any(x in k for x in match):
        continue
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    mx = {c: max(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    print("%s  (n=%d dispatches)" % (k, n))
    for c in sorted(med):
        print("    %-24s median %14.0f   max %14.0f" % (c, med[c], mx[c]))
    g = lambda c: med.get(c, 0.0)
    if g('SQ_LDS_IDX_ACTIVE'):
        print("    -> LDS bank-conflict share of LDS cycles: %.1f %%" % (100 * g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')))
    if g('SQ_WAVE_CYCLES') and g('SQ_WAIT_ANY'):
        print("    -> waves parked (s_waitcnt/barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %% of wave cycles"
              % (100 * g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'), 100 * g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'), 100 * g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')))
    if g('SQ_BUSY_CYCLES') and g('SQ_ACTIVE_INST_VALU'):
        print("    -> VALU-active / SQ-busy: %.2f ; LDS-active / SQ-busy: %.2f" % (g('SQ_ACTIVE_INST_VALU') / g('SQ_BUSY_CYCLES'), g('SQ_ACTIVE_INST_LDS') / g('SQ_BUSY_CYCLES')))
