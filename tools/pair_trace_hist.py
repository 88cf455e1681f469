#!/usr/bin/env python3
"""Per-launch durations of k_pair_tiles and k_rebuild_fused at the END of a rocprofv3 --kernel-trace run (the late-stage leg of
bench.py): usage pair_trace_hist.py <kernel_trace.csv> [last N launches = 499]"""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 499
def dur(name):
    r = [x for x in rows if x["Kernel_Name"].startswith(name)]
    r.sort(key=lambda x: int(x["Start_Timestamp"]))
    return np.array([(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in r[-nlast:]]), r[-nlast:]
dp, rp = dur("void chem::k_pair_tiles<float, 1, false")
dr, rr = dur("void chem::k_rebuild_fused<float")
print("pair launches %d: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (len(dp), dp.mean(), *np.percentile(dp, [10, 50, 90]), dp.max()))
big = dr > 100
print("rebuild launches %d, rebuilding %d: mean %.1f us; idle mean %.1f us" % (len(dr), big.sum(), dr[big].mean() if big.any() else 0, dr[~big].mean()))
# the pair launch right behind a rebuilding launch
after = dp[big[:len(dp)]] if len(dp) == len(dr) else None
if after is not None and len(after):
    print("pair launch behind a rebuild: mean %.1f us (others %.1f us)" % (after.mean(), dp[~big].mean()))
