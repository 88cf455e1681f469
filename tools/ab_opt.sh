#!/bin/bash
# A/B of engine options on the C5 bench (same library): tools/ab_opt.sh OUT_PREFIX "bench args" "name1:--opt a=b" "name2:" ...
out=$1; shift; extra=$1; shift
for spec in "$@"; do
  name=${spec%%:*}; opts=${spec#*:}
  python bench.py --f64-steps 0 --cpu-steps 0 $extra $opts > gpurun_out/${out}_$name.json 2> gpurun_out/${out}_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/${out}_$name.err; continue; }
  python - "$name" gpurun_out/${out}_$name.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = d["roofline"]; dv = r["device_us_per_step"]
s = "%-12s %8.1f steps/s  pair %.1f nb %.1f int %.1f bonded %.1f" % (sys.argv[1], d["value"], dv["pair"], dv["neighbour_kernel"], dv["integrate"], dv["bonded"])
if "late_stage" in d:
    l = d["late_stage"]; lv = l["device_us_per_step"]
    s += "   | late %8.1f steps/s  pair %.1f nb %.1f int %.1f bonded %.1f  conv %.3f" % (l["value"], lv["pair"], lv["neighbour_kernel"], lv["integrate"], lv["bonded"], l["conversion"])
print(s)
PY
done
