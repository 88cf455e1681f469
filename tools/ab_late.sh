#!/bin/bash
# late-stage A/B of engine options: tools/ab_late.sh OUT_PREFIX "bench args" "name1:--opt a=b" ...   (prints rebuild cadence of both legs)
out=$1; shift; extra=$1; shift
for spec in "$@"; do
  name=${spec%%:*}; opts=${spec#*:}
  python bench.py --f64-steps 0 --cpu-steps 0 $extra $opts > gpurun_out/${out}_$name.json 2> gpurun_out/${out}_$name.err || { echo "$name FAILED"; tail -5 gpurun_out/${out}_$name.err; continue; }
  python - "$name" gpurun_out/${out}_$name.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
c = d["config"]; l = d.get("late_stage", {})
print("%-12s melt %8.1f steps/s  list builds %d / %d steps   | late %8.1f steps/s  list builds %s / %s steps  conv %.3f" % (
    sys.argv[1], d["value"], c.get("list_builds_timed", -1), d["steps"], l.get("value", 0), l.get("list_builds_timed"), l.get("steps"), l.get("conversion", 0)))
PY
done
