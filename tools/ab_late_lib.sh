#!/bin/bash
# melt + late-stage A/B of library variants (same box): tools/ab_late_lib.sh OUT_PREFIX "bench args" name1 name2 ...  ("main" = product library)
out=$1; shift; extra=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset CHEM_MI355_LIB; else export CHEM_MI355_LIB=$PWD/chemlab_amd/csrc/variants/libchem_$v.so; fi
  python bench.py --f64-steps 0 --cpu-steps 0 $extra > gpurun_out/${out}_$v.json 2> gpurun_out/${out}_$v.err || { echo "$v FAILED"; tail -5 gpurun_out/${out}_$v.err; continue; }
  python - "$v" gpurun_out/${out}_$v.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = d["roofline"]["device_us_per_step"]; l = d.get("late_stage", {}); lr = l.get("device_us_per_step", {})
print("%-10s melt %8.1f steps/s pair %.1f int %.1f | late %8.1f steps/s pair %.1f int %.1f" % (sys.argv[1], d["value"], r["pair"], r["integrate"], l.get("value", 0), lr.get("pair", 0), lr.get("integrate", 0)))
PY
done
