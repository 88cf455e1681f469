#!/bin/bash
# A/B timing of library builds / engine options on the bench workload: tools/ab.sh <outdir> <label>:<lib or ->:<bench args> ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; args=${rest#*:}
  [ "$lib" = "-" ] && unset CHEM_MI355_LIB || export CHEM_MI355_LIB=$PWD/$lib
  timeout -k 10 240 python3 bench.py --steps ${STEPS:-1000} --cpu-steps 0 --f64-steps 0 $args > "$out/$label.json" 2> "$out/$label.err" || { echo "$label FAILED"; tail -3 "$out/$label.err"; exit 1; }
  python3 - "$out/$label.json" "$label" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("%-14s %8.1f steps/s  %.4f ms/step  react %.2f ms  dev us/step %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"].get("reaction_step_ms", 0),
      {k: round(v, 1) for k, v in r["device_us_per_step"].items()}))
PY
done
