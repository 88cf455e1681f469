#!/bin/bash
python -m pytest tests -m gpu -q > gpurun_out/r3_tests3.log 2>&1; tail -12 gpurun_out/r3_tests3.log
python bench.py --steps 20 --warmup 5 --f64-steps 0 --cpu-steps 0 --late-stage 0 > gpurun_out/b3_20.json 2> gpurun_out/b3_20.err
python bench.py --steps 20 --warmup 5 --f64-steps 0 --cpu-steps 0 --late-stage 0 --no-roofline > gpurun_out/b3_20_noroof.json 2> gpurun_out/b3_20_noroof.err
python bench.py --steps 20 --warmup 5 --f64-steps 0 --cpu-steps 0 --late-stage 0 --opt list_skin=0 > gpurun_out/b3_20_ls0.json 2> gpurun_out/b3_20_ls0.err
python bench.py --f64-steps 0 --cpu-steps 0 > gpurun_out/b3_2000.json 2> gpurun_out/b3_2000.err
python - <<'PY'
import json
for f in ("b3_20", "b3_20_noroof", "b3_20_ls0", "b3_2000"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "FAILED", e, open("gpurun_out/%s.err" % f).read()[-600:]); continue
    r = d.get("roofline", {}); dev = r.get("device_us_per_step")
    print("%-14s %8.1f steps/s %.4f ms/step rebuilds %s builds %s react_ms %s dev %s" % (f, d["value"], d["ms_per_step"], d["config"]["list_rebuilds_timed"], d["config"].get("list_builds_timed"), d["config"].get("reaction_step_ms"), dev))
    if "late_stage" in d: print("   late:", json.dumps(d["late_stage"])[:600])
PY
