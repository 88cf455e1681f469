#!/bin/bash
# internal-skin / tile-shape study on the C5 bench (user-level skin stands in for the internal one: same forces)
run() { # name lib skin extra
  if [ "$2" = main ]; then unset CHEM_MI355_LIB; else export CHEM_MI355_LIB=$PWD/chemlab_amd/csrc/variants/libchem_$2.so; fi
  python bench.py --f64-steps 0 --cpu-steps 0 --late-stage 0 --steps 1000 --skin $3 $4 > gpurun_out/skin_$1.json 2> gpurun_out/skin_$1.err || { echo "$1 FAILED"; tail -3 gpurun_out/skin_$1.err; return; }
  python - "$1" gpurun_out/skin_$1.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = d["roofline"]; ks = {k["kernel"].split(" ")[0]: k for k in r["kernels"]}
print("%-16s %8.1f steps/s  %.4f ms/step  pair %.1f  nb/step %.1f  rebuild %.1f us  int %.1f  bonded %.1f  rebuilds %d  nb_entries %.1f  react_ms %.2f" % (
    sys.argv[1], d["value"], d["ms_per_step"], r["device_us_per_step"]["pair"], r["device_us_per_step"]["neighbour_kernel"],
    ks.get("k_rebuild_fused", {}).get("avg_launch_us", 0), r["device_us_per_step"]["integrate"], r["device_us_per_step"]["bonded"],
    d["config"]["list_rebuilds_timed"], ks["k_pair_tiles"]["mean_neighbours"], d["config"].get("reaction_step_ms", 0)))
PY
}
run main_030 main 0.3 ""
run hz2_049 hz2 0.49 ""
run hz2_058 hz2 0.578 ""
run hz2_067 hz2 0.668 ""
run main_067_b1024 main 0.668 "--opt pair_block=1024"
run main_049 main 0.49 ""
