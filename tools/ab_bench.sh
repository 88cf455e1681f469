#!/bin/bash
# A/B of library variants on the C5 bench: tools/ab_bench.sh OUT_PREFIX "extra bench args" name1 name2 ...   (name "main" = the product library)
out=$1; shift; extra=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset CHEM_MI355_LIB; else export CHEM_MI355_LIB=$PWD/chemlab_amd/csrc/variants/libchem_$v.so; fi
  python bench.py --f64-steps 0 --cpu-steps 0 $extra > gpurun_out/${out}_$v.json 2> gpurun_out/${out}_$v.err || { echo "$v FAILED"; tail -5 gpurun_out/${out}_$v.err; continue; }
  python - "$v" gpurun_out/${out}_$v.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = d["roofline"]; ks = {k["kernel"].split(" ")[0]: k for k in r["kernels"]}
print("%-10s %8.1f steps/s  %.4f ms/step  pair %.1f  nb/step %.1f  rebuild %.1f us  int %.1f  bonded %.1f  rebuilds %d  nb_entries %.1f" % (
    sys.argv[1], d["value"], d["ms_per_step"], r["device_us_per_step"]["pair"], r["device_us_per_step"]["neighbour_kernel"],
    ks.get("k_rebuild_fused", {}).get("avg_launch_us", 0), r["device_us_per_step"]["integrate"], r["device_us_per_step"]["bonded"],
    d["config"]["list_rebuilds_timed"], ks["k_pair_tiles"]["mean_neighbours"]))
PY
done
