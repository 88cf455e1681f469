#!/bin/bash
# Baseline profile of the current build on one MI355X: kernel stats + two SQ counter passes.
# usage (inside gpurun): bash tools/prof_r2.sh <tag>
set -e
TAG=${1:-base}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --cpu-steps 0 --steps 400 --warmup 100 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
echo "stats done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmcA -- python3 $R/bench.py --cpu-steps 0 --no-roofline --steps 40 --warmup 20 > $OUT/pmcA.json 2> $OUT/pmcA.err
echo "pmcA done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmcB -- python3 $R/bench.py --cpu-steps 0 --no-roofline --steps 40 --warmup 20 > $OUT/pmcB.json 2> $OUT/pmcB.err
echo "pmcB done"
