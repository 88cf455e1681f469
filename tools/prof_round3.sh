#!/bin/bash
# Profiles behind the round-3 numbers, one MI355X (run inside gpurun): bench line, kernel stats of the same command,
# SQ counter passes, FETCH_SIZE / WRITE_SIZE passes (each --pmc run on its own: no trace domains beside it).
#   bash tools/prof_round3.sh <tag>
set -e
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --cpu-steps 0 --f64-steps 0 > $OUT/bench_stats.json 2> $OUT/bench_stats.err; echo "stats done"
S="--cpu-steps 0 --f64-steps 0 --no-roofline --steps 120 --warmup 20"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmcA -- python3 $R/bench.py $S > $OUT/pmcA.json 2> $OUT/pmcA.err; echo "pmcA done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmcB -- python3 $R/bench.py $S > $OUT/pmcB.json 2> $OUT/pmcB.err; echo "pmcB done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcF -- python3 $R/bench.py $S > $OUT/pmcF.json 2> $OUT/pmcF.err; echo "pmcF done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcW -- python3 $R/bench.py $S > $OUT/pmcW.json 2> $OUT/pmcW.err; echo "pmcW done"
CHEM_TRACE=1 python3 $R/bench.py --steps 20 --warmup 5 --f64-steps 0 --cpu-steps 0 --late-stage 0 > $OUT/bench20.json 2> $OUT/bench20_trace.txt; echo "bench20 done"
python3 $R/bench.py --steps 20 --warmup 5 --f64-steps 0 --cpu-steps 0 --late-stage 0 --no-roofline > $OUT/bench20_noroofline.json 2> /dev/null; echo "bench20 (no sampling) done"
cd $R
EQUIL=1500 python3 tools/tile_stamps.py 1000000 > $OUT/tile_stamps.txt 2>&1 || true
python3 tools/kstats.py $(find $OUT/stats -name "*kernel_stats.csv" | head -1) 24 > $OUT/kernel_stats.txt
python3 tools/pmc_kernels.py $(find $OUT/pmcA $OUT/pmcB -name "*counter_collection.csv") > $OUT/pmc_sq.txt
python3 tools/pmc_traffic_kernels.py $(find $OUT/pmcF -name "*counter_collection.csv" | head -1) $(find $OUT/pmcW -name "*counter_collection.csv" | head -1) 1000000 > $OUT/kernel_traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/pmcA $OUT/pmcB $OUT/pmcF $OUT/pmcW $OUT/stats     # raw traces: tens of MB per pass (gpurun merges back at most 64 MiB)
echo "summaries done"
