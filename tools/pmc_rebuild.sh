#!/bin/bash
# SQ / instruction-cache counters of the rebuilding k_rebuild_fused launches (the "max" column of the summary) and of
# the force kernel, melted C5 configuration, short run.  bash tools/pmc_rebuild.sh <tag>
set -e
TAG=${1:-pr}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S="--cpu-steps 0 --f64-steps 0 --no-roofline --steps 60 --warmup 10 --equil 1000"
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/bench.py $S > $OUT/p1.json 2> $OUT/p1.err; echo p1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/bench.py $S > $OUT/p2.json 2> $OUT/p2.err; echo p2
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS_LOAD SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p3 -- python3 $R/bench.py $S > $OUT/p3.json 2> $OUT/p3.err; echo p3
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $OUT/p4 -- python3 $R/bench.py $S > $OUT/p4.json 2> $OUT/p4.err; echo p4
cd $R
python3 tools/pmc_kernels.py $(find $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 -name "*counter_collection.csv") --match=rebuild_fused,pair_tiles > $OUT/pmc.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
