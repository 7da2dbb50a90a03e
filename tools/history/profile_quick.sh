#!/bin/bash
# quick PMC pass for K1: instruction mix + wait breakdown (2 passes)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
TAG=${1:-q}
mkdir -p $OUT
cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o k1 -- $CMD > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/prof_sq -o k1 -- $CMD > $OUT/prof_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $OUT/prof_sq2 -o k1 -- $CMD > $OUT/prof_sq2.log 2>&1
