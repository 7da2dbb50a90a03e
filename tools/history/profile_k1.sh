#!/bin/bash
# rocprofv3 evidence for K1 on the GPU box: kernel-trace stats, then PMC passes (separately, as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Writes gpurun_out/prof_*.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o k1 -- $CMD > $OUT/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o k1 -- $CMD > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o k1 -- $CMD > $OUT/prof_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/prof_sq -o k1 -- $CMD > $OUT/prof_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU -d $OUT/prof_sq2 -o k1 -- $CMD > $OUT/prof_sq2.log 2>&1
ls -R $OUT | head -50
