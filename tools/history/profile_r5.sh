#!/bin/bash
# rocprofv3 evidence for the bench workloads on the GPU box.  Kernel-trace stats and the PMC passes are separate runs
# (the MI355X guide: FETCH_SIZE and WRITE_SIZE do not fit one pass; never combine --pmc with trace domains other than
# the kernel trace).  The counters are calibrated against a kernel with a known byte count (tools/calib_counters.py).
# Everything is condensed on the box into gpurun_out/<tag>_* (text + JSON); the databases themselves are dropped.
# Usage: bash tools/profile_r5.sh [all]   (TAG=r05 by default; writes gpurun_out/${TAG}_* AND copies the K1 counters to
# gpurun_out/k1_counters.json = what goes to profiles/k1_counters.json, which tests/test_jit_cpu.py holds against the tree)
export TMPDIR=/tmp
TAG=${TAG:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
P=/tmp/prof
rm -rf $P; mkdir -p $OUT $P
cd /tmp
CMD="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary"
# The profiler's libraries pull in /opt/rocm's code-object manager (ROCm 7.2, LLVM 22) before PyTorch can load the one it
# bundles (ROCm 7.0.2, LLVM 20), and demi_model_specialize would then compile the table with another compiler than in an
# untraced run (another K1: 2 187 instead of 2 455 instructions, 4.81 instead of 4.35 ms).  Preloading PyTorch's comgr keeps
# the traced kernel the one bench.py measures (same demi_model_code_id).
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
PRE=""
if [ -f "$COMGR" ] && [ -z "$DEMI_PROFILE_SYSTEM_COMGR" ]; then PRE="--preload $COMGR"; fi
timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats -o k1 -- $CMD > $OUT/${TAG}_prof_stats.log 2>&1
timeout 300 rocprofv3 $PRE --pmc FETCH_SIZE -d $P/prof_fetch -o k1 -- $CMD > $OUT/${TAG}_prof_fetch.log 2>&1
timeout 300 rocprofv3 $PRE --pmc WRITE_SIZE -d $P/prof_write -o k1 -- $CMD > $OUT/${TAG}_prof_write.log 2>&1
timeout 300 rocprofv3 $PRE --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $P/prof_sq -o k1 -- $CMD > $OUT/${TAG}_prof_sq.log 2>&1
timeout 300 rocprofv3 $PRE --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU -d $P/prof_sq2 -o k1 -- $CMD > $OUT/${TAG}_prof_sq2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $P/calib_fetch -o c -- python $R/tools/calib_counters.py > $OUT/${TAG}_calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $P/calib_write -o c -- python $R/tools/calib_counters.py > $OUT/${TAG}_calib_write.log 2>&1
if [ "$1" = "all" ]; then
  timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats_dpor -o k3 -- python $R/bench.py --workload dpor --no-cpu-baseline > $OUT/${TAG}_prof_stats_dpor.log 2>&1
  timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats_ddmin -o k2 -- python $R/bench.py --workload ddmin --no-cpu-baseline > $OUT/${TAG}_prof_stats_ddmin.log 2>&1
fi
python $R/tools/summarize_prof.py ${TAG} $P $OUT
cp $OUT/${TAG}_k1_counters.json $OUT/k1_counters.json 2>/dev/null
