#!/bin/bash
# Whole GPU suite, then the K2/K3 throughput script under rocprofv3 (only if the tests pass)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q --timeout 120 > $OUT/gpu_tests.log 2>&1
rc=$?
tail -15 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o k23 -- python $R/tools/bench_k2k3.py > $OUT/bench_k2k3.json 2> $OUT/bench_k2k3.err
echo "bench rc $?"
tail -3 $OUT/bench_k2k3.err
cat $OUT/bench_k2k3.json
