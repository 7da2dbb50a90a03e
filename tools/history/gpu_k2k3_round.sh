#!/bin/bash
# K2/K3 parity + throughput in one GPU call (tests first; the bench only if they pass)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests/test_k2_gpu.py tests/test_k3_gpu.py -x -q --timeout 90 > $OUT/k2k3_tests.log 2>&1
rc=$?
tail -15 $OUT/k2k3_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o k23 -- python $R/tools/bench_k2k3.py > $OUT/bench_k2k3.json 2> $OUT/bench_k2k3.err
echo "bench rc $?"
tail -5 $OUT/bench_k2k3.err
cat $OUT/bench_k2k3.json
