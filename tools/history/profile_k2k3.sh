#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o k23 -- python $R/tools/bench_k2k3.py > $OUT/bench_k2k3.json 2> $OUT/bench_k2k3.err
