#!/bin/bash
# Last GPU call of the round: whole GPU suite, smoke(), K2/K3 throughput (no profiler)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 150 python -m pytest tests -m gpu -x -q --timeout 60 > $OUT/gpu_tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/gpu_tests.log
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 90 python tools/bench_k2k3.py > $OUT/bench_k2k3.json 2> $OUT/bench_k2k3.err; echo "k2k3 rc $?"
