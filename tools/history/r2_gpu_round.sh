#!/bin/bash
# round-2 GPU round: the whole GPU suite, the bench line (with the dpor / ddmin records), rocprofv3 evidence
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r02_gpu_tests.log 2>&1; tail -4 gpurun_out/r02_gpu_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; cut -c1-1500 gpurun_out/r02_bench.json; tail -3 gpurun_out/r02_bench.err
if [ "$1" != "noprof" ]; then timeout 1500 bash tools/profile_r2.sh all > gpurun_out/r02_profile.log 2>&1; tail -60 gpurun_out/r02_profile.log; fi
