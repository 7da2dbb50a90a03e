#!/bin/bash
# Round 4, fifth GPU call: config 4's DDMin by K2 counter mode and speculation budget; where config 5's wall time goes
# (DEMI_DPOR_TIMING); the native editDistanceDporDDMin test.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python tools/r4_ddmin_modes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_k2_modes.txt
DEMI_EXPERIMENT=1 DEMI_DPOR_TIMING=1 timeout 600 python bench.py --workload config5 > gpurun_out/r04_config5_timing.json 2> gpurun_out/r04_config5_timing.err; grep -E "dpor (loop|resident)" gpurun_out/r04_config5_timing.err | tail -6
DEMI_EXPERIMENT=1 DEMI_DPOR_TIMING=1 timeout 600 python bench.py --workload dpor > gpurun_out/r04_dpor_timing.json 2> gpurun_out/r04_dpor_timing.err; grep -E "dpor (loop|resident|ref)" gpurun_out/r04_dpor_timing.err | tail -8
timeout 900 python -m pytest tests/test_k3_gpu.py -m gpu -q --timeout 800 -k "edit_distance" > gpurun_out/r04_call5_tests.log 2>&1; tail -3 gpurun_out/r04_call5_tests.log
