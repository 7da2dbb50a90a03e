#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r02_gpu_tests.log 2>&1; grep -v "^Extension modules" gpurun_out/r02_gpu_tests.log | tail -30
for m in lds hbm; do echo "== K2 tests with DEMI_K2_MODE=$m"; DEMI_K2_MODE=$m timeout 900 python -m pytest tests/test_k2_gpu.py -x -q --timeout 600 2>&1 | tail -3; done
echo "== K2 tests with DEMI_K2_SCAN=1"; DEMI_K2_SCAN=1 timeout 900 python -m pytest tests/test_k2_gpu.py -x -q --timeout 600 2>&1 | tail -3
