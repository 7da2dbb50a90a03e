#!/bin/bash
# wide tables through REC / K2 / K3 + everything that touches demi_rec_event: the GPU suites concerned
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_wide_gpu.py tests/test_k2_gpu.py tests/test_k3_gpu.py tests/test_k1_gpu.py -x -q --timeout 900 2>&1 | tail -25
