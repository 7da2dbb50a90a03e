#!/bin/bash
# the communicator tests (sharded K1 / K2 / K3 with two processes on one GPU) on the final code
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python -m pytest tests/test_comm_gpu.py -x -q --timeout 150 2>&1 | grep -E "passed|failed|rror" | tail -3
