#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_provenance_gpu.py tests/test_k2_gpu.py -x -q --timeout 900 -k "provenance or forests or literal or gamut" 2>&1 | tail -12
