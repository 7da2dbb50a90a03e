#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in scan lds hbm; do echo "=== $m"; if [ $m = scan ]; then DEMI_K2_SCAN=1 DEMI_K2_VERBOSE=1 python tools/r2_k2_debug.py 2>&1 | grep -v amdgpu.ids; else DEMI_K2_MODE=$m DEMI_K2_VERBOSE=1 python tools/r2_k2_debug.py 2>&1 | grep -v amdgpu.ids; fi; done
