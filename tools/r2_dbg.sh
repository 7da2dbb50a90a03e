#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in lds hbm; do echo "=== $m"; DEMI_K2_MODE=$m python tools/r2_k2_debug2.py 2>&1 | grep -v amdgpu.ids; done
