#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# per-phase s_memtime split of k3_dpor (diagnostic build of the compiled kernel; the marks cost time: proportions only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
DEMI_K3_PHASES=1 DEMI_JIT_DEFINES="DEMI_K3_PHASES=1" timeout 300 python bench.py --workload dpor --dpor-order rounds --no-cpu-baseline 2>&1 >/dev/null | grep "k3 phases" | tail -8
