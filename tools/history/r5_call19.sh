#!/bin/bash
# Round 5, call 19: config 5 at 2^23 and 2^24 interleavings after the staging area of a round's points became growable and the
# explored-pair table got room for budgets past 2^21 (16 GB); the K3 suite first.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_k3_gpu.py -m gpu -x -q 2>&1 | tail -2
for b in 1048576 8388608 16777216; do
  echo "== config 5, budget $b"
  DEMI_EXPERIMENT=1 DEMI_DPOR_TIMING=1 timeout 400 python bench.py --workload config5 --no-cpu-baseline --config5-budget $b 2> gpurun_out/r05_c5_budget.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.4g/s %.3f s kernels %.1f ms launches %d queued %d d2h %.1f MB h2d %.1f MB digest %s setup %.2f s' % (d['value'], d['seconds'], d['kernel_ms_total'], d['launches'], d['backtrack_points_still_queued'], d['pcie_bytes']['d2h']/1e6, d['pcie_bytes']['h2d']/1e6, d['sequence_digest'], d['setup_s_untimed']))" || tail -3 gpurun_out/r05_c5_budget.err
  grep "staging area\|explored-pair table:" gpurun_out/r05_c5_budget.err | tail -3
  rocm-smi --showmemuse 2>/dev/null | grep -i "VRAM%" | head -1
done
