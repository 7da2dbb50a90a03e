#!/bin/bash
# Round 5, call 18: config 5 at larger budgets (2^21 ... 2^23 interleavings): does the rate hold as the explored-pair table, the
# backtrack queue and the arena grow?  (The table is sized by the budget up to 2^26 entries = 4 GB.)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
for b in 2097152 4194304 8388608; do
  echo "== config 5, budget $b"
  timeout 300 python bench.py --workload config5 --no-cpu-baseline --config5-budget $b 2> gpurun_out/r05_c5_budget.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.4g/s %.3f s kernels %.1f ms launches %d queued %d d2h %.1f MB h2d %.1f MB digest %s setup %.2f s' % (d['value'], d['seconds'], d['kernel_ms_total'], d['launches'], d['backtrack_points_still_queued'], d['pcie_bytes']['d2h']/1e6, d['pcie_bytes']['h2d']/1e6, d['sequence_digest'], d['setup_s_untimed']))" || tail -3 gpurun_out/r05_c5_budget.err
  rocm-smi --showmemuse 2>/dev/null | grep -i "GPU\[0\]" | head -2
done
