#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# K3 after the pair-table pre-checks and the commit's record fetch: tests, then the dpor record per fetch width; K1 at 6 / 7 WG per CU
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_comm_gpu.py -x -q --timeout 600 2>&1 | tail -6
for w in 32 64 128 256; do
DEMI_DPOR_FETCH_WIDTH=$w DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload dpor --no-cpu-baseline 2> gpurun_out/r3_k3b_$w.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d['orders'].items(): print('width $w', k, round(v['value']), 'sec %.4f' % v['seconds'], 'il', v['interleavings'], 'exec', v['executed_on_device'], 'launches', v['launches'], 'kernel_ms %.1f' % v['kernel_ms_total'], 'd2h', v['d2h_bytes'], 'h2d', v['h2d_bytes'], v['sequence_digest'], v.get('record_fetches'))
"
grep -E "dpor|reference" gpurun_out/r3_k3b_$w.err | tail -4
done
for wg in 5 6 7 8; do
  echo "K1 wg/cu cap $wg"
  DEMI_K1_MAX_WG_PER_CU=$wg timeout 300 python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('resident_workgroups'))"
done
