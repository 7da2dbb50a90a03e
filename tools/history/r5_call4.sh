#!/bin/bash
# Round 5, call 4: the reference order with the asynchronous record fetch (A/B: DEMI_DPOR_NO_PREFETCH), per fetch width.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export DEMI_EXPERIMENT=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_k3_gpu.py -m gpu -x -q -k "reference or golden or resident" 2>&1 | tail -2
for v in "" "DEMI_DPOR_NO_PREFETCH=1" "DEMI_DPOR_FETCH_WIDTH=64" "DEMI_DPOR_FETCH_WIDTH=256" "DEMI_DPOR_FETCH_WIDTH=32"; do
  echo "== reference order [$v]"
  for rep in 1 2; do
  env $v DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload dpor --dpor-order reference_order --no-cpu-baseline 2> gpurun_out/r05_ref4.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['orders']['reference_order']; print('  %.4g/s %.4f s kernels %.1f ms launches %d fetches %d d2h %.1f MB digest %s' % (r['value'], r['seconds'], r['kernel_ms_total'], r['launches'], r['record_fetches'], r['d2h_bytes']/1e6, r['sequence_digest']))"
  done
  grep "dpor loop" gpurun_out/r05_ref4.err | tail -1
done
