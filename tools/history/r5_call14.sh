#!/bin/bash
# Round 5, call 14: the reference order with the record fetch as ONE launch (the table's changes applied by the fetch's own
# blocks, a barrier, the answer's flag polled in host memory) and the launch's copies through one pinned area; A/B per piece.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export DEMI_EXPERIMENT=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_k3_gpu.py -m gpu -x -q -k "reference or golden or resident or queue or transliteration" 2>&1 | tail -2
for v in "" "DEMI_DPOR_FETCH_TWO_LAUNCHES=1" "DEMI_DPOR_FETCH_EVENT=1" "DEMI_DPOR_FETCH_TWO_LAUNCHES=1 DEMI_DPOR_FETCH_EVENT=1" "DEMI_DPOR_FETCH_SPINS=0" "DEMI_DPOR_FETCH_WIDTH=64" "DEMI_DPOR_FETCH_WIDTH=96"; do
  echo "== reference order [$v]"
  for rep in 1 2; do
  env $v DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload dpor --dpor-order reference_order --no-cpu-baseline 2> gpurun_out/r05_ref14.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['orders']['reference_order']; print('  %.4g/s %.4f s kernels %.1f ms launches %d fetches %d d2h %.1f MB digest %s' % (r['value'], r['seconds'], r['kernel_ms_total'], r['launches'], r['record_fetches'], r['d2h_bytes']/1e6, r['sequence_digest']))"
  done
  grep "dpor loop\|dpor reference" gpurun_out/r05_ref14.err | tail -2
done
