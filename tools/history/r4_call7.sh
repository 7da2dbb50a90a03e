#!/bin/bash
# Round 4: how the size of the explored-pair table decides config 5's pair kernels (the table is 4 GB for a 2^20 budget: every
# probe is an HBM access; DEMI_K3_TABLE_ENTRIES overrides), and the table's real occupancy.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for e in 67108864 16777216 4194304 1048576; do
  echo "== table entries $e"
  DEMI_EXPERIMENT=1 DEMI_DPOR_TIMING=1 DEMI_K3_TABLE_ENTRIES=$e timeout 300 python bench.py --workload config5 --no-cpu-baseline > gpurun_out/r04_c5_tab_$e.json 2> gpurun_out/r04_c5_tab_$e.err
  grep -E "dpor (loop|resident)" gpurun_out/r04_c5_tab_$e.err | tail -3
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r04_c5_tab_$e.json').read().strip().splitlines()[-1])
    print('   value %.4g  seconds %.3f  kernel_ms %.1f  digest %s' % (d['value'], d['seconds'], d['kernel_ms_total'], d['sequence_digest']))
except Exception as ex:
    print('   failed:', ex, open('gpurun_out/r04_c5_tab_$e.err').read()[-400:])
PY
done 2>&1 | tee gpurun_out/r04_config5_table_sizes.txt
