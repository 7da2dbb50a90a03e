#!/bin/bash
# Round 3: the whole GPU suite, smoke, and the complete bench line (fuzz + dpor + ddmin records) of the current build.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r03_gpu_tests.log 2>&1; tail -3 gpurun_out/r03_gpu_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -2 gpurun_out/r03_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1]); s=d.get('secondary',{})
print('fuzz', 'value %.4g' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('bit_identical_to_gpu'))
if 'orders' in s.get('dpor',{}):
    print('dpor', {k: (round(v['value']), round(v['seconds'],3), v['d2h_bytes']) for k,v in s['dpor']['orders'].items()}, 'cpu', {k: round(v['value']) for k,v in s['dpor']['cpu_baseline']['orders'].items()})
else: print('dpor', s.get('dpor'))
if 'value' in s.get('ddmin',{}):
    print('ddmin', round(s['ddmin']['value']), 'cpu', round(s['ddmin']['cpu_baseline']['value']), s['ddmin']['ddmin_end_to_end'], s['ddmin']['cpu_baseline']['ddmin_end_to_end'], s['ddmin']['launch_floor'])
else: print('ddmin', s.get('ddmin'))
PY
