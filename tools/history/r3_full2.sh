#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r03_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03_gpu_tests.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r03_gpu_tests.log | head -8
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -2 gpurun_out/r03_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1]); s=d.get('secondary',{})
print('fuzz', 'value %.4g' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'cpu', round(d['cpu_baseline']['value']), d['cpu_baseline'].get('bit_identical_to_gpu'), 'single', d['cpu_baseline'].get('single_thread'))
for k in ('config1','config5'):
    r=s.get(k,{}); print(k, {x: r.get(x) for x in ('value','error','seconds','interleavings','exhausted')}, r.get('cpu_baseline'))
r=s.get('dpor',{})
if 'orders' in r:
    print('dpor', {k: (round(v['value']), round(v['seconds'],3), v['d2h_bytes'], v['backtrack_points']) for k,v in r['orders'].items()}, 'roofline frac', r['roofline']['frac'], r['roofline']['algorithmic_bytes_per_launch'], r['cpu_baseline'].get('same_verdict_sequence_as_gpu'))
else: print('dpor', r)
r=s.get('ddmin',{})
if 'value' in r:
    print('ddmin', round(r['value']), r['ddmin_end_to_end'], r.get('random_ddmin_R100'), r['launch_floor'])
else: print('ddmin', r)
PY
timeout 900 bash tools/profile_r3_k2k3.sh > gpurun_out/r03_profile_k2k3.log 2>&1; tail -5 gpurun_out/r03_profile_k2k3.log
cat gpurun_out/r03_ddmin.txt | tail -12
