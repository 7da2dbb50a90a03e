#!/bin/bash
# Round 4, second GPU call: the multi-rank tests on the device (W = 2, 3, 8; configs 4 and 5 over the ranks; failure injection;
# bench.py --gpus 2 with its secondary records), the new one-GPU bench line, and the K1 diagnosis counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_comm_gpu.py -m gpu -x -q --timeout 900 > gpurun_out/r04_comm_tests.log 2>&1; tail -5 gpurun_out/r04_comm_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err; tail -3 gpurun_out/r04_bench_b.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_b.json').read().strip().splitlines()[-1])
print('value %.4g ms_per_step %.3f kernel_ms %.3f bugs_per_hr %.4g timed' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['bugs_per_hr']), d['timed_region'], 'jit', d['config']['jit_compile_s'], 'cpu', (d.get('cpu_baseline') or {}).get('bit_identical_to_gpu'))
for k, v in d.get('secondary', {}).items():
    if 'error' in v: print(k, 'ERROR', v['error']); continue
    print(k, 'value %.4g' % v['value'], {x: v[x] for x in ('seconds', 'interleavings', 'exhausted', 'launches', 'kernel_ms_total', 'backtrack_points_still_queued', 'verdict_flag_histogram') if x in v}, (v.get('cpu_baseline') or {}).get('value'), {x: y for x, y in (v.get('cpu_baseline') or {}).items() if 'same' in x or 'identical' in x})
PY
bash tools/r4_k1_diag.sh > /dev/null 2>&1; cat gpurun_out/r04_k1_diag.txt
