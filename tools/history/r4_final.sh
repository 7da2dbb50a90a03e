#!/bin/bash
# End-of-round evidence in ONE call: the whole GPU suite, smoke, rocprofv3 of the headline kernel (kernel-trace stats + PMC
# passes + calibration), the counters put in place, then the bench lines (specialised with all secondary records, interpreted,
# SrcDstFIFO, wide table) and the K2 / K3 profiles.  Everything lands in gpurun_out/r04_*; copy what is judged to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r04_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_gpu_tests.log | tail -3
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 1500 bash tools/profile_r4.sh > gpurun_out/r04_profile.log 2>&1; tail -3 gpurun_out/r04_profile.log
cp gpurun_out/r04_k1_counters.json profiles/k1_counters.json
# the K2 / K3 profiles first: the dpor record quotes its `traffic` from the counters of THIS build
timeout 1200 bash tools/profile_r4_k2k3.sh > gpurun_out/r04_profile_k2k3.log 2>&1; tail -2 gpurun_out/r04_profile_k2k3.log
for f in r04_dpor_counters.json r04_config5_counters.json r04_ddmin_counters.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f profiles/$f; done
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_1gpu.json 2> gpurun_out/r04_bench_1gpu.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-specialize --no-secondary > gpurun_out/r04_bench_1gpu_interpreter.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --strategy fifo --no-secondary > gpurun_out/r04_bench_1gpu_srcdstfifo.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --wide-term0 1000 --no-secondary > gpurun_out/r04_bench_1gpu_wide.json 2>/dev/null
for f in r04_bench_1gpu r04_bench_1gpu_interpreter r04_bench_1gpu_srcdstfifo r04_bench_1gpu_wide; do python -c "
import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', 'value %.4g' % d['value'], 'kernel_ms %.3f' % r['kernel_ms'], 'frac %.3g' % r['frac'], 'traffic', r.get('traffic'), 'stale', r.get('counters_stale'), 'issue', (r.get('issue_model') or {}).get('issue_frac_straight_line'), 'cpu', (d.get('cpu_baseline') or {}).get('bit_identical_to_gpu'))"; done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_1gpu.json').read().strip().splitlines()[-1]); s=d.get('secondary',{})
print('single-thread', d['cpu_baseline'].get('single_thread'))
for k in ('config1','config5'):
    r=s.get(k,{}); print(k, {x: r.get(x) for x in ('value','error','seconds','interleavings','exhausted')}, (r.get('cpu_baseline') or {}).get('value'))
r=s.get('dpor',{})
if 'orders' in r:
    print('dpor', {k: (round(v['value']), round(v['seconds'],3), v['d2h_bytes'], v['launches'], round(v['kernel_ms_total'],1)) for k,v in r['orders'].items()}, 'frac', r['roofline']['frac'], 'traffic', r['roofline']['traffic'], r['cpu_baseline'].get('same_verdict_sequence_as_gpu'), {k: round(v['value']) for k,v in r['cpu_baseline']['orders'].items()})
else: print('dpor', r)
r=s.get('ddmin',{})
if 'value' in r:
    print('ddmin', round(r['value']), 'frac', r['roofline']['frac'], r['ddmin_end_to_end'], r.get('random_ddmin_R100'), r['launch_floor'], 'cpu', round(r['cpu_baseline']['value']), r['cpu_baseline']['ddmin_end_to_end'])
else: print('ddmin', r)
PY
# the latency-bound kernels' phase splits (diagnostic builds of the compiled kernels; proportions only)
timeout 300 bash tools/k3_phases.sh > gpurun_out/r04_k3_phases.txt 2>&1; tail -2 gpurun_out/r04_k3_phases.txt
timeout 300 bash tools/k2_phases.sh > gpurun_out/r04_k2_phases.txt 2>&1; tail -2 gpurun_out/r04_k2_phases.txt
