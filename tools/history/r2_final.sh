#!/bin/bash
# End-of-round evidence in one call: the whole GPU suite, rocprofv3 (kernel-trace stats + PMC passes + calibration, dpor /
# ddmin stats), then the bench lines with the fresh counters in place (specialised with the dpor / ddmin records and CPU
# baselines, interpreted, SrcDstFIFO, wide table).  Everything lands in gpurun_out/r02_*; copy what is judged to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r02_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r02_gpu_tests.log | tail -3
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 1500 bash tools/profile_r2.sh all > gpurun_out/r02_profile.log 2>&1; tail -5 gpurun_out/r02_profile.log
cp gpurun_out/r02_k1_counters.json profiles/k1_counters.json
bash tools/r2_extra_lines.sh
timeout 300 python bench.py --steps 20 --warmup 5 --wide-term0 1000 --no-secondary > gpurun_out/r02_bench_1gpu_wide.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_1gpu_wide.json').read().strip().splitlines()[-1]); print('wide', 'value %.4g' % d['value'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], d['cpu_baseline']['value'], d['cpu_baseline'].get('bit_identical_to_gpu'))"
python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_1gpu.json').read().strip().splitlines()[-1]); s=d['secondary']
print('dpor', {k: (round(v['value']), round(v['seconds'],3)) for k,v in s['dpor']['orders'].items()}, 'cpu', {k: round(v['value']) for k,v in s['dpor']['cpu_baseline']['orders'].items()})
print('ddmin', round(s['ddmin']['value']), 'cpu', round(s['ddmin']['cpu_baseline']['value']), s['ddmin']['ddmin_end_to_end'])
print('fuzz cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('bit_identical_to_gpu'), 'pcie', d.get('pcie_inclusive'), 'code', d['roofline'].get('kernel_code_id'), 'stale', d['roofline'].get('counters_stale'))"
