#!/bin/bash
# the bench line again with profiles/k1_counters.json in place (traffic + issue model quoted), and the interpreted / SrcDstFIFO lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_1gpu.json 2> gpurun_out/r02_bench_1gpu.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-specialize --no-secondary > gpurun_out/r02_bench_1gpu_interpreter.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --strategy fifo --no-secondary > gpurun_out/r02_bench_1gpu_srcdstfifo.json 2>/dev/null
for f in r02_bench_1gpu r02_bench_1gpu_interpreter r02_bench_1gpu_srcdstfifo; do python -c "
import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', 'value %.4g' % d['value'], 'kernel_ms %.3f' % r['kernel_ms'], 'traffic', r.get('traffic'), 'stale', r.get('counters_stale'), 'issue', (r.get('issue_model') or {}).get('issue_frac_straight_line'))"; done
