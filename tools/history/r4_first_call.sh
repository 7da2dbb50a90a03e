#!/bin/bash
# Round 4, FIRST GPU call: the whole GPU suite with round 3's final host code (its last commits ran on
# the emulator only, apart from the array / wide-variant tests), smoke, the headline line, and the first TIMED lines of what the
# end of round 3 added: a wide table through SrcDstFIFO, and tables with arrays (the raft with a real log).  About 4 GPU-minutes.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 600 > gpurun_out/r04_gpu_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_gpu_tests.log | tail -3
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_1gpu.json 2> gpurun_out/r04_bench_1gpu.err
timeout 300 python bench.py --steps 20 --warmup 5 --wide-term0 1000 --no-secondary > gpurun_out/r04_bench_1gpu_wide.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --wide-term0 1000 --strategy fifo --no-secondary > gpurun_out/r04_bench_1gpu_wide_fifo.json 2>/dev/null
for cap in 4 8 15; do
  timeout 300 python bench.py --steps 20 --warmup 5 --log-cap $cap --no-secondary > gpurun_out/r04_bench_1gpu_log$cap.json 2>/dev/null
done
for f in r04_bench_1gpu r04_bench_1gpu_wide r04_bench_1gpu_wide_fifo r04_bench_1gpu_log4 r04_bench_1gpu_log8 r04_bench_1gpu_log15; do python -c "
import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', 'value %.4g' % d['value'], 'ms_per_step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % r['kernel_ms'], 'cpu', (d.get('cpu_baseline') or {}).get('bit_identical_to_gpu'))"; done
