#!/bin/bash
# After the K2 lock-step default: the bench line again (its ddmin record changes) and the ddmin kernel-trace stats
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_1gpu.json 2> gpurun_out/r02_bench_1gpu.err
python -c "
import json; d=json.loads(open('gpurun_out/r02_bench_1gpu.json').read().strip().splitlines()[-1]); s=d['secondary']; r=d['roofline']
print('fuzz', 'value %.4g' % d['value'], 'kernel_ms %.3f' % r['kernel_ms'], 'stale', r['counters_stale'])
print('ddmin', round(s['ddmin']['value']), {k: round(v['kernel_us']) for k,v in s['ddmin']['frontiers'].items()}, 'cpu', round(s['ddmin']['cpu_baseline']['value']), s['ddmin']['cpu_baseline']['bit_identical_to_gpu'], s['ddmin']['ddmin_end_to_end']['seconds'])
print('dpor', {k: round(v['value']) for k,v in s['dpor']['orders'].items()})"
P=/tmp/profdd; rm -rf $P; mkdir -p $P; cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
rocprofv3 --preload $COMGR --kernel-trace --stats -d $P/prof_stats_ddmin -o k2 -- python $R/bench.py --workload ddmin --no-cpu-baseline > $R/gpurun_out/r02_prof_stats_ddmin.log 2>&1
python $R/tools/summarize_prof.py r02dd $P $R/gpurun_out > /dev/null 2>&1; head -6 $R/gpurun_out/r02dd_ddmin.txt
