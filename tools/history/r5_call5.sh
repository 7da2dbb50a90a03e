#!/bin/bash
# Round 5, call 5: the evidence run - whole GPU suite, smoke, the driver's bench line, the secondary workloads' profiles
# (tools/profile_r5_k2k3.sh: kernel-trace stats + PMC passes of ddmin / dpor / config5) and the headline's (tools/profile_r5.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r05_gpu_tests.log 2>&1
grep -E 'passed|failed|error' gpurun_out/r05_gpu_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
S0=$(date +%s); timeout 900 python bench.py > gpurun_out/r05_bench_1gpu.json 2> gpurun_out/r05_bench_1gpu.err
echo "bench wall $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_1gpu.json').read().strip().splitlines()[-1])
rl = d['roofline']
print('fuzz %.4g/s %.3f ms kernel %.3f ms traffic %s stale=%s' % (d['value'], d['ms_per_step'], rl['kernel_ms'], rl.get('traffic'), rl.get('counters_stale')))
s = d['secondary']
print('dpor rounds %.4g/s  reference %.4g/s' % (s['dpor']['orders']['rounds']['value'], s['dpor']['orders']['reference_order']['value']))
print('ddmin %.4g replays/s  e2e %.3f ms  random_ddmin %s' % (s['ddmin']['value'], 1e3 * s['ddmin']['ddmin_end_to_end']['seconds'], {k: s['ddmin']['random_ddmin_R100'].get(k) for k in ('seconds', 'executions_per_s', 'error')}))
print('config5 %.4g/s %.3f s kernels %.1f ms' % (s['config5']['value'], s['config5']['seconds'], s['config5']['kernel_ms_total']))
PY
timeout 1500 bash tools/profile_r5_k2k3.sh > gpurun_out/r05_profile_k2k3.log 2>&1
ls gpurun_out/ | grep r05_ | tr '\n' ' '
head -12 gpurun_out/r05_config5.txt
cat gpurun_out/r05_config5_counters.json | head -8
