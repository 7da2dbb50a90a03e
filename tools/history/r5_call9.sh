#!/bin/bash
# Round 5, call 9: the evidence run with the final kernels - whole GPU suite, smoke, the driver's bench line, the secondary
# workloads' profiles (tools/profile_r5_k2k3.sh); K1's translation unit is unchanged since call 1 (same code id), its profile stays.
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
head -14 gpurun_out/r05_config5.txt
python -c "
import json
for w in ('config5','dpor','ddmin'):
    d=json.load(open('gpurun_out/r05_%s_counters.json' % w)); print(w, {k: d[k] for k in d if k.startswith('fabric')})"
# beyond the sizes the bench line runs (no profile, one line each): 2^24 schedules per K1 launch, config 5 with a budget of 2^23
timeout 300 python bench.py --schedules 16777216 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K1 2^24 schedules per launch: %.4g/s %.2f ms per step' % (d['value'], d['ms_per_step']))" || echo "K1 2^24: failed"
timeout 300 python bench.py --workload config5 --no-cpu-baseline --config5-budget 8388608 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5, budget 2^23: %.4g/s %.3f s digest %s' % (d['value'], d['seconds'], d['sequence_digest']))" || echo "config 5 2^23: failed"
