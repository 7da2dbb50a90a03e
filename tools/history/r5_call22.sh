#!/bin/bash
# Round 5, call 22: K1's profile (tools/profile_r5.sh) and the driver's bench line with two launches in flight.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 bash tools/profile_r5.sh > gpurun_out/r05_profile_k1.log 2>&1
head -12 gpurun_out/r05_k1.txt
python -c "
import json; d=json.load(open('gpurun_out/k1_counters.json')); print({k: d[k] for k in ('code_id','kernel_ms','kernel_ms_all_launches','launches_profiled','kernel_ms_by_bench_events_in_the_traced_run','fabric_bytes_per_launch') if k in d})"
S0=$(date +%s); timeout 900 python bench.py > gpurun_out/r05_bench_1gpu.json 2> gpurun_out/r05_bench_1gpu.err
echo "bench wall $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_1gpu.json').read().strip().splitlines()[-1])
rl = d['roofline']
print('fuzz %.4g/s %.3f ms per step; kernel_ms %.3f alone %.3f; one at a time %s; traffic %s stale=%s' % (d['value'], d['ms_per_step'], rl['kernel_ms'], rl['kernel_ms_alone'], d.get('one_launch_at_a_time'), rl.get('traffic'), rl.get('counters_stale')))
s = d['secondary']
print('dpor rounds %.4g/s  reference %.4g/s; config5 %.4g/s; ddmin %.4g' % (s['dpor']['orders']['rounds']['value'], s['dpor']['orders']['reference_order']['value'], s['config5']['value'], s['ddmin']['value']))
PY
