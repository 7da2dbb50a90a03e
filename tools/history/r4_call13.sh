#!/bin/bash
# Round 4, call 13 (evidence run after the K3 work): the whole GPU suite, the default bench line as the driver runs it, the
# rocprofv3 stats / PMC passes of the secondary workloads (tools/profile_r4_k2k3.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r04_gpu_tests_final.log 2>&1
grep -E 'passed|failed' gpurun_out/r04_gpu_tests_final.log | tail -2
S0=$(date +%s); timeout 600 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
echo "bench wall $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print('fuzz %.4g/s %.3f ms kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
s = d['secondary']
print('dpor rounds %.4g/s  reference %.4g/s' % (s['dpor']['orders']['rounds']['value'], s['dpor']['orders']['reference_order']['value']))
print('ddmin %.4g replays/s  e2e %.3f ms' % (s['ddmin']['value'], 1e3 * s['ddmin']['ddmin_end_to_end']['seconds']))
print('config5 %.4g/s %.3f s' % (s['config5']['value'], s['config5']['seconds']))
PY
timeout 1200 bash tools/profile_r4_k2k3.sh > gpurun_out/r04_profile_k2k3_final.log 2>&1
ls -la gpurun_out/r04_dpor.txt gpurun_out/r04_config5.txt gpurun_out/r04_dpor_counters.txt gpurun_out/r04_dpor_counters.json gpurun_out/r04_config5_counters.json 2>&1 | tail -6
head -14 gpurun_out/r04_dpor.txt
