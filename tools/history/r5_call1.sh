#!/bin/bash
# Round 5, call 1: the whole GPU suite with this round's new parity tests (config 5's whole exploration against the oracle, the
# one-job table in reference order against the Scala transliteration, bench.py launching its own ranks), the default bench line
# as the driver runs it, and the K1 profile of THIS build (tools/profile_r5.sh -> gpurun_out/k1_counters.json).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1000 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r05_gpu_tests_call1.log 2>&1
grep -E 'passed|failed|error' gpurun_out/r05_gpu_tests_call1.log | tail -3
S0=$(date +%s); timeout 600 python bench.py > gpurun_out/r05_bench_call1.json 2> gpurun_out/r05_bench_call1.err
echo "bench wall $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_call1.json').read().strip().splitlines()[-1])
print('fuzz %.4g/s %.3f ms kernel %.3f ms stale=%s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline'].get('counters_stale')))
s = d['secondary']
print('dpor rounds %.4g/s  reference %.4g/s' % (s['dpor']['orders']['rounds']['value'], s['dpor']['orders']['reference_order']['value']))
print('ddmin %.4g replays/s  e2e %.3f ms' % (s['ddmin']['value'], 1e3 * s['ddmin']['ddmin_end_to_end']['seconds']))
print('config5 %.4g/s %.3f s kernels %.1f ms' % (s['config5']['value'], s['config5']['seconds'], s['config5']['kernel_ms_total']))
PY
timeout 1200 bash tools/profile_r5.sh > gpurun_out/r05_profile_call1.log 2>&1
ls -la gpurun_out/k1_counters.json gpurun_out/r05_k1.txt 2>&1 | tail -3
head -12 gpurun_out/r05_k1.txt
