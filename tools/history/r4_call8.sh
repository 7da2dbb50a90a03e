#!/bin/bash
# Round 4, call 8: the whole GPU suite with the payload-area host code, then timed lines: the headline (must not move), the raft
# with a real log (log-cap 8) with packed payloads and with akka-raft's field sets (DEMI_MODEL_PAYLOADS(5)).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_tests_payloads.log 2>&1
tail -3 gpurun_out/r04_gpu_tests_payloads.log
timeout 300 python bench.py --no-secondary > gpurun_out/r04_bench_after_payloads.json 2> gpurun_out/r04_bench_after_payloads.err
timeout 300 python bench.py --no-secondary --no-cpu-baseline --log-cap 8 > gpurun_out/r04_bench_log8.json 2> gpurun_out/r04_bench_log8.err
timeout 300 python bench.py --no-secondary --no-cpu-baseline --log-cap 8 --real-fields > gpurun_out/r04_bench_log8_fields.json 2> gpurun_out/r04_bench_log8_fields.err
for f in after_payloads log8 log8_fields; do
python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r04_bench_$f.json').read().strip().splitlines()[-1])
    print('$f: value %.4g  ms %.3f  kernel_ms %.3f  jit %.2f s  payloads %s' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['jit_compile_s'], d['config'].get('payload_fields_per_message')))
except Exception as ex:
    print('$f failed:', ex, open('gpurun_out/r04_bench_$f.err').read()[-600:])
PY
done
