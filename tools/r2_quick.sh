#!/bin/bash
# K2: filterKnownAbsents parity + frontier spreading A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_k2_gpu.py -x -q --timeout 900 2>&1 | tail -8
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 600 python bench.py --workload ddmin --no-cpu-baseline 2>gpurun_out/r2_q_$name.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'value %.4g' % d['value'], {k: (round(v['kernel_us']), round(v['wall_us'])) for k, v in d['frontiers'].items()})"
}
run spread A=1
run lanes64 DEMI_K2_LANES_PER_WAVE=64
run spread_hbm DEMI_K2_MODE=hbm
timeout 600 python bench.py --workload ddmin > gpurun_out/r2_q_ddmin.json 2>gpurun_out/r2_q_ddmin.err; tail -c 1500 gpurun_out/r2_q_ddmin.json
