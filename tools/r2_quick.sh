#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/r2_q_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'])"
}
run wg7 A=1
run wg6 DEMI_K1_MAX_WG_PER_CU=6
run wg7b A=1
timeout 1500 python -m pytest tests/test_k1_gpu.py tests/test_blocked_actors_gpu.py -x -q --timeout 900 2>&1 | tail -8
