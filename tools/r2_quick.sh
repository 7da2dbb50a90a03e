#!/bin/bash
# Host-buffer path (demi_random_explore / demi_replay_batch) with and without the pinned staging buffers; the tests that use it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for v in staged plain; do
  if [ $v = plain ]; then export DEMI_NO_STAGED_COPY=1; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>gpurun_out/r2_q_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'], d.get('pcie_inclusive'))"
done
unset DEMI_NO_STAGED_COPY
timeout 600 python -m pytest tests/test_k1_gpu.py tests/test_k2_gpu.py -x -q --timeout 300 2>&1 | grep -E "passed|failed|rror" | tail -3
