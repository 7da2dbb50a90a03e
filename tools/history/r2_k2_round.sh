#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_k2_gpu.py -x -q --timeout 600 > gpurun_out/r02_k2_tests.log 2>&1; grep -v "^Extension modules" gpurun_out/r02_k2_tests.log | tail -40
for m in auto lds hbm; do
  if [ $m = auto ]; then unset DEMI_K2_MODE; else export DEMI_K2_MODE=$m; fi
  DEMI_K2_VERBOSE=1 timeout 600 python bench.py --workload ddmin --no-cpu-baseline 2>gpurun_out/r02_ddmin_$m.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$m:', '%.4g replays/s' % d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],2), d['launch_floor'], d['still_violating'])"
done
