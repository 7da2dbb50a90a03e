#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_k3_gpu.py -x -q --timeout 600 2>&1 | tail -15
DEMI_K1_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep -E "k1 launch" | sort | uniq -c
DEMI_DPOR_TIMING=1 timeout 600 python bench.py --workload dpor > gpurun_out/r02_dpor.json 2> gpurun_out/r02_dpor.err; tail -5 gpurun_out/r02_dpor.err; python -c "
import json; d=json.load(open('gpurun_out/r02_dpor.json')); print(json.dumps({k: d[k] for k in ('value','orders','cpu_baseline')}, indent=1)[:3000])"
