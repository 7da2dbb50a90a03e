#!/bin/bash
# Round 6, call 3: where the submit / wait pipeline keeps or loses the overlap (tools/history/r6_pipeline_ab.py), the K2 / K3 profiles with
# the instruction counters of the issue models (tools/profile_r6_k2k3.sh), the late-feature lines re-measured once.
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call3_build.log 2>&1
timeout 600 python tools/history/r6_pipeline_ab.py > gpurun_out/r06_call3_pipeline_ab.txt 2>&1
cat gpurun_out/r06_call3_pipeline_ab.txt
timeout 600 python -m pytest tests/test_k1_gpu.py -m gpu -x -q -k "two_calls or explore_in_calls or two_streams" > gpurun_out/r06_call3_tests.log 2>&1
tail -3 gpurun_out/r06_call3_tests.log
timeout 2400 bash tools/profile_r6_k2k3.sh > gpurun_out/r06_profile_k2k3.log 2>&1
tail -25 gpurun_out/r06_profile_k2k3.log
for v in "wide --wide-term0 300" "log15 --log-cap 15" "srcdstfifo --strategy fifo" "wide_fifo --wide-term0 300 --strategy fifo" "interpreter --no-specialize"; do
  set -- $v; name=$1; shift
  timeout 400 python bench.py --no-secondary --no-cpu-baseline "$@" > gpurun_out/r06_bench_1gpu_$name.json 2>> gpurun_out/r06_call3_lines.err
  python -c "import json,sys; d=json.load(open('gpurun_out/r06_bench_1gpu_$name.json')); print('$name', '%.4g schedules/s' % d['value'], '%.3f ms per step' % d['ms_per_step'])"
done
