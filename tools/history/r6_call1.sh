#!/bin/bash
# Round 6, call 1: the DPOR workloads that find the seeded bugs (apps.raft5_dpor_config3 / shuffle8_dpor_config5) on the device:
# their parity tests at full size, then the two bench records (both orders each).
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call1_build.log 2>&1
timeout 1500 python -m pytest tests/test_dpor_bug_workloads_gpu.py -m gpu -x -q -k "not transliterations_sequence" > gpurun_out/r06_call1_tests.log 2>&1
tail -3 gpurun_out/r06_call1_tests.log
timeout 900 python -m pytest tests/test_k3_gpu.py -m gpu -x -q -k "bug" >> gpurun_out/r06_call1_tests.log 2>&1
tail -3 gpurun_out/r06_call1_tests.log
timeout 900 python bench.py --workload dpor > gpurun_out/r06_call1_dpor.json 2> gpurun_out/r06_call1_dpor.err
tail -c 3000 gpurun_out/r06_call1_dpor.json
timeout 900 python bench.py --workload config5 > gpurun_out/r06_call1_config5.json 2> gpurun_out/r06_call1_config5.err
tail -c 3000 gpurun_out/r06_call1_config5.json
tail -5 gpurun_out/r06_call1_config5.err
