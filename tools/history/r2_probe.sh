#!/bin/bash
# round-2 probe: current K1 bench (long enough to get past the DVFS ramp), the phase split of the specialised K1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
nproc > gpurun_out/r2_nproc.txt; lscpu | head -20 >> gpurun_out/r2_nproc.txt; free -g >> gpurun_out/r2_nproc.txt
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r2_bench0.json 2> gpurun_out/r2_bench0.err
bash tools/k1_phases.sh > gpurun_out/r2_phases.txt 2>&1
cat gpurun_out/r2_bench0.json | cut -c1-400; cat gpurun_out/r2_phases.txt
