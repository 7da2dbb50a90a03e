#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Diagnostic: build a -DDEMI_K1_PHASES copy of the library, run the bench workload once with the specialised
# kernel and once with the table interpreter, print the per-phase cycle split of K1 (s_memtime deltas summed
# over waves).  Does not touch the product .so.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp demi_amd/libdemi_gpu.so /tmp/libdemi_gpu.so.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -ldl -DDEMI_K1_PHASES -o demi_amd/libdemi_gpu.so demi_amd/csrc/demi_gpu.hip
# (the headline launches only: --no-secondary, and of those the last one)
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prewarm ${BENCH_ARGS} 2>&1 | grep -E "k1 phases" | grep -v "waves=4 " | tail -1 | cut -c1-700
if [ -z "$PHASES_JIT_ONLY" ]; then
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prewarm --no-specialize ${BENCH_ARGS} 2>&1 | grep -E "k1 phases" | grep -v "waves=4 " | tail -1 | cut -c1-700
fi
cp /tmp/libdemi_gpu.so.keep demi_amd/libdemi_gpu.so
