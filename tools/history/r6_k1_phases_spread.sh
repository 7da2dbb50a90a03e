#!/bin/bash
# K1's phases for a launch of 1 024 schedules of config 2: plain (two... sixteen full waves) against SPREAD (one lane per wave)
export DEMI_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp demi_amd/libdemi_gpu.so /tmp/libdemi_gpu.so.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -ldl -DDEMI_K1_PHASES -o demi_amd/libdemi_gpu.so demi_amd/csrc/demi_gpu.hip
for K in "DEMI_K1_NO_SPREAD=1" "DEMI_K1_LANES_PER_WAVE=1"; do
echo "== $K"
env $K python - <<'PY' 2>&1 | grep "k1 phases" | tail -2 | cut -c1-600
import sys
sys.path.insert(0, ".")
from demi_amd import _native
from demi_amd.apps import SEED_BASE, raft5_config2
model, events, limits = raft5_config2()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events); ctx.model_specialize()
for _ in range(2):
    ctx.random_explore(1024, limits, seed_base=SEED_BASE)
PY
done
cp /tmp/libdemi_gpu.so.keep demi_amd/libdemi_gpu.so
