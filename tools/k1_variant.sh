#!/bin/bash
# Experiment helper: build a variant of the library with extra compile flags (VARIANT_FLAGS), check K1 parity on it,
# print the bench line (specialised and interpreted), restore the product library.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp demi_amd/libdemi_gpu.so /tmp/libdemi_gpu.so.keep
for FL in "$@"; do
  echo "=== variant: $FL"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -ldl $FL -o demi_amd/libdemi_gpu.so demi_amd/csrc/demi_gpu.hip || continue
  timeout 200 python -m pytest tests/test_k1_gpu.py -x -q --timeout 90 -k "raft5_parity_all_capacities or limits_matrix" 2>&1 | tail -2
  timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/tmp/jit.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('jit   ', d['roofline']['kernel_ms'], d['value'])"
  tail -3 /tmp/jit.err
  timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-specialize 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('interp', d['roofline']['kernel_ms'], d['value'])"
done
cp /tmp/libdemi_gpu.so.keep demi_amd/libdemi_gpu.so
