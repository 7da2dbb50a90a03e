#!/bin/bash
# End-of-round records: whole GPU suite, the three K1 bench lines, rocprofv3 evidence for the headline kernel
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 700 python -m pytest tests -m gpu -x -q --timeout 120 > $OUT/gpu_tests.log 2>&1
rc=$?
tail -8 $OUT/gpu_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout 200 python bench.py --steps 10 --warmup 2 > $OUT/bench_jit.json 2> $OUT/bench_jit.err; echo "bench(jit) rc $?"
timeout 200 python bench.py --steps 10 --warmup 2 --no-specialize --no-cpu-baseline > $OUT/bench_interp.json 2> $OUT/bench_interp.err; echo "bench(interp) rc $?"
timeout 200 python bench.py --steps 10 --warmup 2 --strategy fifo > $OUT/bench_fifo.json 2> $OUT/bench_fifo.err; echo "bench(fifo) rc $?"
rm -rf $OUT/prof_*
timeout 500 bash tools/profile_k1.sh > $OUT/profile_k1.log 2>&1; echo "profile rc $?"
