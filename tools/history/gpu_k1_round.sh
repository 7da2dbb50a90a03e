#!/bin/bash
# K1 parity (interpreter and specialised kernel) + the bench line both ways + the SrcDstFIFO line + the phase split
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 420 python -m pytest tests/test_k1_gpu.py -x -q --timeout 120 > $OUT/k1_tests.log 2>&1
rc=$?
tail -25 $OUT/k1_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout 200 python bench.py --steps 10 --warmup 2 > $OUT/bench_jit.json 2> $OUT/bench_jit.err
echo "bench(jit) rc $?"; tail -3 $OUT/bench_jit.err
timeout 200 python bench.py --steps 10 --warmup 2 --no-specialize --no-cpu-baseline > $OUT/bench_interp.json 2> $OUT/bench_interp.err
echo "bench(interp) rc $?"
timeout 200 python bench.py --steps 10 --warmup 2 --strategy fifo > $OUT/bench_fifo.json 2> $OUT/bench_fifo.err
echo "bench(fifo) rc $?"; tail -3 $OUT/bench_fifo.err
if [ -n "$PHASES" ]; then bash tools/k1_phases.sh; fi
