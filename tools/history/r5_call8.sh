#!/bin/bash
# Round 5, calls 8 and 10: the pair kernels after a change - K3 suite, configs 5 and 3, per-kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export DEMI_EXPERIMENT=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload config5 --no-cpu-baseline 2> gpurun_out/r05_c8.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config5  %.4g/s  %.3f s  kernels %.1f ms  launches %d  digest %s' % (d['value'], d['seconds'], d['kernel_ms_total'], d['launches'], d['sequence_digest']))"
done
grep "dpor loop\|prepare" gpurun_out/r05_c8.err | sed -n 3,4p
for rep in 1 2; do
timeout 300 python bench.py --workload dpor --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,r in d['orders'].items(): print('config3  %s %.4g/s %.4f s kernels %.1f ms launches %d digest %s' % (k, r['value'], r['seconds'], r['kernel_ms_total'], r['launches'], r['sequence_digest']))"
done
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
rm -rf /tmp/p_c5
timeout 300 rocprofv3 --preload $COMGR --kernel-trace --stats -d /tmp/p_c5 -o k -- python $R/bench.py --workload config5 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import glob, sqlite3
dbs = glob.glob("/tmp/p_c5/*.db") + glob.glob("/tmp/p_c5/*/*.db")
cur = sqlite3.connect(dbs[0]).cursor()
print("== config5: kernel-trace stats")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 10"):
    print("%-70s %6d %12.0f us %10.1f us %6.2f" % (r[0][:70], r[1], r[2], r[3], r[4]))
PY
