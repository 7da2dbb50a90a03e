#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# tools/isa_diff.sh [<git rev>]  - which gfx950 kernels does the working tree compile to different instructions than <rev> (HEAD)?
# No GPU needed.  The generic kernels of libdemi_gpu.so are compared symbol by symbol (device-only compilation of demi_gpu.hip,
# disassembled), the kernels specialised for raft5 by the .text of their code objects (demi_specialize_check under
# DEMI_JIT_DUMP).  A change that is meant to leave a hot kernel alone should show that kernel as SAME here before it goes to
# the GPU; scratch under gpurun_out/isa_diff (git-ignored).
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
G=$ROOT/gpurun_out/isa_diff
LLVM=/opt/rocm/lib/llvm/bin
rm -rf "$G"; mkdir -p "$G"
git -C "$ROOT" worktree add -f "$G/wt" "$REV" -q
trap 'git -C "$ROOT" worktree remove --force "$G/wt"' EXIT
for t in A B; do
  r=$([ $t = A ] && echo "$G/wt" || echo "$ROOT")
  ( cd "$r" && python -c "import __graft_entry__ as G; G._write_jit_sources()" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c demi_amd/csrc/demi_gpu.hip -o "$G/dev$t.o" &&
    $LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$G/dev$t.o" --output="$G/co$t.o" &&
    $LLVM/llvm-objdump -d --no-show-raw-insn "$G/co$t.o" > "$G/dis$t.txt" ) &
done
wait
( cd "$G/wt" && python -c "import __graft_entry__ as G; G.build()" )
( cd "$ROOT" && python -c "import __graft_entry__ as G; G.build()" )
cat > "$G/dump.py" <<'PY'
import sys, os
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); os.chdir(root)
os.environ["DEMI_JIT_DUMP"] = out
from demi_amd import _native, model as M
assert _native.__file__.startswith(root)
_native.specialize_check(M.raft_model(5).to_struct())
PY
python "$G/dump.py" "$G/wt" "$G/jitA"; python "$G/dump.py" "$ROOT" "$G/jitB"
python - "$G" <<'PY'
import re, hashlib, sys, os, subprocess
G = sys.argv[1]
def per(fn):
    d, cur = {}, None
    for l in open(fn):
        m = re.match(r'^[0-9a-f]+ <(.*)>:', l)
        if m: cur = m.group(1); d[cur] = []; continue
        if cur and l.strip(): d[cur].append(re.sub(r'//.*', '', l).strip())
    return {k: (hashlib.md5("\n".join(v).encode()).hexdigest()[:8], len(v)) for k, v in d.items() if k.startswith('_Z')}
a, b = per(G + '/disA.txt'), per(G + '/disB.txt')
for k in sorted(set(a) | set(b)):
    print("generic     %s %-78s %s -> %s" % ("SAME" if a.get(k) == b.get(k) else "DIFF", k[:78], a.get(k, ("-", 0))[1], b.get(k, ("-", 0))[1]))
for k in range(16):
    h = []
    for t in "AB":
        p = "%s/jit%s.%d" % (G, t, k)
        if not os.path.exists(p): h.append(None); continue
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.text", p, p + ".text"])
        h.append(hashlib.md5(open(p + ".text", "rb").read()).hexdigest()[:8])
    if h[0] or h[1]: print("specialised %s kernel %d (raft5) %s -> %s" % ("SAME" if h[0] == h[1] else "DIFF", k, h[0], h[1]))
PY
