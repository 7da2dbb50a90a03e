#!/usr/bin/env python
"""Static size of the main loop of a dumped K1 code object (DEMI_JIT_DUMP=<path> -> <path>.0): the span of the longest
backward branch, instruction counts by unit.  No GPU needed.   python tools/k1_loop_size.py /tmp/img.0"""
import re
import subprocess
import sys
import collections

def main(path):
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", path], capture_output=True, text=True).stdout.splitlines()
    ins = []
    for l in dis:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    best = None
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                off = int(args.split()[-1])
            except ValueError:
                continue
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt < a and tgt in addr_index:
                span = i - addr_index[tgt]
                if best is None or span > best[0]:
                    best = (span, addr_index[tgt], i)
    span, lo, hi = best
    body = [op for _, op, _ in ins[lo:hi + 1] if op != "s_nop"]
    kinds = collections.Counter(op.split("_")[0] for op in body)
    print("%s: total %d instructions (no s_nop); main loop %d: %s" % (path, sum(1 for _, op, _ in ins if op != "s_nop"), len(body), dict(kinds)))

if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
