#!/usr/bin/env python
"""What the run-time specialiser produces for a model, without a GPU: per kernel the instruction count (padding s_nop
excluded), VGPRs / SGPRs / spills / scratch, and the code id bench.py reports (hash of .text) - with PyTorch's bundled
compiler (what `python bench.py` uses) and, with --system, with /opt/rocm's (what a host without PyTorch gets).

  python tools/jit_stats.py [--system] [--wide] [--flags "-O3"]
"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["k1_random_explore<false,false>", "k2_replay (+fp, fp_hbm)", "k3_dpor", "k1_random_explore<false,true> (SrcDstFIFO)",
         "k2 (same TU)", "k2 (same TU)"]


def text_hash(b):
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    stroff = struct.unpack_from("<Q", b, shoff + shstrndx * shentsize + 0x18)[0]
    lo, hi = 0, len(b)
    for i in range(shnum):
        sh = shoff + i * shentsize
        name = struct.unpack_from("<I", b, sh)[0]
        off, size = struct.unpack_from("<QQ", b, sh + 0x18)
        if b[stroff + name:stroff + name + 6] == b".text\0":
            lo, hi = off, off + size
    h = 0xCBF29CE484222325
    for c in b[lo:hi]:
        h = ((h ^ c) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % (h or 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", action="store_true", help="compile with /opt/rocm's comgr (LD_PRELOAD) instead of PyTorch's")
    ap.add_argument("--wide", action="store_true", help="the raft lowered as a wide table")
    ap.add_argument("--log-cap", type=int, default=0, help="the raft with a real log of this many entries (DEMI_MODEL_ARRAY)")
    ap.add_argument("--real-fields", action="store_true", help="with --log-cap: akka-raft's field sets (DEMI_MODEL_PAYLOADS(5))")
    ap.add_argument("--flags", default="", help="DEMI_JIT_FLAGS")
    args = ap.parse_args()
    d = tempfile.mkdtemp()
    code = ("import sys; sys.path.insert(0, %r)\nfrom demi_amd import _native, model as M\n"
            "m = M.raft_model(5, log_cap=%d, real_fields=%r) if %d else M.raft_model(5, term0=1000, loglen0=300) if %r else M.raft_model(5)\n"
            "print(_native.specialize_check(m.to_struct())[0])\n" % (ROOT, args.log_cap, args.real_fields, args.log_cap, args.wide))
    env = dict(os.environ, DEMI_EXPERIMENT="1", DEMI_JIT_DUMP=os.path.join(d, "img"))
    if args.flags:
        env["DEMI_JIT_FLAGS"] = args.flags
    if args.system:
        env["LD_PRELOAD"] = "/opt/rocm/lib/libamd_comgr.so.3"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if out.returncode:
        sys.exit(out.stdout + out.stderr)
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    for k in list(range(6)) + [13]:
        p = os.path.join(d, "img.%d" % k)
        if not os.path.exists(p) or k in (4, 5):
            continue
        b = open(p, "rb").read()
        dis = subprocess.run([objdump, "-d", p], capture_output=True, text=True).stdout.splitlines()
        ins = [l.split()[0] for l in dis if l.startswith("\t") and l.split()]
        notes = subprocess.run([readelf, "--notes", p], capture_output=True, text=True).stdout
        meta = [l.strip() for l in notes.splitlines() if any(x in l for x in (".name:", ".vgpr_count", ".sgpr_count", "spill_count", "private_segment_fixed"))
                and ".name:           hidden" not in l and "value_kind" not in l]
        comp = [l for l in subprocess.run([readelf, "-p", ".comment", p], capture_output=True, text=True).stdout.splitlines() if "clang version" in l]
        print("== kernel %d: %s" % (k, NAMES[k] if k < len(NAMES) else "k1_random_explore<false,false,false,true> (re-binned)"))
        print("   instructions %d (+ %d s_nop), code id %s, %d bytes" % (sum(1 for i in ins if i != "s_nop"), ins.count("s_nop"), text_hash(b), len(b)))
        print("   " + "  ".join(meta))
        if comp and k == 0:
            print("   " + comp[0].split("]")[-1].strip()[:110])


if __name__ == "__main__":
    main()
