#!/usr/bin/env python
"""Condense rocprofv3 rocpd databases (gpurun_out/prof_*/*.db) into the small text/JSON summaries
committed under profiles/.  Usage: summarize_prof.py <tag> [gpurun_out]"""
import glob
import json
import os
import sqlite3
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "gpurun_out")
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)
lines, counters = [], {}
for d in sorted(glob.glob(os.path.join(src, "prof_*"))):
    if not os.path.isdir(d):
        continue
    dbs = glob.glob(os.path.join(d, "*.db"))
    if not dbs:
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    name = os.path.basename(d)
    if name == "prof_stats":
        lines.append("# rocprofv3 --kernel-trace --stats  (durations in ns)")
        lines.append("%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append("%-90s %8d %14.0f %12.0f %7.2f" % (r[0][:90], r[1], r[2] * 1000, r[3] * 1000, r[4]))
        lines.append("")
    else:
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "where kernel_name like '%demi%' group by kernel_name, counter_name")
        for kn, cn, cnt, avg in cur.execute(q):
            counters.setdefault(kn, {})[cn] = {"dispatches": cnt, "avg_per_dispatch": avg}
if counters:
    lines.append("# rocprofv3 --pmc (separate passes), average per dispatch")
    for kn in sorted(counters):
        lines.append(kn)
        for cn in sorted(counters[kn]):
            lines.append("    %-28s %20.1f   (%d dispatches)" % (cn, counters[kn][cn]["avg_per_dispatch"], counters[kn][cn]["dispatches"]))
txt = os.path.join(out_dir, tag + ".txt")
with open(txt, "w") as f:
    f.write("\n".join(lines) + "\n")
with open(os.path.join(out_dir, tag + "_counters.json"), "w") as f:
    json.dump(counters, f, indent=1, sort_keys=True)
print(open(txt).read())
