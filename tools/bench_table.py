"""Prints the records of a bench line (python bench.py > file) as the markdown table DESIGN.md section 5 carries per round:
   python tools/bench_table.py profiles/r06_bench_1gpu.json"""
import json
import sys


def g(d, *ks, default=None):
    for k in ks:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def roof(r):
    if not r:
        return ""
    t = r.get("traffic")
    im = r.get("issue_model") or {}
    s = "%.3g GB/s = %.3g %% of 8 TB/s" % (r.get("achieved", 0.0), 100.0 * r.get("frac", 0.0))
    if r.get("kernel_ms") is not None:
        s = "kernels %.3g ms; " % r["kernel_ms"] + s
    if t:
        s += "; traffic %.3g GB" % (t / 1e9)
    if im:
        lanes = im.get("active_lanes_per_valu_inst")
        if lanes:
            s += "; %.1f of 64 lanes per VALU instruction" % lanes
        if im.get("issue_frac_straight_line") is not None:
            s += "; issue share %.0f-%.0f %%" % (100 * im["issue_frac_straight_line"], 100 * im["issue_frac_branchy"])
        if im.get("wait_share_of_wave_cycles") is not None:
            s += ", %.0f %% of the wave-cycles waiting" % (100 * im["wait_share_of_wave_cycles"])
    return s


def main(path):
    d = json.load(open(path))
    rows = []
    cb = d.get("cpu_baseline") or {}
    rows.append(("K1 fuzz, config 2, 2^20 schedules per step (%s)" % g(d, "config", "workload", default="")[:60],
                 "%.3g %s, %.3f ms per step" % (d["value"], d["unit"], d["ms_per_step"]), roof(d.get("roofline")),
                 "%.3g/s on %s threads" % (cb.get("value", 0), cb.get("cores"))))
    for name, r in (d.get("secondary") or {}).items():
        if not isinstance(r, dict) or "value" not in r:
            continue
        cb = r.get("cpu_baseline") or {}
        extra = ""
        if "seconds" in r:
            extra = ", %.3g s" % r["seconds"]
        if name == "dpor":
            o = r.get("orders", {})
            extra = "; ".join("%s %.3g/s (%.3g s, %d interleavings, %d violating)" % (k, v["value"], v["seconds"], v["interleavings"], v["violations"]) for k, v in o.items())
            extra = " - " + extra
        if name == "config5" and "reference_order" in r:
            ro = r["reference_order"]
            if "value" in ro:
                extra += "; the reference's order %.3g/s (%.3g s)" % (ro["value"], ro["seconds"])
        rows.append((name, "%.3g %s%s" % (r["value"], r.get("unit", ""), extra), roof(r.get("roofline")),
                     ("%.3g/s on %s threads" % (cb.get("value", 0), cb.get("cores"))) if cb else ""))
    print("| record | GPU | roofline (SURVEY 8d: algorithmic bytes / kernel time) | C oracle |")
    print("|---|---|---|---|")
    for r in rows:
        print("| " + " | ".join(r) + " |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_bench_1gpu.json")
