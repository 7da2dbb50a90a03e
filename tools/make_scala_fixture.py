#!/usr/bin/env python
"""Writes tests/golden/scala_boundary.json from the reference's Scala sources (run in the build container, where /root/reference
exists): what a class has to define to BE a `Scheduler with TestOracle` (the abstract members of schedulers/Scheduler.scala and
minification/TestOracle.scala), and which members the reference's drivers use on the scheduler objects they construct
(RunnerUtils.fuzz / stsSchedDDMin / boundedDPOR).  tests/test_scala_boundary_cpu.py holds scala/.../GpuSchedulers.scala
against this file; when /root/reference is present it also checks that the file is current."""
import json
import os
import re
import sys

REF = "/root/reference/src/main/scala/verification"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def split_params(sig):
    """number of parameters of `name(a: T, b: U[V, W])` (0 for `name` / `name()`)"""
    if "(" not in sig:
        return 0
    inner = sig[sig.index("(") + 1:sig.rindex(")")]
    if not inner.strip():
        return 0
    depth, n = 0, 1
    for ch in inner:
        if ch in "[(":
            depth += 1
        elif ch in "])":
            depth -= 1
        elif ch == "," and depth == 0:
            n += 1
    return n


def trait_members(path, trait):
    """(name, n_params, abstract) for every `def` directly in `trait <trait> { ... }`"""
    src = open(path).read()
    start = src.index("trait %s" % trait)
    body_start = src.index("{", start)
    depth, i = 0, body_start
    while True:
        if src[i] == "{":
            depth += 1
        elif src[i] == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    body = src[body_start + 1:i]
    out, depth = [], 0
    lines = body.split("\n")
    k = 0
    while k < len(lines):
        line = lines[k]
        stripped = line.strip()
        if depth == 0 and stripped.startswith("def "):
            decl = stripped
            while decl.count("(") > decl.count(")"):          # a signature spread over several lines
                k += 1
                decl += " " + lines[k].strip()
                line += lines[k]
            name = re.match(r"def (\w+)", decl).group(1)
            after = decl[len("def " + name):]
            if after.lstrip().startswith("("):               # the parameter list: up to ITS closing parenthesis
                d, j = 0, decl.index("(", len("def " + name))
                while True:
                    d += decl[j] == "("
                    d -= decl[j] == ")"
                    if d == 0:
                        break
                    j += 1
                sig, rest = decl[:j + 1], decl[j + 1:]
            else:
                sig, rest = "def " + name, after
            abstract = "=" not in rest and "{" not in rest
            # (a parameter list may itself contain '=' for defaults: look only behind the closing parenthesis / return type)
            out.append({"name": name, "params": split_params(sig), "abstract": abstract})
        depth += line.count("{") - line.count("}")
        k += 1
    return out


def driver_calls(path, lo, hi, var):
    lines = open(path).read().split("\n")[lo - 1:hi]
    return sorted(set(re.findall(r"\b%s\.(\w+)" % var, "\n".join(lines))))


def build():
    sched = trait_members(os.path.join(REF, "schedulers", "Scheduler.scala"), "Scheduler")
    oracle = trait_members(os.path.join(REF, "minification", "TestOracle.scala"), "TestOracle")
    ru = os.path.join(REF, "RunnerUtils.scala")
    return {
        "source": "NetSys/demi: schedulers/Scheduler.scala:13-104, minification/TestOracle.scala:30-55, RunnerUtils.scala:62-147, 642-707, 881-911",
        "Scheduler": sched, "TestOracle": oracle,
        "drivers": {
            "fuzz (RandomScheduler, RunnerUtils.scala:62-147)": {"class": "GpuRandomScheduler", "calls": driver_calls(ru, 62, 147, "sched"),
                                                                 "depTracker": driver_calls(ru, 62, 147, r"sched\.depTracker")},
            "stsSchedDDMin (STSScheduler, RunnerUtils.scala:642-707)": {"class": "GpuSTSScheduler", "calls": driver_calls(ru, 642, 707, "sched")},
            "boundedDPOR (DPORwHeuristics, RunnerUtils.scala:881-911)": {"class": "GpuDPOR", "calls": driver_calls(ru, 881, 911, "dpor")},
        },
        "assigned_to_Instrumenter_scheduler": True,      # RunnerUtils.scala:89, 666: the object must be a Scheduler
    }


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "scala_boundary.json")
    json.dump(build(), open(out, "w"), indent=1, sort_keys=True)
    print(open(out).read()[:3000])
