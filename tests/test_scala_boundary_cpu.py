"""CPU suite: the Scala adapter (scala/akka/dispatch/verification/gpu/GpuSchedulers.scala) against the reference's plugin surface.
There is no scalac in this image, so the check is structural: every abstract member of `trait Scheduler`
(schedulers/Scheduler.scala:13-104) and of `trait TestOracle` (minification/TestOracle.scala:30-55) has a definition with the
same name and parameter count in the adapter, every GPU scheduler class extends `GpuSchedulerBase with TestOracle`
(and GpuSchedulerBase extends Scheduler, because the drivers assign the object to Instrumenter().scheduler), and every member
the reference's drivers use on their scheduler objects - RunnerUtils.fuzz / stsSchedDDMin / boundedDPOR - resolves on the GPU
class that stands in.  The reference-side facts are a committed fixture (tests/golden/scala_boundary.json, written by
tools/make_scala_fixture.py); where /root/reference exists the fixture is also checked to be current."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALA = os.path.join(ROOT, "scala", "akka", "dispatch", "verification", "gpu", "GpuSchedulers.scala")
FIXTURE = os.path.join(ROOT, "tests", "golden", "scala_boundary.json")


def _blocks(src):
    """{name: (header, body)} for the top-level classes / traits / objects of a Scala file."""
    out = {}
    for m in re.finditer(r"^(class|trait|object) (\w+)", src, re.M):
        start = m.start()
        brace = src.index("{", _header_end(src, m.end()))
        depth, i = 0, brace
        while True:
            depth += src[i] == "{"
            depth -= src[i] == "}"
            if depth == 0:
                break
            i += 1
        if m.group(1) == "object" and m.group(2) in out:      # a companion object: the class is what is checked
            continue
        out[m.group(2)] = (src[start:brace], src[brace + 1:i])
    return out


def _header_end(src, i):
    """skips a constructor parameter list (which may contain braces in default arguments: none here, but parentheses nest)"""
    while src[i] in " \t":
        i += 1
    if src[i] != "(":
        return i
    depth = 0
    while True:
        depth += src[i] == "("
        depth -= src[i] == ")"
        i += 1
        if depth == 0:
            return i


def _n_params(sig):
    if "(" not in sig:
        return 0
    inner = sig[sig.index("(") + 1:sig.rindex(")")]
    if not inner.strip():
        return 0
    depth, n = 0, 1
    for ch in inner:
        depth += ch in "[("
        depth -= ch in "])"
        n += ch == "," and depth == 0
    return n


def _members(body):
    """(name, n_params) of every def / val / var at the top level of a class body"""
    out, depth = set(), 0
    lines = body.split("\n")
    k = 0
    while k < len(lines):
        line = lines[k]
        s = line.strip()
        if depth == 0:
            m = re.match(r"(?:override |protected |private )*def (\w+)", s)
            if m:
                decl = s
                while decl.count("(") > decl.count(")"):
                    k += 1
                    decl += " " + lines[k].strip()
                    line += lines[k]
                name = m.group(1)
                after = decl[decl.index("def " + name) + len("def " + name):]
                if after.lstrip().startswith("("):
                    d, j = 0, decl.index("(", decl.index("def " + name))
                    while True:
                        d += decl[j] == "("
                        d -= decl[j] == ")"
                        if d == 0:
                            break
                        j += 1
                    out.add((name, _n_params(decl[decl.index("def " + name):j + 1])))
                else:
                    out.add((name, 0))
            m = re.match(r"(?:override |protected |private )*(?:val|var) (\w+)", s)
            if m:
                out.add((m.group(1), 0))
        depth += line.count("{") - line.count("}")
        k += 1
    return out


@pytest.fixture(scope="module")
def adapter():
    return _blocks(open(SCALA).read())


@pytest.fixture(scope="module")
def ref():
    return json.load(open(FIXTURE))


def test_fixture_is_current_where_the_reference_is_present():
    if not os.path.isdir("/root/reference/src/main/scala/verification"):
        pytest.skip("no /root/reference on this box: the committed fixture stands")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_scala_fixture
    assert make_scala_fixture.build() == json.load(open(FIXTURE)), "re-run tools/make_scala_fixture.py"


def test_base_trait_implements_every_abstract_member_of_scheduler(adapter, ref):
    header, body = adapter["GpuSchedulerBase"]
    assert re.search(r"trait GpuSchedulerBase extends Scheduler\b", header)
    have = _members(body)
    missing = [(m["name"], m["params"]) for m in ref["Scheduler"] if m["abstract"] and (m["name"], m["params"]) not in have
               and m["name"] != "shutdown"]       # shutdown() frees each class's demi_ctx: defined per class, checked below
    assert not missing, "GpuSchedulerBase lacks %s" % missing
    assert ("setInvariant", 1) in have            # TestOracle.setInvariant: stored, and test() / explore() insist on it


@pytest.mark.parametrize("cls", ["GpuRandomScheduler", "GpuSTSScheduler", "GpuDPOR"])
def test_gpu_schedulers_are_schedulers_with_testoracle(adapter, ref, cls):
    header, body = adapter[cls]
    assert re.search(r"extends GpuSchedulerBase with TestOracle\b", header), header[-200:]
    have = _members(body) | _members(adapter["GpuSchedulerBase"][1])
    for m in ref["TestOracle"]:
        if m["abstract"]:
            assert (m["name"], m["params"]) in have, (cls, m)
    assert ("shutdown", 0) in _members(body)
    # "Throws an IllegalArgumentException if setInvariant has not been invoked" (TestOracle.scala:45)
    assert "requireInvariant()" in body
    assert "IllegalArgumentException" in adapter["GpuSchedulerBase"][1]


def test_every_member_the_reference_drivers_use_resolves(adapter, ref):
    base = _members(adapter["GpuSchedulerBase"][1])
    for driver, spec in ref["drivers"].items():
        have = {n for n, _ in _members(adapter[spec["class"]][1]) | base}
        missing = [c for c in spec["calls"] if c not in have]
        assert not missing, "%s: %s lacks %s" % (driver, spec["class"], missing)
    # sched.depTracker.getGraph / getInitialTrace (RunnerUtils.scala:99-100): the adapter's depTracker IS the reference's class
    body = adapter["GpuRandomScheduler"][1]
    assert re.search(r"var depTracker = new DepTracker\(schedulerConfig\)", body)
    assert "GpuDepTracker.fromTrace" in body and "GpuDepTracker" in adapter
    dep_src = adapter["GpuDepTracker"][1]
    for call in ("reportNewlyEnabled", "reportNewlyEnabledExternal", "reportNewlyDelivered", "reportKill", "reportPartition", "reportUnPartition"):
        assert call in dep_src, call


def test_integration_doc_matches_the_source():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(SCALA).read()
    for cls in ("GpuRandomScheduler", "GpuSTSScheduler", "GpuDPOR"):
        assert re.search(r"class %s\([^)]*\)[^{]*extends GpuSchedulerBase with TestOracle" % cls, src, re.S)
    assert "GpuSchedulerBase" in doc and "extends GpuSchedulerBase with TestOracle" in doc
    assert "extends Scheduler with TestOracle" not in doc.replace("trait GpuSchedulerBase extends Scheduler", "")
