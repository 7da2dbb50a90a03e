"""Host-side delta debugging over the GPU replay oracle.

Mirrors (same names and semantics):
  MinificationUtil.split_list            minification/Util.scala:9-37
  AtomicEvent                            minification/Util.scala:46-63
  UnmodifiedEventDag / EventDagView      minification/Util.scala:161-304 (atoms :197-265)
  DDMin.minimize / ddmin2 / verify_mcs   minification/DeltaDebugging.scala:27-109
  RunnerUtils.stsSchedDDMin              RunnerUtils.scala:642-707

An external event is identified by its index in the original external-event list (the reference
identifies ExternalEvents by their unique `_id`, ExternalEvents.scala:14-31), so a subsequence is a
sorted tuple of indices and a candidate is a 256-bit mask.

`ddmin2` is a sequential decision tree: every node consults the oracle on (left half U remainder)
and, if that passes, on (right half U remainder).  One oracle call is one kernel lane, so the GPU
path evaluates the tree SPECULATIVELY: before descending, `SpeculativeDDMin` enumerates every
candidate the next `depth` levels could ask for (all outcomes), tests them in one K2 launch
(sharded across ranks when a process group exists), and then walks the real path through the
cache.  The MCS, the verdict consulted at every step, and MinimizationStats.total_replays are
identical to the sequential algorithm; `speculative_replays` counts what was actually launched.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import types as T
from .schedulers import MinimizationStats, ViolationFingerprint


def split_list(l: Sequence, split_ways: int) -> List[List]:
    """MinificationUtil.split_list: the first `remainder` chunks get the extra element."""
    if split_ways < 1:
        raise ValueError("Split ways must be greater than 0")
    interval, remainder = len(l) // split_ways, len(l) % split_ways
    splits, start = [], 0
    while len(splits) < split_ways:
        end = start + interval
        if remainder > 0:
            end += 1
            remainder -= 1
        splits.append(list(l[start:end]))
        start = end
    return splits


Atom = Tuple[int, ...]       # AtomicEvent: indices of external events that are removed together


class UnmodifiedEventDag:
    """The full external-event list (EventDag over all indices)."""

    def __init__(self, externals: np.ndarray):
        self.externals = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
        self._events: Tuple[int, ...] = tuple(range(len(self.externals)))
        self._conjoined: Dict[int, int] = {}
        # (kind, a, b) of every event as plain ints, and the atoms of every subsequence asked for so far: DDMin's frontier
        # enumeration asks for the atoms of hundreds of views of this one dag
        self._kab = list(zip(self.externals["kind"].tolist(), self.externals["a"].tolist(), self.externals["b"].tolist()))
        self._atoms_of: Dict[Tuple[int, ...], List[Atom]] = {}

    def conjoinAtoms(self, e1: int, e2: int):
        assert e1 not in self._conjoined and e2 not in self._conjoined
        self._conjoined[e1] = e2
        self._conjoined[e2] = e1
        self._atoms_of.clear()

    # -- EventDag trait
    def remove_events(self, to_remove: Sequence[Atom]) -> "EventDagView":
        return EventDagView(self, _remove(to_remove, self._events))

    def union(self, other) -> "UnmodifiedEventDag":
        if other.length != 0:
            raise ValueError("Unknown events")
        return self

    def get_atomic_events(self, given: Optional[Sequence[int]] = None) -> List[Atom]:
        """Util.scala:197-265: explicit pairs first; Kill pairs with the remembered Start of the same
        actor, UnPartition with the remembered Partition of the same ordered pair; the rest are
        singletons; sorted by the index of the first event."""
        given = self._events if given is None else tuple(given)
        cached = self._atoms_of.get(given)
        if cached is not None:
            return list(cached)
        atoms: List[Atom] = []
        done = set()
        for e in given:
            if e in self._conjoined and e not in done:
                o = self._conjoined[e]
                assert o in given
                atoms.append((e, o))
                done.update((e, o))
        prev: Dict[Tuple, int] = {}
        for e in given:
            if e in self._conjoined:
                continue
            kind, a, b = self._kab[e]
            if kind == T.EV_KILL:
                if ("n", a) not in prev:
                    raise RuntimeError("Kill without preceding Start")
                atoms.append((prev.pop(("n", a)), e))
            elif kind == T.EV_PARTITION:
                prev[("p", a, b)] = e
            elif kind == T.EV_START:
                prev[("n", a)] = e
            elif kind == T.EV_UNPARTITION:
                if ("p", a, b) not in prev:
                    raise RuntimeError("UnPartition without preceding Partition")
                atoms.append((prev.pop(("p", a, b)), e))
            else:
                atoms.append((e,))
        atoms.extend((e,) for e in prev.values())
        # `assume(atomics...flatten.length == given_events.length)`: a Start/Partition overwritten in the
        # map (two Starts of one actor with no Kill between them) trips the reference's assumption
        if sum(len(a) for a in atoms) != len(given):
            raise AssertionError("assumption failed: atoms do not partition the events")
        atoms = sorted(atoms, key=lambda a: a[0])
        if len(self._atoms_of) < 65536:
            self._atoms_of[given] = atoms
        return list(atoms)

    def get_all_events(self) -> Tuple[int, ...]:
        return self._events

    events = property(get_all_events)

    @property
    def length(self) -> int:
        return len(self._events)


class EventDagView:
    """A subsequence of an UnmodifiedEventDag (Util.scala:271-304)."""

    def __init__(self, parent: UnmodifiedEventDag, events: Sequence[int]):
        self.parent = parent
        self._events = tuple(events)

    def remove_events(self, to_remove: Sequence[Atom]) -> "EventDagView":
        return EventDagView(self.parent, _remove(to_remove, self._events))

    def union(self, other) -> "EventDagView":
        u = sorted(set(self._events) | set(other.get_all_events()))
        assert len(self._events) + other.length == len(u)
        return EventDagView(self.parent, u)

    def get_atomic_events(self) -> List[Atom]:
        return self.parent.get_atomic_events(self._events)

    def get_all_events(self) -> Tuple[int, ...]:
        return self._events

    events = property(get_all_events)

    @property
    def length(self) -> int:
        return len(self._events)


def _remove(to_remove: Sequence[Atom], events: Sequence[int]) -> Tuple[int, ...]:
    flat = [e for atom in to_remove for e in atom]
    gone = set(flat)
    assert len(flat) == len(gone)
    return tuple(e for e in events if e not in gone)


def events_to_mask(events: Sequence[int]) -> np.ndarray:
    return events_to_masks([events])[0]


def events_to_masks(subseqs) -> np.ndarray:
    """[n][4] u64: bit e of row r = external event e is part of candidate r (the 256-bit masks K2 takes)."""
    bits = np.zeros((len(subseqs), 256), dtype=np.uint8)
    for r, sub in enumerate(subseqs):
        if len(sub):
            bits[r, np.fromiter(sub, dtype=np.int64, count=len(sub))] = 1
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(len(subseqs), 4)


class DDMin:
    """DDMin (DeltaDebugging.scala:7-110), Zeller '99 ddmin2, one oracle.test per candidate."""

    def __init__(self, oracle, checkUnmodifed: bool = False, stats: Optional[MinimizationStats] = None):
        self.oracle = oracle
        self.checkUnmodifed = checkUnmodifed
        self._stats = stats or MinimizationStats()
        self.violation_fingerprint: Optional[ViolationFingerprint] = None
        self.original_num_events = 0
        self.total_inputs_pruned = 0
        self.consulted: List[Tuple[Tuple[int, ...], bool]] = []    # (candidate, passes) in consultation order

    # oracle.test(...) == None  <=>  "passes" (the violation is NOT reproduced)
    def _passes(self, events: Tuple[int, ...]) -> bool:
        trace = self.oracle.test(events, self.violation_fingerprint, self._stats)
        passes = trace is None
        self.consulted.append((tuple(events), passes))
        return passes

    def minimize(self, dag, _violation_fingerprint: ViolationFingerprint):
        self.violation_fingerprint = _violation_fingerprint
        if self.checkUnmodifed:
            if self.oracle.test(dag.events, self.violation_fingerprint, self._stats) is None:
                raise ValueError("Unmodified trace does not trigger violation")
        self._stats.total_replays = 0          # _stats.reset()
        self.original_num_events = dag.length
        self.total_inputs_pruned = 0
        self.consulted = []
        mcs_dag = self.ddmin2(dag, EventDagView(_parent_of(dag), ()))
        assert self.original_num_events - self.total_inputs_pruned == mcs_dag.length
        return mcs_dag

    def verify_mcs(self, mcs, _violation_fingerprint: ViolationFingerprint):
        return self.oracle.test(mcs.events, _violation_fingerprint, MinimizationStats())

    def ddmin2(self, dag, remainder):
        atoms = dag.get_atomic_events()
        if len(atoms) <= 1:
            return dag
        halves = split_list(atoms, 2)
        # `.map(split => dag.remove_events(split)).reverse`: splits(0) keeps the first half
        splits = [dag.remove_events(h) for h in halves][::-1]
        for split in splits:
            union = split.union(remainder)
            if not self._passes(union.get_all_events()):
                self.total_inputs_pruned += dag.length - split.length
                return self.ddmin2(split, remainder)
        left = self.ddmin2(splits[0], splits[1].union(remainder))
        right = self.ddmin2(splits[1], splits[0].union(remainder))
        return left.union(right)


def _parent_of(dag) -> UnmodifiedEventDag:
    return dag if isinstance(dag, UnmodifiedEventDag) else dag.parent


class SpeculativeDDMin(DDMin):
    """ddmin2 with the oracle consulted through a cache that is filled `depth` levels ahead by
    batched (and, across ranks, sharded) K2 launches.  Same MCS, same consultation sequence."""

    def __init__(self, oracle, depth: int = 3, checkUnmodifed: bool = False,
                 stats: Optional[MinimizationStats] = None):
        super().__init__(oracle, checkUnmodifed, stats)
        self.depth = depth
        self.cache: Dict[Tuple[int, ...], bool] = {}
        self.speculative_replays = 0
        self.batches: List[int] = []

    def _frontier(self, dag, remainder, depth: int, out: Dict[Tuple[int, ...], None]):
        atoms = dag.get_atomic_events()
        if len(atoms) <= 1 or depth == 0:
            return
        splits = [dag.remove_events(h) for h in split_list(atoms, 2)][::-1]
        cands = [tuple(s.union(remainder).get_all_events()) for s in splits]
        known = [self.cache.get(c) for c in cands]
        for c, k in zip(cands, known):
            if k is None:
                out.setdefault(c)
        # follow every outcome that is still possible, one level deeper
        k0, k1 = known
        if k0 is not True:                              # left may fail -> ddmin2(left, remainder)
            self._frontier(splits[0], remainder, depth - 1, out)
        if k0 is not False and k1 is not True:          # left passes, right may fail -> ddmin2(right, remainder)
            self._frontier(splits[1], remainder, depth - 1, out)
        if k0 is not False and k1 is not False:         # both may pass -> interference
            self._frontier(splits[0], splits[1].union(remainder), depth - 1, out)
            self._frontier(splits[1], splits[0].union(remainder), depth - 1, out)

    def _passes(self, events: Tuple[int, ...]) -> bool:
        events = tuple(events)
        if events not in self.cache:
            todo: Dict[Tuple[int, ...], None] = {events: None}
            dag, remainder = self._node
            self._frontier(dag, remainder, self.depth, todo)
            cands = list(todo)
            results = self.oracle.test_batch(cands, self.violation_fingerprint, None)
            self.speculative_replays += len(cands)
            self.batches.append(len(cands))
            for c, reproduced in zip(cands, results):
                self.cache[c] = not reproduced
        self._stats.increment_replays()
        passes = self.cache[events]
        self.consulted.append((events, passes))
        return passes

    def ddmin2(self, dag, remainder):
        self._node = (dag, remainder)
        atoms = dag.get_atomic_events()
        if len(atoms) <= 1:
            return dag
        splits = [dag.remove_events(h) for h in split_list(atoms, 2)][::-1]
        for split in splits:
            self._node = (dag, remainder)
            if not self._passes(split.union(remainder).get_all_events()):
                self.total_inputs_pruned += dag.length - split.length
                return self.ddmin2(split, remainder)
        left = self.ddmin2(splits[0], splits[1].union(remainder))
        right = self.ddmin2(splits[1], splits[0].union(remainder))
        return left.union(right)


def stsSchedDDMin(oracle, externals: np.ndarray, violation: ViolationFingerprint, speculative_depth: int = 3,
                  stats: Optional[MinimizationStats] = None, checkUnmodified: bool = True):
    """RunnerUtils.stsSchedDDMin (:642-707): strip WaitQuiescence from the externals, minimise with
    DDMin over the STSSched oracle, verify the MCS.  Returns (mcs indices, ddmin, verified trace).
    checkUnmodified (default true, as in the reference, :653, 671): the unmodified trace must reproduce the violation under
    STSSched, else ValueError - a non-reproducing input would otherwise "minimise" to an arbitrary single atom.  (The
    reference validates the MCS only when it is smaller than the input, :689; here it is always replayed once: the
    verified trace is what the later stages start from.)"""
    dag = UnmodifiedEventDag(externals)
    keep = tuple(i for i in dag.events if int(externals[i]["kind"]) != T.EV_WAIT_QUIESCENCE)
    view = EventDagView(dag, keep)
    ddmin = SpeculativeDDMin(oracle, depth=speculative_depth, checkUnmodifed=checkUnmodified, stats=stats) if speculative_depth > 0 else \
        DDMin(oracle, checkUnmodifed=checkUnmodified, stats=stats)
    mcs = ddmin.minimize(view, violation)
    verified = ddmin.verify_mcs(mcs, violation)
    return mcs.get_all_events(), ddmin, verified


class _SubsequenceOracle:
    """DDMin hands the oracle index tuples (EventDag events); a scheduler's test() wants the events themselves."""

    def __init__(self, sched, externals: np.ndarray):
        self.sched = sched
        self.externals = externals

    def getName(self) -> str:
        return self.sched.getName()

    def test(self, events, violation_fingerprint: ViolationFingerprint, stats: Optional[MinimizationStats] = None):
        return self.sched.test(self.externals[list(events)], violation_fingerprint, stats)


def randomDDMin(schedulerConfig, trace, violation: ViolationFingerprint, max_executions: int = 100, seed_base: int = 0,
                stats: Optional[MinimizationStats] = None, device: int = 0, p_max: int = 64, native: bool = False,
                specialize: Optional[bool] = None, max_candidates: int = 256, sequential: bool = False):
    """RunnerUtils.randomDDMin (RunnerUtils.scala:601-623): DDMin whose oracle is the RandomScheduler itself —
    a candidate subsequence "fails" iff one of `max_executions` random interleavings of it (the reference constructs the
    scheduler with max_executions = 1) reproduces the violation fingerprint.  maxMessages is the length of the recorded trace,
    as in the reference.  Returns (mcs indices, ddmin, verified trace or None).

    native=False: the reference's loop written out here - sequential DDMin, one K1 launch per consultation (the table
    interpreted unless `specialize`).  native=True: demi_random_ddmin - decision tree, speculative frontier and launches inside
    the library, one launch for (frontier candidates x max_executions) with a workgroup per candidate, the table compiled
    (unless specialize=False); `ddmin` is then a record with the same fields the mirror's DDMin exposes (consulted,
    total_inputs_pruned is not kept) plus `.stats` (demi_ddmin_stats) and `.batches`."""
    from .schedulers import RandomScheduler
    if specialize is None:
        specialize = native
    sched = RandomScheduler(schedulerConfig, max_executions, 0, seed_base=seed_base, device=device, p_max=p_max,
                            specialize=specialize)
    sched.setMaxMessages(len(trace.events))
    try:
        externals = trace.original_externals
        if native:
            sched._prepare(externals)
            lim = sched._limits(violation)
            par = T.RandomDdminParams(executions=max_executions, max_candidates=max_candidates, check_unmodified=0, verify_mcs=1,
                                      sequential=1 if sequential else 0)
            mcs_idx, consulted, batches, st = sched._ctx.random_ddmin(lim, par, seed_base=seed_base)
            if stats is not None:
                stats.increment_replays(int(st.consultations) * max_executions)
            ddmin = _NativeDdminRecord(consulted, batches, st)
            if len(mcs_idx) < len(externals):
                verified = sched.test(externals[list(mcs_idx)], violation) if st.verified else None
            else:
                verified = trace
            return tuple(mcs_idx), ddmin, verified
        ddmin = DDMin(_SubsequenceOracle(sched, externals), checkUnmodifed=False, stats=stats)
        mcs = ddmin.minimize(UnmodifiedEventDag(externals), violation)
        verified = ddmin.verify_mcs(mcs, violation) if mcs.length < len(externals) else trace
    finally:
        sched.shutdown()
    return mcs.get_all_events(), ddmin, verified


class _NativeDdminRecord:
    """What demi_random_ddmin reports, under the names the mirror's DDMin uses."""

    def __init__(self, consulted, batches, st):
        self.consulted = [(tuple(int(i) for i in c), bool(p)) for c, p in consulted]
        self.batches = list(batches)
        self.stats = st
        self.total_consultations = int(st.consultations)
