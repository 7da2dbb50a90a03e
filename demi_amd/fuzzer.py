"""External-event trace generation with the distribution of DEMi's Fuzzer.

Follows src/main/scala/verification/fuzzing/Fuzzer.scala (weights :24-29, event choice :44-57,
generateNextEvent :83-120, generateFuzzTest :122-175).  The reference reseeds from the wall clock
on every call (:67-68, :177-179), so its traces are not reproducible; here one seeded
java.util.Random drives every choice and the generated traces are frozen as golden files.
Trace generation is off the hot path.
"""
from dataclasses import dataclass
from typing import Callable, List, Tuple

import numpy as np

from . import types as T

_MULT = 0x5DEECE66D
_MASK = (1 << 48) - 1


class JavaRandom:
    """java.util.Random (JDK javadoc LCG)."""

    def __init__(self, seed: int):
        self.s = (seed ^ _MULT) & _MASK

    def next(self, bits: int) -> int:
        self.s = (self.s * _MULT + 0xB) & _MASK
        v = self.s >> (48 - bits)
        if v >= 1 << 31:
            v -= 1 << 32
        return v

    def next_int(self, bound: int = None) -> int:
        if bound is None:
            return self.next(32)
        r = self.next(31)
        m = bound - 1
        if bound & m == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            t = (u - r + m) & 0xFFFFFFFF
            if t < (1 << 31):
                return r
            u = self.next(31)

    def next_double(self) -> float:
        return ((self.next(26) << 27) + self.next(27)) * (1.0 / (1 << 53))


@dataclass
class FuzzerWeights:          # Fuzzer.scala:24-29
    kill: float = 0.01
    send: float = 0.3
    wait_quiescence: float = 0.1
    partition: float = 0.1
    unpartition: float = 0.1


Event = Tuple[int, int, int, int, int, int]   # kind, a, b, msg_type, p0, p1


def start(a):
    return (T.EV_START, a, 0, 0, 0, 0)


def kill(a):
    return (T.EV_KILL, a, 0, 0, 0, 0)


def send(a, msg_type, p0=0, p1=0):
    return (T.EV_SEND, a, 0, msg_type, p0, p1)


def partition(a, b):
    return (T.EV_PARTITION, a, b, 0, 0, 0)


def unpartition(a, b):
    return (T.EV_UNPARTITION, a, b, 0, 0, 0)


def wait_quiescence():
    return (T.EV_WAIT_QUIESCENCE, 0, 0, 0, 0, 0)


def events_to_array(events: List[Event]) -> np.ndarray:
    arr = np.zeros(len(events), dtype=T.EXT_EVENT_DTYPE)
    for i, e in enumerate(events):
        kind, a, b, msg_type, p0, p1 = e
        arr[i]["kind"], arr[i]["a"], arr[i]["b"], arr[i]["msg_type"] = kind, a, b, msg_type
        arr[i]["p0"], arr[i]["p1"] = p0 & 0xFF, p1 & 0xFF
        arr[i]["p0_hi"], arr[i]["p1_hi"] = p0 >> 8, p1 >> 8          # 16-bit payloads: wide models only
    return arr


def array_to_events(arr: np.ndarray) -> List[Event]:
    return [(int(e["kind"]), int(e["a"]), int(e["b"]), int(e["msg_type"]), int(e["p0"]) | (int(e["p0_hi"]) << 8),
             int(e["p1"]) | (int(e["p1_hi"]) << 8)) for e in arr]


class _RandSet:
    """RandomizedHashSet (schedulers/Util.scala:110-185) driven by the shared RNG."""

    def __init__(self, rng):
        self.arr, self.rng = [], rng

    def insert(self, v):
        self.arr.append(v)

    def remove_random(self):
        i = self.rng.next_int(len(self.arr))
        v = self.arr[i]
        self.arr[i] = self.arr[-1]
        self.arr.pop()
        return v

    def get_random(self):
        return self.arr[self.rng.next_int(len(self.arr))]

    def __len__(self):
        return len(self.arr)


def generate_fuzz_test(num_events: int, weights: FuzzerWeights, message_gen: Callable, prefix: List[Event],
                       seed: int, postfix: List[Event] = ()) -> List[Event]:
    """Fuzzer.generateFuzzTest (Fuzzer.scala:122-175).  message_gen(rng, alive_set) -> Send event."""
    rng = JavaRandom(seed)
    nodes = [e[1] for e in prefix if e[0] == T.EV_START]
    alive = _RandSet(rng)
    for n in nodes:
        alive.insert(n)
    parted, unparted = _RandSet(rng), _RandSet(rng)
    for i in range(len(nodes)):
        for j in range(i + 1, len(nodes)):
            unparted.insert((nodes[i], nodes[j]))
    weights_list = [weights.kill, weights.send, weights.partition, weights.unpartition]
    total = sum(weights_list) + weights.wait_quiescence

    def next_event():
        while True:
            scaled = rng.next_double() * total
            cur, cls = 0.0, None
            for idx, w in enumerate(weights_list):
                cur += w
                if scaled < cur:
                    cls = idx
                    break
            if cls is None:
                return wait_quiescence()
            if cls == 0:
                if len(alive) == 0:
                    return None
                return kill(alive.remove_random())
            if cls == 1:
                return message_gen(rng, alive)
            if cls == 2:
                if len(unparted) == 0:
                    continue
                pair = unparted.remove_random()
                parted.insert(pair)
                return partition(*pair)
            if len(parted) == 0:
                continue
            pair = parted.remove_random()
            unparted.insert(pair)
            return unpartition(*pair)

    out = list(prefix)
    just_wq = bool(out) and out[-1][0] == T.EV_WAIT_QUIESCENCE
    for _ in range(num_events):
        ev = next_event()
        while ev is not None and ev[0] == T.EV_WAIT_QUIESCENCE and just_wq:
            ev = next_event()
        if ev is None:
            return out
        just_wq = ev[0] == T.EV_WAIT_QUIESCENCE
        out.append(ev)
    out.extend(postfix)
    if out and out[-1][0] != T.EV_WAIT_QUIESCENCE:
        out.append(wait_quiescence())
    return out


def raft_trace(n_actors: int, n_events: int, seed: int, weights: FuzzerWeights = None, exact: bool = True) -> List[Event]:
    """Start x A, Bootstrap x A, then Fuzzer-distributed events; exactly n_events long (exact=False: at
    most n_events + 1; the Fuzzer stops early once every node is killed)."""
    from .model import M_BOOTSTRAP, M_CLIENT
    weights = weights or FuzzerWeights()
    prefix = [start(a) for a in range(n_actors)] + [send(a, M_BOOTSTRAP) for a in range(n_actors)]
    counter = [0]

    def gen(rng, alive):
        counter[0] += 1
        target = alive.get_random() if len(alive) else 0
        return send(target, M_CLIENT, counter[0] & 0xFF, 0)

    if not exact:
        return generate_fuzz_test(n_events - len(prefix), weights, gen, prefix, seed)
    # the Fuzzer appends a final WaitQuiescence only when the last event is not one, so a given seed
    # may not hit the requested length exactly: retry with a derived seed (deterministic)
    for attempt in range(64):
        for k in (n_events - len(prefix), n_events - len(prefix) - 1):
            counter[0] = 0
            tr = generate_fuzz_test(k, weights, gen, prefix, seed + attempt * 0x9E3779B9)
            if len(tr) == n_events:
                return tr
    raise ValueError("cannot build a trace of exactly %d events" % n_events)
