"""Sharding of the schedule-index space across the GPUs of one node.

K1 executions are independent given per-execution seeds, so rank r of W takes the contiguous index
range shard_range(n, r, W) with the transition table and the trace replicated; there is no
data-path collective.  The only exchange is the found-violation set: one fixed-size all-gather
(RCCL over xGMI when the tensors live on GPUs, gloo in the CPU tests) of each rank's compacted
`demi_violation` list, [count row] + cap entries, so every rank ends with the merged set.
Payloads are tens of KB: latency-bound, link bandwidth is irrelevant here.
"""
from typing import List, Tuple

import numpy as np

from . import types as T


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank's slice of n schedule indices (strong scaling of a fixed range)."""
    return n * rank // world, n * (rank + 1) // world


def pack_violation_list(entries: np.ndarray, cap: int) -> np.ndarray:
    """Host-side equivalent of the device layout: int64[(cap+1), 2], row 0 = [count, 0]."""
    buf = np.zeros((cap + 1, 2), dtype=np.int64)
    k = min(len(entries), cap)
    buf[0, 0] = len(entries)
    if k:
        buf[1:k + 1] = np.ascontiguousarray(entries[:k]).view(np.int64).reshape(k, 2)
    return buf


def merge_violation_sets(parts: List[np.ndarray], cap: int) -> np.ndarray:
    """Merge per-rank lists (device layout) into one VIOLATION_DTYPE array sorted by index."""
    out = []
    for p in parts:
        p = np.ascontiguousarray(p, dtype=np.int64)
        count = int(p[0, 0])
        k = min(count, cap)
        if k:
            out.append(p[1:k + 1].copy().view(T.VIOLATION_DTYPE).reshape(-1))
    if not out:
        return np.zeros(0, dtype=T.VIOLATION_DTYPE)
    all_ = np.concatenate(out)
    return all_[np.argsort(all_["index"], kind="stable")]


def allgather_violation_sets(local: "torch.Tensor", cap: int) -> np.ndarray:
    """all_gather the fixed-size per-rank list and merge (works on nccl/RCCL and gloo)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_violation_sets([local.cpu().numpy()], cap)
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local)
    return merge_violation_sets([p.cpu().numpy() for p in parts], cap)


def sharded_map(items: list, fn) -> list:
    """Evaluate fn(list) -> list of bool over `items`, dealt round-robin to the ranks of the default
    process group, and all-gather the result bits so every rank returns the full list.  Without a
    process group this is fn(items).  Used for DDMin frontiers and DPOR rounds: payloads are a few
    hundred bytes, i.e. latency-bound."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(fn(items))
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = items[rank::world]
    res = list(fn(mine)) if mine else []
    per = (len(items) + world - 1) // world
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = torch.zeros(per, dtype=torch.uint8, device=dev)
    if res:
        buf[:len(res)] = torch.tensor([1 if r else 0 for r in res], dtype=torch.uint8, device=dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    parts = [p.cpu().numpy() for p in parts]
    out = [False] * len(items)
    for r in range(world):
        n_r = len(items[r::world])
        for k in range(n_r):
            out[r + k * world] = bool(parts[r][k])
    return out


def sharded_batch(items: list, fn):
    """fn(list of prefixes) -> (verdicts ndarray, [trace arrays], [pair arrays]); items are dealt
    round-robin to the ranks and the results all-gathered (as one fixed-size byte row per item:
    verdict 16 B | trace_len 4 B | n_pairs 4 B | trace 256 x 16 B | pairs), so every rank keeps an
    identical backtrack queue and explored set while the interleavings themselves are sharded."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return fn(items)
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = items[rank::world]
    v, tr, pr = fn(mine) if mine else (np.zeros(0, dtype=T.VERDICT_DTYPE), [], [])
    max_pairs = max([len(p) for p in pr] + [0])
    mp = torch.tensor([max_pairs], dtype=torch.int64)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mp = mp.to(dev)
    dist.all_reduce(mp, op=dist.ReduceOp.MAX)
    max_pairs = int(mp.item())
    row = 24 + T.DPOR_MAX_TRACE * 16 + max_pairs * 4
    per = (len(items) + world - 1) // world
    buf = np.zeros((per, row), dtype=np.uint8)
    for k in range(len(mine)):
        buf[k, :16] = np.frombuffer(v[k:k + 1].tobytes(), dtype=np.uint8)
        buf[k, 16:24] = np.frombuffer(np.array([len(tr[k]), len(pr[k])], dtype=np.uint32).tobytes(), dtype=np.uint8)
        buf[k, 24:24 + 16 * len(tr[k])] = np.frombuffer(tr[k].tobytes(), dtype=np.uint8)
        o = 24 + T.DPOR_MAX_TRACE * 16
        buf[k, o:o + 4 * len(pr[k])] = np.frombuffer(pr[k].tobytes(), dtype=np.uint8)
    tb = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(tb) for _ in range(world)]
    dist.all_gather(parts, tb)
    parts = [p.cpu().numpy() for p in parts]
    verdicts = np.zeros(len(items), dtype=T.VERDICT_DTYPE)
    traces, pairs = [None] * len(items), [None] * len(items)
    for r in range(world):
        for k in range(len(items[r::world])):
            i = r + k * world
            rowb = parts[r][k]
            verdicts[i] = np.frombuffer(rowb[:16].tobytes(), dtype=T.VERDICT_DTYPE)[0]
            tl, npr = np.frombuffer(rowb[16:24].tobytes(), dtype=np.uint32)
            traces[i] = np.frombuffer(rowb[24:24 + 16 * int(tl)].tobytes(), dtype=T.DPOR_TRACE_DTYPE).copy()
            o = 24 + T.DPOR_MAX_TRACE * 16
            pairs[i] = np.frombuffer(rowb[o:o + 4 * int(npr)].tobytes(), dtype=T.DPOR_PAIR_DTYPE).copy()
    return verdicts, traces, pairs
