"""Multi-GPU glue for the sharded corpus passes.

The path shards by samples with NO data-path collective: each rank (one process per
GPU) encodes / scans its own shard with a replica of the trie.  The only exchange is
one vector per rank and pass — `u64[V]` frequencies (reference src/prune.rs:231-236),
`f64[V]` expected counts (src/prune.rs:104-112) — which the reference merges under a
lock in completion order; here every rank gathers all vectors and sums them in RANK
order, so the result is deterministic and identical on all ranks.  Works with any
torch.distributed backend ("nccl" = RCCL over xGMI on the GPU box, "gloo" on CPU);
at <= 4 MB per rank the exchange is latency-bound and never on the critical path.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(offs: np.ndarray, world: int) -> list[tuple[int, int]]:
    """Contiguous sample ranges balanced by cumulative BYTES (not sample count)."""
    n_samples = int(offs.shape[0] - 1)
    total = int(offs[-1] - offs[0])
    cuts = [0]
    for r in range(1, world):
        target = int(offs[0]) + total * r // world
        k = int(np.searchsorted(offs, target, side="left"))
        cuts.append(min(max(k, cuts[-1]), n_samples))
    cuts.append(n_samples)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def take_shard(flat: np.ndarray, offs: np.ndarray, lo: int, hi: int):
    base = int(offs[lo])
    return flat[base:int(offs[hi])], (offs[lo:hi + 1] - offs[lo]).astype(np.uint64)


def allreduce_vector(vec: np.ndarray, dist=None, device: str = "cpu") -> np.ndarray:
    """Sum of one per-rank vector over all ranks, in rank order (bit-reproducible)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return vec.copy()
    import torch
    if vec.dtype == np.uint64:
        t = torch.from_numpy(vec.view(np.int64).copy()).to(device)
    else:
        t = torch.from_numpy(np.ascontiguousarray(vec)).to(device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    acc = parts[0].clone()
    for p in parts[1:]:
        acc += p
    out = acc.cpu().numpy()
    return out.view(np.uint64) if vec.dtype == np.uint64 else out


def aggregate_timing(elapsed: float, n_bytes: int, n_tokens: int, dist=None, device: str = "cpu"):
    """bench.py contract: MAX of the elapsed time over ranks, SUM of bytes and tokens."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return elapsed, float(n_bytes), float(n_tokens)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(n_bytes), float(n_tokens)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t.item()), float(s[0].item()), float(s[1].item())


def allreduce_scalar(x: int, dist=None, device: str = "cpu") -> int:
    """Sum of one integer per rank."""
    return int(allreduce_vector(np.array([x], np.uint64), dist, device)[0])


def allreduce_pairs(keys: np.ndarray, counts: np.ndarray, dist=None, device: str = "cpu"):
    """Merge of the per-rank (key, count) tables of the pair scan (reference src/merge.rs:66-71 merges its
    per-chunk maps under a lock): every rank gathers all tables (padded to the longest) and adds the counts
    of equal keys -> the same sorted table on every rank.  A few MB per rank, once per merge round."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return keys.copy(), counts.copy()
    import torch
    world = dist.get_world_size()
    n = torch.tensor([keys.size], dtype=torch.int64, device=device)
    sizes = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    width = max(max(sizes), 1)
    buf = np.zeros((2, width), np.int64)
    buf[0, :keys.size] = keys.view(np.int64)
    buf[1, :counts.size] = counts.view(np.int64)
    t = torch.from_numpy(buf).to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    allk = np.concatenate([parts[r][0, :sizes[r]].cpu().numpy().view(np.uint64) for r in range(world)])
    allc = np.concatenate([parts[r][1, :sizes[r]].cpu().numpy().view(np.uint64) for r in range(world)])
    uk, inv = np.unique(allk, return_inverse=True)
    uc = np.zeros(uk.size, np.uint64)
    np.add.at(uc, inv, allc)
    return uk, uc
