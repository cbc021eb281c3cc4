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
    of equal keys -> the same sorted table on every rank.  The WHOLE tables cross the link: ~5 M pairs per
    256 MiB shard, 80 MB per rank and round, and a host-side unique over world x that — fine for tests and
    small corpora; the merge driver uses top_pairs_exchange below, which moves ~1 MB per round."""
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


def top_pairs_exchange(keys: np.ndarray, counts: np.ndarray, k: int, dist=None, device: str = "cpu"):
    """The head of the GLOBAL pair table without exchanging the per-rank tables (reference src/merge.rs:66-84: the
    merged map is sorted by descending count and only its first few hundred entries are ever looked at).

    Every rank contributes its k most frequent local pairs and the count of the first pair it leaves out (its
    cut-off); the union of the contributed keys is the candidate set, for which every rank then reports its EXACT
    local count (a binary search in its own sorted table), summed over ranks.  A pair outside the candidate set is
    below the cut-off on every rank, so its global count is at most the sum of the cut-offs.

    -> (cand_keys ascending, exact global counts, bound, bytes_exchanged): candidates whose global count exceeds
    `bound` are provably the head of the global table, in the same order on every rank; the caller widens k when its
    selection would have to go below that (ModelVocabularyMerger.merge).  Two collectives per call: an all-gather of
    world x (k keys + k counts + 1 cut-off) and an all-reduce of one count per candidate."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return keys.copy(), counts.copy(), 0, 0
    import torch
    world = dist.get_world_size()
    n = int(keys.size)
    n_max = torch.tensor([n], dtype=torch.int64, device=device)
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
    k = int(max(1, min(int(k), int(n_max.item()))))  # (the gathered buffers have one size on every rank)
    if n > k:
        head = np.argpartition(counts.astype(np.int64), n - k)[n - k:]   # the k largest local counts (ties: any)
        cut = int(np.partition(counts.astype(np.int64), n - k - 1)[n - k - 1])  # the largest count left out
    else:
        head, cut = np.arange(n), 0
    buf = np.zeros(2 * k + 2, np.int64)
    buf[0] = head.size
    buf[1] = cut
    buf[2:2 + head.size] = keys[head].view(np.int64)
    buf[2 + k:2 + k + head.size] = counts[head].view(np.int64)
    t = torch.from_numpy(buf).to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    parts = [p.cpu().numpy() for p in parts]
    bound = int(sum(int(p[1]) for p in parts))
    cand = np.unique(np.concatenate([p[2:2 + int(p[0])].view(np.uint64) for p in parts]))
    # exact local counts of every candidate (0 where this rank has not seen the pair); keys are sorted ascending
    idx = np.searchsorted(keys, cand)
    idx_c = np.minimum(idx, max(n - 1, 0))
    local = np.where((idx < n) & (keys[idx_c] == cand), counts[idx_c], 0).astype(np.int64) if n else np.zeros(cand.size, np.int64)
    tot = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    exchanged = world * buf.nbytes + cand.size * 8
    return cand, tot.cpu().numpy().view(np.uint64).copy(), bound, exchanged
