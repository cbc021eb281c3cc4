"""world_size-2 gloo test of the multi-rank glue (sharding, rank-ordered count
reduction, bench timing aggregation) on CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from tokengeex_amd import dist as tdist
from tokengeex_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, offs = synth.make_corpus(512 << 10, "mixed")
        toks, scores = synth.build_vocab(flat, 1500, 12)
        lo, hi = tdist.shard_bounds(offs, world)[rank]
        sflat, soffs = tdist.take_shard(flat, offs, lo, hi)
        model = orc.OracleModel(toks, scores)  # stands in for the per-GPU pass in this CPU-only test
        freq = model.count_tokens_flat(sflat, soffs)
        st, expected, z, _ = model.estep_flat(sflat, soffs)
        tot_freq = tdist.allreduce_vector(freq, dist)
        tot_exp = tdist.allreduce_vector(expected, dist)
        elapsed, nb, nt = tdist.aggregate_timing(1.0 + rank, int(sflat.size), int(freq.sum()), dist)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), freq=tot_freq, exp=tot_exp, t=elapsed, nb=nb, nt=nt,
                 lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_reduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    flat, offs = synth.make_corpus(512 << 10, "mixed")
    toks, scores = synth.build_vocab(flat, 1500, 12)
    model = orc.OracleModel(toks, scores)
    # shards are contiguous, disjoint, cover everything, byte-balanced
    assert (int(r0["lo"]), int(r1["hi"])) == (0, offs.size - 1) and int(r0["hi"]) == int(r1["lo"])
    b0 = int(offs[int(r0["hi"])])
    assert abs(b0 - flat.size / 2) < 70000
    # integer counts: identical on both ranks and equal to the single-process pass
    want = model.count_tokens_flat(flat, offs)
    np.testing.assert_array_equal(r0["freq"], want)
    np.testing.assert_array_equal(r1["freq"], want)
    # f64 counts: rank-ordered sum -> bitwise identical on both ranks, equal to one pass up to rounding
    np.testing.assert_array_equal(r0["exp"], r1["exp"])
    _, exp1, _, _ = model.estep_flat(flat, offs)
    np.testing.assert_allclose(r0["exp"], exp1, rtol=1e-12, atol=1e-12)
    # bench aggregation: max of times, sums of units
    assert float(r0["t"]) == 2.0 and float(r1["t"]) == 2.0
    assert float(r0["nb"]) == float(flat.size) and float(r0["nt"]) == float(want.sum())


def test_shard_bounds_edge_cases():
    offs = np.array([0, 10, 10, 500, 1000], dtype=np.uint64)
    assert tdist.shard_bounds(offs, 1) == [(0, 4)]
    b = tdist.shard_bounds(offs, 3)
    assert b[0][0] == 0 and b[-1][1] == 4 and all(b[i][1] == b[i + 1][0] for i in range(2))
    b = tdist.shard_bounds(np.array([0, 5], dtype=np.uint64), 4)  # more ranks than samples
    assert sum(hi - lo for lo, hi in b) == 1


def _pairs_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, offs = synth.make_corpus(256 << 10, "mixed")
        toks, scores = synth.build_vocab(flat, 800, 12)
        lo, hi = tdist.shard_bounds(offs, world)[rank]
        sflat, soffs = tdist.take_shard(flat, offs, lo, hi)
        keys, counts = orc.OracleModel(toks, scores).count_pairs_flat(sflat, soffs)  # stands in for the GPU scan
        k, c = tdist.allreduce_pairs(keys, counts, dist)
        n = tdist.allreduce_scalar(hi - lo, dist)
        np.savez(os.path.join(out_dir, f"p{rank}.npz"), k=k, c=c, n=n)
    finally:
        dist.destroy_process_group()


def test_two_rank_pair_table_merge(tmp_path):
    """dist.allreduce_pairs: the merged (key, count) table is the table of the whole corpus, on every rank
    (pairs never span samples, src/merge.rs:60-63, so sharding by samples loses none)."""
    world, port = 2, _free_port()
    mp.spawn(_pairs_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / "p0.npz"), np.load(tmp_path / "p1.npz")
    flat, offs = synth.make_corpus(256 << 10, "mixed")
    toks, scores = synth.build_vocab(flat, 800, 12)
    wk, wc = orc.OracleModel(toks, scores).count_pairs_flat(flat, offs)
    for p in (p0, p1):
        np.testing.assert_array_equal(p["k"], wk)
        np.testing.assert_array_equal(p["c"], wc)
        assert int(p["n"]) == offs.size - 1


def _top_pairs_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, offs = synth.make_corpus(256 << 10, "mixed")
        toks, scores = synth.build_vocab(flat, 800, 12)
        lo, hi = tdist.shard_bounds(offs, world)[rank]
        sflat, soffs = tdist.take_shard(flat, offs, lo, hi)
        keys, counts = orc.OracleModel(toks, scores).count_pairs_flat(sflat, soffs)  # stands in for the GPU scan
        out = {}
        for k in (64, 1000, 10 ** 7):
            ck, cc, bound, nbytes = tdist.top_pairs_exchange(keys, counts, k, dist)
            out[f"k{k}"], out[f"c{k}"], out[f"b{k}"], out[f"n{k}"] = ck, cc, bound, nbytes
        np.savez(os.path.join(out_dir, f"t{rank}.npz"), local=keys.size, **out)
    finally:
        dist.destroy_process_group()


def test_two_rank_top_pairs_exchange(tmp_path):
    """dist.top_pairs_exchange (what `merge` exchanges per round with several ranks): every candidate's count is its
    exact global count, every pair whose global count exceeds the bound is among the candidates — the head of the
    single-process table, identical on both ranks — and the bytes exchanged are a small fraction of the tables."""
    world, port = 2, _free_port()
    mp.spawn(_top_pairs_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    t0, t1 = np.load(tmp_path / "t0.npz"), np.load(tmp_path / "t1.npz")
    flat, offs = synth.make_corpus(256 << 10, "mixed")
    toks, scores = synth.build_vocab(flat, 800, 12)
    wk, wc = orc.OracleModel(toks, scores).count_pairs_flat(flat, offs)
    truth = dict(zip(wk.tolist(), wc.tolist()))
    for k in (64, 1000, 10 ** 7):
        ck, cc, bound = t0[f"k{k}"], t0[f"c{k}"], int(t0[f"b{k}"])
        np.testing.assert_array_equal(ck, t1[f"k{k}"])
        np.testing.assert_array_equal(cc, t1[f"c{k}"])
        assert all(truth[int(a)] == int(b) for a, b in zip(ck, cc))                     # exact global counts
        head = {int(a) for a, c in truth.items() if c > bound}
        assert head <= set(ck.tolist())                                                  # nothing above the bound is missing
        if k == 64:
            assert 0 < bound and 10 < len(head) < wk.size and int(t0["n64"]) < 16 * int(t0["local"]) // 10
        if k == 10 ** 7:
            assert bound == 0 and ck.size == wk.size                                     # whole tables: the whole result
