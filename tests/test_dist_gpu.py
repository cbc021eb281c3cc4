"""Two ranks (one process each, both on cuda:0 of the one-GPU box, host-side gloo exchange) running the
sharded `prune` and `merge` drivers: the only cross-rank traffic is one vector / one pair table per pass
(SURVEY.md §8e), and every rank must arrive at the vocabulary of the single-process run."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    from util import corpus_and_vocab
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 3000, 12, max_len=16384)
    vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
    return flat, offs, vocab


def _worker(rank, world, port, out_dir, tests_dir):
    import sys
    sys.path.insert(0, tests_dir)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from tokengeex_amd import dist as tdist
    from tokengeex_amd.merge import ModelVocabularyMerger
    from tokengeex_amd.prune import ModelVocabularyPruner
    from test_merge_cpu import ALLOW
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, offs, vocab = _inputs()
        lo, hi = tdist.shard_bounds(offs, world)[rank]
        sflat, soffs = tdist.take_shard(flat, offs, lo, hi)
        pruned = ModelVocabularyPruner(1500, 0.75, 2, 0.0, dist=dist).prune(vocab, sflat, soffs)
        merged = ModelVocabularyMerger(ALLOW, 60, 25, 0.9, 16, dist=dist).merge(vocab, sflat, soffs)
        with open(os.path.join(out_dir, f"r{rank}.pkl"), "wb") as f:
            pickle.dump((pruned, merged), f)
    finally:
        dist.destroy_process_group()


def test_two_rank_prune_and_merge_match_single_process(tmp_path):
    from tokengeex_amd.merge import ModelVocabularyMerger
    from tokengeex_amd.prune import ModelVocabularyPruner
    from test_merge_cpu import ALLOW
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), os.path.dirname(os.path.abspath(__file__))), nprocs=world, join=True)
    res = [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(world)]
    assert res[0] == res[1]                                   # every rank derives the same vocabularies, bit for bit
    flat, offs, vocab = _inputs()
    want_merge = ModelVocabularyMerger(ALLOW, 60, 25, 0.9, 16).merge(vocab, flat, offs)
    assert res[0][1] == want_merge                            # integer pair counts: exact
    want_prune = ModelVocabularyPruner(1500, 0.75, 2, 0.0).prune(vocab, flat, offs)
    got = {t[0]: t for t in res[0][0]}
    assert sorted(got) == sorted(t[0] for t in want_prune)    # f64 expected counts: summed per shard, so to rounding
    for t in want_prune:
        assert abs(got[t[0]][1] - t[1]) <= 1e-7 * abs(t[1]) + 1e-9 and got[t[0]][2] == t[2]
