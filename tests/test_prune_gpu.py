"""End-to-end `prune` (src/prune.rs:23-57) on the GPU path against the same loop driven through the
oracle: E-step + frequency pass on the device, M-step / alternatives / selection on the host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
from tokengeex_amd.prune import ModelVocabularyPruner

from util import corpus_and_vocab


def oracle_prune(vocab, flat, offs, vocab_size, shrink_factor, em_subiters, dropout, seed=0):
    """The reference loop over oracle functions only (checker)."""
    S = offs.shape[0] - 1
    while len(vocab) > vocab_size:
        for _ in range(em_subiters):
            m = orc.OracleModel([t[0] for t in vocab], [t[1] for t in vocab])
            seed += 1
            st, expected, _, _ = m.estep_flat(flat, offs, 81920, dropout, seed, threads=8)
            assert st == orc.OK
            st, idx, sc = orc.m_step(expected, np.array([1 if t[2] else 0 for t in vocab], np.uint8))
            assert st == 0
            vocab = [(vocab[int(i)][0], float(s), vocab[int(i)][2]) for i, s in zip(idx, sc)]
        m = orc.OracleModel([t[0] for t in vocab], [t[1] for t in vocab])
        pruned_size = max(int(len(vocab) * shrink_factor), vocab_size)
        ak, ao, ai = orc.prune_alternatives(m)
        freq = m.count_tokens_flat(flat, offs, threads=8)
        st, out = orc.prune_select(freq, np.array([1 if t[2] else 0 for t in vocab], np.uint8), ak, ao, ai,
                                   m.scores, S, pruned_size)
        assert st == 0
        vocab = [vocab[int(i)] for i in out]
    return vocab


@pytest.mark.parametrize("dropout", [0.0, 0.05])
def test_prune_end_to_end_matches_oracle_loop(dropout):
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 4000, 16, max_len=16384)
    vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
    pruner = ModelVocabularyPruner(800, 0.75, 2, dropout)
    got = pruner.prune(vocab, flat, offs)
    want = oracle_prune(vocab, flat, offs, 800, 0.75, 2, dropout)
    assert len(got) == 800
    # Same tokens, same scores to E-step tolerance, same keep flags.  The ORDER of tokens whose expected
    # counts are equal up to f64 summation noise is not comparable (the reference's own rayon reduction
    # order, src/prune.rs:104-113, makes it run-dependent too), so compare as token -> (score, keep) maps
    # and check each result is ordered by score.
    gd, wd = {t[0]: t for t in got}, {t[0]: t for t in want}
    assert len(gd) == len(got) and sorted(gd) == sorted(wd)
    for k in gd:
        assert gd[k][2] == wd[k][2]
        assert abs(gd[k][1] - wd[k][1]) <= 1e-7 * abs(wd[k][1]) + 1e-9, k
    assert np.all(np.diff([t[1] for t in got]) <= 0)
    # positions may differ only between tokens whose scores tie to the E-step tolerance: wherever the two orders
    # disagree, the token the oracle has there and the token the GPU path has there have (oracle) scores within 1e-7
    ws = {t[0]: t[1] for t in want}
    for a, b in zip(got, want):
        if a[0] != b[0]:
            assert abs(ws[a[0]] - b[1]) <= 1e-7 * abs(b[1]) + 1e-9, (a, b)
    assert len(pruner.timings) >= 2 and all(r["to"] < r["from"] for r in pruner.timings)


def test_model_derived_from_its_parent_equals_a_model_built_from_scratch():
    """tgx_model_create_derived (round 3): a model for a subset of another model's vocabulary with new scores, on the
    other's double-arrays (tokens that are gone lose their terminal marks, nothing is rebuilt).  Encode ids, token
    frequencies, expected counts, log Z and the 2-best alternatives equal those of a model built from the subset; a
    second derivation from the derived model too; a parent with duplicate tokens is refused."""
    import tokengeex_amd as tgx
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 6000, 16, seed_offset=91, max_len=20000)
    scores = np.asarray(scores, np.float64)
    rng = np.random.default_rng(17)
    parent = tgx.NativeModel(toks, scores, for_estep=True)
    corpus = tgx.NativeCorpus(flat, offs)

    def subset(n_from, frac):
        keep = np.sort(np.concatenate([np.arange(256), 256 + rng.choice(n_from - 256, int((n_from - 256) * frac), replace=False)])).astype(np.uint32)
        return keep

    keep1 = subset(len(toks), 0.6)  # (the single bytes stay: every text remains coverable)
    toks1 = [toks[i] for i in keep1]
    sc1 = scores[keep1] - rng.random(keep1.size) * 0.2
    for for_estep in (True, False):
        d = parent.derive(keep1, sc1, for_estep=for_estep)
        f = tgx.NativeModel(toks1, sc1, for_estep=for_estep)
        assert d.vocab_size == f.vocab_size == keep1.size
        rd, rf = d.encode_corpus(corpus), f.encode_corpus(corpus)
        assert np.array_equal(rd.ids(), rf.ids()) and np.array_equal(rd.offsets(), rf.offsets())
        rd.free(); rf.free()
        assert np.array_equal(d.count_tokens(corpus), f.count_tokens(corpus))
        ed, zd = d.estep(corpus, 81920, 0.0, 1)
        ef, zf = f.estep(corpus, 81920, 0.0, 1)
        np.testing.assert_allclose(ed, ef, rtol=1e-11, atol=1e-13)
        assert abs(zd - zf) <= 1e-13 * abs(zf)
        ed, _ = d.estep(corpus, 81920, 0.1, 7)
        ef, _ = f.estep(corpus, 81920, 0.1, 7)
        np.testing.assert_allclose(ed, ef, rtol=1e-11, atol=1e-13)
        ad, af = d.prune_alternatives(), f.prune_alternatives()
        assert all(np.array_equal(x, y) for x, y in zip(ad, af))
        # a subset of the subset, from the derived model
        keep2 = subset(keep1.size, 0.5)
        sc2 = sc1[keep2] - 0.1
        d2 = d.derive(keep2, sc2)
        f2 = tgx.NativeModel([toks1[i] for i in keep2], sc2)
        rd, rf = d2.encode_corpus(corpus), f2.encode_corpus(corpus)
        assert np.array_equal(rd.ids(), rf.ids())
        rd.free(); rf.free()
        for mm in (d, f, d2, f2):
            mm.free()
    dup = tgx.NativeModel(toks + [toks[300]], np.concatenate([scores, [-3.0]]))
    with pytest.raises(tgx.TokenGeeXError):
        dup.derive(np.arange(100, dtype=np.uint32), scores[:100])
    with pytest.raises(tgx.TokenGeeXError):
        parent.derive(np.array([5, 3], np.uint32), scores[:2])  # not ascending
