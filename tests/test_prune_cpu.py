"""Host half of `prune` (src/prune.rs): C-ABI functions vs the oracle restatement, plus
independent pins (scipy digamma, brute-force enumeration of segmentations, hand-worked cases).
No GPU needed: these ABI entry points are host-only."""
import itertools

import numpy as np
import pytest
from scipy.special import digamma as sp_digamma

from oracle import oracle as orc
from tokengeex_amd import _lib, synth


def test_digamma_matches_scipy_and_oracle():
    xs = [0.5, 0.75, 1.0, 2.5, 6.999, 7.0, 10.0, 100.0, 1e3, 1e4, 1e5, 111111.0, 3.3e9]
    for x in xs:
        got = _lib.digamma(x)
        assert got == orc.digamma(x)  # same formula, same operation order
        assert abs(got - sp_digamma(x)) < 2e-9 * max(1.0, abs(sp_digamma(x)))


def test_m_step_hand_case():
    # src/prune.rs:124-170: below 0.5 dropped unless keep; kept ones clamp to 0.5
    expected = np.array([10.0, 0.2, 0.2, 0.5, 3.0])
    keep = np.array([0, 0, 1, 0, 0], np.uint8)
    idx, sc = _lib.prune_m_step(expected, keep)
    assert idx.tolist() == [0, 2, 3, 4]
    total = 10.0 + 0.5 + 0.5 + 3.0
    want = [sp_digamma(v) - sp_digamma(total) for v in (10.0, 0.5, 0.5, 3.0)]
    assert np.allclose(sc, want, atol=1e-8)
    st, oidx, osc = orc.m_step(expected, keep)
    assert st == 0 and oidx.tolist() == idx.tolist() and np.array_equal(osc, sc)


def test_m_step_random_bit_exact():
    rng = np.random.default_rng(5)
    expected = rng.gamma(0.3, 50.0, 20000)
    keep = (rng.random(20000) < 0.02).astype(np.uint8)
    idx, sc = _lib.prune_m_step(expected, keep)
    st, oidx, osc = orc.m_step(expected, keep)
    assert st == 0
    assert np.array_equal(idx, oidx) and np.array_equal(sc, osc)
    assert 0 < idx.size < 20000


def test_m_step_invalid_scores():
    # all-zero + keep → digamma(0.5) fine; NaN propagates through max() as 0.5 like f64::max
    idx, sc = _lib.prune_m_step(np.array([np.nan, 1.0]), np.array([0, 0], np.uint8))
    assert idx.tolist() == [0, 1] and np.isfinite(sc).all()
    with pytest.raises(_lib.TokenGeeXError):
        _lib.prune_m_step(np.array([np.inf, 1.0]), np.array([0, 0], np.uint8))  # the reference panics
    assert orc.m_step(np.array([np.inf, 1.0]), np.array([0, 0], np.uint8))[0] == -1


def _segmentations(tok: bytes, index: dict):
    """Every way to write `tok` as vocabulary tokens -> list of id tuples."""
    out = []

    def rec(pos, acc):
        if pos == len(tok):
            out.append(tuple(acc))
            return
        for ln in range(1, len(tok) - pos + 1):
            i = index.get(tok[pos:pos + ln])
            if i is not None:
                rec(pos + ln, acc + [i])
    rec(0, [])
    return out


def test_alternatives_hand_case():
    tokens = [b"a", b"b", b"c", b"ab", b"bc", b"abc", b"zz", b"z"]
    scores = np.array([-2.0, -2.0, -2.0, -2.5, -3.9, -3.0, -9.0, -1.0])
    trie = _lib.FlatTrie(tokens, scores)
    ak, ao, ai = trie.prune_alternatives(tokens, scores)
    alts = [ai[ao[i]:ao[i + 1]].tolist() for i in range(len(tokens))]
    # single bytes: one segmentation only -> always_keep, no alternatives
    assert alts[0] == alts[1] == alts[2] == alts[7] == []
    assert ak[[0, 1, 2, 7]].tolist() == [1, 1, 1, 1]
    assert alts[3] == [0, 1] and ak[3] == 1              # ab -> a b (-4.0)
    assert alts[4] == [1, 2] and ak[4] == 1              # bc -> b c
    assert alts[5] == [3, 2] and ak[5] == 1              # abc: ab c (-4.5) beats a bc (-5.9), a b c (-6)
    assert alts[6] == [] and ak[6] == 0                  # zz (-9) loses to z z (-2): not its own first choice
    oak, oao, oai = orc.prune_alternatives(orc.OracleModel(tokens, scores))
    assert np.array_equal(ak, oak) and np.array_equal(ao, oao) and np.array_equal(ai, oai)


def test_alternatives_vs_bruteforce_and_oracle():
    rng = np.random.default_rng(11)
    alphabet = [bytes([c]) for c in b"abcd"]
    toks = set(alphabet)
    while len(toks) < 300:
        ln = int(rng.integers(2, 9))
        toks.add(bytes(rng.choice(list(b"abcd"), ln).tolist()))
    tokens = sorted(toks)
    scores = -rng.random(len(tokens)) * 6 - 0.5 * np.array([len(t) for t in tokens])  # distinct, no ties
    index = {t: i for i, t in enumerate(tokens)}
    trie = _lib.FlatTrie(tokens, scores)
    ak, ao, ai = trie.prune_alternatives(tokens, scores)
    oak, oao, oai = orc.prune_alternatives(orc.OracleModel(tokens, scores))
    assert np.array_equal(ak, oak) and np.array_equal(ao, oao) and np.array_equal(ai, oai)
    for i, t in enumerate(tokens):
        segs = sorted(_segmentations(t, index), key=lambda s: -sum(scores[j] for j in s))
        got = ai[ao[i]:ao[i + 1]].tolist()
        if len(segs) == 1:
            assert ak[i] == 1 and got == []
        elif len(segs[0]) > 1:
            assert ak[i] == 0 and got == []
        else:
            # permutations of the same tokens tie exactly; the second path must be a real
            # segmentation with the second-best score
            assert ak[i] == 1 and tuple(got) in segs[1:], (t, segs[:3], got)
            assert abs(sum(scores[j] for j in got) - sum(scores[j] for j in segs[1])) < 1e-12


def test_alternatives_synthetic_vocab_matches_oracle():
    flat, _ = synth.make_corpus(1 << 20, seed_offset=3)
    tokens, scores = synth.build_vocab(flat, 6000, 16)
    trie = _lib.FlatTrie(tokens, scores)
    ak, ao, ai = trie.prune_alternatives(tokens, scores)
    oak, oao, oai = orc.prune_alternatives(orc.OracleModel(tokens, scores))
    assert np.array_equal(ak, oak) and np.array_equal(ao, oao) and np.array_equal(ai, oai)
    assert ai.size > 0 and (ak == 0).any()


def test_alternatives_missing_byte():
    # a token containing a byte that is no token by itself: viterbi stops early (src/lattice.rs:131-133),
    # nbest still walks the agenda; product and oracle must agree on the outcome
    tokens = [b"a", b"ab", b"abq", b"b"]
    scores = np.array([-1.0, -1.5, -2.0, -1.0])
    trie = _lib.FlatTrie(tokens, scores)
    ak, ao, ai = trie.prune_alternatives(tokens, scores)
    oak, oao, oai = orc.prune_alternatives(orc.OracleModel(tokens, scores))
    assert np.array_equal(ak, oak) and np.array_equal(ao, oao) and np.array_equal(ai, oai)
    assert ai[ao[2]:ao[3]].tolist() == []  # only one segmentation of "abq"


def _select_inputs(seed, V=500):
    rng = np.random.default_rng(seed)
    freq = rng.integers(0, 1000, V).astype(np.uint64)
    freq[rng.random(V) < 0.2] = 0
    keep = (rng.random(V) < 0.05).astype(np.uint8)
    always_keep = (rng.random(V) < 0.7).astype(np.uint8)
    counts = rng.integers(0, 4, V)
    counts[counts == 1] = 2
    alt_offs = np.zeros(V + 1, np.uint32)
    alt_offs[1:] = np.cumsum(counts)
    alt_ids = rng.integers(0, V, int(alt_offs[-1])).astype(np.uint32)
    scores = -rng.random(V) * 10
    return freq, keep, always_keep, alt_offs, alt_ids, scores


def test_select_matches_oracle_and_python():
    for seed in range(4):
        freq, keep, ak, ao, ai, scores = _select_inputs(seed)
        V, n_samples, size = freq.shape[0], 77, 300
        got = _lib.prune_select(freq, keep, ak, ao, ai, scores, n_samples, size)
        st, want = orc.prune_select(freq, keep, ak, ao, ai, scores, n_samples, size)
        assert st == 0 and np.array_equal(got, want)
        # independent numpy restatement of src/prune.rs:246-318
        total = float(freq.sum())
        pruned, cand = [], []
        for i in range(V):
            alts = ai[ao[i]:ao[i + 1]]
            if keep[i]:
                pruned.append(i)
            elif freq[i] == 0 and not ak[i]:
                pass
            elif alts.size == 0:
                pruned.append(i)
            elif freq[i] != 0:
                f = float(freq[i])
                logprob = np.log(f) - np.log(total)
                alt_logsum = np.log(total + f * (V - 1))
                alt_logprob = sum(np.log(float(freq[a]) + f) - alt_logsum for a in alts)
                cand.append((i, (f / n_samples) * (logprob - alt_logprob)))
        cand.sort(key=lambda c: -c[1])
        for i, _ in cand:
            if len(pruned) == size:
                break
            pruned.append(i)
        assert len(got) == size
        assert sorted(got.tolist()) == sorted(pruned)
        assert np.all(np.diff(scores[got]) <= 0)  # final order: score descending


def test_select_no_cut_when_kept_exceed_target():
    # `if pruned_vocab.len() == pruned_size { break }` never fires once len > pruned_size
    freq, keep, ak, ao, ai, scores = _select_inputs(9, V=200)
    keep[:] = 0
    keep[:60] = 1
    got = _lib.prune_select(freq, keep, ak, ao, ai, scores, 10, 50)
    st, want = orc.prune_select(freq, keep, ak, ao, ai, scores, 10, 50)
    assert st == 0 and np.array_equal(got, want)
    assert len(got) > 60


def test_select_abnormal_loss_is_an_error():
    # `if !loss.is_normal() { panic!(..) }` — src/prune.rs:290-295; zero samples make the loss infinite
    freq = np.array([5, 3], np.uint64)
    z = np.zeros(2, np.uint8)
    args = (freq, z, z + 1, np.array([0, 1, 1], np.uint32), np.array([1], np.uint32), np.array([-1.0, -2.0]))
    with pytest.raises(_lib.TokenGeeXError):
        _lib.prune_select(*args, 0, 1)
    assert orc.prune_select(*args, 0, 1)[0] == -1
    assert _lib.prune_select(*args, 4, 1).tolist() == [1]   # id 1 has no alternatives: kept first; target reached
