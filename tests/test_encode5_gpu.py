"""GPU parity of encode5_kernel / encode6_kernel (csrc/encode5.hip): score values by rank — the hottest in the
block's LDS, the others read from L2 by the relaxing lane (COLD builds) —, child masks in the trie records, the
long-sample kernel; all bit-exact against the CPU oracle (reference src/model.rs:59-129)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth

from util import assert_same_encoding, corpus_and_vocab


def _distinct_scores(scores, rng):
    """Every token its own score (what an M-step leaves behind, src/prune.rs:143-151), same ranking."""
    return np.asarray(scores, np.float64) - rng.random(len(scores)) * 1e-3


@pytest.mark.parametrize("hot", ["6600", "2000", "300", "0"])
def test_values_outside_the_lds_copy_are_read_by_the_relaxing_lane(monkeypatch, hot):
    """8 000 tokens with 8 000 distinct scores and an LDS copy of 6 600 / 2 000 / 300 / 0 values: the relaxing lane
    reads the others from the table in HBM / L2; nothing is redone, no other encode kernel runs."""
    rng = np.random.default_rng(7)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16, seed_offset=11, max_len=30000)
    scores = _distinct_scores(scores, rng)
    monkeypatch.setenv("TGX_E5_HOT", hot)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode5_kernel" in kt and "encode4_kernel" not in kt and nat.last_encode_redo_samples() == 0
    assert nat.score_values() == 8000 and nat.last_encode_hot_values() == int(hot)
    assert_same_encoding(nat, ora, flat, offs, dropout=0.25, seed=3)
    for ppl in ("1", "2", "3"):
        monkeypatch.setenv("TGX_PPL", ppl)
        assert_same_encoding(nat, ora, flat, offs)


def test_every_match_cold_and_sixteen_matches_per_position(monkeypatch):
    """No value in the LDS copy and a text in which every position has sixteen matches: every one of the 1 024
    entries of a group of 16 positions is read from L2 by the relaxing lane."""
    monkeypatch.setenv("TGX_E5_HOT", "0")
    toks = [bytes([c]) for c in range(256)] + [b"a" * k for k in range(2, 17)] + [b"ab", b"ba", b"abab"]
    rng = np.random.default_rng(41)
    scores = -(rng.random(len(toks)) * 6.0 + 1.0)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    texts = [b"a" * 5000, b"ab" * 700 + b"a" * 900, b"a" * 15, b"a" * 16, b"a" * 17, b"x" + b"a" * 300 + b"y", b""]
    flat, offs = synth.make_corpus(1 << 20, "mixed", seed_offset=77, max_len=20000)
    f, o = tgx.pack(texts + [flat[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(40)])
    assert_same_encoding(nat, ora, f, o)
    assert "encode5_kernel" in nat.last_kernel_times() and nat.last_encode_hot_values() == 0
    assert_same_encoding(nat, ora, f, o, dropout=0.3, seed=5)
    for ppl in ("1", "2", "3"):
        monkeypatch.setenv("TGX_PPL", ppl)
        assert_same_encoding(nat, ora, f, o)


def test_which_kernel_serves_which_vocabulary(monkeypatch):
    """generate-style vocabularies (scores are logs of integer counts: a few thousand distinct values) AND
    vocabularies in which every token has its own score (after an M-step or merge: src/prune.rs:143-151,
    src/merge.rs:102 — any trained vocabulary) run encode5_kernel: the first with all values in LDS, the second
    with the hottest in LDS and the rest read from L2.  More than 65 535 distinct values (a match index is a 16-bit
    rank) leave encode4_kernel."""
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 32000, 16, seed_offset=13)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode5_kernel" in kt and "encode4_kernel" not in kt
    assert 0 < nat.score_values() == nat.last_encode_hot_values() < 16000   # every value in LDS
    rng = np.random.default_rng(9)
    sc2 = _distinct_scores(scores, rng)
    nat2, ora2 = tgx.NativeModel(toks, sc2), orc.OracleModel(toks, sc2)
    assert_same_encoding(nat2, ora2, flat, offs)
    kt = nat2.last_kernel_times()
    assert "encode5_kernel" in kt and "encode4_kernel" not in kt and nat2.last_encode_redo_samples() == 0
    assert nat2.score_values() == len(set(sc2.tolist())) > nat2.last_encode_hot_values() >= 3000   # (the 2 MiB batch has rows for 16 waves)
    # TGX_PATH=rows4 still forces round 1's kernel (A/B timing): same ids
    monkeypatch.setenv("TGX_PATH", "rows4")
    assert_same_encoding(nat2, ora2, flat, offs)
    assert "encode4_kernel" in nat2.last_kernel_times() and "encode5_kernel" not in nat2.last_kernel_times()
    monkeypatch.delenv("TGX_PATH")
    # 70 000 tokens with 70 000 distinct scores: no 8-byte records
    flat3, offs3 = synth.make_corpus(6 << 20, "mixed", seed_offset=14)
    toks3, scores3 = synth.build_vocab(flat3, 70000, 16)
    scores3 = -rng.random(len(toks3)) * 12.0 - 1.0
    nat3, ora3 = tgx.NativeModel(toks3, scores3), orc.OracleModel(toks3, scores3)
    sub_f, sub_o = flat3[: int(offs3[150])], offs3[:151]
    assert_same_encoding(nat3, ora3, sub_f, sub_o)
    assert "encode4_kernel" in nat3.last_kernel_times() and "encode5_kernel" not in nat3.last_kernel_times()
    assert nat3.score_values() == 0


def test_exactly_65535_distinct_values_still_take_the_rank_kernel():
    """Ranks are 16 bits with 0 for "no token": 65 535 values is the last vocabulary the rows5 kernels serve."""
    rng = np.random.default_rng(31)
    flat, offs = synth.make_corpus(6 << 20, "mixed", seed_offset=15)
    toks, _ = synth.build_vocab(flat, 65535, 16)
    assert len(toks) == 65535
    scores = -(np.arange(65535, dtype=np.float64) * 1e-4 + 1.0)
    rng.shuffle(scores)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    sub_f, sub_o = flat[: int(offs[200])], offs[:201]
    assert_same_encoding(nat, ora, sub_f, sub_o)
    assert "encode5_kernel" in nat.last_kernel_times() and nat.score_values() == 65535


def test_edge_bytes_and_block_boundaries(monkeypatch):
    """Bytes 0xFE / 0xFF (the label check's reserved base values), tokens made of them, samples around the
    multiples of 16 and 64, unreachable ends, empty samples."""
    toks = [bytes([c]) for c in range(255)] + [b"\xff\xfe", b"\xfe\xfe\xfe", b"\xfe" * 16, b"ab", b"abc" * 5, b"\x00\x00"]
    scores = np.array([-6.0] * 255 + [-2.0, -3.0, -20.0, -7.0, -9.5, -1.0])
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    rng = np.random.default_rng(10)
    texts = [bytes(rng.integers(0, 255, size=n, dtype=np.uint8)) for n in (1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1000, 5000)]
    texts += [b"", b"\xfe" * 100, b"\xff\xfe" * 33, b"\xfe\xff" * 20 + b"\xfe", b"ab" * 40, b"abc" * 21, b"\x00" * 77]
    f, o = tgx.pack(texts)
    assert_same_encoding(nat, ora, f, o)
    # 0xFF alone is no token: the sample fails with the reference's message, the lowest failing sample first
    bad, ob = tgx.pack([b"ab", b"\xfe\xff\xff", b"abc", b"\xff"])
    with pytest.raises(tgx.TokenGeeXError) as e:
        nat.encode_batch_flat(bad, ob)
    assert str(e.value) == "no path to position 3/3" and e.value.sample == 1


@pytest.mark.parametrize("threshold", ["1", "300", "5000"])
def test_long_sample_kernel_walkers_and_relaxer(monkeypatch, threshold):
    """encode6_kernel (a block per long sample: seven walker waves fill a ring of match-index buffers ahead of
    one relaxing wave) with the length threshold forced low, so that every / most samples take it: sample
    lengths around the multiples of 16 and 64, empty samples, unreachable ends, dropout, more samples than
    blocks, a 1 MiB sample — bit-exact against the oracle; the rest of the batch runs encode5_kernel."""
    monkeypatch.setenv("TGX_LONG_THRESHOLD", threshold)
    flat, offs, toks, scores = corpus_and_vocab(3 << 20, "mixed", 4000, 16, seed_offset=51, max_len=40000)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode6_kernel" in kt and nat.last_encode_long_samples() > 0
    assert_same_encoding(nat, ora, flat, offs, dropout=0.3, seed=7)
    rng = np.random.default_rng(5)
    raw = flat.tobytes()
    texts = [raw[:k] for k in (0, 1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 4096, 70000)] + [b"", b"", raw[5000:5000 + (1 << 20)]]
    texts += [raw[int(a):int(a) + int(k)] for a, k in zip(rng.integers(0, 1 << 20, 300), rng.integers(1, 3000, 300))]
    f2, o2 = tgx.pack(texts)
    assert_same_encoding(nat, ora, f2, o2)
    # unreachable ends (no single-byte cover) are reported as by the other kernels
    sparse = tgx.NativeModel([b"a", b"b", b"ab"], [-1.0, -1.0, -1.5])
    f3, o3 = tgx.pack([b"ab" * 5000, b"ab" * 3000 + b"c" + b"ab" * 100, b"ba" * 2000])
    with pytest.raises(tgx.TokenGeeXError) as e:
        sparse.encode_batch_flat(f3, o3)
    assert e.value.sample == 1 and str(e.value) == "no path to position 6201/6201"


@pytest.mark.parametrize("hot,pool", [("2000", None), ("300", None), ("0", None), ("2000", "8"), ("300", "0"), ("0", "192"), ("4000", "16")])
def test_long_sample_kernel_with_values_outside_its_lds_copy(monkeypatch, hot, pool):
    """encode6_kernel with score values outside the LDS copy: the walkers fetch them into the pool entries of their
    ring slot (round 3: the relaxer finds them in LDS), and what a full pool leaves behind the relaxer reads from L2
    a quarter group ahead of the steps that use them — small copies and small pools (none at all: round 2's path)
    force the second, the default pool the first; nothing is redone."""
    rng = np.random.default_rng(21)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16, seed_offset=54, max_len=30000)
    scores = _distinct_scores(scores, rng)
    monkeypatch.setenv("TGX_E5_HOT", hot)
    if pool is not None:
        monkeypatch.setenv("TGX_E6_POOL", pool)
    monkeypatch.setenv("TGX_LONG_THRESHOLD", "2000")
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode6_kernel" in kt and "encode5_kernel" in kt and "encode4_kernel" not in kt and nat.last_encode_long_samples() > 0
    assert nat.last_encode_redo_samples() == 0
    assert_same_encoding(nat, ora, flat, offs, dropout=0.3, seed=11)


def test_vocabulary_with_distinct_scores_by_default(monkeypatch):
    """After an M-step every token has its own score (src/prune.rs:143-151): by default such a model runs
    encode5_kernel with the LDS copy that fits, and its long samples go to encode6_kernel where the estimate says
    so (a 16 MiB batch of samples <= 64 KiB is bound by the serial chains of its longest samples)."""
    rng = np.random.default_rng(22)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 12000, 16, seed_offset=55)
    nat, ora = tgx.NativeModel(toks, _distinct_scores(scores, rng)), orc.OracleModel(toks, _distinct_scores(scores, np.random.default_rng(22)))
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode5_kernel" in kt and "encode4_kernel" not in kt and nat.last_encode_redo_samples() == 0
    assert nat.last_encode_hot_values() <= nat.score_values() == 12000  # (a batch this small gets few waves per block: every value fits)
    f2, o2 = synth.make_corpus(16 << 20, "mixed", seed_offset=56)
    assert_same_encoding(nat, ora, f2, o2)
    kt = nat.last_kernel_times()
    assert "encode6_kernel" in kt and "encode5_kernel" in kt and 0 < nat.last_encode_long_samples() < o2.size - 1
    assert_same_encoding(nat, ora, f2, o2, dropout=0.2, seed=4)


def test_long_sample_threshold_default():
    """By default the long samples of a batch get a block of their own only where the estimate says the pass gets
    shorter: the long ones of a 16 MiB batch of samples <= 64 KiB, none of a 64 MiB batch of samples <= 4 KiB."""
    flat, offs, toks, scores = corpus_and_vocab(16 << 20, "mixed", 4000, 16, seed_offset=52)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    assert 0 < nat.last_encode_long_samples() < offs.size - 1 and "encode6_kernel" in nat.last_kernel_times()
    f2, o2 = synth.make_corpus(64 << 20, "mixed", max_len=4096, seed_offset=53)
    res = nat.encode_batch_flat(f2, o2)
    res.free()
    assert nat.last_encode_long_samples() == 0 and "encode6_kernel" not in nat.last_kernel_times()


@pytest.mark.parametrize("cus", ["32", "96", "192"])
def test_long_and_short_samples_at_once(monkeypatch, cus):
    """Co-run (round 3): the long-sample kernel on CUs of its own (second stream, a work queue of its own) while
    encode5_kernel takes the rest of the batch on the others — forced here with a low threshold; every value in LDS and
    distinct scores (cold values through the walkers' pools), dropout; bit-exact against the oracle."""
    monkeypatch.setenv("TGX_LONG_THRESHOLD", "3000")
    monkeypatch.setenv("TGX_CORUN", cus)
    rng = np.random.default_rng(23)
    flat, offs, toks, scores = corpus_and_vocab(4 << 20, "mixed", 8000, 16, seed_offset=71, max_len=40000)
    for sc in (scores, _distinct_scores(scores, rng)):
        nat, ora = tgx.NativeModel(toks, sc), orc.OracleModel(toks, sc)
        assert_same_encoding(nat, ora, flat, offs)
        kt = nat.last_kernel_times()
        assert nat.last_encode_corun_cus() == int(cus) and "encode6_kernel" in kt and "encode5_kernel" in kt
        assert 0 < nat.last_encode_long_samples() < offs.size - 1
        assert_same_encoding(nat, ora, flat, offs, dropout=0.2, seed=9)
    monkeypatch.setenv("TGX_CORUN", "0")
    assert_same_encoding(nat, ora, flat, offs)
    assert nat.last_encode_corun_cus() == 0


def test_values_reranked_by_match_counts_after_an_m_step(monkeypatch):
    """A vocabulary as an M-step leaves it (src/prune.rs:124-170: one value per token, kept single-byte tokens with tiny
    scores that still match everywhere): encode5_kernel's values are re-ranked by how often a sample of the first corpus
    reads them (tgx_api.cpp ensure_value_ranks).  Ids must not depend on the order: bit-exact against the oracle with the
    re-ranked values, with build_trie8's order (TGX_VALUE_RANK=model), with dropout, on a second corpus, and after an
    E-step on the chained kernels rewrote the tables."""
    from tokengeex_amd import _lib
    flat, offs = synth.make_corpus(6 << 20, "mixed", seed_offset=71)
    toks, scores = synth.build_vocab(flat[: 2 << 20], 30000, 16)
    base = tgx.NativeModel(toks, scores, for_estep=True)
    exp, _ = base.estep(tgx.NativeCorpus(flat, offs))
    keep = np.array([1 if len(t) == 1 else 0 for t in toks], np.uint8)
    idx, sc2 = _lib.prune_m_step(exp, keep)
    idx, sc2 = np.asarray(idx, np.int64), np.asarray(sc2, np.float64)
    toks2 = [toks[i] for i in idx]
    assert len(set(sc2.tolist())) > 9000  # more values than LDS holds: the COLD build
    ora = orc.OracleModel(toks2, sc2)
    monkeypatch.setenv("TGX_E5_HOT", "2000")  # (a batch this small would get a table that holds every value)
    nat = tgx.NativeModel(toks2, sc2)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode5_kernel" in kt and 0 < nat.last_encode_hot_values() < nat.score_values()
    assert_same_encoding(nat, ora, flat, offs, dropout=0.1, seed=3)
    f2, o2 = synth.make_corpus(1 << 20, "mixed", seed_offset=72, max_len=3000)
    assert_same_encoding(nat, ora, f2, o2)
    monkeypatch.setenv("TGX_ESTEP", "chain")  # the chained E-step kernels rewrite the 8-byte records and the value tables
    got, _ = nat.estep(tgx.NativeCorpus(f2, o2))
    monkeypatch.delenv("TGX_ESTEP")
    assert_same_encoding(nat, ora, f2, o2)
    monkeypatch.setenv("TGX_VALUE_RANK", "model")
    plain = tgx.NativeModel(toks2, sc2)
    assert_same_encoding(plain, ora, flat, offs)
