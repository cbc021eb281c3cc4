"""GPU parity of encode5_kernel's score-table paths (csrc/encode5.hip): values in the LDS table, cold values
through the per-wave pool, and pool overflow with the samples concerned redone by encode4_kernel — all
bit-exact against the CPU oracle (reference src/model.rs:59-129)."""
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


@pytest.mark.parametrize("max_hot", ["6600", "2000", "300"])
def test_cold_values_go_through_the_pool(monkeypatch, max_hot):
    """8 000 tokens with 8 000 distinct scores and a table of 6 600 / 2 000 / 300 values: the rest are fetched
    from HBM into the wave's pool while the walk goes on."""
    rng = np.random.default_rng(7)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16, seed_offset=11, max_len=30000)
    scores = _distinct_scores(scores, rng)
    monkeypatch.setenv("TGX_E5_MAX_HOT", max_hot)
    monkeypatch.setenv("TGX_PATH", "rows5")
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    assert "encode5_kernel" in nat.last_kernel_times()
    assert_same_encoding(nat, ora, flat, offs, dropout=0.25, seed=3)
    for ppl in ("2", "4"):
        monkeypatch.setenv("TGX_PPL", ppl)
        assert_same_encoding(nat, ora, flat, offs)


def test_pool_overflow_redoes_only_the_samples_concerned(monkeypatch):
    """A table of 8 values: nearly every match is cold, a block of 64 positions needs far more than the 64 pool
    entries of its wave, so the samples are redone by encode4_kernel; ids still bit-exact; a model whose values
    all fit never syncs for a redo list."""
    rng = np.random.default_rng(8)
    flat, offs, toks, scores = corpus_and_vocab(1 << 20, "mixed", 5000, 16, seed_offset=12, max_len=8000)
    scores = _distinct_scores(scores, rng)
    monkeypatch.setenv("TGX_PATH", "rows5")
    monkeypatch.setenv("TGX_E5_MAX_HOT", "8")
    monkeypatch.setenv("TGX_E5_POOL", "64")  # the default geometry gives a wave as many entries as LDS allows
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode5_kernel" in kt and "encode4_kernel" in kt
    assert nat.last_encode_redo_samples() > 0
    assert_same_encoding(nat, ora, flat, offs, dropout=0.4, seed=9)
    monkeypatch.delenv("TGX_E5_MAX_HOT")
    monkeypatch.delenv("TGX_E5_POOL")
    nat2 = tgx.NativeModel(toks, scores)
    assert_same_encoding(nat2, ora, flat, offs)
    assert "encode4_kernel" not in nat2.last_kernel_times() and nat2.last_encode_redo_samples() == 0


def test_default_path_and_coverage_rule(monkeypatch):
    """generate-style vocabularies (integer counts: a few thousand distinct values) run encode5_kernel by
    default; a vocabulary of 60 000 distinct values, of which the table holds a small share, stays on
    encode4_kernel unless forced."""
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 32000, 16, seed_offset=13)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    assert "encode5_kernel" in nat.last_kernel_times()
    rng = np.random.default_rng(9)
    flat2, offs2 = synth.make_corpus(6 << 20, "mixed", seed_offset=14)
    toks2, scores2 = synth.build_vocab(flat2, 60000, 16)
    scores2 = -rng.random(len(toks2)) * 12.0 - 1.0          # flat random scores: the table covers ~11 % of the mass
    nat2, ora2 = tgx.NativeModel(toks2, scores2), orc.OracleModel(toks2, scores2)
    sub_f, sub_o = flat2[: int(offs2[150])], offs2[:151]
    assert_same_encoding(nat2, ora2, sub_f, sub_o)
    assert "encode4_kernel" in nat2.last_kernel_times() and "encode5_kernel" not in nat2.last_kernel_times()
    monkeypatch.setenv("TGX_PATH", "rows5")
    assert_same_encoding(nat2, ora2, sub_f, sub_o)
    assert "encode5_kernel" in nat2.last_kernel_times()


def test_edge_bytes_and_block_boundaries(monkeypatch):
    """Bytes 0xFE / 0xFF (the label check's reserved base values), tokens made of them, samples around the
    multiples of 16 and 64, unreachable ends, empty samples."""
    monkeypatch.setenv("TGX_PATH", "rows5")
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


@pytest.mark.parametrize("max_hot,pool", [("2000", None), ("300", None), ("40", "16")])
def test_long_sample_kernel_with_cold_values(monkeypatch, max_hot, pool):
    """encode6_kernel's walkers with score values outside the table: each ring slot has a pool, filled after the
    walk in one batch of loads; a slot that runs out flags the sample, which the redo pass encodes."""
    rng = np.random.default_rng(21)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16, seed_offset=54, max_len=30000)
    scores = _distinct_scores(scores, rng)
    monkeypatch.setenv("TGX_PATH", "rows5")
    monkeypatch.setenv("TGX_E5_MAX_HOT", max_hot)
    monkeypatch.setenv("TGX_LONG_THRESHOLD", "2000")
    if pool:
        monkeypatch.setenv("TGX_E5_POOL", pool)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode6_kernel" in kt and "encode5_kernel" in kt and nat.last_encode_long_samples() > 0
    if pool:
        assert nat.last_encode_redo_samples() > 0 and "encode4_kernel" in kt
    assert_same_encoding(nat, ora, flat, offs, dropout=0.3, seed=11)


def test_vocabulary_with_distinct_scores_takes_encode4_by_default(monkeypatch):
    """After an M-step every token has its own score (src/prune.rs:143-151): more values than the LDS table holds.
    Such a model runs encode4_kernel (16-byte records, f64 scores in the match buffer) unless forced."""
    rng = np.random.default_rng(22)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 12000, 16, seed_offset=55)
    nat, ora = tgx.NativeModel(toks, _distinct_scores(scores, rng)), orc.OracleModel(toks, _distinct_scores(scores, np.random.default_rng(22)))
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode4_kernel" in kt and "encode5_kernel" not in kt and nat.last_encode_redo_samples() == 0
    # ... and its long samples go to encode6_kernel (the build with cold-value pools) where the estimate says so:
    # a 16 MiB batch of samples <= 64 KiB is bound by the serial chains of its longest samples
    f2, o2 = synth.make_corpus(16 << 20, "mixed", seed_offset=56)
    assert_same_encoding(nat, ora, f2, o2)
    kt = nat.last_kernel_times()
    assert "encode6_kernel" in kt and "encode4_kernel" in kt and 0 < nat.last_encode_long_samples() < o2.size - 1
    monkeypatch.setenv("TGX_E5_POOL", "8")          # pools this small overflow: those samples come back to encode4
    monkeypatch.setenv("TGX_LONG_THRESHOLD", "3000")
    assert_same_encoding(nat, ora, f2, o2)
    assert nat.last_encode_redo_samples() > 0
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
