"""GPU parity: the HIP encode path (through the C ABI) must produce token ids
bit-identical to the CPU oracle on the same inputs."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth

from util import assert_same_encoding, corpus_and_vocab


def _pair(tokens, scores):
    return tgx.NativeModel(tokens, scores), orc.OracleModel(tokens, scores)


def _enc(native, texts, dropout=0.0, seed=0):
    flat, offs = tgx.pack(texts)
    res = native.encode_batch_flat(flat, offs, dropout, seed)
    ids, oo = res.ids(), res.offsets()
    return [ids[int(oo[i]):int(oo[i + 1])].tolist() for i in range(len(texts))]


def test_reference_kats_on_gpu(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json"), encoding="utf-8") as f:
        kats = json.load(f)
    for k in kats["encode"]:
        nat = tgx.NativeModel([t.encode() for t, _ in k["vocab"]], [s for _, s in k["vocab"]])
        assert _enc(nat, [k["input"].encode()], k["dropout"], seed=99) == [k["ids"]], k["source"]
    k = kats["default_vocab_roundtrip"]
    toks = [bytes([i]) for i in range(256)]
    nat = tgx.NativeModel(toks, [1.0 / 256.0] * 256)
    raw = k["input"].encode("utf-8")
    assert _enc(nat, [raw]) == [list(raw)]


def test_hf_golden_on_gpu(golden_dir):
    with open(os.path.join(golden_dir, "hf_ascii.json"), encoding="utf-8") as f:
        g = json.load(f)
    for case in g["cases"]:
        nat = tgx.NativeModel([t.encode() for t, _ in case["vocab"]], [s for _, s in case["vocab"]])
        got = _enc(nat, [t.encode("ascii") for t in case["texts"]])
        assert got == case["ids"]


def test_edge_cases():
    nat, ora = _pair([b"a", b"b", b"ab", b"", b"ab"], [-3.0, -3.0, -6.0, -1.0, -6.0])
    texts = [b"", b"a", b"ab", b"abab", b"", b"b" * 63, b"a" * 64, b"ab" * 32 + b"a", b"a" * 65, b"ab" * 64, b"b" * 129]
    assert _enc(nat, texts) == ora.encode_batch(texts)
    # ties: longest token wins; duplicate bytes: last id wins (id 4, never 2)
    assert _enc(nat, [b"ab"]) == [[4]]
    assert nat.common_prefix_search(b"abz") == ora.common_prefix_search(b"abz") == [(0, 1), (4, 2)]


def test_no_path_reports_lowest_sample():
    nat, _ = _pair([b"a", b"b"], [-1.0, -1.0])
    flat, offs = tgx.pack([b"ab", b"abc", b"a", b"!!", b""])
    with pytest.raises(tgx.TokenGeeXError) as e:
        nat.encode_batch_flat(flat, offs)
    assert str(e.value) == "no path to position 3/3"  # src/lib.rs:243-245
    assert (e.value.status, e.value.sample, e.value.pos, e.value.length) == (4, 1, 3, 3)
    # the handle stays usable after an error
    assert _enc(nat, [b"abba"]) == [[0, 1, 1, 0]]


def test_positive_scores_and_invalid_utf8_tokens():
    toks = [bytes([i]) for i in range(256)] + [b"\xff\xfe", b"\xe4\xb8", b"\x80\x80\x80"]
    scores = [1.0 / 256.0] * 256 + [5.0, 0.25, -1.0]
    nat, ora = _pair(toks, scores)
    rng = np.random.default_rng(3)
    texts = [bytes(rng.integers(0, 256, size=int(n)).astype(np.uint8)) for n in [1, 5, 64, 100, 1000]]
    texts += [b"\xff\xfe" * 40, "你好，我叫罗杰斯".encode()]
    assert _enc(nat, texts) == ora.encode_batch(texts)


@pytest.mark.parametrize("max_len", [2, 3, 7, 16, 17, 24, 33, 64])
def test_random_vocab_parity_across_token_lengths(max_len):
    rng = np.random.default_rng(1000 + max_len)
    flat, offs = synth.make_corpus(256 << 10, "mixed", seed_offset=max_len)
    toks, scores = synth.random_vocab(rng, bytes(flat[: 64 << 10]), n_multi=3000, max_len=max_len)
    nat, ora = _pair(toks, scores)
    assert nat.max_token_len <= max_len
    assert_same_encoding(nat, ora, flat, offs)


def test_sparse_vocab_unreachable_positions():
    # no single-byte cover: many positions are unreachable, some samples fail
    rng = np.random.default_rng(5)
    flat, offs = synth.make_corpus(64 << 10, "ascii", max_len=256)
    toks, scores = synth.random_vocab(rng, bytes(flat), n_multi=4000, max_len=6, all_bytes=False)
    nat, ora = _pair(toks, scores)
    ok = []
    for i in range(len(offs) - 1):
        t = bytes(flat[int(offs[i]):int(offs[i + 1])])
        try:
            ora.encode(t)
            ok.append(t)
        except orc.NoPath:
            pass
    # build a few guaranteed-reachable samples by concatenating tokens
    for _ in range(50):
        ok.append(b"".join(toks[int(j)] for j in rng.integers(0, len(toks), size=int(rng.integers(1, 60)))))
    assert _enc(nat, ok) == ora.encode_batch(ok)


def test_realistic_vocab_parity_with_ties():
    flat, offs, toks, scores = corpus_and_vocab(4 << 20, "mixed", 8000, 16)
    nat, ora = _pair(toks, scores)
    ids, oo = assert_same_encoding(nat, ora, flat, offs)
    assert ids.size > 0 and int(oo[-1]) == ids.size


def test_long_samples():
    flat, offs, toks, scores = corpus_and_vocab(3 << 20, "ascii", 4000, 16)
    nat, ora = _pair(toks, scores)
    offs2 = np.array([0, 1 << 20, (1 << 20) + 300000, flat.size], dtype=np.uint64)  # 1 MiB, 300 KB, rest
    assert_same_encoding(nat, ora, flat, offs2)


def test_dropout_parity_and_extremes():
    flat, offs, toks, scores = corpus_and_vocab(512 << 10, "mixed", 3000, 12)
    nat, ora = _pair(toks, scores)
    assert_same_encoding(nat, ora, flat, offs, dropout=0.1, seed=42)
    assert_same_encoding(nat, ora, flat, offs, dropout=0.5, seed=7)
    ids, oo = assert_same_encoding(nat, ora, flat, offs, dropout=1.0, seed=1)
    assert ids.size == flat.size  # every multi-byte token dropped (model.rs:217-236)


def test_corpus_resident_passes_and_count_tokens():
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 6000, 16)
    nat, ora = _pair(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    r1 = nat.encode_corpus(corpus)
    r2 = nat.encode_corpus(corpus)
    want_ids, want_offs = ora.encode_batch_flat(flat, offs, threads=8)
    np.testing.assert_array_equal(r1.ids(), want_ids)
    np.testing.assert_array_equal(r2.ids(), want_ids)
    np.testing.assert_array_equal(r2.offsets(), want_offs)
    freq = nat.count_tokens(corpus)
    np.testing.assert_array_equal(freq, np.bincount(want_ids, minlength=len(toks)).astype(np.uint64))
    np.testing.assert_array_equal(freq, ora.count_tokens_flat(flat, offs, threads=8))
    times = nat.last_kernel_times()
    assert times.get("ids_histogram_kernel", 0) > 0, times  # the counters of 6 256 ids fit a block's LDS
    os.environ["TGX_FREQ_SORT"] = "1"                         # larger vocabularies: radix sort + run-length encode
    try:
        np.testing.assert_array_equal(nat.count_tokens(corpus), freq)
        assert nat.last_kernel_times().get("ids_sort+rle", 0) > 0
    finally:
        del os.environ["TGX_FREQ_SORT"]
    f2 = nat.count_tokens(corpus, freq.copy())                # accumulates into the caller's vector
    np.testing.assert_array_equal(f2, 2 * freq)
    # a second model on the same resident corpus (prune rebuilds the model every sub-iteration)
    toks2, scores2 = toks[:3000], scores[:3000] * 1.01
    keep = [bytes([b]) for b in range(256)]
    toks2 = toks2 + [k for k in keep if k not in set(toks2)]
    scores2 = np.concatenate([scores2, np.full(len(toks2) - 3000, -12.0)])
    nat2, ora2 = _pair(toks2, scores2)
    np.testing.assert_array_equal(nat2.encode_corpus(corpus).ids(), ora2.encode_batch_flat(flat, offs, threads=8)[0])


def test_tokenizer_surface_on_gpu():
    vocab = [(bytes([i]), -8.0, True) for i in range(256)] + [(b"Hello", -3.0, False), (b"lo", -4.0, False),
                                                                (b" world", -3.5, False), ("你好".encode(), -2.0, False)]
    tk = tgx.Tokenizer(vocab, [tgx.CrlfProcessor()], ["<EOS>", "random", "<EOS_2>"])
    base = tk.base_vocab_size()
    ids = tk.encode("<EOS>Hello world\r\n你好<EOS_2>", 0.0)
    assert ids == [base + 0, 256, 258, 10, 259, base + 2]
    assert tk.decode(ids, True) == "<EOS>Hello world\n你好<EOS_2>"
    assert tk.decode(ids, False) == "Hello world\n你好"
    batch = tk.encode_batch(["Hello", "", "randomHello"], 0.0)
    assert batch == [[256], [], [base + 1, 256]]
    assert tk.encode_ordinary("<EOS>", 0.0) == list(b"<EOS>")
    assert tk.common_prefix_search("Hello") == [ord("H"), 256]
    tk2 = tgx.Tokenizer.from_str(tk.to_string())
    assert tk2.encode_batch(["Hello world<EOS>"], 0.0) == tk.encode_batch(["Hello world<EOS>"], 0.0)
    with pytest.raises(tgx.TokenGeeXError):
        tgx.Tokenizer([(b"a", -1.0, False)]).encode("b", 0.0)


def test_large_vocabularies_64k_and_200k():
    """BASELINE configs[2] (64 K vocab, max token 16) and a prune-start sized vocabulary."""
    flat, offs = synth.make_corpus(12 << 20, "mixed", seed_offset=77)
    for size in (65536, 200000):
        toks, scores = synth.build_vocab(flat[: 6 << 20], size, 16)
        assert len(toks) == size
        nat, ora = _pair(toks, scores)
        sub_flat, sub_offs = flat[: int(offs[400])], offs[:401]
        assert_same_encoding(nat, ora, sub_flat, sub_offs)
        corpus = tgx.NativeCorpus(sub_flat, sub_offs)
        np.testing.assert_array_equal(nat.count_tokens(corpus), ora.count_tokens_flat(sub_flat, sub_offs, threads=8))


@pytest.mark.parametrize("size", [32000, 65536])
def test_committed_spec_vocabularies_against_the_oracle(size):
    """The vocabularies bench.py runs on (the generate stand-in over a 64 MiB slice, committed as
    tests/golden/vocab_32000.npz / vocab_65536.npz; configs[1] and configs[2]): encode, frequency pass, pair scan and
    E-step against the oracle on a few MiB."""
    toks, scores, slice_mib = synth.load_spec_vocab(size)
    assert len(toks) == size and slice_mib == 64
    flat, offs = synth.make_corpus(6 << 20, "mixed", seed_offset=1000)
    nat, ora = _pair(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    assert "encode5_kernel" in nat.last_kernel_times() or "encode6_kernel" in nat.last_kernel_times()
    corpus = tgx.NativeCorpus(flat, offs)
    np.testing.assert_array_equal(nat.count_tokens(corpus), ora.count_tokens_flat(flat, offs, threads=8))
    keys, counts = nat.count_pairs(corpus)
    wk, wc = ora.count_pairs_flat(flat, offs, threads=8)
    np.testing.assert_array_equal(keys, wk)
    np.testing.assert_array_equal(counts, wc)
    got, gz = nat.estep(corpus)
    st, want, wz, _ = ora.estep_flat(flat, offs, threads=8)
    assert st == orc.OK
    longest = int(np.diff(offs.astype(np.int64)).max())
    np.testing.assert_allclose(got, want, rtol=1.2e-8 * max(1.0, longest / 4096.0), atol=1e-12)
    assert np.array_equal(got != 0, want != 0) and abs(gz - wz) <= 1e-12 * abs(wz) + 1e-9


def test_both_kernel_paths_agree(monkeypatch):
    """The four-samples-per-wave path and the one-sample-per-wave path (TGX_PATH=fused)."""
    flat, offs, toks, scores = corpus_and_vocab(1 << 20, "mixed", 5000, 16, seed_offset=3)
    nat, ora = _pair(toks, scores)
    want_ids, want_offs = ora.encode_batch_flat(flat, offs, threads=8)
    for path in ("rows5", "rows4", "fused"):
        monkeypatch.setenv("TGX_PATH", path)
        res = nat.encode_batch_flat(flat, offs)
        np.testing.assert_array_equal(res.ids(), want_ids)
        np.testing.assert_array_equal(res.offsets(), want_offs)
        assert ("encode4_kernel" in nat.last_kernel_times()) == (path == "rows4")
        assert ("encode5_kernel" in nat.last_kernel_times()) == (path == "rows5")


def test_non_finite_scores_use_the_exact_generic_path():
    # -inf / +inf / NaN scores are legal f64 values for the reference's DP (model.rs:98-101)
    inf = float("inf")
    toks = [b"a", b"b", b"ab", b"ba", b"c", b"abc"]
    for scores in ([-1.0, -inf, -0.5, -2.0, -1.0, -3.0], [-1.0, -1.0, inf, -2.0, -1.0, -inf],
                   [-1.0, float("nan"), -0.5, -2.0, -1.0, -1.0]):
        nat, ora = _pair(toks, scores)
        texts = [b"abcab", b"bbbb", b"abab" * 20, b"cab", b"b"]
        assert _enc(nat, texts) == ora.encode_batch(texts), scores
        assert "encode4_kernel" not in nat.last_kernel_times()


@pytest.mark.parametrize("path,kernel,min_waves", [("rows5", "encode5_kernel", 28), ("rows4", "encode4_kernel", 18)])
def test_launch_geometry_fits_the_device(monkeypatch, path, kernel, min_waves):
    """The four-samples-per-wave kernels must really have their planned waves resident: two blocks of nine or
    ten waves put six waves on some SIMD, which needs encode4_kernel to stay within 80 VGPRs (a build that
    drifted to 82 ran at half occupancy, 28 ms instead of 18 ms per GiB, without failing any parity test)."""
    flat, offs, toks, scores = corpus_and_vocab(16 << 20, "mixed", 4000, 16, max_len=256)   # more samples than the chip has rows
    nat = tgx.NativeModel(toks, scores)
    monkeypatch.setenv("TGX_PPL", "1")       # one position per lane: the variant with the most waves per CU
    monkeypatch.setenv("TGX_PATH", path)
    res = nat.encode_batch_flat(flat, offs)
    res.free()
    assert kernel in nat.last_kernel_times()
    assert nat.last_encode_waves_per_cu() >= min_waves
    if path == "rows5":   # the default: four positions per lane, one block of sixteen waves per CU
        monkeypatch.delenv("TGX_PPL")
        nat.encode_batch_flat(flat, offs).free()
        assert nat.last_encode_waves_per_cu() == 16


@pytest.mark.parametrize("path", ["rows5", "rows4"])
@pytest.mark.parametrize("ppl", ["1", "2", "4"])
def test_every_positions_per_lane_variant(monkeypatch, ppl, path):
    """encode5_kernel and encode4_kernel exist for 1, 2 and 4 positions per lane (the host normally picks by
    corpus shape): each bit-exact against the oracle, with samples that end on and off block boundaries, with
    and without dropout."""
    monkeypatch.setenv("TGX_PPL", ppl)
    monkeypatch.setenv("TGX_PATH", path)
    flat, offs, toks, scores = corpus_and_vocab(512 << 10, "mixed", 3000, 16, seed_offset=23, max_len=20000)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    assert ("encode5_kernel" if path == "rows5" else "encode4_kernel") in nat.last_kernel_times()
    assert_same_encoding(nat, ora, flat, offs, dropout=0.2, seed=5)
    texts = [b"", b"a", b"ab" * 8, b"ab" * 8 + b"c", b"ab" * 16, b"ab" * 32, b"ab" * 32 + b"a", b"hello world " * 30]
    f2, o2 = tgx.pack(texts)
    assert_same_encoding(nat, ora, f2, o2)


@pytest.mark.parametrize("path", ["default", "rows4l", "rows2"])
@pytest.mark.parametrize("max_len", [17, 20, 24, 32])
def test_long_token_vocabularies(monkeypatch, max_len, path):
    """Vocabularies whose longest token has 17..32 bytes (after `merge`): the LONG build of encode5_kernel by default
    (round 3: rank indices, a list of long matches per wave, the slow steps only where a long match is pending),
    encode4l_kernel (round 1: 16-byte records, scores through the match buffer) with TGX_PATH=rows4l, encode2_kernel
    (two samples per wave) with TGX_PATH=rows2; all bit-exact against the oracle incl. dropout, ties, block-boundary
    lengths and unreachable ends; the default also with every token its own score and a small LDS copy."""
    if path != "default":
        monkeypatch.setenv("TGX_PATH", path)
    rng = np.random.default_rng(4000 + max_len)
    flat, offs = synth.make_corpus(384 << 10, "mixed", seed_offset=50 + max_len, max_len=20000)
    toks, scores = synth.random_vocab(rng, bytes(flat[: 96 << 10]), n_multi=4000, max_len=max_len, tie_fraction=0.5)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert 16 < nat.max_token_len <= max_len
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert {"default": "encode5_kernel", "rows4l": "encode4l_kernel", "rows2": "encode2_kernel"}[path] in kt and "trace32_kernel" in kt
    assert_same_encoding(nat, ora, flat, offs, dropout=0.3, seed=9)
    longest = max(toks, key=len)
    texts = [b"", b"a", longest, longest * 3, b"ab" * 16, b"ab" * 16 + b"a", b"ab" * 32, b"q" * 31, b"q" * 33, b"hello world " * 40]
    f2, o2 = tgx.pack(texts)
    assert_same_encoding(nat, ora, f2, o2)
    if path == "default":
        sc3 = np.asarray(scores, np.float64) - rng.random(len(toks)) * 1e-3
        nat3, ora3 = tgx.NativeModel(toks, sc3), orc.OracleModel(toks, sc3)
        monkeypatch.setenv("TGX_E5_HOT", "500")   # most values outside the LDS copy
        assert_same_encoding(nat3, ora3, flat, offs)
        assert "encode5_kernel" in nat3.last_kernel_times() and nat3.last_encode_hot_values() == 500
        assert_same_encoding(nat3, ora3, f2, o2)


@pytest.mark.parametrize("max_len", [16, 24])
def test_token_end_mask_pipeline(monkeypatch, max_len):
    """TGX_TRACE=mask (trace2.hip): the back-trace only marks token ends in a per-sample bit mask, a popcount scan
    numbers the tokens, emit_kernel looks the ids up and writes them in place — no `tmp`, no compaction.  Kept as the
    measured alternative (1.3 ms per GiB slower than trace + compact); ids, offsets and errors as the
    default's: bit-exact against the oracle, with empty samples, dropout, samples of 1 .. 200 bytes (a word of the mask
    and less) and an unreachable end."""
    monkeypatch.setenv("TGX_TRACE", "mask")
    rng = np.random.default_rng(77 + max_len)
    flat, offs = synth.make_corpus(512 << 10, "mixed", seed_offset=31 + max_len, max_len=20000)
    if max_len == 16:
        toks, scores = synth.build_vocab(flat[: 256 << 10], 3000, 16)
    else:
        toks, scores = synth.random_vocab(rng, bytes(flat[: 96 << 10]), n_multi=4000, max_len=max_len, tie_fraction=0.5)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "mark_kernel" in kt and "emit_kernel" in kt and "compact_kernel" not in kt
    assert_same_encoding(nat, ora, flat, offs, dropout=0.3, seed=9)
    texts = [b"", b"a", b"", b"ab" * 16, b"ab" * 32, b"ab" * 32 + b"a", b"q" * 63, b"q" * 64, b"q" * 65, b"", b"hello world " * 40, b""]
    texts += [bytes(flat[i * 211: i * 211 + n]) for i, n in enumerate(range(1, 200))]
    f2, o2 = tgx.pack(texts)
    assert_same_encoding(nat, ora, f2, o2)
    nat2, _ = _pair([b"a", b"b"], [-1.0, -1.0])  # test_no_path_reports_lowest_sample's case
    f3, o3 = tgx.pack([b"ab", b"abc", b"a", b"!!", b""])
    with pytest.raises(tgx.TokenGeeXError) as e:
        nat2.encode_batch_flat(f3, o3)
    assert str(e.value) == "no path to position 3/3"
    assert (e.value.status, e.value.sample, e.value.pos, e.value.length) == (4, 1, 3, 3)
    assert _enc(nat2, [b"abba", b"", b"b"]) == [[0, 1, 1, 0], [], [1]]


@pytest.mark.parametrize("carry", ["0", "1"])
@pytest.mark.parametrize("max_len", [16, 24])
def test_trace_keeps_or_flushes_its_waiting_tokens(monkeypatch, max_len, carry):
    """trace_kernel / trace32_kernel in both modes (trace_body.h): tokens that wait for their lookup carry over from one sample
    to the next (TGX_TRACE_CARRY=1: the default for samples of less than 2 KiB on average) or are flushed per sample; on a
    corpus of short samples with empty ones in between and on one of long samples, ids and offsets against the oracle."""
    monkeypatch.setenv("TGX_TRACE_CARRY", carry)
    rng = np.random.default_rng(500 + max_len)
    flat, offs = synth.make_corpus(384 << 10, "mixed", seed_offset=61 + max_len, max_len=180)
    if max_len == 16:
        toks, scores = synth.build_vocab(flat[: 256 << 10], 3000, 16)
    else:
        toks, scores = synth.random_vocab(rng, bytes(flat[: 96 << 10]), n_multi=4000, max_len=max_len, tie_fraction=0.5)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    assert_same_encoding(nat, ora, flat, offs)
    assert ("trace_kernel" if max_len == 16 else "trace32_kernel") in nat.last_kernel_times()
    texts = [b"", b"a", b"", b"ab" * 16, b"", b"", b"q" * 63, b"q" * 64, b"q" * 65, b""] + [bytes(flat[i * 97: i * 97 + n]) for i, n in enumerate(range(1, 150))]
    f2, o2 = tgx.pack(texts)
    assert_same_encoding(nat, ora, f2, o2)
    assert_same_encoding(nat, ora, f2, o2, dropout=0.2, seed=3)
    f3, o3 = synth.make_corpus(256 << 10, "mixed", seed_offset=5, max_len=30000)
    assert_same_encoding(nat, ora, f3, o3)


def test_long_token_overflow_redoes_only_the_samples_concerned():
    """Every position of "aaaa..." matches sixteen tokens of 17..32 bytes: far more than a wave's overflow list
    holds, so the samples of such a wave are redone by encode2_kernel (and only those: the batch also has
    hundreds of samples without a long match); ties everywhere (scores proportional to length)."""
    toks = [bytes([c]) for c in range(256)] + [b"a" * k for k in range(2, 33)]
    scores = np.array([-3.0] * 256 + [-3.0 * k for k in range(2, 33)])   # a^k scores like k single a's: ties
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    rng = np.random.default_rng(11)
    plain = [bytes(rng.integers(98, 123, size=int(rng.integers(1, 400)), dtype=np.uint8)) for _ in range(600)]
    texts = [b"a" * 700, b"a" * 33 + b"b" + b"a" * 64, b"xyz", b"a" * 17] + plain + [b"q" * 50 + b"a" * 90]
    flat, offs = tgx.pack(texts)
    assert_same_encoding(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "encode5_kernel" in kt and "encode2_kernel" in kt
    # the three samples with long runs of "a" and at most the three others of each of their waves
    assert 3 <= nat.last_encode_redo_samples() <= 40   # (a wave works on four samples at a time, 256 positions per iteration)
    # dropout: the redone samples draw the same per-(sample, position, length) numbers in either kernel
    assert_same_encoding(nat, ora, flat, offs, dropout=0.3, seed=5)
    assert nat.last_encode_redo_samples() >= 1
    # a batch without long matches leaves nothing to redo
    f2, o2 = tgx.pack(plain)
    assert_same_encoding(nat, ora, f2, o2)
    assert nat.last_encode_redo_samples() == 0 and "encode2_kernel" not in nat.last_kernel_times()


def test_concurrent_calls_on_one_handle():
    """The reference's Tokenizer is Sync: rayon workers call `&self` methods concurrently (tokenizer.rs:107-110).
    A model handle serialises its passes internally; four host threads running different passes over different
    batches on ONE handle get the results of the serial calls."""
    import threading
    flat, offs, toks, scores = corpus_and_vocab(n_bytes=3 << 20, vocab_size=3000, max_len=8192)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    cuts = [0, offs.size // 4, offs.size // 2, 3 * offs.size // 4, offs.size - 1]
    parts = [(flat[int(offs[a]):int(offs[b])], (offs[a:b + 1] - offs[a]).astype(np.uint64))
             for a, b in zip(cuts[:-1], cuts[1:])]
    want = [ora.encode_batch_flat(f, o, threads=8) for f, o in parts]
    want_freq = [ora.count_tokens_flat(f, o, threads=8) for f, o in parts]
    got, errors = [None] * 4, []

    def work(i):
        try:
            out = []
            for _ in range(3):
                res = nat.encode_batch_flat(*parts[i])
                out.append((res.ids().copy(), res.offsets().copy()))
                res.free()
                corpus = tgx.NativeCorpus(*parts[i])
                out.append(nat.count_tokens(corpus))
                corpus.free()
            got[i] = out
        except Exception as e:  # surfaced below, in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(4):
        for k in range(0, 6, 2):
            np.testing.assert_array_equal(got[i][k][0], want[i][0])
            np.testing.assert_array_equal(got[i][k][1], want[i][1])
            np.testing.assert_array_equal(got[i][k + 1], want_freq[i])


@pytest.mark.parametrize("max_len", [16, 24])
def test_trace_ring_extremes(max_len):
    """The trace looks tokens up 64 at a time from a ring of 128 entries per wave (trace_body.h).  Dropout 1.0
    makes every token one byte (KAT 2, src/model.rs:217-236): 64 tokens per window, the ring wraps on every
    window; a vocabulary whose multi-byte tokens are all long gives a few tokens per window and partial last
    batches.  Sample lengths around the multiples of 64."""
    rng = np.random.default_rng(3)
    toks = [bytes([c]) for c in range(256)] + [bytes(rng.integers(97, 101, size=max_len - int(rng.integers(0, 3)), dtype=np.uint8))
                                               for _ in range(300)]
    toks = list(dict.fromkeys(toks))
    scores = np.concatenate([np.full(256, -8.0), -1.0 - rng.random(len(toks) - 256)])
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    texts = [bytes(rng.integers(97, 101, size=n, dtype=np.uint8)) for n in (1, 63, 64, 65, 127, 128, 129, 191, 192, 4095, 4096, 4097, 20000)]
    texts += [b"".join(toks[256 + int(i)] for i in rng.integers(0, len(toks) - 256, size=400)) for _ in range(4)]
    flat, offs = tgx.pack(texts)
    assert_same_encoding(nat, ora, flat, offs)
    ids, _ = assert_same_encoding(nat, ora, flat, offs, dropout=1.0, seed=1)
    np.testing.assert_array_equal(ids, flat.astype(np.uint32))   # one token per byte, id = byte value


def test_result_accessors_agree():
    """tgx_result_ids / _offsets (library-owned host copy) and tgx_result_copy_ids / _copy_offsets (caller-owned
    memory) return the same arrays; a destination that is too small is refused."""
    import ctypes as C
    from tokengeex_amd import _lib
    flat, offs, toks, scores = corpus_and_vocab(n_bytes=1 << 20, vocab_size=2000, max_len=4096)
    nat = tgx.NativeModel(toks, scores)
    res = nat.encode_batch_flat(flat, offs)
    ids, oo = res.ids(), res.offsets()
    t, n = res.num_tokens, res.num_samples
    p_ids = C.cast(_lib.lib.tgx_result_ids(res._h), C.POINTER(C.c_uint32))
    p_off = C.cast(_lib.lib.tgx_result_offsets(res._h), C.POINTER(C.c_uint64))
    np.testing.assert_array_equal(np.ctypeslib.as_array(p_ids, shape=(t,)), ids)
    np.testing.assert_array_equal(np.ctypeslib.as_array(p_off, shape=(n + 1,)), oo)
    assert oo[-1] == t and oo[0] == 0
    small = np.empty(max(t - 1, 1), np.uint32)
    assert _lib.lib.tgx_result_copy_ids(res._h, _lib.ptr(small), t - 1) == _lib.ERR_INVALID
    res.free()


def test_full_size_properties():
    """BASELINE.json configs[1] at full size (1 GiB, 32 000-entry vocabulary) through properties that need no
    oracle pass of that size: decoding the ids gives back every byte of the text (src/model.rs:146-160, checked
    chunk by chunk), a second pass and a pass over the two halves of the batch give the same ids (samples are
    independent, src/tokenizer.rs:107-110), and the first 8 MiB agree with the CPU oracle bit for bit."""
    toks, scores, _ = synth.load_spec_vocab(32000)   # the committed 64 MiB-slice vocabulary (bench.py's)
    flat, offs = synth.make_corpus(1024 << 20, "mixed", seed_offset=1000)
    nat = tgx.NativeModel(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    res = nat.encode_corpus(corpus)
    ids, oo = res.ids(), res.offsets()
    res.free()
    res = nat.encode_corpus(corpus)
    np.testing.assert_array_equal(res.ids(), ids)          # deterministic
    res.free()
    corpus.free()
    # token bytes as a flat table
    tlen = np.array([len(t) for t in toks], np.int64)
    toff = np.concatenate([[0], np.cumsum(tlen)])
    tbytes = np.frombuffer(b"".join(toks), np.uint8)
    o = offs.astype(np.int64)
    t = oo.astype(np.int64)
    assert t[-1] == ids.size and ids.max() < len(toks)
    # per sample: the token lengths add up to the sample's length
    lens = tlen[ids]
    per_sample = np.add.reduceat(lens, t[:-1].clip(max=max(ids.size - 1, 0)))
    per_sample[t[1:] == t[:-1]] = 0                         # reduceat's convention for empty slices
    np.testing.assert_array_equal(per_sample, o[1:] - o[:-1])
    # decode == text, in chunks of about 64 MiB of text
    S = o.size - 1
    a = 0
    while a < S:
        b = min(int(np.searchsorted(o, o[a] + (64 << 20), side="right")), S)
        b = max(b, a + 1)
        cid = ids[t[a]:t[b]]
        cl = tlen[cid]
        starts = np.cumsum(cl) - cl
        idx = np.repeat(toff[cid] - starts, cl) + np.arange(int(cl.sum()), dtype=np.int64)
        np.testing.assert_array_equal(tbytes[idx], flat[o[a]:o[b]])
        a = b
    # the two halves of the batch on their own
    h = S // 2
    for lo, hi in ((0, h), (h, S)):
        sub = nat.encode_batch_flat(flat[o[lo]:o[hi]], (offs[lo:hi + 1] - offs[lo]).astype(np.uint64))
        np.testing.assert_array_equal(sub.ids(), ids[t[lo]:t[hi]])
        sub.free()
    # and the oracle on the head of the corpus
    k = int(np.searchsorted(o, 8 << 20))
    want_ids, want_offs = orc.OracleModel(toks, scores).encode_batch_flat(flat[:o[k]], offs[:k + 1], threads=8)
    np.testing.assert_array_equal(ids[:t[k]], want_ids)
    np.testing.assert_array_equal(oo[:k + 1], want_offs)


def test_native_front_end_equals_the_per_sample_python_path(monkeypatch):
    """encode_batch through the packed-buffer front end (native special-token split, CRLF, id assembly:
    csrc/frontback.cpp) against the per-sample Python mirror of src/tokenizer.rs:65-123, and a decode round
    trip through the native decode_batch."""
    flat, offs, toks, scores = corpus_and_vocab(1 << 20, "mixed", 3000, 16, seed_offset=31, max_len=4000)
    vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
    specials = ["<EOS>", "<|pad|>", "\n\n\n", "<EOS_2>"]
    tk = tgx.Tokenizer(vocab, [tgx.CrlfProcessor()], specials)
    raw = flat.tobytes()
    rng = np.random.default_rng(4)
    texts = []
    for i in range(min(400, offs.size - 1)):
        t = raw[int(offs[i]):int(offs[i + 1])].decode("utf-8", "ignore")
        if i % 3 == 0:
            k = int(rng.integers(0, max(1, len(t))))
            t = t[:k] + specials[i % 4] + t[k:] + ("\r\n" if i % 2 else "") + specials[(i + 1) % 4]
        texts.append(t)
    texts += ["", "<EOS>", "<EOS><EOS>", "\r\n", "a\r\nb<EOS>\r", "<EO", "<EOS_2>"]
    got = tk.encode_batch(texts, 0.0)
    got_ord = tk.encode_ordinary_batch(texts, 0.0)
    monkeypatch.setattr(tgx.Tokenizer, "_native_front", lambda self: False)
    assert got == tk.encode_batch(texts, 0.0)
    assert got_ord == tk.encode_ordinary_batch(texts, 0.0)
    monkeypatch.undo()
    dec = tk.decode_batch(got, True)
    assert dec == [tk.decode(g, True) for g in got]                     # native decode_batch == per-sample decode
    # round trip, CRLF-normalised (a "\r" directly in front of a special token that begins with "\n" stays: the
    # processor sees the segments, src/tokenizer.rs:78-84)
    assert all(d == t.replace("\r\n", "\n") for d, t in zip(dec, texts) if "\r\n\n\n" not in t)
    base = tk.base_vocab_size()
    assert got[-1] == [base + 3] and got[-6] == [base + 0] and got[-5] == [base + 0, base + 0]


@pytest.mark.parametrize("procs", [("nfc",), ("crlf", "nfkc"), ("nfd", "crlf")])
def test_native_front_end_with_unicode_processors(monkeypatch, procs):
    """The Unicode normalisation forms of src/processor.rs:124-137 on packed buffers (csrc/unicode_norm.cpp), alone and in
    either order with the CRLF processor (src/tokenizer.rs:79-82 applies the list in order, per segment between special
    tokens): encode_batch / encode_ordinary_batch equal the per-segment Python path (unicodedata)."""
    flat, offs, toks, scores = corpus_and_vocab(1 << 20, "mixed", 3000, 16, seed_offset=33, max_len=3000)
    vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
    specials = ["<EOS>", "é", "\n\n\n"]
    tk = tgx.Tokenizer(vocab, [tgx.CrlfProcessor() if p == "crlf" else tgx.UnicodeProcessor(p) for p in procs], specials)
    raw = flat.tobytes()
    texts = []
    for i in range(min(300, offs.size - 1)):
        t = raw[int(offs[i]):int(offs[i + 1])].decode("utf-8", "ignore")
        if i % 3 == 0:
            t = t[: len(t) // 2] + "e\u0301 \ufb01 \u212b\r\n" + specials[i % 3] + "\uac01\u1100\u1161" + t[len(t) // 2:]
        texts.append(t)
    texts += ["", "e\u0301", "\u0301", "<EOS>\u0301", "a\r\n\u0301"]
    got = tk.encode_batch(texts, 0.0)
    got_ord = tk.encode_ordinary_batch(texts, 0.0)
    monkeypatch.setattr(tgx.Tokenizer, "_native_front", lambda self: False)
    assert got == tk.encode_batch(texts, 0.0)
    assert got_ord == tk.encode_ordinary_batch(texts, 0.0)


def test_host_to_host_entry_point_chunks_and_errors(monkeypatch):
    """tgx_encode_batch_host: the batch in chunks through upload / kernels / download on three host threads —
    same ids and offsets as the one-piece path whatever the chunk size, the lowest failing sample reported,
    a destination that is too small refused."""
    flat, offs, toks, scores = corpus_and_vocab(6 << 20, "mixed", 4000, 16, seed_offset=41, max_len=30000)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    want_ids, want_offs = ora.encode_batch_flat(flat, offs, threads=8)
    for mb in ("1", "2", "64"):
        monkeypatch.setenv("TGX_E2E_CHUNK_MB", mb)
        ids, oo = nat.encode_batch_host(flat, offs)
        np.testing.assert_array_equal(ids, want_ids)
        np.testing.assert_array_equal(oo, want_offs)
    # caller buffers in page-locked memory (tgx_host_alloc): same result, and the memory is given back
    from tokengeex_amd import _lib
    pf, po = _lib.pinned_empty(flat.shape, np.uint8), _lib.pinned_empty(offs.shape, np.uint64)
    pi = _lib.pinned_empty(int(want_ids.size) + 8, np.uint32)
    pf[:], po[:] = flat, offs
    ids, oo = nat.encode_batch_host(pf, po, ids_out=pi)
    np.testing.assert_array_equal(ids, want_ids)
    np.testing.assert_array_equal(oo, want_offs)
    del ids, pf, po, pi
    ids, oo = nat.encode_batch_host(flat, offs, dropout=0.2, seed=5)      # dropout: one chunk, global sample indices
    w2, o2 = ora.encode_batch_flat(flat, offs, 0.2, 5, threads=8)
    np.testing.assert_array_equal(ids, w2)
    np.testing.assert_array_equal(oo, o2)
    monkeypatch.setenv("TGX_E2E_CHUNK_MB", "1")
    with pytest.raises(tgx.TokenGeeXError):
        nat.encode_batch_host(flat, offs, ids_out=np.empty(1000, np.uint32))
    sparse = tgx.NativeModel([b"a", b"b"], [-1.0, -1.0])
    texts = [b"ab" * 300000, b"ba" * 300000, b"abc", b"a" * 700000, b"!!", b"b"]
    f2, o2 = tgx.pack(texts)
    with pytest.raises(tgx.TokenGeeXError) as e:
        sparse.encode_batch_host(f2, o2)
    assert str(e.value) == "no path to position 3/3" and e.value.sample == 2
    ids, oo = nat.encode_batch_host(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert ids.size == 0 and oo.tolist() == [0]


def test_one_process_several_device_handles():
    """tgx_encode_batch_multi: one batch from ONE host process over several model handles (one per GPU on a node; two
    and three handles on the test box's one GPU here): byte-balanced shards, a host thread per handle, ids and offsets
    packed in sample order — equal to the single call's; the lowest failing sample is reported across shards."""
    flat, offs, toks, scores = corpus_and_vocab(8 << 20, "mixed", 6000, 16, seed_offset=17)
    nat, ora = _pair(toks, scores)
    want_ids, want_oo = ora.encode_batch_flat(flat, offs, threads=8)
    for n in (1, 2, 3):
        models = [nat] + [tgx.NativeModel(toks, scores) for _ in range(n - 1)]
        ids, oo = tgx.NativeModel.encode_batch_multi(models, flat, offs)
        np.testing.assert_array_equal(oo, want_oo)
        np.testing.assert_array_equal(ids, want_ids)
        for m in models[1:]:
            m.free()
    # more handles than samples, empty samples, an empty batch
    f2, o2 = tgx.pack([b"ab", b"", b"c"])
    small = tgx.NativeModel([b"a", b"b", b"c", b"ab"], [-1.0, -1.0, -1.0, -1.5])
    ms = [small] + [tgx.NativeModel([b"a", b"b", b"c", b"ab"], [-1.0, -1.0, -1.0, -1.5]) for _ in range(4)]
    ids, oo = tgx.NativeModel.encode_batch_multi(ms, f2, o2)
    assert ids.tolist() == [3, 2] and oo.tolist() == [0, 1, 1, 2]
    ids, oo = tgx.NativeModel.encode_batch_multi(ms, np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert ids.size == 0 and oo.tolist() == [0]
    # NoPath in the second shard: the batch's sample index
    f3, o3 = tgx.pack([b"ab" * 50, b"abc" * 10, b"ab" * 50, b"a" * 20 + b"x", b"b" * 90])
    with pytest.raises(tgx.TokenGeeXError) as e:
        tgx.NativeModel.encode_batch_multi(ms[:2], f3, o3)
    assert e.value.status == 4 and e.value.sample == 3
