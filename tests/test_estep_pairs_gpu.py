"""GPU parity for the prune / merge corpus passes: E-step expected counts
(tolerance: the backward fold order and exp/log differ from the reference, see
estep.hip), and the pair scan (bit-exact)."""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth

from util import corpus_and_vocab

# Parity criterion for expected[] (1e-9 relative, 1e-12 absolute floor) holds for
# snippets up to a few KiB.  For longer snippets the reference's quantity exp(A + s + B - z) is a difference
# of log-probabilities of magnitude |z| (2.4 .. 4.7 x bytes, depending on the vocabulary's scores; ulp(3.7e5) =
# 5.8e-11 at 80 KB) whose rounding errors random-walk along the recursion: the reference's own f64 result is only
# defined to ~sqrt(n) * ulp(|z|) in the exponent.  test_estep_error_budget_against_extended_precision measures
# it: at 64 - 80 KiB the oracle is 5e-9 .. 1e-7 away from an 80-bit evaluation (the latter figure from
# tests/measure/fuzz_gpu.py, a random vocabulary with z = -4.65 per byte), the linear-domain kernels
# (estep4l.hip) 1e-13.  The tolerance against the oracle therefore scales with the snippet length.
ATOL = 1e-12


def rtol_for(snippet_bytes):
    return 1.2e-8 * max(1.0, snippet_bytes / 4096.0)


def _pair(tokens, scores):
    return tgx.NativeModel(tokens, scores), orc.OracleModel(tokens, scores)


def _check_estep(nat, ora, flat, offs, snippet_len=81920, dropout=0.0, seed=0, rtol=None):
    if rtol is None:
        longest = int(np.diff(offs.astype(np.int64)).max()) if offs.size > 1 else 0
        rtol = rtol_for(min(snippet_len, longest))
    corpus = tgx.NativeCorpus(flat, offs)
    got, gz = nat.estep(corpus, snippet_len, dropout, seed)
    st, want, wz, _ = ora.estep_flat(flat, offs, snippet_len, dropout, seed, threads=8)
    assert st == orc.OK
    np.testing.assert_allclose(got, want, rtol=rtol, atol=ATOL)
    assert np.array_equal(got != 0, want != 0)  # same set of ids with mass
    assert abs(gz - wz) <= 1e-12 * abs(wz) + 1e-9
    return got, gz


def test_marginal_kat_on_gpu(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json"), encoding="utf-8") as f:
        k = json.load(f)["marginal"]
    toks = [t.encode() for t, _ in k["vocab"]]
    nat = tgx.NativeModel(toks, [s for _, s in k["vocab"]])
    flat, offs = tgx.pack([k["input"].encode()])
    exp, z = nat.estep(tgx.NativeCorpus(flat, offs))
    names = [t for t, _ in k["vocab"]]
    for name, want in k["expected"].items():
        assert abs(exp[names.index(name)] - want) < 5e-7, name
    assert abs(z - (-12.0 + math.log(1.0 + math.exp(-1.0) + math.exp(-2.0)))) < 1e-12


def test_estep_small_cases_and_quirks():
    # no single-byte cover: positions without incoming / outgoing tokens keep 0.0 (lattice.rs:255-256)
    toks = [b"a", b"b", b"ab", b"ba", b"aba", b"c", b"bcb"]
    scores = [-1.0, -1.5, -1.7, -2.0, -2.2, -3.0, -0.5]
    nat, ora = _pair(toks, scores)
    texts = [b"abaabab", b"a", b"abcba", b"cbcb", b"ab" * 100, b"aba" * 43 + b"c", b"bcbcb"]
    flat, offs = tgx.pack(texts)
    _check_estep(nat, ora, flat, offs)
    _check_estep(nat, ora, flat, offs, snippet_len=3)    # "aba","aba","b": cuts through tokens (prune.rs:83)
    _check_estep(nat, ora, flat, offs, snippet_len=64)   # exactly one block
    _check_estep(nat, ora, flat, offs, snippet_len=65)


def test_estep_realistic_corpus():
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16)
    nat, ora = _pair(toks, scores)
    got, _ = _check_estep(nat, ora, flat, offs)
    # every byte of the corpus is covered with total mass 1
    mass = float(sum(got[i] * len(toks[i]) for i in range(len(toks))))
    assert abs(mass - flat.size) < 1e-6 * flat.size
    assert "estep7_kernel" in nat.last_kernel_times()   # round 4: the default for tokens of at most 16 bytes


def test_estep_snippets_long_samples_and_token_lengths():
    flat, offs = synth.make_corpus(1 << 20, "mixed", seed_offset=5)
    rng = np.random.default_rng(9)
    for max_len in (5, 24, 40):
        toks, scores = synth.random_vocab(rng, bytes(flat[: 64 << 10]), 2500, max_len, tie_fraction=0.0)
        nat, ora = _pair(toks, scores)
        offs2 = np.array([0, 300000, 300001, flat.size], dtype=np.uint64)  # long samples -> several snippets
        _check_estep(nat, ora, flat, offs2, snippet_len=81920)
        _check_estep(nat, ora, flat[: 200000], np.array([0, 200000], dtype=np.uint64), snippet_len=1000)


def test_estep_dropout_matches_oracle_decisions():
    flat, offs, toks, scores = corpus_and_vocab(512 << 10, "mixed", 3000, 12)
    nat, ora = _pair(toks, scores)
    _check_estep(nat, ora, flat, offs, dropout=0.1, seed=42)
    _check_estep(nat, ora, flat, offs, snippet_len=5000, dropout=0.5, seed=7)


def test_estep_z_not_normal_is_reported():
    nat, ora = _pair([b"a", b"b"], [0.0, -1.0])
    flat, offs = tgx.pack([b"bb", b"aa", b"ab"])  # z("aa") == 0.0 -> not normal (prune.rs:90-96)
    corpus = tgx.NativeCorpus(flat, offs)
    with pytest.raises(tgx.TokenGeeXError) as e:
        nat.estep(corpus)
    assert e.value.status == 6 and e.value.sample == 1
    st, _, _, es = ora.estep_flat(flat, offs)
    assert st == orc.ERR_Z_NOT_NORMAL and es == 1


def test_count_pairs_parity():
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 6000, 16)
    nat, ora = _pair(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    keys, counts = nat.count_pairs(corpus)
    wk, wc = ora.count_pairs_flat(flat, offs, threads=8)
    np.testing.assert_array_equal(keys, wk)
    np.testing.assert_array_equal(counts, wc)
    assert "pair_keys_kernel" in nat.last_kernel_times()


def test_count_pairs_small_and_empty():
    nat, ora = _pair([b"a", b"b", b"c", b"ab"], [-3.0, -3.0, -3.0, -4.0])
    texts = [b"abc", b"abab", b"c", b"", b"ababab"]
    flat, offs = tgx.pack(texts)
    keys, counts = nat.count_pairs(tgx.NativeCorpus(flat, offs))
    got = {(int(k) >> 32, int(k) & 0xFFFFFFFF): int(c) for k, c in zip(keys, counts)}
    assert got == {(3, 2): 1, (3, 3): 3}
    flat, offs = tgx.pack([b"", b"c"])
    keys, counts = nat.count_pairs(tgx.NativeCorpus(flat, offs))
    assert keys.size == 0 and counts.size == 0


def test_estep_all_kernel_paths_agree(monkeypatch):
    """Linear-domain rows4 E-step (estep4l.hip, the default when every byte is a token), log-domain rows4
    E-step (estep4.hip, TGX_ESTEP=log) and the generic kernel (TGX_PATH=fused) against each other and the oracle."""
    flat, offs, toks, scores = corpus_and_vocab(1 << 20, "mixed", 5000, 16, seed_offset=9)
    nat, ora = _pair(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    monkeypatch.setenv("TGX_PATH", "rows4")
    lin, zl = nat.estep(corpus, 20000, 0.05, 3)
    assert "estep4l_fwd_kernel" in nat.last_kernel_times()
    monkeypatch.setenv("TGX_ESTEP", "log")
    a, za = nat.estep(corpus, 20000, 0.05, 3)
    assert "estep4_fwd_kernel" in nat.last_kernel_times()
    monkeypatch.setenv("TGX_PATH", "fused")
    b, zb = nat.estep(corpus, 20000, 0.05, 3)
    assert "estep_kernel" in nat.last_kernel_times()
    st, want, wz, _ = ora.estep_flat(flat, offs, 20000, 0.05, 3, threads=8)
    assert st == orc.OK
    for got, gz in ((lin, zl), (a, za), (b, zb)):
        np.testing.assert_allclose(got, want, rtol=rtol_for(20000), atol=ATOL)
        assert np.array_equal(got != 0, want != 0)
        assert abs(gz - wz) <= 1e-12 * abs(wz)
    np.testing.assert_allclose(a, b, rtol=rtol_for(20000), atol=ATOL)


def test_log_domain_kernels_hold_the_tightest_tolerance_they_meet(monkeypatch):
    """The literal log-domain restatement (estep4.hip, TGX_ESTEP=log) against the CPU oracle at the tolerances of
    BASELINE.md section 3: SURVEY's 1e-9 up to 16 KiB snippets (measured 2.8e-10), 6e-9 at 64 KiB (measured 3.4e-9;
    tests/measure/estep_log_tolerance.py, profiles/r02/v_estep_tolerance_by_snippet.txt)."""
    flat, offs = synth.make_corpus(4 << 20, "mixed", seed_offset=5)
    toks, scores = synth.build_vocab(flat[: 2 << 20], 8000, 16)
    monkeypatch.setenv("TGX_ESTEP", "log")
    nat, ora = _pair(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    for snip, rtol in ((4096, 1e-9), (16384, 1e-9), (65536, 6e-9)):
        got, gz = nat.estep(corpus, snip)
        assert "estep4_fwd_kernel" in nat.last_kernel_times()
        st, want, wz, _ = ora.estep_flat(flat, offs, snip, threads=8)
        assert st == orc.OK
        np.testing.assert_allclose(got, want, rtol=rtol, atol=ATOL)
        assert np.array_equal(got != 0, want != 0)
        assert abs(gz - wz) <= 1e-12 * abs(wz)


@pytest.mark.parametrize("max_len", [24, 32])
def test_estep_with_tokens_of_17_to_32_bytes(monkeypatch, max_len):
    """Vocabularies after `merge` (src/merge.rs: tokens of up to 24 bytes by default): the linear-domain rows4
    kernels in their long-token builds (a second accumulator per lane + a per-wave overflow list, estep4l.hip)
    against the oracle, with and without dropout, and against the generic kernel."""
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 6000, max_len, seed_offset=31)
    assert max(len(t) for t in toks) > 16
    nat, ora = _pair(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    for snip, dropout in ((20000, 0.0), (4096, 0.1), (81920, 0.0)):
        got, gz = nat.estep(corpus, snip, dropout, 5)
        kt = nat.last_kernel_times()
        assert "estep4l_fwd_kernel" in kt and "estep4l_bwd_kernel" in kt and "estep_kernel" not in kt, kt
        st, want, wz, _ = ora.estep_flat(flat, offs, snip, dropout, 5, threads=8)
        assert st == orc.OK
        np.testing.assert_allclose(got, want, rtol=rtol_for(snip), atol=ATOL)
        assert np.array_equal(got != 0, want != 0)
        assert abs(gz - wz) <= 1e-12 * abs(wz)
    monkeypatch.setenv("TGX_PATH", "fused")
    gen, zg = nat.estep(corpus, 20000, 0.0, 5)
    assert "estep_kernel" in nat.last_kernel_times()
    monkeypatch.delenv("TGX_PATH")
    lin, zl = nat.estep(corpus, 20000, 0.0, 5)
    np.testing.assert_allclose(lin, gen, rtol=rtol_for(20000), atol=ATOL)


def test_estep_long_token_overflow_list_falls_back_to_the_generic_kernel():
    """A run of blanks under a vocabulary with a blank token of every length up to 32: every position has sixteen
    long matches, the per-wave overflow list (62 entries per block) fills up, and the pass is redone by the
    generic kernel — same expected counts as the oracle's."""
    toks = [bytes([b]) for b in range(256)] + [b" " * k for k in range(2, 33)] + [b"ab", b"abc"]
    scores = np.concatenate([np.full(256, -6.0), -2.0 - 0.05 * np.arange(31), [-4.0, -5.0]])
    nat, ora = _pair(toks, scores)
    texts = [b" " * 700 + b"abc" * 20, b"ab" * 50 + b" " * 300, b"x"]
    flat, offs = tgx.pack(texts)
    got, gz = nat.estep(tgx.NativeCorpus(flat, offs), 81920)
    assert "estep_kernel" in nat.last_kernel_times()
    st, want, wz, _ = ora.estep_flat(flat, offs, 81920, threads=2)
    assert st == orc.OK
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=ATOL)
    assert abs(gz - wz) <= 1e-12 * abs(wz)


def test_estep_long_token_overflow_of_the_backward_list_alone_falls_back():
    """The backward kernel counts matches of 17..32 bytes by END position (windows of 16 counted from the snippet's
    end), the forward kernel by START position: 80 long matches whose ends share one backward window while their
    starts straddle two forward windows (39 + 41, the list holds 62) overflow the backward list only.  The forward
    kernel raises nothing, the backward kernel does, and the pass must be discarded and redone by the generic
    kernel — not returned with the excess matches dropped."""
    n, j = 160, 4
    region = bytes(range(0x30, 0x30 + 48))            # 48 distinct bytes at [16 j - 32, 16 j + 16)
    text = bytearray(b"." * n)
    text[16 * j - 32:16 * j + 16] = region
    text = bytes(text)
    long_toks = [text[q - L:q] for q in range(16 * j + 1, 16 * j + 17) for L in (17, 21, 25, 29, 32)]
    assert len(set(long_toks)) == 80
    starts = [q - L for q in range(16 * j + 1, 16 * j + 17) for L in (17, 21, 25, 29, 32)]
    per_fwd_window = np.bincount(np.array(starts) // 16)
    assert per_fwd_window.max() <= 62 and len(long_toks) > 62  # forward lists fit, the backward one does not
    toks = [bytes([b]) for b in range(256)] + long_toks
    scores = np.concatenate([np.full(256, -5.0), -3.0 - 0.01 * np.arange(80)])
    nat, ora = _pair(toks, scores)
    flat, offs = tgx.pack([text])
    got, gz = nat.estep(tgx.NativeCorpus(flat, offs), 81920)
    kt = nat.last_kernel_times()
    assert "estep4l_bwd_kernel" in kt and "estep_kernel" in kt, kt  # the backward kernel ran, then the fallback
    st, want, wz, _ = ora.estep_flat(flat, offs, 81920, threads=1)
    assert st == orc.OK
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=ATOL)
    assert np.array_equal(got != 0, want != 0)
    assert abs(gz - wz) <= 1e-12 * abs(wz)


def test_estep_falls_back_to_log_domain_when_a_position_has_no_incoming_token():
    """A text byte that is no token leaves a position without an incoming token (lattice.rs:255: it then
    counts as log-probability 0.0), which the linear-domain kernels cannot express: their forward kernel
    flags the pass and the log-domain kernels redo it.  Texts the vocabulary covers stay linear."""
    toks = [bytes([c]) for c in b"abcdefgh"] + [b"ab", b"abc", b"cd", b"efg", b"gh", b"hh"]
    scores = -np.linspace(1.0, 4.0, len(toks))
    nat, ora = _pair(toks, scores)
    for texts, bwd in (([b"abcdefgh" * 40, b"hhgfedcba" * 13, b"a"], "estep7_kernel"),
                       ([b"abcdefgh" * 40, b"abcXdefgh" * 9, b"a"], "estep4_bwd_kernel")):
        flat, offs = tgx.pack(texts)
        got, gz = nat.estep(tgx.NativeCorpus(flat, offs), 81920, 0.0, 0)
        assert bwd in nat.last_kernel_times()
        st, want, wz, _ = ora.estep_flat(flat, offs, 81920, 0.0, 0, threads=1)
        assert st == orc.OK
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=ATOL)
        assert np.array_equal(got != 0, want != 0)
        assert abs(gz - wz) <= 1e-12 * abs(wz)


def test_estep_error_budget_against_extended_precision(monkeypatch):
    """Where the E-step tolerance comes from: one 64 KiB snippet evaluated in 80-bit extended precision
    (util.estep_longdouble), by the oracle (the reference's f64 log-domain arithmetic) and by the kernels.
    The linear-domain kernels must sit within 1e-11 of the extended-precision values; the oracle's own
    distance from them is what the tolerance of the other tests allows for."""
    from util import estep_longdouble
    flat, offs, toks, scores = corpus_and_vocab(1 << 20, "mixed", 3000, 16, seed_offset=9)
    nat, ora = _pair(toks, scores)
    lens = np.diff(offs.astype(np.int64))
    i = int(np.argmax(lens))
    text = flat[int(offs[i]):int(offs[i + 1])].tobytes()
    assert len(text) > 60000
    truth, zt = estep_longdouble(ora, text)
    want, zo = ora.marginal(text)
    f, o = tgx.pack([text])
    big = np.abs(truth) > 1e-9

    def err(x):
        return float((np.abs(x - truth)[big] / np.abs(truth)[big]).max())
    e_ora = err(want)
    assert e_ora < rtol_for(len(text)), e_ora
    for mode, kernel in ((None, "estep7_kernel"), ("chain", "estep4l_bwd_kernel")):
        if mode:
            monkeypatch.setenv("TGX_ESTEP", mode)
        got, zg = nat.estep(tgx.NativeCorpus(f, o))
        assert kernel in nat.last_kernel_times()
        e_gpu = err(got)
        assert e_gpu < 1e-11, (kernel, e_gpu)
        assert e_gpu * 50 < e_ora                      # the deviation between the two is the oracle's rounding
        assert abs(zg - zt) <= 1e-13 * abs(zt) and abs(zo - zt) <= 1e-13 * abs(zt)


@pytest.mark.parametrize("eppl", ["1", "2", "4"])
def test_estep_every_positions_per_lane_variant(monkeypatch, eppl):
    """The linear-domain kernels exist for 1, 2 and 4 positions per lane (the host normally picks by the shape of
    the pass): each against the oracle, with snippets that end on and off block boundaries and dropout."""
    monkeypatch.setenv("TGX_EPPL", eppl)
    monkeypatch.setenv("TGX_ESTEP", "chain")   # the chained forward / backward kernels (rounds 1 - 3)
    flat, offs, toks, scores = corpus_and_vocab(512 << 10, "mixed", 3000, 16, seed_offset=21, max_len=20000)
    nat, ora = _pair(toks, scores)
    _check_estep(nat, ora, flat, offs, snippet_len=4096, dropout=0.1, seed=11)
    assert "estep4l_bwd_kernel" in nat.last_kernel_times()
    texts = [b"", b"a", b"ab" * 8, b"ab" * 8 + b"c", b"xyz" * 21 + b"x", b"q" * 64, b"q" * 65, b"hello world " * 30]
    f2, o2 = tgx.pack(texts)
    _check_estep(nat, ora, f2, o2, snippet_len=48, dropout=0.0, seed=0)


def test_count_pairs_top_is_the_head_of_the_full_table():
    """tgx_count_pairs_top: the k most frequent pairs by descending count, ascending key among equal counts."""
    flat, offs, toks, scores = corpus_and_vocab(1 << 20, "mixed", 3000, 16, seed_offset=31)
    nat, _ = _pair(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    keys, counts = nat.count_pairs(corpus)
    order = np.lexsort((keys, -counts.astype(np.int64)))
    for k in (1, 100, 5000, keys.size, keys.size + 10):
        tk, tc, total = nat.count_pairs_top(corpus, k)
        assert total == keys.size and tk.size == min(k, keys.size)
        np.testing.assert_array_equal(tk, keys[order[:tk.size]])
        np.testing.assert_array_equal(tc, counts[order[:tk.size]])
    e_flat, e_offs = tgx.pack([b"", b"a"])
    tk, tc, total = nat.count_pairs_top(tgx.NativeCorpus(e_flat, e_offs), 10)   # no pair at all
    assert tk.size == 0 and total == 0


# ---- round 3: snippets cut where no token match crosses (csrc/cuts.hip) ---------------------------------------------

@pytest.mark.parametrize("window", ["256", "2048"])
def test_estep_on_pieces_equals_the_oracle(monkeypatch, window):
    """The lattice factorises at a position no match crosses, so a pass over the pieces gives the expected counts
    and log Z of the pass over whole snippets: long samples (one of 300 000 bytes -> four reference snippets), forced
    piece mode with a small and the default window, with and without dropout, tokens up to 16 and up to 24 bytes.
    (Tolerance as for whole snippets: it is the ORACLE's log-domain rounding that grows with its snippet length.)"""
    monkeypatch.setenv("TGX_ESTEP_PIECES", "1")
    monkeypatch.setenv("TGX_ESTEP_WINDOW", window)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16, seed_offset=61)
    nat, ora = _pair(toks, scores)
    n_samples = offs.size - 1
    _check_estep(nat, ora, flat, offs)
    assert nat.last_estep_pieces() > 2 * n_samples and "cut_windows_kernel" in nat.last_kernel_times()
    _check_estep(nat, ora, flat, offs, dropout=0.3, seed=11)
    assert nat.last_estep_pieces() > 2 * n_samples
    offs2 = np.array([0, 300000, 300001, 1 << 20], dtype=np.uint64)
    _check_estep(nat, ora, flat[: 1 << 20], offs2)
    assert nat.last_estep_pieces() > 100
    rng = np.random.default_rng(10)
    toks24, scores24 = synth.random_vocab(rng, bytes(flat[: 64 << 10]), 2500, 24, tie_fraction=0.0)
    nat24, ora24 = _pair(toks24, scores24)
    _check_estep(nat24, ora24, flat[: 1 << 20], offs2)
    assert nat24.last_estep_pieces() > 100
    _check_estep(nat24, ora24, flat[: 1 << 20], offs2, dropout=0.2, seed=3)


def test_estep_pieces_where_the_text_has_no_cut(monkeypatch):
    """A run of one byte under the tokens a, aa, aaaa is crossed everywhere: no window finds a boundary, the snippets
    stay whole (and a text with cuts only every ~1 000 bytes gets pieces of that size); results as without pieces."""
    monkeypatch.setenv("TGX_ESTEP_PIECES", "1")
    monkeypatch.setenv("TGX_ESTEP_WINDOW", "256")
    toks = [b"a", b"aa", b"aaaa", b"b", b"ab"]
    scores = [-1.0, -1.6, -2.5, -2.0, -2.2]
    nat, ora = _pair(toks, scores)
    texts = [b"a" * 5000, b"a" * 300 + b"b" + b"a" * 2000, (b"a" * 999 + b"b") * 8, b"ab" * 700]
    flat, offs = tgx.pack(texts)
    _check_estep(nat, ora, flat, offs)
    # sample 0: none; sample 1: the cut before "b" is inside window 1 (a cut AFTER "b" does not exist: "ab" crosses it... and
    # "b" + "a" has no token across) — only the totals are asserted: more pieces than samples, far fewer than windows
    assert len(texts) <= nat.last_estep_pieces() < 40


def test_estep_pieces_fall_back_to_whole_snippets_for_the_log_domain(monkeypatch):
    """A byte that is no token leaves positions nothing reaches (lattice.rs:255): the linear-domain pass over the pieces
    raises its flag and the log-domain kernels redo the pass on the uncut snippets — the reference's values."""
    monkeypatch.setenv("TGX_ESTEP_PIECES", "1")
    monkeypatch.setenv("TGX_ESTEP_WINDOW", "256")
    toks = [b"a", b"b", b"ab", b"ba", b"aba", b"c", b"bcb"]
    scores = [-1.0, -1.5, -1.7, -2.0, -2.2, -3.0, -0.5]
    nat, ora = _pair(toks, scores)
    texts = [b"abaabab" * 300, b"ab" * 1000 + b"\xff" + b"abc" * 400, b"cbcb" * 500]
    flat, offs = tgx.pack(texts)
    _check_estep(nat, ora, flat, offs)
    assert any(k.startswith("estep4_") for k in nat.last_kernel_times())


def test_estep_pieces_by_default_on_a_chain_bound_batch():
    """By default a pass is cut into pieces when the estimate says its longest snippets bound it: a 16 MiB batch of
    samples up to 64 KiB is, a batch of samples up to 2 KiB is not."""
    flat, offs, toks, scores = corpus_and_vocab(16 << 20, "mixed", 8000, 16, seed_offset=62)
    nat, ora = _pair(toks, scores)
    _check_estep(nat, ora, flat, offs)
    assert nat.last_estep_pieces() > offs.size - 1
    f2, o2 = synth.make_corpus(4 << 20, "mixed", max_len=2048, seed_offset=63)
    corpus = tgx.NativeCorpus(f2, o2)
    nat.estep(corpus)
    assert nat.last_estep_pieces() == 0


# ---- round 3: the forward sweep on the 8-byte ranked records (encode5.hip: estep5_fwd_kernel) -------------------------

@pytest.mark.parametrize("eppl,hot", [(None, None), ("1", None), ("2", None), ("3", "300"), ("4", "0"), ("4", None)])
def test_estep_forward_sweep_on_ranked_records(monkeypatch, eppl, hot):
    """estep5_fwd_kernel (encode5_kernel's staggered walk and 2-byte match indices, w = exp(score value) by rank in LDS or
    read from L2) against the oracle and against estep4l_fwd_kernel (TGX_ESTEP_FWD=rows4): same matches, same weights,
    same steps, so the two agree to the rounding of the backward kernel's unordered sums; every positions-per-lane
    build, tables smaller than the vocabulary's values (COLD builds), dropout, pieces, distinct scores."""
    monkeypatch.setenv("TGX_ESTEP", "chain")   # the chained forward / backward kernels (rounds 1 - 3)
    if eppl:
        monkeypatch.setenv("TGX_EPPL", eppl)
    if hot:
        monkeypatch.setenv("TGX_E5_HOT", hot)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16, seed_offset=81, max_len=30000)
    rng = np.random.default_rng(31)
    for sc in (scores, np.asarray(scores) + rng.uniform(-0.3, 0.3, len(toks))):
        nat, ora = _pair(toks, sc)
        # a model's FIRST pass over a small corpus keeps estep4l_fwd_kernel (the tables cost more than they save there) ...
        _check_estep(nat, ora, flat, offs)
        assert "estep4l_fwd_kernel" in nat.last_kernel_times()
        # ... its second pass builds them
        got, gz = _check_estep(nat, ora, flat, offs)
        assert "estep5_fwd_kernel" in nat.last_kernel_times()
        _check_estep(nat, ora, flat, offs, dropout=0.2, seed=5)
        monkeypatch.setenv("TGX_ESTEP_PIECES", "1")
        monkeypatch.setenv("TGX_ESTEP_WINDOW", "512")
        _check_estep(nat, ora, flat, offs)
        monkeypatch.delenv("TGX_ESTEP_PIECES")
        monkeypatch.delenv("TGX_ESTEP_WINDOW")
        monkeypatch.setenv("TGX_ESTEP_FWD", "rows4")
        corpus = tgx.NativeCorpus(flat, offs)
        old, oz = nat.estep(corpus)
        assert "estep4l_fwd_kernel" in nat.last_kernel_times() and "estep5_fwd_kernel" not in nat.last_kernel_times()
        np.testing.assert_allclose(got, old, rtol=1e-11, atol=1e-13)
        assert abs(gz - oz) <= 1e-13 * abs(oz)
        monkeypatch.delenv("TGX_ESTEP_FWD")


# ---- round 4: one walk per position, every trip a lattice of its own (csrc/estep7.hip) -------------------------------

@pytest.mark.parametrize("eppl,hot,waves", [(None, None, None), ("1", None, None), ("2", None, "12"), ("3", "300", None), ("4", "0", "6"), ("4", None, "3")])
def test_estep7_every_build_against_the_oracle_and_the_chained_kernels(monkeypatch, eppl, hot, waves):
    """estep7_kernel: every positions-per-lane build, tables smaller than the vocabulary (COLD builds: weights from L2,
    expected counts to HBM), dropout, forced pieces, every token its own score — against the oracle and against the
    chained kernels of rounds 1 - 3 (both sum the same products, in different orders: 1e-11)."""
    if eppl:
        monkeypatch.setenv("TGX_EPPL", eppl)
    if hot:
        monkeypatch.setenv("TGX_E7_HOT", hot)
    if waves:
        monkeypatch.setenv("TGX_E7_WAVES", waves)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16, seed_offset=81, max_len=30000)
    rng = np.random.default_rng(31)
    for sc in (scores, np.asarray(scores) + rng.uniform(-0.3, 0.3, len(toks))):
        nat, ora = _pair(toks, sc)
        got, gz = _check_estep(nat, ora, flat, offs)
        assert "estep7_kernel" in nat.last_kernel_times()
        if eppl in ("1", "2"):   # short trips find no cut now and then: the redo kernel takes those stretches
            assert nat.last_estep_redo() > 0 and "estep7_redo_kernel" in nat.last_kernel_times()
        _check_estep(nat, ora, flat, offs, dropout=0.2, seed=5)
        assert "estep7_kernel" in nat.last_kernel_times()
        monkeypatch.setenv("TGX_ESTEP_PIECES", "1")
        monkeypatch.setenv("TGX_ESTEP_WINDOW", "512")
        _check_estep(nat, ora, flat, offs)
        assert "estep7_kernel" in nat.last_kernel_times() and nat.last_estep_pieces() > offs.size
        monkeypatch.delenv("TGX_ESTEP_PIECES")
        monkeypatch.delenv("TGX_ESTEP_WINDOW")
        monkeypatch.setenv("TGX_ESTEP", "chain")
        corpus = tgx.NativeCorpus(flat, offs)
        old, oz = nat.estep(corpus)
        assert "estep4l_bwd_kernel" in nat.last_kernel_times() and "estep7_kernel" not in nat.last_kernel_times()
        np.testing.assert_allclose(got, old, rtol=1e-11, atol=1e-13)
        assert abs(gz - oz) <= 1e-13 * abs(oz)
        monkeypatch.delenv("TGX_ESTEP")


def test_estep7_small_cases_snippet_ends_and_texts_without_cuts():
    """Trips that end on and off group boundaries, pieces of exactly 16 / 48 / 64 / 65 positions, snippets cut through
    tokens, and runs of one byte under the tokens a, aa, aaaa (no position is a cut: everything goes through the redo
    kernel, whose trips are spilled to scratch) with and without dropout."""
    toks = [b"a", b"aa", b"aaaa", b"b", b"ab", b"c", b"abc"]
    scores = [-1.0, -1.6, -2.5, -2.0, -2.2, -3.0, -2.4]
    nat, ora = _pair(toks, scores)
    texts = [b"a" * 5000, b"a" * 300 + b"b" + b"a" * 2000, (b"a" * 999 + b"b") * 8, b"ab" * 700, b"", b"a", b"b" * 15, b"b" * 16, b"b" * 17,
             b"cb" * 24, b"c" * 63, b"c" * 64, b"c" * 65, b"abc" * 21 + b"a", b"b" * 47 + b"a" * 40 + b"c" * 30]
    flat, offs = tgx.pack(texts)
    _check_estep(nat, ora, flat, offs, rtol=1e-9)
    assert "estep7_kernel" in nat.last_kernel_times() and "estep7_redo_kernel" in nat.last_kernel_times()
    assert nat.last_estep_redo() >= 3
    _check_estep(nat, ora, flat, offs, snippet_len=1000, rtol=1e-9)
    _check_estep(nat, ora, flat, offs, snippet_len=48, rtol=1e-9)
    _check_estep(nat, ora, flat, offs, dropout=0.3, seed=3, rtol=1e-9)
    _check_estep(nat, ora, flat, offs, snippet_len=100, dropout=0.5, seed=9, rtol=1e-9)


def test_estep7_with_more_than_65535_tokens():
    """32-bit match entries: the committed 500 000-entry vocabulary of BASELINE.json configs[3]."""
    from util import load_vocab_500k
    toks, scores = load_vocab_500k()
    nat, ora = _pair(toks, scores)
    flat, offs = synth.make_corpus(2 << 20, "mixed", seed_offset=77)
    _check_estep(nat, ora, flat, offs)
    assert "estep7_kernel" in nat.last_kernel_times()
    _check_estep(nat, ora, flat, offs, dropout=0.1, seed=2)


def test_estep7_after_an_m_step_ranks_by_match_counts(monkeypatch):
    """prune's second sub-iteration (src/prune.rs:36-56): the E-step of a DERIVED model with the scores an M-step leaves
    — kept single-byte tokens with tiny scores that still match at every occurrence of their byte.  The ranks that stay in
    LDS come from match counts over a sample of the corpus (tgx_api.cpp ensure_estep_trie8t); the result must not depend
    on that choice: the same expected counts with the model's own order (TGX_E7_RANK=model), with a small table
    (TGX_E7_HOT) and against the oracle."""
    from tokengeex_amd import _lib
    flat, offs = synth.make_corpus(3 << 20, "mixed", seed_offset=91)
    toks, scores = synth.build_vocab(flat[: 1 << 20], 20000, 16)
    nat, _ = _pair(toks, scores)
    corpus = tgx.NativeCorpus(flat, offs)
    exp, _ = nat.estep(corpus)
    keep = np.array([1 if len(t) == 1 else 0 for t in toks], np.uint8)
    idx, sc2 = _lib.prune_m_step(exp, keep)
    idx, sc2 = np.asarray(idx, np.uint32), np.asarray(sc2, np.float64)
    assert 1000 < idx.size < len(toks)
    toks2 = [toks[i] for i in idx]
    ora2 = orc.OracleModel(toks2, sc2)
    derived = nat.derive(idx, sc2, for_estep=True)
    got, gz = _check_estep(derived, ora2, flat, offs, dropout=0.05, seed=4)
    assert "estep7_kernel" in derived.last_kernel_times()
    monkeypatch.setenv("TGX_E7_RANK", "model")
    plain = nat.derive(idx, sc2, for_estep=True)
    got2, gz2 = plain.estep(corpus, 81920, 0.05, 4)
    np.testing.assert_allclose(got2, got, rtol=1e-9, atol=ATOL)
    assert abs(gz2 - gz) <= 1e-12 * abs(gz)
    monkeypatch.delenv("TGX_E7_RANK")
    monkeypatch.setenv("TGX_E7_HOT", "300")
    small = tgx.NativeModel(toks2, sc2, for_estep=True)
    got3, _ = small.estep(corpus, 81920, 0.05, 4)
    np.testing.assert_allclose(got3, got, rtol=1e-9, atol=ATOL)


def test_estep7_overflow_build_for_vocabularies_of_a_few_more_than_65535_tokens(monkeypatch):
    """16-bit match entries hold 65 535 ranks; a vocabulary of up to 1 024 more tokens (the usual "64 K": the committed
    65 536-entry spec vocabulary) keeps them and hands the trips in which a token of a rank beyond matches to the redo
    kernel, which then has 32-bit entries (tgx_api.cpp estep_fused, estep7.hip OVF).  Against the oracle: the spec
    vocabulary as it is, and a small vocabulary whose ranks beyond 1 500 / 40 are treated that way (TGX_E7_OVF_AT), so
    that a large share of the trips takes the path — with dropout, with the model's own rank order and with a tiny table."""
    toks, scores, _ = synth.load_spec_vocab(65536)
    assert len(toks) == 65536
    nat, ora = _pair(toks, scores)
    flat, offs = synth.make_corpus(3 << 20, "mixed", seed_offset=88)
    _check_estep(nat, ora, flat, offs)
    kt = nat.last_kernel_times()
    assert "estep7_kernel" in kt
    _check_estep(nat, ora, flat, offs, dropout=0.1, seed=5)
    # a small vocabulary, many overflow matches
    f2, o2 = synth.make_corpus(2 << 20, "mixed", seed_offset=89)
    t2, s2 = synth.build_vocab(f2[: 1 << 20], 4000, 16)
    ora2 = orc.OracleModel(t2, s2)
    for at, extra in (("1500", {}), ("40", {"TGX_E7_HOT": "20"}), ("1500", {"TGX_E7_RANK": "model"})):
        monkeypatch.setenv("TGX_E7_OVF_AT", at)
        for k, v in extra.items():
            monkeypatch.setenv(k, v)
        nat2 = tgx.NativeModel(t2, s2, for_estep=True)
        _check_estep(nat2, ora2, f2, o2)
        assert "estep7_kernel" in nat2.last_kernel_times() and nat2.last_estep_redo() > 100
        _check_estep(nat2, ora2, f2, o2, dropout=0.2, seed=9)
        for k in extra:
            monkeypatch.delenv(k)
