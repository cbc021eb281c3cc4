"""Pins the CPU oracle (oracle/tgx_oracle.c) to every known-answer vector the
reference's own tests hold for the hot path."""
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as orc


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json"), encoding="utf-8") as f:
        return json.load(f)


def _model(vocab):
    return orc.OracleModel([t.encode("utf-8") for t, _ in vocab], [s for _, s in vocab])


def test_encode_kats(kats):
    for k in kats["encode"]:
        m = _model(k["vocab"])
        assert m.encode(k["input"].encode(), k["dropout"], seed=1234) == k["ids"], k["source"]


def test_encode_dropout_zero_takes_whole_token(kats):
    # same code path as model.rs:217-236 with dropout 0.0 -> the single 6-byte token (id 10)
    k = kats["encode"][1]
    assert _model(k["vocab"]).encode(k["input"].encode(), 0.0) == [10]


def test_default_vocab_roundtrip(kats):
    k = kats["default_vocab_roundtrip"]
    toks = [bytes([i]) for i in range(256)]
    m = orc.OracleModel(toks, [1.0 / 256.0] * 256)
    raw = k["input"].encode("utf-8")
    ids = m.encode(raw)
    assert len(ids) == k["n_ids"] == len(raw)
    assert b"".join(toks[i] for i in ids).decode("utf-8") == k["input"]


def test_marginal_kat(kats):
    k = kats["marginal"]
    m = _model(k["vocab"])
    expected, z = m.marginal(k["input"].encode())
    names = [t for t, _ in k["vocab"]]
    for name, want in k["expected"].items():
        assert abs(expected[names.index(name)] - want) < 5e-7, name
    # three paths with log-probs -12, -13, -14
    assert abs(z - (-12.0 + math.log(1.0 + math.exp(-1.0) + math.exp(-2.0)))) < 1e-12


def test_splitter_kats(kats):
    for c in kats["splitter"]["cases"]:
        got = orc.split_specials(c["input"].encode(), [s.encode() for s in c["specials"]])
        assert got == [(s.encode(), b) for s, b in c["segments"]], c["input"]


# ---- behaviours from SURVEY.md Appendix A, each established by reference code ----

def test_longest_token_wins_ties():
    # model.rs:100-101 strict '>' with ascending starts: a:-3,b:-3,ab:-6 on "ab" -> [ab]
    m = orc.OracleModel([b"a", b"b", b"ab"], [-3.0, -3.0, -6.0])
    assert m.encode(b"ab") == [2]


def test_duplicate_token_last_id_wins():
    # trie.rs:19 overwrites data
    m = orc.OracleModel([b"a", b"ab", b"ab", b"b"], [-1.0, -1.0, -1.0, -1.0])
    assert m.encode(b"ab") == [2]
    assert m.common_prefix_search(b"abz") == [(0, 1), (2, 2)]


def test_empty_token_never_matches_and_empty_input():
    m = orc.OracleModel([b"", b"a"], [-1.0, -2.0])
    assert m.encode(b"aa") == [1, 1]
    assert m.encode(b"") == []


def test_no_path():
    m = orc.OracleModel([b"a", b"b"], [-1.0, -1.0])
    with pytest.raises(orc.NoPath) as e:
        m.encode(b"abc")
    assert str(e.value) == "no path to position 3/3"  # lib.rs:243-245
    # an unmatched byte in the middle makes everything after it unreachable
    with pytest.raises(orc.NoPath):
        m.encode(b"acb")


def test_positive_scores_and_byte_tokens():
    # lib.rs:206-210 default vocab has positive scores; invalid-UTF-8 bytes are plain tokens
    toks = [bytes([i]) for i in range(256)] + [b"\xff\xfe"]
    m = orc.OracleModel(toks, [1.0 / 256.0] * 256 + [5.0])
    assert m.encode(b"\xff\xfe\x00") == [256, 0]


def test_common_prefix_search_stops_at_missing_child():
    m = orc.OracleModel([b"a", b"abc", b"abcde"], [-1.0, -1.0, -1.0])
    assert m.common_prefix_search(b"abcdx") == [(0, 1), (1, 3)]
    assert m.common_prefix_search(b"xbc") == []


def test_marginals_sum_and_quirk_free_case():
    # with all single bytes present every position has end nodes; expected mass of
    # tokens covering any byte position sums to 1
    toks = [b"a", b"b", b"ab", b"ba", b"aba"]
    m = orc.OracleModel(toks, [-1.0, -1.5, -1.7, -2.0, -2.2])
    text = b"abaabab"
    expected, z = m.marginal(text)
    mass = sum(expected[i] * len(toks[i]) for i in range(len(toks)))
    assert abs(mass - len(text)) < 1e-9
    assert z < 0


def test_estep_snippets_are_independent():
    toks = [b"a", b"b", b"ab"]
    m = orc.OracleModel(toks, [-1.0, -1.0, -1.5])
    flat, offs = orc.pack([b"abab", b"ab"])
    st, ex, z, _ = m.estep_flat(flat, offs, snippet_len=3)  # "aba","b","ab"
    e1, z1 = m.marginal(b"aba")
    e2, z2 = m.marginal(b"b")
    e3, z3 = m.marginal(b"ab")
    assert st == orc.OK
    np.testing.assert_allclose(ex, e1 + e2 + e3, rtol=1e-15)
    assert abs(z - (z1 + z2 + z3)) < 1e-12


def test_count_tokens_and_pairs_follow_encode():
    toks = [b"a", b"b", b"c", b"ab"]
    m = orc.OracleModel(toks, [-3.0, -3.0, -3.0, -4.0])
    texts = [b"abc", b"abab", b"c", b""]
    flat, offs = orc.pack(texts)
    freq = m.count_tokens_flat(flat, offs, threads=2)
    assert freq.tolist() == [0, 0, 2, 3]
    keys, counts = m.count_pairs_flat(flat, offs, threads=2)
    assert {(int(k) >> 32, int(k) & 0xFFFFFFFF): int(c) for k, c in zip(keys, counts)} == {(3, 2): 1, (3, 3): 1}


def test_batch_threads_agree_and_lowest_error_reported():
    toks = [bytes([i]) for i in range(97, 123)] + [b"th", b"he", b"the", b"in"]
    rng = np.random.default_rng(7)
    scores = -rng.random(len(toks)) * 5 - 1
    m = orc.OracleModel(toks, scores)
    texts = [bytes(rng.integers(97, 123, size=int(rng.integers(0, 200))).astype(np.uint8)) for _ in range(64)]
    a = m.encode_batch(texts, threads=1)
    b = m.encode_batch(texts, threads=4)
    assert a == b == [m.encode(t) for t in texts]
    texts[10] = b"ab!"
    texts[40] = b"!"
    with pytest.raises(orc.NoPath) as e:
        m.encode_batch(texts, threads=4)
    assert e.value.sample == 10


def test_dropout_is_deterministic_and_only_hits_multibyte():
    toks = [b"a", b"b", b"ab"]
    m = orc.OracleModel(toks, [-3.0, -3.0, -1.0])
    assert m.encode(b"abab", 1.0, seed=5) == [0, 1, 0, 1]
    x = m.encode(b"ab" * 200, 0.5, seed=5)
    assert x == m.encode(b"ab" * 200, 0.5, seed=5)
    assert 2 in x and 0 in x
    u = orc.dropout_u01(1, 2, 3, 4)
    assert 0.0 <= u < 1.0 and u == orc.dropout_u01(1, 2, 3, 4)


def test_hf_unigram_ascii_crosscheck(golden_dir):
    """Independent cross-check: HF `tokenizers` Unigram ids on ASCII."""
    with open(os.path.join(golden_dir, "hf_ascii.json"), encoding="utf-8") as f:
        g = json.load(f)
    n = 0
    for case in g["cases"]:
        m = _model(case["vocab"])
        for text, ids in zip(case["texts"], case["ids"]):
            assert m.encode(text.encode("ascii")) == ids
            n += 1
    assert n >= 20
