"""The oracle's restatement of VocabularyGenerator (oracle/generate_oracle.py, src/generate.rs:12-243) — the checker
of the device path (tests/test_generate_gpu.py) — pinned by the reference's own test case (src/generate.rs:258-284)
and by hand-checkable document frequencies; plus the invariants of generate()."""
import math
import re

from oracle.generate_oracle import OracleVocabularyGenerator, fnv1a64, keep_u01


def _reference_case():
    # src/generate.rs:258-276 (test_generate)
    g = OracleVocabularyGenerator(6, 1.0, None, re.compile(r"^ ?[a-z]+$"), ["goodbye", "vec"], ["string", "map"])
    g.feed(["hello my name is diego and i like std::string", "i also like std::vector",
            "and std::vector<std::string>", "and std::map<int, std::string>"])
    return g


def test_reference_generate_case():
    g = _reference_case()
    vocab = g.generate(256 + 10)
    multi = [t for t in vocab if len(t[0]) > 1]
    assert any(t[0] == b"string" for t in multi)                 # the reference's assertion
    assert len(vocab) == 266 and sum(1 for t in vocab if len(t[0]) == 1) == 255
    assert {bytes([b]) for b in range(255)} <= {t[0] for t in vocab} and bytes([255]) not in {t[0] for t in vocab}
    by = {t[0]: t for t in vocab}
    assert by[b"vec"][2] is True and by[b"string"][2] is False    # added tokens are kept, suggested are not
    assert all(t[2] for t in vocab if len(t[0]) == 1)
    assert abs(sum(math.exp(t[1]) for t in vocab) - 1.0) < 1e-12  # log-probabilities
    assert all(vocab[i][1] >= vocab[i + 1][1] for i in range(len(vocab) - 1))
    # document frequencies by hand: "string" is in samples 0, 2, 3 (+ 1 from the constructor); " like" in 0 and 1
    assert g.frequencies["string"] == 4 and g.frequencies[" like"] == 2 and g.frequencies["vec"] == 3
    assert "std" in g.frequencies and "::" not in g.frequencies   # the allow pattern


def test_document_frequencies_and_limits():
    g = OracleVocabularyGenerator(4, 1.0, None, None)
    g.feed(["abab", "ab", "xyz"])
    assert g.frequencies["ab"] == 2 and g.frequencies["abab"] == 1 and g.frequencies["b"] == 2   # once per sample
    assert "ababa" not in g.frequencies and max(len(k.encode()) for k in g.frequencies) <= 4
    g2 = OracleVocabularyGenerator(4, 1.0, None, None)
    g2.feed(["中文中"])                                                                           # 3-byte chars
    assert set(g2.frequencies) == {"中", "文"}                                                   # 6 bytes > 4: no pairs
    # insert_probability: reproducible, monotone in p
    a = OracleVocabularyGenerator(8, 0.3, None, None, seed=7); a.feed(["the quick brown fox"] * 3)
    b = OracleVocabularyGenerator(8, 0.3, None, None, seed=7); b.feed(["the quick brown fox"] * 3)
    c = OracleVocabularyGenerator(8, 1.0, None, None, seed=7); c.feed(["the quick brown fox"] * 3)
    assert a.frequencies == b.frequencies and 0 < len(a.frequencies) < len(c.frequencies)


def test_keep_rule_is_the_librarys():
    """The seeded stand-in for the reference's thread RNG is one function in the oracle and in the library."""
    from tokengeex_amd import _lib
    assert fnv1a64(b"") == 0xCBF29CE484222325 and fnv1a64(b"a") == 0xAF63DC4C8601EC8C   # FNV-1a test vectors
    for seed, sample, occ in [(0, 0, 1), (7, 123456, (17 << 8) | 5), (2**63 + 5, 99, (1 << 63) | (3 << 8) | 6), (1, 2**26, (2**30 << 8) | 32)]:
        assert keep_u01(seed, sample, occ) == _lib.generate_u01(seed, sample, occ)


def test_one_draw_per_occurrence():
    """src/generate.rs:84-89, 108-113, 122-127 draw inside the loops over positions, lengths and matches: a substring with k
    occurrences in a sample is counted for it with probability 1 - (1 - p)^k, not p.  4 000 one-line samples with "ab"
    eight times each and "xy" once: the counts follow those probabilities (binomial, 5 sigma)."""
    p, n = 0.1, 4000
    g = OracleVocabularyGenerator(2, p, None, None, ["ab"], [], seed=3)  # "ab" is an added token too: the draws of its matches add to the windows'
    h = OracleVocabularyGenerator(2, p, None, None, seed=3)
    samples = ["ab ab ab ab ab ab ab ab xy"] * n
    g.feed(samples)
    h.feed(samples)
    def near(count, prob):
        return abs(count - n * prob) <= 5.0 * (n * prob * (1.0 - prob)) ** 0.5
    assert near(h.frequencies["xy"], p) and near(h.frequencies["ab"], 1.0 - (1.0 - p) ** 8)
    assert near(g.frequencies["ab"] - 1, 1.0 - (1.0 - p) ** 16)  # (- 1: added tokens start at 1, src/generate.rs:33-41)


def test_pass_sizing_keeps_a_device_pass_below_its_window_limit():
    """tokengeex_amd/generate.py: pass_bytes — tgx_substring_df takes fewer than 2^32 kept windows per call, and a byte
    position of ASCII text starts max_token_length of them: the text of one pass is sized by ENCODED bytes accordingly."""
    import importlib.util
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tokengeex_amd", "generate.py")).read()
    ns = {}
    exec(src[src.index("def pass_bytes"):src.index("def _fnv1a64")], ns)   # (importing the package needs the built library)
    for mtl in (1, 8, 16, 24, 32):
        b = ns["pass_bytes"](mtl)
        assert b * mtl < (1 << 32) and b <= (256 << 20) and b >= (64 << 20)
