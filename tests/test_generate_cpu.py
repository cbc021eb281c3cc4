"""Host mirror of VocabularyGenerator (src/generate.rs): the reference's own test case plus the invariants of
generate()."""
import math

from tokengeex_amd.generate import VocabularyGenerator


def _reference_case():
    # src/generate.rs:258-276 (test_generate)
    g = VocabularyGenerator(6, 1.0, None, r"^ ?[a-z]+$", ["goodbye", "vec"], ["string", "map"])
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


def test_document_frequencies_and_limits():
    g = VocabularyGenerator(4, 1.0, None, None)
    g.feed(["abab", "ab", "xyz"])
    assert g.frequencies["ab"] == 2 and g.frequencies["abab"] == 1 and g.frequencies["b"] == 2   # once per sample
    assert "ababa" not in g.frequencies and max(len(k.encode()) for k in g.frequencies) <= 4
    g2 = VocabularyGenerator(4, 1.0, None, None)
    g2.feed(["中文中"])                                                                           # 3-byte chars
    assert set(g2.frequencies) == {"中", "文"}                                                   # 6 bytes > 4: no pairs
    # insert_probability: reproducible, monotone in p
    a = VocabularyGenerator(8, 0.3, None, None, seed=7); a.feed(["the quick brown fox"] * 3)
    b = VocabularyGenerator(8, 0.3, None, None, seed=7); b.feed(["the quick brown fox"] * 3)
    c = VocabularyGenerator(8, 1.0, None, None, seed=7); c.feed(["the quick brown fox"] * 3)
    assert a.frequencies == b.frequencies and 0 < len(a.frequencies) < len(c.frequencies)
