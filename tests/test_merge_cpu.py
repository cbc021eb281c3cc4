"""Host half of `merge` (src/merge.rs:78-135): allow-regex translation and candidate selection."""
import os

import numpy as np
import pytest

from tokengeex_amd.merge import ModelVocabularyMerger, compile_rust_regex

# data/exact.regex of the reference is `build_allow_regex` over nine patterns (src/regex.rs:413-425,
# README.md:49-50); this is the same construction over four of them, written out here as test data
ALLOW = r"^(?:.)$|^(?:[a-z]+)$|^(?:[A-Z][a-z]+)$|^(?:[㐀-䶿一-鿿]+)$|^(?: ?[[:punct:]] ?)$"


def test_rust_regex_semantics():
    r = compile_rust_regex(ALLOW)
    yes = ["a", "ab", "Ab", "中文", "!", " ! ", "é", "1", " ", "\t"]
    no = ["aB", "a1", "ab\n", "\n", "", "a b", " é ", "AB"]
    for s in yes:
        assert r.search(s), s
    for s in no:
        assert not r.search(s), s
    # `$` is the end of the text in Rust, never "before a final newline"
    assert compile_rust_regex("^a$").search("a\n") is None
    assert compile_rust_regex(r"^a\z").search("a")
    # POSIX classes are ASCII: a Unicode punctuation mark is not [[:punct:]]
    assert compile_rust_regex("^[[:punct:]]$").search("，") is None
    assert compile_rust_regex(r"^[\u{4E00}-\u{9FFF}]+$").search("中")
    assert compile_rust_regex(r"^[[:alpha:][:digit:]_]+$").search("a_1")
    assert compile_rust_regex(r"^\$$").search("$")
    with pytest.raises(ValueError):
        compile_rust_regex(r"\p{L}+")


def test_allow_by_hand_agrees_with_the_translation():
    """util.AllowByHand (the checker's evaluation of ALLOW, no regex engine) against compile_rust_regex(ALLOW)."""
    from util import AllowByHand
    hand, rx = AllowByHand(), compile_rust_regex(ALLOW)
    cases = ["a", "ab", "Ab", "中文", "!", " ! ", "é", "1", " ", "\t", "aB", "a1", "ab\n", "\n", "", "a b", " é ", "AB",
             " !", "! ", "  !", "!!", " ", "  ", "，", "A", "Abc", "ABc", "中a", "~", " ~ ", "a ", " a"]
    for s in cases:
        assert bool(hand.search(s)) == bool(rx.search(s)), repr(s)


def test_reference_allow_file_and_examples():
    """tests/golden/exact.regex is a byte copy of the reference's data/exact.regex (build_allow_regex over nine named
    patterns, src/regex.rs:413-425, README.md:49-50); tests/golden/exact_regex_examples.json holds the examples the
    reference's own test_regexes checks for those patterns (src/regex.rs:178-411, 449-480).  Every accepted example
    must match the union; plus cases of the Rust semantics that Python's re does not share."""
    import json
    from tokengeex_amd.merge import load_regex
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    r = load_regex(os.path.join(gold, "exact.regex"))
    with open(os.path.join(gold, "exact_regex_examples.json"), encoding="utf-8") as f:
        doc = json.load(f)
    assert len(doc["patterns"]) == 8
    for name, ex in doc["patterns"].items():
        for s in ex["accept"]:
            assert r.search(s), (name, s)
    for s in ["'re", " !==", " + ", "...", "  ", "\t\t", "HELLO", "Hello", "()", " [] ", ":="]:
        assert r.search(s), s
    for s in ["heLLo", " \t", "a+", "ab\n", "\t ", "HeLlO", "مرحبا", "123", " WORLD", "'rex"]:
        assert not r.search(s), s


def test_select_order_limits_and_ignore():
    vocab = [(b"a", -1.0, True), (b"b", -2.0, True), (b"c", -3.0, True), (b"ab", -2.5, False), (b"1", -4.0, True)]
    m = ModelVocabularyMerger(ALLOW, num_merges=10, step=2, scale_factor=0.9, max_token_length=3)
    key = lambda a, b: (a << 32) | b
    keys = np.array(sorted([key(0, 1), key(1, 2), key(0, 4), key(3, 2), key(3, 3), key(2, 0)]), np.uint64)
    counts = {key(0, 1): 50, key(1, 2): 70, key(0, 4): 90, key(3, 2): 70, key(3, 3): 99, key(2, 0): 10}
    cnt = np.array([counts[int(k)] for k in keys], np.uint64)
    ignore = set()
    new = m.select(vocab, keys, cnt, 2, ignore)
    # (3,3) "abab" is too long, (0,4) "a1" fails the regex; of the two 70s the smaller key (1,2) goes first
    assert [t[0] for t in new] == [b"bc", b"abc"]
    assert new[0][1] == (-2.0 + -3.0) * 0.9 and new[0][2] is False
    assert ignore == {key(3, 3), key(0, 4)}
    # budget 0 takes nothing; remembered rejections are skipped without touching the regex
    assert m.select(vocab, keys, cnt, 0, ignore) == []
    m.allow = None
    assert [t[0] for t in m.select(vocab, keys[[1, 5]], cnt[[1, 5]], 5, ignore)] == []
