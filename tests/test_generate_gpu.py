"""VocabularyGenerator::feed / generate on the device path (tokengeex_amd/generate.py over csrc/generate.hip:
tgx_substring_df[_top]) against the oracle's restatement of src/generate.rs:54-243 (oracle/generate_oracle.py):
identical document frequencies — with and without a split regex, an allow regex, added / suggested tokens,
insert_probability < 1, multi-byte characters, several feed calls — and identical generate() output, also when only
the top_k most frequent substrings leave the device."""
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import tokengeex_amd as tgx
from oracle.generate_oracle import OracleVocabularyGenerator
from tokengeex_amd import _lib, synth
from tokengeex_amd.generate import VocabularyGenerator
from test_merge_cpu import ALLOW
from util import AllowByHand


def _samples(n_bytes, kind="mixed", max_len=1500, seed_offset=0):
    flat, offs = synth.make_corpus(n_bytes, kind, max_len=max_len, seed_offset=seed_offset)
    o = offs.astype(np.int64)
    return [flat[o[i]:o[i + 1]].tobytes().decode("utf-8") for i in range(o.size - 1)]


def _both(samples, mtl, p, split, allow, added=(), suggested=(), chunks=1, seed=0, top_k=None):
    # the checker's allow pattern never goes through the product's translation: ALLOW is evaluated by hand, the
    # reference's own test pattern by Python's re (plain syntax, `$` made explicit)
    pat = AllowByHand() if allow == ALLOW else (re.compile(allow.replace('$', r'\Z')) if isinstance(allow, str) else allow)
    dev = VocabularyGenerator(mtl, p, split, allow, added, suggested, seed=seed, top_k=top_k)
    ora = OracleVocabularyGenerator(mtl, p, split, pat, added, suggested, seed=seed)
    step = (len(samples) + chunks - 1) // chunks
    for a in range(0, len(samples), step):
        dev.feed(samples[a:a + step])
        ora.feed(samples[a:a + step])
    return dev, ora


def test_reference_case_on_the_device():
    # src/generate.rs:258-276 (test_generate)
    dev, ora = _both(["hello my name is diego and i like std::string", "i also like std::vector",
                      "and std::vector<std::string>", "and std::map<int, std::string>"],
                     6, 1.0, None, r"^ ?[a-z]+$", ["goodbye", "vec"], ["string", "map"])
    assert dev.frequencies == ora.frequencies
    assert any(t[0] == b"string" for t in dev.generate(266)) and dev.generate(266) == ora.generate(266)


def test_document_frequencies_equal_the_oracle():
    samples = _samples(160 << 10) + ["", "a", "中文中文", "é" * 9, "x" * 40, "\n\n    return", "𝒳𝒴 math"]
    dev, ora = _both(samples, 16, 1.0, None, None)
    assert dev.frequencies == ora.frequencies and len(dev.frequencies) > 50000
    dev, ora = _both(samples, 12, 1.0, None, ALLOW, chunks=3)                     # allow regex, three feed calls
    assert dev.frequencies == ora.frequencies
    split = re.compile(r"[A-Za-z_]+|[0-9]+|\s+|[^\sA-Za-z_0-9]+")
    dev, ora = _both(samples, 8, 1.0, split, None, ["std::", "return"], ["    ", "中文"])
    assert dev.frequencies == ora.frequencies                                      # split parts + added / suggested
    dev, ora = _both(samples, 10, 0.05, None, None, ["def"], [], chunks=2, seed=11)
    assert dev.frequencies == ora.frequencies and 0 < len(dev.frequencies)         # the seeded keep rule
    assert dev.generate(3000) == ora.generate(3000)                                 # same tokens, scores, order


def test_top_k_leaves_the_vocabulary_exact_or_refuses():
    """Only the top_k most frequent substrings of a device pass reach the host.  With top_k well above the vocabulary
    size the generated vocabulary equals the oracle's (which counts everything); with top_k too small the selection
    would depend on substrings that were cut off, and generate() refuses instead of returning something else."""
    samples = _samples(512 << 10, seed_offset=5)
    dev, ora = _both(samples, 16, 1.0, None, ALLOW, ["std::"], ["    "], top_k=60000)
    want = ora.generate(4000)
    assert dev.generate(4000) == want and dev.passes == 1
    assert len(dev.frequencies) < len(ora.frequencies) // 4                        # far fewer strings crossed the link
    dev2, _ = _both(samples, 16, 1.0, None, ALLOW, top_k=3000)
    with pytest.raises(tgx.TokenGeeXError, match="top_k too small"):
        dev2.generate(4000)
    # several device passes (forced: 64 KiB of text per pass): frequencies add, every pass reports what it cut off
    import tokengeex_amd.generate as gen_mod
    old = gen_mod.pass_bytes
    gen_mod.pass_bytes = lambda mtl: 64 << 10
    try:
        dev3, _ = _both(samples, 16, 1.0, None, ALLOW, ["std::"], ["    "], top_k=1000000)   # nothing is cut in a 64 KiB pass
        assert dev3.generate(1500) == ora.generate(1500) and dev3.passes > 4
        # ... and when every small pass cuts something, the bounds add up: the result is the oracle's or a refusal,
        # never a different vocabulary
        dev4, _ = _both(samples, 16, 1.0, None, ALLOW, ["std::"], ["    "], top_k=150000)
        try:
            assert dev4.generate(1500) == ora.generate(1500)
        except tgx.TokenGeeXError as e:
            assert "top_k too small" in str(e)
    finally:
        gen_mod.pass_bytes = old


@pytest.mark.parametrize("mtl", [20, 24, 32])
def test_windows_of_up_to_32_bytes(mtl):
    """max_token_length beyond 16 (the reference's CLI default is 24, src/cli.rs:675; src/generate.rs:78-83 has no limit):
    frequencies and vocabulary equal to the oracle's, with multi-byte characters that end exactly at the limit."""
    samples = _samples(96 << 10, seed_offset=9) + ["中文字符" * 12, "x" * 70, "é" * 40, "ab" * 33, "𝒳" * 9]
    dev, ora = _both(samples, mtl, 1.0, None, None)
    assert dev.frequencies == ora.frequencies
    assert max(len(k.encode("utf-8")) for k in dev.frequencies) == mtl
    dev, ora = _both(samples, mtl, 0.3, None, ALLOW, ["std::"], ["    "], chunks=2, seed=4)
    assert dev.frequencies == ora.frequencies and dev.generate(2000) == ora.generate(2000)


def test_key_collisions_are_resolved(monkeypatch):
    """Two different substrings in one sort key: every entry of a run is compared with the run's first, byte by byte,
    and a pass with any such entry is sorted again under a second, independent hash.  Forced here by keeping 12 bits
    of the first attempt's keys (TGX_GENERATE_COLLIDE): the counts are the oracle's all the same, and the call reports
    how many entries had met a foreign run."""
    monkeypatch.setenv("TGX_GENERATE_COLLIDE", "12")
    samples = _samples(64 << 10, seed_offset=3) + ["中文中文", "abcabc"]
    dev, ora = _both(samples, 12, 1.0, None, None)
    assert dev.frequencies == ora.frequencies and len(dev.frequencies) > 10000
    flat, offs = _lib.pack([b"abab", b"ab", b"", b"xyz"])
    keep = np.flatnonzero(offs[1:] > offs[:-1])
    monkeypatch.setenv("TGX_GENERATE_COLLIDE", "2")
    pos, ln, df, n_windows, collisions = _lib.substring_df(flat, offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32), 4, with_collisions=True)
    got = {flat[int(p):int(p + l)].tobytes(): int(d) for p, l, d in zip(pos, ln, df)}
    assert got[b"ab"] == 2 and got[b"abab"] == 1 and len(got) == 13 and collisions > 0


def test_substring_df_raw_interface():
    flat, offs = _lib.pack([b"abab", b"ab", b"", b"xyz"])
    keep = np.flatnonzero(offs[1:] > offs[:-1])
    pos, ln, df, n_windows = _lib.substring_df(flat, offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32), 4)
    got = {flat[int(p):int(p + l)].tobytes(): int(d) for p, l, d in zip(pos, ln, df)}
    assert got == {b"a": 2, b"b": 2, b"ab": 2, b"ba": 1, b"aba": 1, b"bab": 1, b"abab": 1, b"x": 1, b"y": 1, b"z": 1,
                   b"xy": 1, b"yz": 1, b"xyz": 1}
    assert n_windows == 10 + 3 + 6
    with pytest.raises(Exception):
        _lib.substring_df(flat, offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32), 33)
    # top 3: the three substrings that occur in two samples, and the cut-off frequency 1
    pos, ln, df, _, n_distinct, cut = _lib.substring_df_top(flat, offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32), 4, 3)
    assert {flat[int(p):int(p + l)].tobytes() for p, l in zip(pos, ln)} == {b"a", b"b", b"ab"} and df.tolist() == [2, 2, 2]
    assert n_distinct == 13 and cut == 1
    pos, ln, df, _, n_distinct, cut = _lib.substring_df_top(flat, offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32), 4, 100)
    assert pos.size == 13 and cut == 0
