"""VocabularyGenerator::feed on the device (csrc/generate.hip: tgx_substring_df) against the per-sample host
restatement of src/generate.rs:54-139 (device=None): identical document frequencies — with and without a split
regex, an allow regex, added / suggested tokens, insert_probability < 1, multi-byte characters, several feed
calls — and identical generate() output up to the order of equal scores."""
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tokengeex_amd import _lib, synth
from tokengeex_amd.generate import VocabularyGenerator

from test_merge_cpu import ALLOW


def _samples(n_bytes, kind="mixed", max_len=1500, seed_offset=0):
    flat, offs = synth.make_corpus(n_bytes, kind, max_len=max_len, seed_offset=seed_offset)
    o = offs.astype(np.int64)
    return [flat[o[i]:o[i + 1]].tobytes().decode("utf-8") for i in range(o.size - 1)]


def _both(samples, *args, chunks=1, **kw):
    dev, host = VocabularyGenerator(*args, device=0, **kw), VocabularyGenerator(*args, device=None, **kw)
    step = (len(samples) + chunks - 1) // chunks
    for a in range(0, len(samples), step):
        dev.feed(samples[a:a + step])
        host.feed(samples[a:a + step])
    return dev, host


def test_reference_case_on_the_device():
    # src/generate.rs:258-276 (test_generate)
    dev, host = _both(["hello my name is diego and i like std::string", "i also like std::vector",
                       "and std::vector<std::string>", "and std::map<int, std::string>"],
                      6, 1.0, None, r"^ ?[a-z]+$", ["goodbye", "vec"], ["string", "map"])
    assert dev.frequencies == host.frequencies
    assert any(t[0] == b"string" for t in dev.generate(266))


def test_document_frequencies_equal_the_host_restatement():
    samples = _samples(160 << 10) + ["", "a", "中文中文", "é" * 9, "x" * 40, "\n\n    return", "𝒳𝒴 math"]
    dev, host = _both(samples, 16, 1.0, None, None)
    assert dev.frequencies == host.frequencies and len(dev.frequencies) > 50000
    dev, host = _both(samples, 12, 1.0, None, ALLOW, chunks=3)                    # allow regex, three feed calls
    assert dev.frequencies == host.frequencies
    split = re.compile(r"[A-Za-z_]+|[0-9]+|\s+|[^\sA-Za-z_0-9]+")
    dev, host = _both(samples, 8, 1.0, split, None, ["std::", "return"], ["    ", "中文"])
    assert dev.frequencies == host.frequencies                                     # split parts + added / suggested
    dev, host = _both(samples, 10, 0.05, None, None, ["def"], [], chunks=2, seed=11)
    assert dev.frequencies == host.frequencies and 0 < len(dev.frequencies)        # the seeded keep rule
    assert dev.generate(3000) == host.generate(3000)                                # same tokens, scores, order


def test_substring_df_raw_interface():
    flat, offs = _lib.pack([b"abab", b"ab", b"", b"xyz"])
    keep = np.flatnonzero(offs[1:] > offs[:-1])
    pos, ln, df, n_windows = _lib.substring_df(flat, offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32), 4)
    got = {flat[int(p):int(p + l)].tobytes(): int(d) for p, l, d in zip(pos, ln, df)}
    assert got == {b"a": 2, b"b": 2, b"ab": 2, b"ba": 1, b"aba": 1, b"bab": 1, b"abab": 1, b"x": 1, b"y": 1, b"z": 1,
                   b"xy": 1, b"yz": 1, b"xyz": 1}
    assert n_windows == 10 + 3 + 6
    with pytest.raises(Exception):
        _lib.substring_df(flat, offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32), 17)
