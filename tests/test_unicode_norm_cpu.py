"""The native Unicode normalisation forms (csrc/unicode_norm.cpp; UnicodeProcessor::preprocess, src/processor.rs:124-137)
against `unicodedata.normalize` — the module the tables were generated from (tools/make_unicode_tables.py): every code
point alone and between ASCII letters, random sequences rich in combining marks, Hangul, segments with bytes that are not
UTF-8, a batch large enough for the threaded path."""
import unicodedata as ud

import numpy as np
import pytest

from tokengeex_amd import _lib

FORMS = ("nfd", "nfc", "nfkd", "nfkc")


def _norm(form, texts):
    flat, offs = _lib.pack([t if isinstance(t, bytes) else t.encode("utf-8") for t in texts])
    f, o = _lib.normalize_flat(form, flat, offs)
    assert o.shape[0] == len(texts) + 1 and int(o[0]) == 0
    return [bytes(f[int(o[i]):int(o[i + 1])]) for i in range(len(texts))]


def test_the_tables_are_those_of_this_interpreter():
    assert _lib.unidata_version() == ud.unidata_version


@pytest.mark.parametrize("form", FORMS)
def test_every_code_point(form):
    cps = [cp for cp in range(0x110000) if not 0xD800 <= cp <= 0xDFFF]
    for wrap in ("{}", "a{}b", "e{}́"):
        texts = [wrap.format(chr(cp)) for cp in cps]
        got = _norm(form, texts)
        want = [ud.normalize(form.upper(), t).encode("utf-8") for t in texts]
        bad = [i for i in range(len(texts)) if got[i] != want[i]]
        assert not bad, (form, wrap, [hex(cps[i]) for i in bad[:10]])


@pytest.mark.parametrize("form", FORMS)
def test_random_sequences_with_combining_marks(form):
    rng = np.random.default_rng(7)
    marks = [cp for cp in range(0x300, 0x1E8D7) if ud.combining(chr(cp))]
    starters = list(range(0x41, 0x5B)) + list(range(0xC0, 0x250)) + list(range(0x391, 0x3CA)) + list(range(0x1100, 0x1113)) + \
        list(range(0x1161, 0x1176)) + list(range(0x11A8, 0x11C3)) + list(range(0xAC00, 0xAC00 + 600)) + [0x0F71, 0x0F72, 0x0F74, 0x1100, 0x09C7, 0x09BE, 0x0B47, 0x0B56]
    compat = [cp for cp in range(0xA0, 0x30000) if ud.decomposition(chr(cp)).startswith("<")][::7]
    texts = []
    for _ in range(20000):
        n = int(rng.integers(1, 12))
        s = []
        for _ in range(n):
            r = rng.random()
            pool = marks if r < 0.45 else starters if r < 0.85 else compat
            s.append(chr(pool[int(rng.integers(len(pool)))]))
        texts.append("".join(s))
    got = _norm(form, texts)
    want = [ud.normalize(form.upper(), t).encode("utf-8") for t in texts]
    bad = [i for i in range(len(texts)) if got[i] != want[i]]
    assert not bad, (form, [texts[i].encode("unicode_escape") for i in bad[:5]])


def test_bytes_that_are_not_utf8_pass_through():
    raw = [b"\xff\xfea\xcc\x81", b"\xe4\xb8", b"abc\x80\x80def", b"\xf5\x80\x80\x80", b"\xed\xa0\x80x", b"e\xcc\x81\xc0\xaf"]
    for form in FORMS:
        got = _norm(form, raw)
        for g, r in zip(got, raw):
            # the well-formed stretches are normalised, the offending bytes stay where they were
            want = b"".join(ud.normalize(form.upper(), part.decode("utf-8")).encode("utf-8") if ok else part for ok, part in _split_valid(r))
            assert g == want, (form, r, g, want)


def _split_valid(b: bytes):
    """[(is_valid_utf8, bytes)] with maximal valid stretches (an invalid byte is a part of its own)."""
    out, i = [], 0
    while i < len(b):
        for j in range(len(b), i, -1):
            try:
                b[i:j].decode("utf-8")
                out.append((True, b[i:j]))
                i = j
                break
            except UnicodeDecodeError:
                continue
        else:
            out.append((False, b[i:i + 1]))
            i += 1
    return out


def test_a_large_batch_runs_on_threads_and_keeps_the_segment_order():
    base = ["def f(x):\n    return x  # café ＡＢ", "中文 ẛ̣", "", "plain ascii " * 40, "각" * 30]
    texts = [base[i % len(base)] + str(i) for i in range(60000)]
    for form in ("nfc", "nfkd"):
        got = _norm(form, texts)
        assert got == [ud.normalize(form.upper(), t).encode("utf-8") for t in texts]
