"""The Tokenizer's front and back over packed buffers (csrc/frontback.cpp) against the per-sample Python
mirror of the reference (tokengeex_amd/tokenizer.py), the oracle's splitter and Python's UTF-8 decoder:
special-token splitting (src/tokenizer.rs:299-347), CRLF (src/processor.rs:46-54), assembly of ids
(src/tokenizer.rs:65-90), decode_batch (src/tokenizer.rs:126-187, src/model.rs:146-160).  No GPU needed."""
import json
import os

import numpy as np
import pytest

import tokengeex_amd as tgx
from tokengeex_amd import _lib


def test_utf8_lossy_matches_pythons_decoder():
    """String::from_utf8_lossy and bytes.decode("utf-8", "replace") both substitute maximal subparts."""
    cases = [b"", b"abc", "你好".encode(), b"\xff", b"\xc3", b"\xc3\x28", b"\xe4\xbd", b"\xe4\xbd\x41", b"\xf0\x9f\x98", b"\xf0\x9f",
             b"\xed\xa0\x80", b"\xe0\x80\x80", b"\xf4\x90\x80\x80", b"\xc0\xaf", b"\xc1\xbf", b"\xf5\x80", b"a\x80\x80b", b"\xe4\xbd\xa0\xe4",
             b"\xf0\x90\x80\x80", b"\xef\xbf\xbd", b"\x80" * 5, b"\xe1\x80\xe1\x80\x80"]
    rng = np.random.default_rng(1)
    for _ in range(3000):
        n = int(rng.integers(1, 24))
        # bytes drawn from the interesting classes: ASCII, continuation, 2/3/4-byte leads, invalid leads
        pool = np.array([0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC2, 0xDF, 0xE0, 0xE1, 0xED, 0xEF, 0xF0, 0xF1, 0xF4, 0xF5, 0xFF], np.uint8)
        cases.append(bytes(pool[rng.integers(0, pool.size, size=n)]))
    for c in cases:
        assert _lib.utf8_lossy(c) == c.decode("utf-8", "replace").encode("utf-8"), c


def _segments(flat, offs, specials):
    seg_offs, sb, se, ss = _lib.split_specials_flat(flat, offs, [s.encode() for s in specials])
    raw = flat.tobytes()
    out = []
    for i in range(offs.size - 1):
        out.append([(raw[int(sb[k]):int(se[k])].decode(), bool(ss[k] >= 0)) for k in range(int(seg_offs[i]), int(seg_offs[i + 1]))])
    return out, (seg_offs, sb, se, ss)


def test_native_splitter_kats_and_random(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json"), encoding="utf-8") as f:
        kats = json.load(f)
    for c in kats["splitter"]["cases"]:
        flat, offs = tgx.pack([c["input"].encode()])
        got, _ = _segments(flat, offs, c["specials"])
        assert got == [[(s, b) for s, b in c["segments"]]]
    from oracle import oracle as orc
    rng = np.random.default_rng(2)
    specials = ["<EOS>", "ab", "<EOS_2>", "你", "b", "<EOS"]
    alphabet = ["a", "b", "<", "EOS", ">", "_2", "你", "好", " ", "<EOS>", "\r\n", "é"]
    texts = ["".join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), size=int(rng.integers(0, 40)))) for _ in range(2000)]
    flat, offs = tgx.pack([t.encode() for t in texts])
    got, (seg_offs, sb, se, ss) = _segments(flat, offs, specials)
    for t, g in zip(texts, got):
        assert g == tgx.split_special_tokens(t, specials)
        assert [(s.encode(), b) for s, b in g] == orc.split_specials(t.encode(), [x.encode() for x in specials])
    # first-listed special wins at one position, not the longest (src/tokenizer.rs:325-338)
    f2, o2 = tgx.pack([b"<EOS_2>"])
    assert _segments(f2, o2, ["<EOS", "<EOS_2>"])[0] == [[("<EOS", True), ("_2>", False)]]
    with pytest.raises(tgx.TokenGeeXError):
        _lib.split_specials_flat(f2, o2, [b""])
    # CRLF normalisation of the non-special segments and the assembly of ids
    pflat, poffs = _lib.pack_segments(flat, sb, se, ss, True)
    want = [s.replace("\r\n", "\n").encode() for g in got for s, sp in g if not sp]
    praw = pflat.tobytes()
    assert [praw[int(poffs[i]):int(poffs[i + 1])] for i in range(poffs.size - 1)] == want
    ids = np.concatenate([np.frombuffer(w, np.uint8).astype(np.uint32) for w in want] + [np.zeros(0, np.uint32)])   # "encode" = bytes
    out, oo = _lib.assemble_ids(seg_offs, ss, ids, poffs, 1000)
    for i, g in enumerate(got):
        exp = []
        for s, sp in g:
            exp += [1000 + specials.index(s)] if sp else list(s.replace("\r\n", "\n").encode())
        assert out[int(oo[i]):int(oo[i + 1])].tolist() == exp


def test_native_decode_batch_matches_the_per_sample_path():
    vocab = [(bytes([i]), -8.0, True) for i in range(256)] + [(b"Hello", -3.0, False), ("你好".encode(), -2.0, False),
                                                                (b"\xe4\xbd", -9.0, False), (b"\xa0", -9.0, False), (b"\xf0\x9f", -9.0, False)]
    tk = tgx.Tokenizer(vocab, [tgx.CrlfProcessor()], ["<EOS>", "<pad>", "é"])
    base = tk.base_vocab_size()
    rng = np.random.default_rng(3)
    rows = [[], [256], [258, 259], [258, base + 0, 259], [base + 1], [base + 2, base + 2], [260, 65, 0x98, 0x80]]
    for _ in range(500):
        n = int(rng.integers(0, 30))
        rows.append([int(x) for x in rng.integers(0, base + 3, size=n)])
    for inc in (True, False):
        assert tk.decode_batch(rows, inc) == [tk.decode(r, inc) for r in rows]
    # truncated character | special | continuation bytes: the runs are lossy-decoded separately
    assert tk.decode_batch([[258, base + 1, 259]], False) == ["��"]
    ids = np.array([x for r in rows for x in r], np.uint32)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
    text, to = tk.decode_batch_flat(ids, offs, True)
    assert text.tobytes()[int(to[3]):int(to[4])].decode() == tk.decode(rows[3], True)
    with pytest.raises(tgx.TokenGeeXError) as e:
        tk.decode_batch([[1], [2, base + 3], [base + 7]], True)
    assert str(e.value) == f"token id {base + 3} is out of bounds" and e.value.sample == 1   # src/lib.rs:246-248
    assert tk.decode_batch([], True) == []


def test_tokengeex_import_alias_has_the_stubs_surface():
    """`from tokengeex import Tokenizer` (bindings/python/example.py:1) with every method of tokengeex.pyi."""
    import tokengeex
    assert tokengeex.Tokenizer is tgx.Tokenizer and issubclass(tokengeex.TokenGeeXError, Exception)
    methods = ["encode", "encode_ordinary", "encode_batch", "encode_ordinary_batch", "decode", "decode_batch", "token_to_id",
               "base_token_to_id", "special_token_to_id", "id_to_token", "id_to_base_token", "id_to_special_token",
               "add_special_tokens", "special_tokens", "is_special", "vocab_size", "base_vocab_size", "special_vocab_size",
               "save", "common_prefix_search", "from_file", "from_str", "is_base", "to_string"]
    for name in methods:
        assert callable(getattr(tokengeex.Tokenizer, name)), name
    tk = tokengeex.Tokenizer.from_str('{"version":"2.0","special_tokens":["<EOS>"],"processors":[{"type":"crlf"}],'
                                      '"vocab":[{"value":"a","score":-1.0},{"value":"b","score":-2.0,"keep":true}]}')
    assert tk.vocab_size() == 3 and tk.decode([0, 1, 2], True) == "ab<EOS>" and tk.base_token_to_id(b"b") == 1


def test_native_ends_of_the_list_surface():
    """csrc/pyfast.c: pack_strs == the samples' str.encode("utf-8") back to back (1-, 2- and 4-byte string kinds, empty
    samples, thread slices), with str.encode's error for a lone surrogate; rows_from_flat == the per-sample lists, with
    and without the shared-int cache, ids beyond the cache as fresh ints."""
    from tokengeex_amd import _tgxfast
    rng = np.random.default_rng(3)
    alphabet = ["a", "Z", " ", "\n", "é", "ß", "中", "文", "，", "😀", "\U0001F600", "\x00", "\x7f", "\x80", "߿", "ࠀ", "￿"]
    texts = ["".join(rng.choice(alphabet, size=int(n))) for n in rng.integers(0, 200, size=3000)] + ["", "x" * 100000, "中" * 50000]
    for threads in (1, 3, 8):
        tb, ob = _tgxfast.pack_strs(texts, threads)
        want = [t.encode("utf-8") for t in texts]
        assert tb == b"".join(want)
        assert np.frombuffer(ob, np.uint64).tolist() == np.concatenate([[0], np.cumsum([len(w) for w in want])]).tolist()
    assert _tgxfast.pack_strs([]) == (b"", (0).to_bytes(8, "little"))
    assert _tgxfast.pack_strs(("ab", "c")) [0] == b"abc"          # any sequence
    with pytest.raises(UnicodeEncodeError):
        _tgxfast.pack_strs(["fine", "bad \ud800 surrogate"])
    with pytest.raises(TypeError):
        _tgxfast.pack_strs(["fine", b"bytes"])
    ids = rng.integers(0, 5000, size=20000).astype(np.uint32)
    cuts = np.sort(rng.integers(0, ids.size + 1, size=300))
    offs = np.concatenate([[0], cuts, [ids.size]]).astype(np.uint64)
    want = [ids[int(offs[i]):int(offs[i + 1])].tolist() for i in range(offs.size - 1)]
    cache = list(range(4000))                                      # ids 4000 .. 4999 lie beyond it
    assert _tgxfast.rows_from_flat(ids, offs, cache) == want
    assert _tgxfast.rows_from_flat(ids, offs, None) == want
    rows = _tgxfast.rows_from_flat(ids, offs, cache)
    k = int(np.flatnonzero(ids < 4000)[0]); r = int(np.searchsorted(offs, k, side="right") - 1)
    assert rows[r][k - int(offs[r])] is cache[int(ids[k])]         # the shared object
    assert _tgxfast.rows_from_flat(np.zeros(0, np.uint32), np.zeros(1, np.uint64), None) == []
    with pytest.raises(ValueError):
        _tgxfast.rows_from_flat(ids, np.array([0, ids.size + 1], np.uint64), None)
