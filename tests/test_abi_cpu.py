"""CPU-side checks: the C-ABI library loads and exports every symbol declared in
include/tgx.h, fails loudly without a GPU, and the host glue mirrors the
reference's Tokenizer semantics."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest

import tokengeex_amd as tgx
from tokengeex_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "tgx.h"), encoding="utf-8") as f:
        hdr = f.read()
    declared = set(re.findall(r"\b(tgx_[a-z_0-9]+)\s*\(", hdr)) - {"tgx_dropout_u01"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(so, name) is not None
    assert so.tgx_abi_version() == 1


def test_fails_loudly_without_gpu():
    if tgx.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(tgx.TokenGeeXError) as e:
        tgx.NativeModel([b"a"], [0.0])
    assert e.value.status == _lib.ERR_DEVICE
    tk = tgx.Tokenizer([(b"a", -1.0, False)])
    with pytest.raises(tgx.TokenGeeXError):
        tk.encode("a", 0.0)
    with pytest.raises(tgx.TokenGeeXError):  # page-locked host memory is the HIP runtime's too
        _lib.pinned_empty(16, np.uint8)
    from tokengeex_amd.generate import VocabularyGenerator
    with pytest.raises(tgx.TokenGeeXError):  # generate's substring counting has no host fallback either
        VocabularyGenerator(8, 1.0).feed(["abc abc"])


def test_packed_vocabulary_container():
    """_lib.Packed: the prune driver's vocabulary between passes — subsets without a Python loop."""
    rng = np.random.default_rng(3)
    items = [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in rng.integers(0, 20, 500)]
    p = _lib.Packed.of(items)
    assert len(p) == 500 and p.tolist() == items and _lib.Packed.of(p) is p
    idx = rng.integers(0, 500, 300)
    assert p.take(idx).tolist() == [items[int(i)] for i in idx]
    assert p.take([]).tolist() == [] and _lib.Packed.of([]).tolist() == []
    flat, offs = _lib.pack(p.take(idx))
    assert flat.dtype == np.uint8 and offs.dtype == np.uint64 and int(offs[-1]) == flat.size


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "tokengeex_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                with open(os.path.join(dirpath, fn), encoding="utf-8") as f:
                    src = f.read()
                assert "oracle" not in src.replace("CPU oracle", "").replace("the oracle", ""), fn


def test_dropout_hash_matches_oracle():
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = [int(x) for x in rng.integers(0, 2**63, size=3)]
        l = int(rng.integers(1, 65))
        assert _lib.dropout_u01(a[0], a[1], a[2], l) == orc.dropout_u01(a[0], a[1], a[2], l)


def test_flat_trie_matches_oracle_prefix_search():
    """The XOR double-array (host twin of the device walk) against the oracle's
    hash-map trie on every suffix of real text, incl. duplicates and empty tokens."""
    from oracle import oracle as orc
    rng = np.random.default_rng(11)
    flat, _ = synth.make_corpus(128 << 10, "mixed")
    text = bytes(flat)
    for n_multi, max_len in [(50, 4), (3000, 16), (20000, 24), (500, 64)]:
        toks, scores = synth.random_vocab(rng, text, n_multi, max_len)
        toks = toks + [toks[5], b"", toks[17]]  # duplicates + empty token
        scores = np.concatenate([scores, [-1.0, -2.0, -3.0]])
        ft = _lib.FlatTrie(toks, scores)
        ora = orc.OracleModel(toks, scores)
        assert ft.max_token_len == max(len(t) for t in toks)
        for _ in range(3000):
            i = int(rng.integers(0, len(text) - 1))
            s = text[i:i + 80]
            assert ft.common_prefix_search(s) == ora.common_prefix_search(s)
        st = ft.stats()
        # (block 0 holds the root alone — build_trie8's leaves point into it — so tiny tables are half empty)
        assert st["n_slots"] % 256 == 0 and (st["fill"] > 0.5 or st["n_slots"] <= 1024), st


def test_label_checked_records_match_the_oracle_trie():
    """The 8-byte records encode5_kernel walks (label check only, unique bases, leaves and unused slots that can
    never pass) against the oracle's hash-map trie: random text incl. bytes 0xFE / 0xFF and tokens that contain
    them, dense single-child chains, duplicates; and the score table: distinct values, hottest first."""
    from oracle import oracle as orc
    rng = np.random.default_rng(12)
    flat, _ = synth.make_corpus(128 << 10, "mixed")
    text = bytes(flat)
    raw = bytes(rng.integers(0, 256, size=4096, dtype=np.uint8)) + b"\xff\xfe" * 50 + b"\xfe" * 40 + b"\xff" * 40
    for n_multi, max_len, src in [(50, 4, text), (3000, 16, text), (20000, 16, text), (3000, 8, raw)]:
        toks, scores = synth.random_vocab(rng, src, n_multi, max_len)
        toks = toks + [toks[5], b"", toks[17], b"\xff\xfe", b"\xfe\xfe\xfe", b"\xff" * 5]
        scores = np.concatenate([scores, [-1.0, -2.0, -3.0, -4.0, -5.0, -6.0]])
        ft = _lib.FlatTrie(toks, scores)
        ora = orc.OracleModel(toks, scores)
        for k in range(2000):
            pool = src if k % 2 == 0 else raw
            i = int(rng.integers(0, len(pool) - 1))
            s8 = pool[i:i + 40]
            got, st = ft.common_prefix_search8(s8)
            assert got == ora.common_prefix_search(s8) == ft.common_prefix_search(s8)
        distinct = len(set(np.asarray(scores, np.float64).tobytes()[8 * i:8 * i + 8] for i in range(len(toks)) if toks[i]))
        assert st["n_hot"] == min(distinct, 6600) or st["n_hot"] <= distinct
        _, st_small = ft.common_prefix_search8(b"ab", max_hot=16)
        assert st_small["n_hot"] <= 16 and st_small["n_cold"] > 0 and st_small["hot_coverage"] < 1.0
    # a generate-style vocabulary scores tokens by integer counts: few distinct values, all of them in the table
    toks, scores = synth.build_vocab(flat, 8000, 16)
    _, st = _lib.FlatTrie(toks, scores).common_prefix_search8(b"return")
    assert st["n_cold"] == 0 and st["hot_coverage"] == 1.0 and st["n_hot"] < 6600


def test_flat_trie_large_vocab_builds_fast():
    flat, _ = synth.make_corpus(8 << 20, "mixed")
    toks, scores = synth.build_vocab(flat[: 4 << 20], 200000, 16)
    import time
    t = time.time()
    ft = _lib.FlatTrie(toks, scores)
    dt = time.time() - t
    st = ft.stats()
    assert dt < 20.0, dt
    assert st["fill"] > 0.6, st
    # configs[3]'s 500 000-entry vocabulary (628 436 nodes): 0.3 s on the build container since single-child nodes
    # keep a first candidate block per edge byte (1.35 s before: the scan over blocks with free slots but claimed bases)
    from util import load_vocab_500k
    toks5, scores5 = load_vocab_500k()
    packed = _lib.Packed.of(toks5)
    t = time.time()
    ft5 = _lib.FlatTrie(packed, scores5)
    dt5 = time.time() - t
    st5 = ft5.stats()
    assert dt5 < 5.0, dt5
    assert st5["n_nodes"] == 628436 and st5["fill"] > 0.7, st5


def test_special_splitter_kats(golden_dir):
    import json
    with open(os.path.join(golden_dir, "reference_kats.json"), encoding="utf-8") as f:
        kats = json.load(f)
    for c in kats["splitter"]["cases"]:
        assert tgx.split_special_tokens(c["input"], c["specials"]) == [(s, b) for s, b in c["segments"]]
    # first-listed special wins at one position, not the longest (src/tokenizer.rs:325-338)
    assert tgx.split_special_tokens("<EOS_2>", ["<EOS", "<EOS_2>"]) == [("<EOS", True), ("_2>", False)]


def test_splitter_matches_oracle_on_random_inputs():
    from oracle import oracle as orc
    rng = np.random.default_rng(2)
    specials = ["<EOS>", "ab", "<EOS_2>", "你", "b"]
    alphabet = ["a", "b", "<", "EOS", ">", "_2", "你", "好", " ", "<EOS>"]
    for _ in range(300):
        s = "".join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), size=int(rng.integers(0, 30))))
        want = orc.split_specials(s.encode(), [x.encode() for x in specials])
        assert [(a.encode(), b) for a, b in tgx.split_special_tokens(s, specials)] == want


def test_tokenizer_json_roundtrip_and_queries(tmp_path):
    vocab = [(b"a", -1.0, True), (b"\xff\xfe", -2.5, False), ("你好".encode(), -3.0, False), (b"a", -9.0, False)]
    tk = tgx.Tokenizer(vocab, [tgx.CrlfProcessor(), tgx.UnicodeProcessor("nfc")], ["<EOS>", "<PAD>", "<EOS>"])
    assert tk.special_tokens() == ["<EOS>", "<PAD>"]
    assert (tk.vocab_size(), tk.base_vocab_size(), tk.special_vocab_size()) == (6, 4, 2)
    assert tk.base_token_to_id(b"a") == 3  # later duplicate wins (src/model.rs:21)
    assert tk.token_to_id(b"<PAD>") == 5 and tk.special_token_to_id("<EOS>") == 4
    assert tk.id_to_token(1) == b"\xff\xfe" and tk.id_to_token(4) == b"<EOS>" and tk.id_to_token(9) is None
    assert tk.id_to_base_token(2) == ("你好".encode(), -3.0)
    assert tk.is_special(4) and not tk.is_special(3) and tk.is_base(3) and not tk.is_special(99)
    assert tk.decode([0, 2, 4, 1], True) == "a你好<EOS>��"
    assert tk.decode([0, 2, 4, 1], False) == "a你好��"
    with pytest.raises(tgx.TokenGeeXError) as e:
        tk.decode([77], True)
    assert str(e.value) == "token id 77 is out of bounds"
    p = tmp_path / "tok.json"
    tk.save(str(p))
    tk2 = tgx.Tokenizer.from_file(str(p))
    assert tk2.to_string() == tk.to_string()
    assert '"encoded":true' in tk.to_string() and '"keep":true' in tk.to_string()
    tk3 = pickle.loads(pickle.dumps(tk))
    assert tk3.vocab() == tk.vocab() and tk3.special_tokens() == tk.special_tokens()
    with pytest.raises(tgx.TokenGeeXError):
        tgx.Tokenizer.from_str('{"version":"1.0","vocab":[]}')
    with pytest.raises(tgx.TokenGeeXError):
        tgx.Tokenizer.from_str('{"version":"2.0","vocab":[],"bogus":1}')
    with pytest.raises(tgx.TokenGeeXError):
        tgx.Tokenizer.from_file(str(tmp_path / "missing.json"))


def test_tok_hash_table_is_exact_on_vocabularies():
    """The bytes -> id table of the trace kernel (32-bit-multiply hash, csrc/trie_build.h) must map every
    token back to its id: checked on a synthetic 6 K vocabulary, on adversarial near-duplicates, and on the
    cached 500 K vocabulary when tools/make_vocab_cache.py has been run."""
    import os
    import numpy as np
    from tokengeex_amd import _lib, synth
    flat, _ = synth.make_corpus(1 << 20, seed_offset=5)
    toks, _ = synth.build_vocab(flat, 6000, 16)
    assert _lib.tok_hash_selftest(toks)[1] == 0
    # every 1- and 2-byte string, zero bytes, duplicates (the later id must win), 16-byte tokens
    adv = [bytes([a]) for a in range(256)] + [bytes([a, b]) for a in range(0, 256, 3) for b in range(256)]
    adv += [b"\x00" * k for k in range(1, 17)] + [b"ab", b"ab", b"abcdefghijklmnop", b"abcdefghijklmnoq"]
    assert _lib.tok_hash_selftest(adv)[1] == 0
    cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cache", "vocab_500000.npz")
    if os.path.exists(cache):
        z = np.load(cache)
        o = z["offs"].astype(np.int64); fb = z["flat"].tobytes()
        big = [fb[o[i]:o[i + 1]] for i in range(o.size - 1)]
        assert _lib.tok_hash_selftest(big)[1] == 0
    # tokens of 17..32 bytes (vocabularies after `merge`): hashed over eight dwords
    long_toks = adv + [b"abcdefghijklmnopq", b"abcdefghijklmnopqr", b"abcdefghijklmnop" * 2, b"abcdefghijklmnop" + b"abcdefghijklmnoq",
                       b"\x00" * 17, b"\x00" * 32]
    assert _lib.tok_hash_selftest(long_toks)[1] == 0
    with pytest.raises(_lib.TokenGeeXError):
        _lib.tok_hash_selftest([b"x" * 33])
