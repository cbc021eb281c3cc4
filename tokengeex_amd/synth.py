"""Deterministic synthetic corpora and vocabularies.

The reference's real data (`hub/`) is git-ignored and needs network access, so
benchmarks and parity tests run on seeded synthetic text with the statistics that
matter for the hot path: code-like ASCII lines drawn Zipf from an identifier
lexicon (the lower / Upper / UPPER classes of data/exact.regex), keywords,
operators, digits, indentation and newlines, optionally mixed 70/30 by bytes with
runs of CJK characters (3-byte UTF-8) — and vocabularies built the way
`VocabularyGenerator::generate` scores them (reference src/generate.rs:148-243:
all single bytes 0..254 at the top frequency, then the most frequent allowed
substrings, score = freq * len turned into log-probabilities).

Bench / test infrastructure: the product path never imports this module.
"""
from __future__ import annotations

import numpy as np

SEED = 0x544F4B47454558  # "TOKGEEX"

KEYWORDS = ["def", "return", "if", "else", "elif", "for", "while", "in", "import", "from", "class", "self",
            "None", "True", "False", "int", "str", "float", "bool", "void", "const", "static", "public",
            "private", "function", "var", "let", "new", "this", "null", "struct", "fn", "pub", "use",
            "impl", "match", "async", "await", "try", "except", "catch", "throw", "lambda", "yield"]
OPERATORS = [" = ", " == ", " != ", " + ", " - ", " * ", " / ", " < ", " > ", " <= ", " >= ", " += ", " -= ",
             " -> ", " => ", " && ", " || ", "(", ")", "()", "[", "]", "[]", "{", "}", ", ", ".", ":", ";",
             "::", "...", "\"", "'", "#", "//", " & ", " | ", "!", "?", "@", "_", "%"]
SYLLABLES = ["a", "e", "i", "o", "u", "an", "en", "in", "on", "er", "re", "le", "st", "nd", "th", "tion", "al",
             "ar", "or", "at", "to", "ing", "ed", "es", "it", "is", "con", "de", "pro", "ex", "com", "get",
             "set", "val", "key", "idx", "num", "buf", "ptr", "len", "max", "min", "tmp", "ctx", "cfg", "arg",
             "data", "node", "list", "item", "name", "type", "size", "count", "index", "value", "error",
             "result", "token", "model", "state", "input", "output", "file", "path", "line", "text", "code"]


def _rng(seed_offset: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(SEED + seed_offset))


def _zipf_weights(n: int, s: float = 1.1) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), s)
    return w / w.sum()


def make_lexicon(kind: str = "mixed", n_identifiers: int = 50000, n_cjk_words: int = 20000):
    """Returns (items: list[bytes], cumulative probability array)."""
    rng = _rng(1)
    idents: dict[str, None] = {}  # insertion-ordered set
    while len(idents) < n_identifiers:
        m = n_identifiers
        ks = rng.integers(1, 4, size=m)
        syl = rng.integers(0, len(SYLLABLES), size=(m, 3))
        rs = rng.random(m)
        for k, row, r in zip(ks.tolist(), syl.tolist(), rs.tolist()):
            w = "".join(SYLLABLES[j] for j in row[:k])
            if r >= 0.90:
                w = w.upper()
            elif r >= 0.70:
                w = w.capitalize()
            idents[w] = None
    idents = list(idents)[:n_identifiers]
    items: list[bytes] = []
    weights: list[np.ndarray] = []

    def add(group, total_weight, zipf_s=1.1):
        items.extend(g.encode("utf-8") if isinstance(g, str) else g for g in group)
        weights.append(_zipf_weights(len(group), zipf_s) * total_weight)

    add(idents, 0.34)
    add(KEYWORDS, 0.10, 0.8)
    add(OPERATORS, 0.20, 0.7)
    add([" "], 0.17)
    add(["\n", "\n    ", "\n        ", "\n            ", "\n\n", "\n\t", "\n                "], 0.10, 0.9)
    add([str(i) for i in list(range(0, 33)) + [64, 100, 128, 255, 256, 1000, 1024, 4096, 65535]], 0.04, 0.9)
    add(["\r\n", "\t", "  ", "    "], 0.02)
    add(["\"%s\"" % s for s in ["ok", "error", "name", "id", "utf-8", "hello world", "foo", "bar"]], 0.03)
    if kind == "mixed":
        # scale the ASCII part to 70 % of bytes, CJK words 30 % (3 bytes per char)
        cjk_chars = [chr(c) for c in range(0x4E00, 0x4E00 + 3000)]
        cw = _zipf_weights(len(cjk_chars), 1.0)
        wordset: dict[str, None] = {}
        ccdf = np.cumsum(cw)
        while len(wordset) < n_cjk_words:
            m = n_cjk_words
            ks = rng.integers(1, 5, size=m)
            ch = np.minimum(np.searchsorted(ccdf, rng.random((m, 4)), side="right"), len(cjk_chars) - 1)
            for k, row in zip(ks.tolist(), ch.tolist()):
                wordset["".join(cjk_chars[j] for j in row[:k])] = None
        words = list(wordset)[:n_cjk_words]
        ascii_bytes = sum(float(w.sum()) for w in weights)  # == 1.0 probability mass
        mean_ascii = sum(float((np.array([len(i) for i in items[a:a + len(w)]]) * w).sum())
                         for a, w in zip(np.cumsum([0] + [len(w) for w in weights[:-1]]), weights)) / ascii_bytes
        wz = _zipf_weights(len(words), 1.05)
        mean_cjk = float((np.array([len(w.encode()) for w in words]) * wz).sum()) * 0.9 + 3 * 0.1
        # choose CJK probability q so that q*mean_cjk / (q*mean_cjk + (1-q)*mean_ascii) = 0.30
        q = 0.30 * mean_ascii / (0.70 * mean_cjk + 0.30 * mean_ascii)
        weights = [w * (1.0 - q) for w in weights]
        add(words, q * 0.9, 1.05)
        add(["，", "。", "、", "：", "（", "）"], q * 0.1, 0.8)
    p = np.concatenate(weights)
    p = p / p.sum()
    return items, np.cumsum(p)


_LEX_CACHE: dict = {}


def make_corpus(n_bytes: int, kind: str = "mixed", min_len: int = 64, max_len: int = 65536, seed_offset: int = 0):
    """-> (flat uint8[N], offsets uint64[S+1]), N within one lexicon item of n_bytes.
    Uses the C stream generator (csrc/synth_gen.c) — about 1 GB in a few seconds."""
    import ctypes as C
    import os

    from . import build as _build
    lib = C.CDLL(_build.build_synth())
    lib.synth_fill.restype = C.c_uint64
    lib.synth_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64,
                               C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    if kind not in _LEX_CACHE:
        items, cdf = make_lexicon(kind)
        lex_offs = np.zeros(len(items) + 1, dtype=np.uint32)
        np.cumsum(np.array([len(i) for i in items], dtype=np.uint32), out=lex_offs[1:])
        _LEX_CACHE[kind] = (np.frombuffer(b"".join(items), dtype=np.uint8).copy(), lex_offs,
                            np.ascontiguousarray(cdf, dtype=np.float64), len(items))
    lex_flat, lex_offs, cdf, n_items = _LEX_CACHE[kind]
    cap = n_bytes + 64
    text = np.empty(cap, dtype=np.uint8)
    max_samples = n_bytes // max(1, min_len) + 2
    offs = np.zeros(max_samples + 1, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    ns = lib.synth_fill(p(lex_flat), p(lex_offs), p(cdf), n_items, SEED + 100 + seed_offset, n_bytes, min_len,
                        max_len, p(text), cap, p(offs), max_samples)
    offs = offs[: ns + 1].copy()
    return text[: int(offs[-1])], offs


# ---- vocabulary stand-in for VocabularyGenerator (reference src/generate.rs) -----------

def _run_remaining(mask: np.ndarray) -> np.ndarray:
    """rem[i] = number of consecutive True values starting at i."""
    n = mask.shape[0]
    idx = np.arange(n, dtype=np.int64)
    nxt_false = np.where(~mask, idx, n)
    nxt_false = np.minimum.accumulate(nxt_false[::-1])[::-1]
    return np.where(mask, nxt_false - idx, 0)


def _count_windows(text: np.ndarray, valid_len: np.ndarray, L: int, step_ok: np.ndarray | None = None):
    n = text.shape[0]
    if n < L:
        return {}
    ok = valid_len[: n - L + 1] >= L
    if step_ok is not None:
        ok &= step_ok[: n - L + 1]
    pos = np.nonzero(ok)[0]
    if pos.size == 0:
        return {}
    win = np.lib.stride_tricks.sliding_window_view(text, L)[pos]
    keys = np.ascontiguousarray(win).view(f"V{L}").ravel()
    uniq, counts = np.unique(keys, return_counts=True)
    return {bytes(u.tobytes()): int(c) for u, c in zip(uniq, counts) if c >= 2}


def build_vocab(text: np.ndarray, size: int, max_token_length: int = 16):
    """Counts allowed substrings (the classes of data/exact.regex:1 that occur in the
    synthetic corpora: [a-z]+, [A-Z]+, [A-Z][a-z]+, CJK runs, space/tab runs,
    operators and punctuation with optional spaces) in `text` and keeps the `size`
    best by freq * len, scored like reference src/generate.rs:148-243.
    -> (tokens: list[bytes], scores: float64[size])"""
    t = np.ascontiguousarray(text, dtype=np.uint8)
    lower = (t >= 97) & (t <= 122)
    upper = (t >= 65) & (t <= 90)
    space = t == 32
    tab = t == 9
    cjk_lead = (t >= 0xE4) & (t <= 0xE9)
    rem_lower, rem_upper = _run_remaining(lower), _run_remaining(upper)
    rem_space, rem_tab = _run_remaining(space), _run_remaining(tab)
    # CJK: bytes of chars whose lead byte is E4..E9; run length counted in bytes from a lead byte
    cjk_byte = np.zeros_like(cjk_lead)
    cjk_byte |= cjk_lead
    cjk_byte[1:] |= cjk_lead[:-1]
    cjk_byte[2:] |= cjk_lead[:-2]
    rem_cjk = np.where(cjk_lead, _run_remaining(cjk_byte), 0)
    cap_len = np.zeros(t.shape[0], dtype=np.int64)  # [A-Z][a-z]+ starting here
    cap_len[:-1] = np.where(upper[:-1] & lower[1:], 1 + rem_lower[1:], 0)
    punct = ((t >= 33) & (t <= 47)) | ((t >= 58) & (t <= 64)) | ((t >= 91) & (t <= 96)) | ((t >= 123) & (t <= 126))
    rem_punct = _run_remaining(punct)

    freq: dict[bytes, int] = {}

    def merge(d):
        for k, v in d.items():
            freq[k] = freq.get(k, 0) + v

    for L in range(2, max_token_length + 1):
        merge(_count_windows(t, rem_lower, L))
        merge(_count_windows(t, rem_upper, L))
        merge(_count_windows(t, cap_len, L))
        merge(_count_windows(t, rem_space, L))
        merge(_count_windows(t, rem_tab, L))
        if L % 3 == 0:
            merge(_count_windows(t, rem_cjk, L))
    # operators / punctuation with optional surrounding single spaces: ' ?p{1,3} ?'
    n = t.shape[0]
    for pl in (1, 2, 3):
        for lead in (0, 1):
            for trail in (0, 1):
                L = pl + lead + trail
                if L < 2 or n < L:
                    continue
                ok = rem_punct[lead: n - L + 1 + lead] >= pl
                if lead:
                    ok &= space[: n - L + 1]
                if trail:
                    ok &= space[lead + pl: n - L + 1 + lead + pl]
                valid = np.where(ok, L, 0)
                merge(_count_windows(t, np.concatenate([valid, np.zeros(L - 1, np.int64)]), L))
    # newline + indentation idioms (the reference feeds these as --suggested-tokens-file)
    nl = t == 10
    for L in range(2, max_token_length + 1):
        ok = np.zeros(n, dtype=np.int64)
        ok[:-1] = np.where(nl[:-1], 1 + rem_space[1:], 0)
        merge(_count_windows(t, ok, L))

    cand = sorted(freq.items(), key=lambda kv: (-kv[1] * len(kv[0]), kv[0]))
    highest = max((v for _, v in cand), default=1)
    vocab: list[tuple[bytes, float]] = [(bytes([b]), float(highest)) for b in range(255)]  # generate.rs:164-169
    for tok, f in cand:
        if len(vocab) >= size:
            break
        vocab.append((tok, float(f * len(tok))))
    vocab.sort(key=lambda kv: -kv[1])  # generate.rs:208-212 (stable here)
    scores = np.array([s for _, s in vocab], dtype=np.float64)
    scores = np.log(scores) - np.log(scores.sum())  # logprobs, generate.rs:236-243
    return [tkn for tkn, _ in vocab], scores


def load_spec_vocab(size: int):
    """The benchmark vocabulary of SURVEY.md section 8(d): build_vocab over a fixed 64 MiB slice of the mixed corpus
    (seed offset 0), 32 000 or 65 536 entries of at most 16 bytes — minutes of numpy, so it is committed compressed
    (tests/golden/vocab_<size>.npz, written by tools/make_spec_vocab.py).  -> (tokens, scores, slice MiB)"""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"vocab_{size}.npz")
    z = np.load(path)
    fb = z["flat"].tobytes()
    o = np.concatenate([[0], np.cumsum(z["lens"].astype(np.int64))])
    toks = [fb[o[i]:o[i + 1]] for i in range(o.size - 1)]
    return toks, z["uscores"][z["inv"]].astype(np.float64), int(z["slice_mib"][0])


def random_vocab(rng: np.random.Generator, text: bytes, n_multi: int, max_len: int = 16,
                 all_bytes: bool = True, tie_fraction: float = 0.2):
    """Small random vocabulary for parity tests: (all) single bytes + substrings
    sampled from `text`, random scores with deliberate exact ties."""
    toks: list[bytes] = [bytes([b]) for b in range(256)] if all_bytes else []
    seen = set(toks)
    tries = 0
    while len(toks) < (256 if all_bytes else 0) + n_multi and tries < n_multi * 50 and len(text) > 2:
        tries += 1
        L = int(rng.integers(2, max_len + 1))
        i = int(rng.integers(0, max(1, len(text) - L)))
        tok = text[i:i + L]
        if len(tok) >= 2 and tok not in seen:
            seen.add(tok)
            toks.append(tok)
    scores = -(rng.random(len(toks)) * 10.0 + 0.5)
    ties = rng.random(len(toks)) < tie_fraction
    scores[ties] = np.round(scores[ties])  # integers: many exact ties between paths
    perm = rng.permutation(len(toks))
    return [toks[i] for i in perm], scores[perm]
