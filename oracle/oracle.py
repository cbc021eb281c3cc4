"""ctypes loader for the CPU oracle (oracle/tgx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (tokengeex_amd/) never
imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle.so")

OK, ERR_NO_PATH, ERR_Z_NOT_NORMAL = 0, 4, 6

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no-op when the .so is newer than the source)."""
    srcs = [os.path.join(_HERE, f) for f in ("tgx_oracle.c", "tgx_prune_oracle.c", "tgx_oracle.h")]
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(f) for f in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    L.orc_model_new.restype = C.c_void_p
    L.orc_model_new.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.orc_model_free.argtypes = [C.c_void_p]
    L.orc_dropout_u01.restype = C.c_double
    L.orc_dropout_u01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
    L.orc_encode.restype = C.c_int
    L.orc_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_uint64, C.c_uint64,
                             C.POINTER(_u32p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t),
                             C.POINTER(C.c_size_t)]
    L.orc_encode_batch.restype = C.c_int
    L.orc_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_double,
                                   C.c_uint64, C.c_int, C.POINTER(_u32p), C.c_void_p,
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_common_prefix_search.restype = C.c_size_t
    L.orc_common_prefix_search.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                           C.c_void_p, C.c_size_t]
    L.orc_marginal.restype = C.c_double
    L.orc_marginal.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_uint64,
                               C.c_uint64, C.c_uint64, C.c_void_p]
    L.orc_estep.restype = C.c_int
    L.orc_estep.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_double,
                            C.c_uint64, C.c_int, C.c_void_p, C.POINTER(C.c_double),
                            C.POINTER(C.c_uint64)]
    L.orc_count_tokens.restype = C.c_int
    L.orc_count_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                   C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_count_pairs.restype = C.c_int
    L.orc_count_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                  C.POINTER(_u64p), C.POINTER(_u64p), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_split_specials.restype = C.c_size_t
    L.orc_split_specials.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_digamma.restype = C.c_double
    L.orc_digamma.argtypes = [C.c_double]
    L.orc_m_step.restype = C.c_int
    L.orc_m_step.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, _u32p]
    L.orc_prune_alternatives.restype = C.c_int
    L.orc_prune_alternatives.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.POINTER(_u32p)]
    L.orc_prune_select.restype = C.c_int
    L.orc_prune_select.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, _u32p]
    _lib = L
    return L


class NoPath(Exception):
    """Error::NoPath(pos, len) — src/lib.rs:223, message src/lib.rs:243-245."""

    def __init__(self, pos: int, length: int, sample: int = 0):
        super().__init__(f"no path to position {pos}/{length}")
        self.pos, self.length, self.sample = pos, length, sample


def pack(items) -> tuple[np.ndarray, np.ndarray]:
    """list[bytes] -> (uint8 flat, uint64 offsets[len+1])."""
    offs = np.zeros(len(items) + 1, dtype=np.uint64)
    if len(items):
        offs[1:] = np.cumsum([len(t) for t in items], dtype=np.uint64)
    flat = np.frombuffer(b"".join(items), dtype=np.uint8).copy() if len(items) else np.zeros(0, np.uint8)
    if flat.size == 0:
        flat = np.zeros(1, np.uint8)[:0]
    return flat, offs


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class OracleModel:
    """Model — src/model.rs:8-200 (hot-path methods only)."""

    def __init__(self, tokens: list[bytes], scores):
        self.tokens = [bytes(t) for t in tokens]
        self.scores = np.ascontiguousarray(scores, dtype=np.float64)
        assert len(self.tokens) == self.scores.shape[0]
        flat, offs = pack(self.tokens)
        self._flat, self._offs = flat, offs
        self._h = lib().orc_model_new(_ptr(flat), _ptr(offs), _ptr(self.scores), len(self.tokens))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_model_free(self._h)
            self._h = None

    @property
    def vocab_size(self) -> int:
        return len(self.tokens)

    def encode(self, text: bytes, dropout: float = 0.0, seed: int = 0, sample_index: int = 0) -> list[int]:
        buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(0, np.uint8)
        ids = _u32p()
        n = C.c_size_t()
        ep, el = C.c_size_t(), C.c_size_t()
        st = lib().orc_encode(self._h, _ptr(buf) if len(text) else None, len(text), dropout, seed,
                              sample_index, C.byref(ids), C.byref(n), C.byref(ep), C.byref(el))
        if st == ERR_NO_PATH:
            raise NoPath(ep.value, el.value)
        out = [ids[i] for i in range(n.value)]
        lib().orc_free(ids)
        return out

    def encode_batch_flat(self, flat: np.ndarray, offs: np.ndarray, dropout: float = 0.0,
                          seed: int = 0, threads: int = 1):
        S = offs.shape[0] - 1
        out_offs = np.zeros(S + 1, dtype=np.uint64)
        ids = _u32p()
        es, ep = C.c_uint64(), C.c_uint64()
        st = lib().orc_encode_batch(self._h, _ptr(flat), _ptr(offs), S, dropout, seed, threads,
                                    C.byref(ids), _ptr(out_offs), C.byref(es), C.byref(ep))
        if st == ERR_NO_PATH:
            raise NoPath(ep.value, ep.value, es.value)
        T = int(out_offs[S])
        arr = np.ctypeslib.as_array(ids, shape=(max(T, 1),))[:T].copy()
        lib().orc_free(ids)
        return arr, out_offs

    def encode_batch(self, texts: list[bytes], dropout: float = 0.0, seed: int = 0, threads: int = 1):
        flat, offs = pack(texts)
        ids, oo = self.encode_batch_flat(flat, offs, dropout, seed, threads)
        return [ids[int(oo[i]):int(oo[i + 1])].tolist() for i in range(len(texts))]

    def common_prefix_search(self, s: bytes) -> list[tuple[int, int]]:
        buf = np.frombuffer(s, dtype=np.uint8) if len(s) else np.zeros(0, np.uint8)
        cap = len(s) + 1
        ids = np.zeros(cap, np.uint32)
        lens = np.zeros(cap, np.uint32)
        k = lib().orc_common_prefix_search(self._h, _ptr(buf), len(s), _ptr(ids), _ptr(lens), cap)
        return [(int(ids[i]), int(lens[i])) for i in range(k)]

    def marginal(self, snippet: bytes, dropout: float = 0.0, seed: int = 0, sample_index: int = 0,
                 snippet_base: int = 0):
        buf = np.frombuffer(snippet, dtype=np.uint8) if len(snippet) else np.zeros(0, np.uint8)
        expected = np.zeros(self.vocab_size, np.float64)
        z = lib().orc_marginal(self._h, _ptr(buf), len(snippet), dropout, seed, sample_index,
                               snippet_base, _ptr(expected))
        return expected, z

    def estep_flat(self, flat, offs, snippet_len: int = 81920, dropout: float = 0.0, seed: int = 0,
                   threads: int = 1):
        S = offs.shape[0] - 1
        expected = np.zeros(self.vocab_size, np.float64)
        z = C.c_double()
        es = C.c_uint64()
        st = lib().orc_estep(self._h, _ptr(flat), _ptr(offs), S, snippet_len, dropout, seed, threads,
                             _ptr(expected), C.byref(z), C.byref(es))
        return st, expected, z.value, es.value

    def count_tokens_flat(self, flat, offs, threads: int = 1):
        S = offs.shape[0] - 1
        freq = np.zeros(self.vocab_size, np.uint64)
        es, ep = C.c_uint64(), C.c_uint64()
        st = lib().orc_count_tokens(self._h, _ptr(flat), _ptr(offs), S, threads, _ptr(freq),
                                    C.byref(es), C.byref(ep))
        if st == ERR_NO_PATH:
            raise NoPath(ep.value, ep.value, es.value)
        return freq

    def count_pairs_flat(self, flat, offs, threads: int = 1):
        S = offs.shape[0] - 1
        keys, counts = _u64p(), _u64p()
        n = C.c_uint64()
        es, ep = C.c_uint64(), C.c_uint64()
        st = lib().orc_count_pairs(self._h, _ptr(flat), _ptr(offs), S, threads, C.byref(keys),
                                   C.byref(counts), C.byref(n), C.byref(es), C.byref(ep))
        k = int(n.value)
        ka = np.ctypeslib.as_array(keys, shape=(max(k, 1),))[:k].copy()
        ca = np.ctypeslib.as_array(counts, shape=(max(k, 1),))[:k].copy()
        lib().orc_free(keys)
        lib().orc_free(counts)
        if st == ERR_NO_PATH:
            raise NoPath(ep.value, ep.value, es.value)
        return ka, ca


def dropout_u01(seed: int, sample: int, pos: int, length: int) -> float:
    return lib().orc_dropout_u01(seed, sample, pos, length)


def split_specials(text: bytes, specials: list[bytes]) -> list[tuple[bytes, bool]]:
    """SpecialTokenSplitter — src/tokenizer.rs:299-347."""
    buf = np.frombuffer(text, dtype=np.uint8) if len(text) else np.zeros(0, np.uint8)
    sp_flat, sp_offs = pack(specials)
    cap = len(text) + 1
    st = np.zeros(cap, np.uint64)
    en = np.zeros(cap, np.uint64)
    sp = np.zeros(cap, np.int32)
    k = lib().orc_split_specials(_ptr(buf), len(text), _ptr(sp_flat), _ptr(sp_offs), len(specials),
                                 _ptr(st), _ptr(en), _ptr(sp), cap)
    if k == C.c_size_t(-1).value:
        raise ValueError("empty special token")
    return [(text[int(st[i]):int(en[i])], bool(sp[i] >= 0)) for i in range(k)]


# ---- host half of `prune` (tgx_prune_oracle.c) ----

def digamma(x: float) -> float:
    return lib().orc_digamma(x)


def m_step(expected, keep):
    """src/prune.rs:124-170 -> (status, ids, scores)."""
    expected = np.ascontiguousarray(expected, dtype=np.float64)
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    V = expected.shape[0]
    idx, sc, n = np.zeros(max(V, 1), np.uint32), np.zeros(max(V, 1), np.float64), C.c_uint32()
    st = lib().orc_m_step(_ptr(expected), _ptr(keep), V, _ptr(idx), _ptr(sc), C.byref(n))
    return st, idx[:n.value].copy(), sc[:n.value].copy()


def prune_alternatives(model: "OracleModel"):
    """src/prune.rs:179-203 -> (always_keep, alt_offs, alt_ids)."""
    V = model.vocab_size
    always_keep = np.zeros(V, np.uint8)
    alt_offs = np.zeros(V + 1, np.uint32)
    p = _u32p()
    lib().orc_prune_alternatives(model._h, _ptr(model._flat), _ptr(model._offs), _ptr(model.scores), V,
                                 _ptr(always_keep), _ptr(alt_offs), C.byref(p))
    k = int(alt_offs[V])
    ids = np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].copy()
    lib().orc_free(p)
    return always_keep, alt_offs, ids


def prune_select(freq, keep, always_keep, alt_offs, alt_ids, scores, n_samples: int, pruned_size: int):
    """src/prune.rs:246-318 -> (status, ids in final order)."""
    freq = np.ascontiguousarray(freq, dtype=np.uint64)
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    always_keep = np.ascontiguousarray(always_keep, dtype=np.uint8)
    alt_offs = np.ascontiguousarray(alt_offs, dtype=np.uint32)
    alt_ids = np.ascontiguousarray(alt_ids, dtype=np.uint32)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    V = freq.shape[0]
    out, n = np.zeros(max(V, 1), np.uint32), C.c_uint32()
    st = lib().orc_prune_select(_ptr(freq), _ptr(keep), _ptr(always_keep), _ptr(alt_offs), _ptr(alt_ids),
                                _ptr(scores), V, n_samples, pruned_size, _ptr(out), C.byref(n))
    return st, out[:n.value].copy()
