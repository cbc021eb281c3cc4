"""ctypes binding of the C ABI declared in include/tgx.h (tokengeex_amd/libtgx.so).

The extension is mandatory: importing this module raises if libtgx.so is missing
or does not export a declared symbol, and every compute call raises
TokenGeeXError when no gfx950 device is usable.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtgx.so")

# tgx_status (include/tgx.h)
OK, ERR_IO, ERR_JSON, ERR_TOKEN_ID_OOB, ERR_NO_PATH, ERR_DEVICE, ERR_Z_NOT_NORMAL, ERR_INVALID, \
    ERR_UNSUPPORTED = range(9)
MAX_TOKEN_LEN = 64
ESTEP_SNIPPET_LEN = 81920

# every exported symbol of include/tgx.h: name -> (restype, argtypes)
_vp, _u64, _u32, _i, _d = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
_pvp, _pu64 = C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)
SYMBOLS = {
    "tgx_last_error": (C.c_char_p, []),
    "tgx_last_error_detail": (None, [_pu64, _pu64, _pu64]),
    "tgx_abi_version": (_i, []),
    "tgx_device_count": (_i, []),
    "tgx_model_create": (_i, [_vp, _vp, _vp, _u32, _i, _pvp]),
    "tgx_model_create_ex": (_i, [_vp, _vp, _vp, _u32, _i, _u32, _pvp]),
    "tgx_model_create_derived": (_i, [_vp, _vp, _u32, _vp, _u32, _pvp]),
    "tgx_model_destroy": (None, [_vp]),
    "tgx_model_vocab_size": (_u32, [_vp]),
    "tgx_model_max_token_len": (_u32, [_vp]),
    "tgx_model_trie_bytes": (_u64, [_vp]),
    "tgx_model_device": (_i, [_vp]),
    "tgx_common_prefix_search": (_i, [_vp, _vp, _u64, _vp, _vp, _u64, _pu64]),
    "tgx_flat_trie_build": (_i, [_vp, _vp, _vp, _u32, _pvp]),
    "tgx_flat_trie_free": (None, [_vp]),
    "tgx_flat_trie_search": (_u64, [_vp, _vp, _u64, _vp, _vp, _u64]),
    "tgx_flat_trie_search8": (_u64, [_vp, _vp, _vp, _vp, _u32, _vp, _u64, _vp, _vp, _u64, C.POINTER(C.c_uint32), _pu64,
                                     C.POINTER(C.c_double)]),
    "tgx_flat_trie_stats": (None, [_vp, _pu64, _pu64, C.POINTER(C.c_uint32)]),
    "tgx_flat_trie_copy": (None, [_vp, _vp, _vp, _vp]),
    "tgx_tok_hash_selftest": (_i, [_vp, _vp, _u32, C.POINTER(C.c_uint32), _pu64]),
    "tgx_dropout_u01_host": (_d, [_u64, _u64, _u64, _u32]),
    "tgx_encode_batch": (_i, [_vp, _vp, _vp, _u64, _d, _u64, _pvp]),
    "tgx_encode_batch_host": (_i, [_vp, _vp, _vp, _u64, _d, _u64, _vp, _u64, _vp, _pu64]),
    "tgx_encode_batch_multi": (_i, [_vp, _u32, _vp, _vp, _u64, _d, _u64, _vp, _u64, _vp, _pu64]),
    "tgx_result_num_samples": (_u64, [_vp]),
    "tgx_result_num_tokens": (_u64, [_vp]),
    "tgx_result_ids": (_vp, [_vp]),
    "tgx_result_offsets": (_vp, [_vp]),
    "tgx_result_copy_ids": (_i, [_vp, _vp, _u64]),
    "tgx_result_copy_offsets": (_i, [_vp, _vp, _u64]),
    "tgx_result_ids_device": (_vp, [_vp]),
    "tgx_result_offsets_device": (_vp, [_vp]),
    "tgx_result_free": (None, [_vp]),
    "tgx_corpus_upload": (_i, [_i, _vp, _vp, _u64, _pvp]),
    "tgx_corpus_free": (None, [_vp]),
    "tgx_corpus_num_samples": (_u64, [_vp]),
    "tgx_corpus_num_bytes": (_u64, [_vp]),
    "tgx_encode_corpus": (_i, [_vp, _vp, _d, _u64, _pvp]),
    "tgx_count_tokens": (_i, [_vp, _vp, _vp]),
    "tgx_count_pairs": (_i, [_vp, _vp, _pvp, _pvp, _pu64]),
    "tgx_count_pairs_top": (_i, [_vp, _vp, _u64, _pvp, _pvp, _pu64, _pu64]),
    "tgx_estep": (_i, [_vp, _vp, _u64, _d, _u64, _vp, C.POINTER(C.c_double)]),
    "tgx_split_specials": (_i, [_vp, _vp, _u64, _vp, _vp, _u32, _vp, _pvp, _pvp, _pvp, _pu64]),
    "tgx_pack_segments": (_i, [_vp, _vp, _vp, _vp, _u64, _i, _vp, _vp, _pu64]),
    "tgx_normalize_segments": (_i, [_u32, _vp, _vp, _vp, _u64, _pvp, _pvp]),
    "tgx_unidata_version": (C.c_char_p, []),
    "tgx_assemble_ids": (_i, [_vp, _vp, _u64, _vp, _vp, _u32, _vp, _vp]),
    "tgx_decode_batch": (_i, [_vp, _vp, _u32, _vp, _vp, _u32, _vp, _vp, _u64, _i, _pvp, _vp, _pu64, _pu64]),
    "tgx_utf8_lossy": (_u64, [_vp, _u64, _vp]),
    "tgx_substring_df": (_i, [_i, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _u32, _d, _u64, _pvp, _pvp, _pvp, _pu64, _pu64, _pu64]),
    "tgx_substring_df_top": (_i, [_i, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _u32, _d, _u64, _u64, _pvp, _pvp, _pvp, _pu64, _pu64, _pu64,
                                  _pu64, C.POINTER(C.c_uint32)]),
    "tgx_generate_u01": (_d, [_u64, _u64, _u64]),
    "tgx_free": (None, [_vp]),
    "tgx_pool_trim": (None, [_i]),
    "tgx_host_alloc": (_vp, [_u64]),
    "tgx_host_free": (None, [_vp]),
    "tgx_digamma": (_d, [_d]),
    "tgx_prune_m_step": (_i, [_vp, _vp, _u32, _vp, _vp, C.POINTER(C.c_uint32)]),
    "tgx_prune_alternatives": (_i, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _pvp]),
    "tgx_model_prune_alternatives": (_i, [_vp, _vp, _vp, _pvp]),
    "tgx_prune_select": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u64, _u32, _vp, C.POINTER(C.c_uint32)]),
    "tgx_last_kernel_times": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), _i]),
    "tgx_last_algorithmic_bytes": (_u64, [_vp]),
    "tgx_last_encode_waves_per_cu": (_u32, [_vp]),
    "tgx_last_encode_redo_samples": (_u64, [_vp]),
    "tgx_model_score_values": (_u32, [_vp]),
    "tgx_last_encode_hot_values": (_u32, [_vp]),
    "tgx_last_encode_long_samples": (_u64, [_vp]),
    "tgx_last_estep_pieces": (_u64, [_vp]),
    "tgx_last_estep_redo": (_u64, [_vp]),
    "tgx_last_encode_corun_cus": (_u32, [_vp]),
    "tgx_encode_corun_timeouts": (_u32, [_vp]),
}


class TokenGeeXError(Exception):
    """tokengeex.TokenGeeXError — bindings/python/src/lib.rs:9,33-37; carries the
    reference's Display string (src/lib.rs:238-249)."""

    def __init__(self, message: str, status: int = -1, sample: int | None = None,
                 pos: int | None = None, length: int | None = None):
        super().__init__(message)
        self.status, self.sample, self.pos, self.length = status, sample, pos, length


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is mandatory (no CPU fallback). "
            "Build it with `python tokengeex_amd/build.py`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.tgx_abi_version() != 1:
        raise ImportError("libtgx.so ABI version mismatch")
    return lib


lib = _load()


def check(status: int) -> None:
    if status == OK:
        return
    msg = (lib.tgx_last_error() or b"").decode("utf-8", "replace")
    if status in (ERR_NO_PATH, ERR_Z_NOT_NORMAL):
        s, p, l = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib.tgx_last_error_detail(C.byref(s), C.byref(p), C.byref(l))
        raise TokenGeeXError(msg, status, s.value, p.value, l.value)
    raise TokenGeeXError(msg, status)


class _PinnedBlock:
    """Owner of one tgx_host_alloc block; numpy views keep it alive through their base chain."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.addr = lib.tgx_host_alloc(self.nbytes)
        if not self.addr:
            raise TokenGeeXError(lib.tgx_last_error().decode("utf-8", "replace"))
        self.buf = (C.c_ubyte * max(1, self.nbytes)).from_address(self.addr)

    def __del__(self):
        addr, self.addr = getattr(self, "addr", None), None
        if addr:
            lib.tgx_host_free(addr)


def pinned_empty(shape, dtype) -> np.ndarray:
    """numpy array over page-locked host memory (tgx_host_alloc): uploads from it and downloads into it are
    DMA transfers at the PCIe link's rate."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    blk = _PinnedBlock(n * dt.itemsize)
    arr = np.frombuffer(blk.buf, dtype=dt, count=n).reshape(shape)
    blk.buf._tgx_owner = blk  # the array's base is blk.buf: the block lives as long as any view of it
    return arr


def pool_trim(device: int = -1) -> None:
    """Returns the library's pooled device buffers to the HIP runtime (all devices if negative)."""
    lib.tgx_pool_trim(device)


def device_count() -> int:
    return lib.tgx_device_count()


def ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Packed:
    """A list of byte strings kept in the ABI's batch format (flat bytes + offsets), with subsets taken without a
    Python loop: what the prune driver carries its vocabulary in between passes (500 000 tokens: a list
    comprehension plus pack() per model cost as much as the native trie build)."""

    __slots__ = ("flat", "offs")

    def __init__(self, flat: np.ndarray, offs: np.ndarray):
        self.flat, self.offs = flat, offs

    @classmethod
    def of(cls, items) -> "Packed":
        return items if isinstance(items, Packed) else cls(*pack(items))

    def __len__(self) -> int:
        return int(self.offs.shape[0]) - 1

    def take(self, idx) -> "Packed":
        idx = np.asarray(idx, dtype=np.int64)
        beg, end = self.offs[:-1][idx].astype(np.int64), self.offs[1:][idx].astype(np.int64)
        lens = end - beg
        offs = np.zeros(idx.shape[0] + 1, np.uint64)
        np.cumsum(lens, out=offs[1:])
        total = int(offs[-1])
        # byte j of the output comes from beg[i] + (j - offs[i]) of the token i it belongs to
        src = np.repeat(beg - offs[:-1].astype(np.int64), lens) + np.arange(total, dtype=np.int64)
        return Packed(self.flat[src] if total else np.zeros(0, np.uint8), offs)

    def tolist(self) -> list[bytes]:
        raw, o = self.flat.tobytes(), self.offs.tolist()
        return [raw[o[i]:o[i + 1]] for i in range(len(o) - 1)]


def pack(items) -> tuple[np.ndarray, np.ndarray]:
    """list[bytes] (or a Packed) -> (uint8 flat, uint64 offsets[len+1]) — the ABI's batch format."""
    if isinstance(items, Packed):
        return items.flat, items.offs
    offs = np.zeros(len(items) + 1, dtype=np.uint64)
    if len(items):
        np.cumsum(np.fromiter((len(t) for t in items), dtype=np.uint64, count=len(items)), out=offs[1:])
    flat = np.frombuffer(b"".join(items), dtype=np.uint8)
    return flat, offs


def _take(ptr_, n, ctype, dtype):
    """Copies a malloc'd array of n elements out of the library and frees it."""
    if not ptr_ or n == 0:
        if ptr_:
            lib.tgx_free(ptr_)
        return np.zeros(0, dtype)
    a = np.ctypeslib.as_array(C.cast(ptr_, C.POINTER(ctype)), shape=(n,)).copy()
    lib.tgx_free(ptr_)
    return a


def split_specials_flat(flat: np.ndarray, offs: np.ndarray, specials: list[bytes]):
    """SpecialTokenSplitter over a packed batch -> (seg_offs u64[S+1], seg_begin, seg_end u64[M], seg_special i32[M])."""
    sflat, soffs = pack(specials)
    n = offs.shape[0] - 1
    seg_offs = np.zeros(n + 1, np.uint64)
    sb, se, ss, m = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
    check(lib.tgx_split_specials(ptr(flat) if flat.size else None, ptr(offs), n, ptr(sflat) if sflat.size else None, ptr(soffs),
                                 len(specials), ptr(seg_offs), C.byref(sb), C.byref(se), C.byref(ss), C.byref(m)))
    k = m.value
    return seg_offs, _take(sb, k, C.c_uint64, np.uint64), _take(se, k, C.c_uint64, np.uint64), _take(ss, k, C.c_int32, np.int32)


def pack_segments(flat: np.ndarray, seg_begin: np.ndarray, seg_end: np.ndarray, seg_special, crlf: bool):
    """The non-special segments back to back, CRLF-normalised on the way if asked -> (flat, offs)."""
    n = seg_begin.shape[0]
    total = int((seg_end.astype(np.int64) - seg_begin.astype(np.int64)).sum()) if n else 0
    out = np.empty(max(total, 1), np.uint8)
    out_offs = np.zeros(n + 1, np.uint64)
    m = C.c_uint64()
    check(lib.tgx_pack_segments(ptr(flat) if flat.size else None, ptr(seg_begin), ptr(seg_end),
                                None if seg_special is None else ptr(seg_special), n, 1 if crlf else 0, ptr(out), ptr(out_offs),
                                C.byref(m)))
    out_offs = out_offs[: m.value + 1]
    return out[: int(out_offs[-1])], out_offs


NORMAL_FORMS = {"nfd": 0, "nfc": 1, "nfkd": 2, "nfkc": 3}


def normalize_flat(form: str, flat: np.ndarray, offs: np.ndarray):
    """UnicodeProcessor::preprocess over a packed batch (src/processor.rs:124-137) -> (flat, offs), native (csrc/unicode_norm.cpp)."""
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    n = offs.shape[0] - 1
    beg, end = np.ascontiguousarray(offs[:-1]), np.ascontiguousarray(offs[1:])
    ot, oo = C.c_void_p(), C.c_void_p()
    check(lib.tgx_normalize_segments(NORMAL_FORMS[form], ptr(flat) if flat.size else None, ptr(beg) if n else None, ptr(end) if n else None, n,
                                     C.byref(ot), C.byref(oo)))
    out_offs = _take(oo, n + 1, C.c_uint64, np.uint64)
    return _take(ot, int(out_offs[-1]), C.c_uint8, np.uint8), out_offs


def unidata_version() -> str:
    return lib.tgx_unidata_version().decode()


def assemble_ids(seg_offs: np.ndarray, seg_special: np.ndarray, ids: np.ndarray, id_offs: np.ndarray, vocab_size: int):
    n = seg_offs.shape[0] - 1
    n_special = int((seg_special >= 0).sum()) if seg_special.size else 0
    out = np.empty(max(1, ids.shape[0] + n_special), np.uint32)
    out_offs = np.zeros(n + 1, np.uint64)
    check(lib.tgx_assemble_ids(ptr(seg_offs), ptr(seg_special) if seg_special.size else None, n, ptr(ids) if ids.size else None,
                               ptr(id_offs), vocab_size, ptr(out), ptr(out_offs)))
    return out[: int(out_offs[-1])], out_offs


def decode_batch_flat(vocab_flat, vocab_offs, vocab_size: int, special_flat, special_offs, n_specials: int,
                      ids: np.ndarray, id_offs: np.ndarray, include_special: bool):
    """decode_batch over packed ids -> (utf-8 bytes, offsets u64[S+1])."""
    n = id_offs.shape[0] - 1
    out_offs = np.zeros(n + 1, np.uint64)
    txt, bs, bi = C.c_void_p(), C.c_uint64(), C.c_uint64()
    st = lib.tgx_decode_batch(ptr(vocab_flat) if vocab_flat.size else None, ptr(vocab_offs), vocab_size,
                              ptr(special_flat) if special_flat.size else None, ptr(special_offs), n_specials,
                              ptr(ids) if ids.size else None, ptr(id_offs), n, 1 if include_special else 0, C.byref(txt),
                              ptr(out_offs), C.byref(bs), C.byref(bi))
    if st != OK:
        msg = (lib.tgx_last_error() or b"").decode("utf-8", "replace")
        raise TokenGeeXError(msg, st, bs.value, bi.value, None)
    return _take(txt, int(out_offs[-1]), C.c_uint8, np.uint8), out_offs


def substring_df(flat: np.ndarray, part_begin: np.ndarray, part_end: np.ndarray, part_sample: np.ndarray,
                 max_token_length: int, insert_probability: float = 1.0, seed: int = 0, device: int = 0, with_collisions: bool = False,
                 part_origin=None):
    """Document frequencies of char-aligned substrings on the device -> (pos u64[D], len u32[D], df u32[D], n_windows
    [, entries that met a foreign run in a discarded attempt]).  part_origin: where every part's sample begins in `flat`
    (None: the part is its own sample) — the keep rule hashes the occurrence's offset in its sample."""
    flat = np.ascontiguousarray(flat, np.uint8)
    pb, pe = np.ascontiguousarray(part_begin, np.uint64), np.ascontiguousarray(part_end, np.uint64)
    ps = np.ascontiguousarray(part_sample, np.uint32)
    po = None if part_origin is None else np.ascontiguousarray(part_origin, np.uint64)
    pos, ln, df, n, nw, nc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    check(lib.tgx_substring_df(device, ptr(flat) if flat.size else None, flat.size, ptr(pb), ptr(pe), ptr(ps), None if po is None else ptr(po), pb.shape[0],
                               max_token_length, float(insert_probability), seed & (2**64 - 1), C.byref(pos), C.byref(ln),
                               C.byref(df), C.byref(n), C.byref(nw), C.byref(nc)))
    k = n.value
    out = (_take(pos, k, C.c_uint64, np.uint64), _take(ln, k, C.c_uint32, np.uint32), _take(df, k, C.c_uint32, np.uint32), nw.value)
    return out + (nc.value,) if with_collisions else out


def substring_df_top(flat: np.ndarray, part_begin: np.ndarray, part_end: np.ndarray, part_sample: np.ndarray,
                     max_token_length: int, top_k: int, insert_probability: float = 1.0, seed: int = 0, device: int = 0, part_origin=None):
    """The top_k most frequent substrings only (descending frequency; 0 = all)
    -> (pos, len, df, n_windows, n_distinct, cutoff_df): whatever was cut off occurs in at most cutoff_df samples."""
    flat = np.ascontiguousarray(flat, np.uint8)
    pb, pe = np.ascontiguousarray(part_begin, np.uint64), np.ascontiguousarray(part_end, np.uint64)
    ps = np.ascontiguousarray(part_sample, np.uint32)
    po = None if part_origin is None else np.ascontiguousarray(part_origin, np.uint64)
    pos, ln, df, n, nw, nc = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    nd, cut = C.c_uint64(), C.c_uint32()
    check(lib.tgx_substring_df_top(device, ptr(flat) if flat.size else None, flat.size, ptr(pb), ptr(pe), ptr(ps), None if po is None else ptr(po), pb.shape[0],
                                   max_token_length, float(insert_probability), seed & (2**64 - 1), int(top_k), C.byref(pos),
                                   C.byref(ln), C.byref(df), C.byref(n), C.byref(nw), C.byref(nc), C.byref(nd), C.byref(cut)))
    k = n.value
    return (_take(pos, k, C.c_uint64, np.uint64), _take(ln, k, C.c_uint32, np.uint32), _take(df, k, C.c_uint32, np.uint32),
            nw.value, nd.value, cut.value)


def generate_u01(seed: int, sample: int, window_hash: int) -> float:
    return lib.tgx_generate_u01(seed & (2**64 - 1), sample, window_hash & (2**64 - 1))


def utf8_lossy(data: bytes) -> bytes:
    """String::from_utf8_lossy (the library's own implementation; tests compare it with Python's)."""
    src = np.frombuffer(data, np.uint8)
    out = np.empty(max(1, 3 * len(data)), np.uint8)
    n = lib.tgx_utf8_lossy(ptr(src) if len(data) else None, len(data), ptr(out))
    return out[:n].tobytes()


class NativeResult:
    """Owns a tgx_result (ids + offsets of one encode pass)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        self.free()

    def free(self):
        if getattr(self, "_h", None):
            lib.tgx_result_free(self._h)
            self._h = None

    @property
    def num_tokens(self) -> int:
        return lib.tgx_result_num_tokens(self._h)

    @property
    def num_samples(self) -> int:
        return lib.tgx_result_num_samples(self._h)

    def offsets(self) -> np.ndarray:
        out = np.empty(self.num_samples + 1, np.uint64)
        check(lib.tgx_result_copy_offsets(self._h, ptr(out), out.size))
        return out

    def ids(self) -> np.ndarray:
        t = self.num_tokens
        if t == 0:
            return np.zeros(0, np.uint32)
        out = np.empty(t, np.uint32)   # the device copy lands in the array itself
        check(lib.tgx_result_copy_ids(self._h, ptr(out), t))
        return out

    def ids_into(self, out: np.ndarray) -> int:
        """Copies the ids into caller memory (uint32, at least num_tokens long); -> number of ids."""
        t = self.num_tokens
        assert out.dtype == np.uint32 and out.flags.c_contiguous and out.size >= t
        if t:
            check(lib.tgx_result_copy_ids(self._h, ptr(out), t))
        return t

    def ids_device_ptr(self) -> int:
        return lib.tgx_result_ids_device(self._h) or 0


class NativeCorpus:
    """Owns a tgx_corpus: a packed batch resident in HBM across passes."""

    def __init__(self, flat: np.ndarray, offs: np.ndarray, device: int = 0):
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        h = C.c_void_p()
        check(lib.tgx_corpus_upload(device, ptr(flat) if flat.size else None, ptr(offs),
                                    offs.shape[0] - 1, C.byref(h)))
        self._h = h
        self.device = device

    def __del__(self):
        self.free()

    def free(self):
        if getattr(self, "_h", None):
            lib.tgx_corpus_free(self._h)
            self._h = None

    @property
    def num_samples(self) -> int:
        return lib.tgx_corpus_num_samples(self._h)

    @property
    def num_bytes(self) -> int:
        return lib.tgx_corpus_num_bytes(self._h)


class NativeModel:
    """Owns a tgx_model: Model::from(vocab) flattened into HBM (src/model.rs:16-30)."""

    def __init__(self, tokens: list[bytes], scores, device: int = 0, for_estep: bool = False):
        flat, offs = pack(tokens)
        self._scores = np.ascontiguousarray(scores, dtype=np.float64)
        if self._scores.shape[0] != len(tokens):
            raise ValueError("scores and tokens differ in length")
        h = C.c_void_p()
        check(lib.tgx_model_create_ex(ptr(flat) if flat.size else None, ptr(offs), ptr(self._scores),
                                      len(tokens), device, 1 if for_estep else 0, C.byref(h)))
        self._h = h
        self.device = device

    def __del__(self):
        self.free()

    def free(self):
        if getattr(self, "_h", None):
            lib.tgx_model_destroy(self._h)
            self._h = None

    def derive(self, keep_ids, scores, for_estep: bool = False) -> "NativeModel":
        """A model for the subset `keep_ids` (ascending ids of this model's tokens) with new scores, on this model's
        tables (tgx_model_create_derived): what prune builds per EM sub-iteration, without rebuilding the tries.
        Raises TokenGeeXError (unsupported) when this vocabulary has duplicate tokens."""
        keep = np.ascontiguousarray(keep_ids, dtype=np.uint32)
        sc = np.ascontiguousarray(scores, dtype=np.float64)
        if sc.shape[0] != keep.shape[0]:
            raise ValueError("scores and keep_ids differ in length")
        h = C.c_void_p()
        check(lib.tgx_model_create_derived(self._h, ptr(keep), keep.shape[0], ptr(sc), 1 if for_estep else 0, C.byref(h)))
        m = NativeModel.__new__(NativeModel)
        m._scores = sc
        m._h = h
        m.device = self.device
        return m

    @property
    def vocab_size(self) -> int:
        return lib.tgx_model_vocab_size(self._h)

    @property
    def max_token_len(self) -> int:
        return lib.tgx_model_max_token_len(self._h)

    @property
    def trie_bytes(self) -> int:
        return lib.tgx_model_trie_bytes(self._h)

    def encode_batch_flat(self, flat: np.ndarray, offs: np.ndarray, dropout: float = 0.0,
                          seed: int = 0) -> NativeResult:
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        h = C.c_void_p()
        check(lib.tgx_encode_batch(self._h, ptr(flat) if flat.size else None, ptr(offs),
                                   offs.shape[0] - 1, float(dropout), seed & (2**64 - 1), C.byref(h)))
        return NativeResult(h)

    def encode_batch_host(self, flat: np.ndarray, offs: np.ndarray, dropout: float = 0.0, seed: int = 0,
                          ids_out: np.ndarray | None = None):
        """Host buffers in, host buffers out, with upload / kernels / download of the batch's chunks overlapped
        (tgx_encode_batch_host) -> (ids uint32[T] — a view of ids_out when given —, offsets uint64[S+1])."""
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = offs.shape[0] - 1
        if ids_out is None:
            ids_out = np.empty(max(1, int(offs[-1] - offs[0])), np.uint32)
        assert ids_out.dtype == np.uint32 and ids_out.flags.c_contiguous
        out_offs = np.zeros(n + 1, np.uint64)
        t = C.c_uint64()
        check(lib.tgx_encode_batch_host(self._h, ptr(flat) if flat.size else None, ptr(offs), n, float(dropout),
                                        seed & (2**64 - 1), ptr(ids_out), ids_out.size, ptr(out_offs), C.byref(t)))
        return ids_out[: t.value], out_offs

    @staticmethod
    def encode_batch_multi(models: "list[NativeModel]", flat: np.ndarray, offs: np.ndarray, dropout: float = 0.0, seed: int = 0):
        """One batch over several model handles — one per GPU — from this one process (tgx_encode_batch_multi): byte-balanced
        shards, a host thread per handle, ids and offsets packed in sample order -> (ids uint32[T], offsets uint64[S+1])."""
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = offs.shape[0] - 1
        ids_out = np.empty(max(1, int(offs[-1] - offs[0])), np.uint32)
        out_offs = np.zeros(n + 1, np.uint64)
        handles = (C.c_void_p * len(models))(*[m._h for m in models])
        t = C.c_uint64()
        check(lib.tgx_encode_batch_multi(handles, len(models), ptr(flat) if flat.size else None, ptr(offs), n, float(dropout),
                                         seed & (2**64 - 1), ptr(ids_out), ids_out.size, ptr(out_offs), C.byref(t)))
        return ids_out[: t.value], out_offs

    def prune_alternatives(self) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """FlatTrie.prune_alternatives for this model's vocabulary, over the model's own table
        -> (always_keep u8[V], alt_offs u32[V+1], alt_ids u32[...]) — src/prune.rs:179-203."""
        V = self.vocab_size
        always_keep = np.zeros(V, np.uint8)
        alt_offs = np.zeros(V + 1, np.uint32)
        p = C.c_void_p()
        check(lib.tgx_model_prune_alternatives(self._h, ptr(always_keep), ptr(alt_offs), C.byref(p)))
        k = int(alt_offs[V])
        ids = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(max(k, 1),))[:k].copy()
        lib.tgx_free(p)
        return always_keep, alt_offs, ids

    def encode_corpus(self, corpus: NativeCorpus, dropout: float = 0.0, seed: int = 0) -> NativeResult:
        h = C.c_void_p()
        check(lib.tgx_encode_corpus(self._h, corpus._h, float(dropout), seed & (2**64 - 1), C.byref(h)))
        return NativeResult(h)

    def count_tokens(self, corpus: NativeCorpus, freq: np.ndarray | None = None) -> np.ndarray:
        if freq is None:
            freq = np.zeros(self.vocab_size, np.uint64)
        assert freq.dtype == np.uint64 and freq.shape[0] == self.vocab_size and freq.flags.c_contiguous
        check(lib.tgx_count_tokens(self._h, corpus._h, ptr(freq)))
        return freq

    def count_pairs(self, corpus: NativeCorpus) -> tuple[np.ndarray, np.ndarray]:
        keys, counts, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        check(lib.tgx_count_pairs(self._h, corpus._h, C.byref(keys), C.byref(counts), C.byref(n)))
        k = n.value
        ka = np.ctypeslib.as_array(C.cast(keys, C.POINTER(C.c_uint64)), shape=(max(k, 1),))[:k].copy()
        ca = np.ctypeslib.as_array(C.cast(counts, C.POINTER(C.c_uint64)), shape=(max(k, 1),))[:k].copy()
        lib.tgx_free(keys)
        lib.tgx_free(counts)
        return ka, ca

    def count_pairs_top(self, corpus: NativeCorpus, max_pairs: int) -> tuple[np.ndarray, np.ndarray, int]:
        """The max_pairs most frequent pairs, by descending count then ascending key -> (keys, counts, n_total)."""
        keys, counts, n, tot = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        check(lib.tgx_count_pairs_top(self._h, corpus._h, int(max_pairs), C.byref(keys), C.byref(counts),
                                      C.byref(n), C.byref(tot)))
        k = n.value
        ka = np.ctypeslib.as_array(C.cast(keys, C.POINTER(C.c_uint64)), shape=(max(k, 1),))[:k].copy()
        ca = np.ctypeslib.as_array(C.cast(counts, C.POINTER(C.c_uint64)), shape=(max(k, 1),))[:k].copy()
        lib.tgx_free(keys)
        lib.tgx_free(counts)
        return ka, ca, tot.value

    def estep(self, corpus: NativeCorpus, snippet_len: int = ESTEP_SNIPPET_LEN, dropout: float = 0.0,
              seed: int = 0, expected: np.ndarray | None = None) -> tuple[np.ndarray, float]:
        if expected is None:
            expected = np.zeros(self.vocab_size, np.float64)
        assert expected.dtype == np.float64 and expected.shape[0] == self.vocab_size
        z = C.c_double()
        check(lib.tgx_estep(self._h, corpus._h, snippet_len, float(dropout), seed & (2**64 - 1),
                            ptr(expected), C.byref(z)))
        return expected, z.value

    def common_prefix_search(self, s: bytes) -> list[tuple[int, int]]:
        buf = np.frombuffer(s, dtype=np.uint8)
        cap = len(s) + 1
        ids, lens = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        cnt = C.c_uint64()
        check(lib.tgx_common_prefix_search(self._h, ptr(buf) if len(s) else None, len(s), ptr(ids),
                                           ptr(lens), cap, C.byref(cnt)))
        return [(int(ids[i]), int(lens[i])) for i in range(cnt.value)]

    def last_kernel_times(self) -> dict[str, float]:
        names = (C.c_char_p * 8)()
        ms = (C.c_float * 8)()
        n = lib.tgx_last_kernel_times(self._h, names, ms, 8)
        return {names[i].decode(): float(ms[i]) for i in range(n)}

    def last_algorithmic_bytes(self) -> int:
        return lib.tgx_last_algorithmic_bytes(self._h)

    def last_encode_waves_per_cu(self) -> int:
        return lib.tgx_last_encode_waves_per_cu(self._h)

    def last_encode_redo_samples(self) -> int:
        return lib.tgx_last_encode_redo_samples(self._h)

    def score_values(self) -> int:
        return lib.tgx_model_score_values(self._h)

    def last_encode_hot_values(self) -> int:
        return lib.tgx_last_encode_hot_values(self._h)

    def last_encode_long_samples(self) -> int:
        return lib.tgx_last_encode_long_samples(self._h)

    def last_estep_pieces(self) -> int:
        return lib.tgx_last_estep_pieces(self._h)

    def last_estep_redo(self) -> int:
        return lib.tgx_last_estep_redo(self._h)

    def last_encode_corun_cus(self) -> int:
        return lib.tgx_last_encode_corun_cus(self._h)

    def encode_corun_timeouts(self) -> int:
        return lib.tgx_encode_corun_timeouts(self._h)


class FlatTrie:
    """Host-only build of the device trie layout (tgx_flat_trie_*), no GPU needed."""

    def __init__(self, tokens: list[bytes], scores):
        flat, offs = pack(tokens)
        sc = np.ascontiguousarray(scores, dtype=np.float64)
        h = C.c_void_p()
        check(lib.tgx_flat_trie_build(ptr(flat) if flat.size else None, ptr(offs), ptr(sc), len(tokens),
                                      C.byref(h)))
        self._h = h
        self._flat, self._offs, self._sc = flat, offs, sc

    def common_prefix_search8(self, s: bytes, max_hot: int = 6600):
        """The search over the 8-byte label-checked records of encode5_kernel -> (matches, stats)."""
        buf = np.frombuffer(s, dtype=np.uint8)
        cap = len(s) + 1
        ids, lens = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        nh, nc, cov = C.c_uint32(), C.c_uint64(), C.c_double()
        k = lib.tgx_flat_trie_search8(self._h, ptr(self._flat) if self._flat.size else None, ptr(self._offs), ptr(self._sc),
                                      max_hot, ptr(buf) if len(s) else None, len(s), ptr(ids), ptr(lens), cap,
                                      C.byref(nh), C.byref(nc), C.byref(cov))
        if k == 2**64 - 1:
            raise TokenGeeXError("8-byte records need fewer than 2^23 slots")
        return [(int(ids[i]), int(lens[i])) for i in range(k)], {"n_hot": nh.value, "n_cold": nc.value, "hot_coverage": cov.value}

    def __del__(self):
        if getattr(self, "_h", None):
            lib.tgx_flat_trie_free(self._h)
            self._h = None

    def common_prefix_search(self, s: bytes) -> list[tuple[int, int]]:
        buf = np.frombuffer(s, dtype=np.uint8)
        cap = len(s) + 1
        ids, lens = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        k = lib.tgx_flat_trie_search(self._h, ptr(buf) if len(s) else None, len(s), ptr(ids), ptr(lens), cap)
        return [(int(ids[i]), int(lens[i])) for i in range(k)]

    def stats(self) -> dict:
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        lib.tgx_flat_trie_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"n_slots": a.value, "n_nodes": b.value, "max_token_len": c.value,
                "fill": b.value / max(1, a.value)}

    @property
    def max_token_len(self) -> int:
        return self.stats()["max_token_len"]

    def table(self):
        """-> (check, base_flags, tokid) uint32 arrays of n_slots entries."""
        n = self.stats()["n_slots"]
        check, base, tokid = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        lib.tgx_flat_trie_copy(self._h, ptr(check), ptr(base), ptr(tokid))
        return check, base, tokid


    # ---- host half of `prune` (src/prune.rs) ----
    def prune_alternatives(self, tokens: list[bytes], scores) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """-> (always_keep u8[V], alt_offs u32[V+1], alt_ids u32[...]) — src/prune.rs:179-203."""
        flat, offs = pack(tokens)
        sc = np.ascontiguousarray(scores, dtype=np.float64)
        V = len(tokens)
        always_keep = np.zeros(V, np.uint8)
        alt_offs = np.zeros(V + 1, np.uint32)
        p = C.c_void_p()
        check(lib.tgx_prune_alternatives(self._h, ptr(flat) if flat.size else None, ptr(offs), ptr(sc), V,
                                         ptr(always_keep), ptr(alt_offs), C.byref(p)))
        k = int(alt_offs[V])
        ids = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(max(k, 1),))[:k].copy()
        lib.tgx_free(p)
        return always_keep, alt_offs, ids


def tok_hash_selftest(tokens: list[bytes]) -> tuple[int, int]:
    """-> (seed, mismatches) of the bytes -> id table built for `tokens` (host only)."""
    flat, offs = pack(tokens)
    seed, bad = C.c_uint32(), C.c_uint64()
    check(lib.tgx_tok_hash_selftest(ptr(flat) if flat.size else None, ptr(offs), len(tokens), C.byref(seed), C.byref(bad)))
    return seed.value, bad.value


def digamma(x: float) -> float:
    return lib.tgx_digamma(x)


def prune_m_step(expected: np.ndarray, keep: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """run_m_step — src/prune.rs:124-170 -> (surviving ids, new scores)."""
    expected = np.ascontiguousarray(expected, dtype=np.float64)
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    V = expected.shape[0]
    idx, sc, n = np.zeros(max(V, 1), np.uint32), np.zeros(max(V, 1), np.float64), C.c_uint32()
    check(lib.tgx_prune_m_step(ptr(expected), ptr(keep), V, ptr(idx), ptr(sc), C.byref(n)))
    return idx[:n.value].copy(), sc[:n.value].copy()


def prune_select(freq, keep, always_keep, alt_offs, alt_ids, scores, n_samples: int,
                 pruned_size: int) -> np.ndarray:
    """Loss-based selection — src/prune.rs:246-318 -> ids of the pruned vocabulary, final order."""
    freq = np.ascontiguousarray(freq, dtype=np.uint64)
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    always_keep = np.ascontiguousarray(always_keep, dtype=np.uint8)
    alt_offs = np.ascontiguousarray(alt_offs, dtype=np.uint32)
    alt_ids = np.ascontiguousarray(alt_ids, dtype=np.uint32)
    scores = np.ascontiguousarray(scores, dtype=np.float64)
    V = freq.shape[0]
    out, n = np.zeros(max(V, 1), np.uint32), C.c_uint32()
    check(lib.tgx_prune_select(ptr(freq), ptr(keep), ptr(always_keep), ptr(alt_offs),
                               ptr(alt_ids) if alt_ids.size else None, ptr(scores), V, n_samples,
                               pruned_size, ptr(out), C.byref(n)))
    return out[:n.value].copy()


def dropout_u01(seed: int, sample: int, pos: int, length: int) -> float:
    return lib.tgx_dropout_u01_host(seed, sample, pos, length)
