"""Host-side mirror of the reference's `Tokenizer` (src/tokenizer.rs:6-435) and of
its PyO3 surface (bindings/python/src/lib.rs:41-223, tokengeex.pyi:10-255).

Same method names, argument order and error strings as the PyO3 class; the hot
path (Model::encode over a batch) goes through the C ABI to the HIP kernels, the
rest (special-token splitting, processors, decode, JSON) is host glue in Python.
"""
from __future__ import annotations

import base64
import json
import random
import re
import unicodedata

import numpy as np

from . import _lib
from . import _tgxfast as _fast  # csrc/pyfast.c, built by tokengeex_amd/build.py beside libtgx.so
from ._lib import TokenGeeXError

SERIALIZATION_VERSION = "2.0"  # src/tokenizer.rs:347


# ---- processors: src/processor.rs ---------------------------------------------

class CrlfProcessor:
    """src/processor.rs:37-54: plain "\\r\\n" -> "\\n"; postprocess is identity."""

    def preprocess(self, s: str) -> str:
        return s.replace("\r\n", "\n")

    def postprocess(self, s: str) -> str:
        return s

    def to_json(self):
        return {"type": "crlf"}


class UnicodeProcessor:
    """src/processor.rs:124-137.  The reference delegates to the
    unicode-normalization crate; here unicodedata does the same UAX #15 forms
    (parity with the crate's Unicode version is not pinned by any reference test)."""

    FORMS = {"nfc": "NFC", "nfd": "NFD", "nfkc": "NFKC", "nfkd": "NFKD"}

    def __init__(self, form: str):
        if form not in self.FORMS:
            raise TokenGeeXError(f"unknown variant `{form}`, expected one of `nfc`, `nfd`, `nfkc`, `nfkd`",
                                 _lib.ERR_JSON)
        self.form = form

    def preprocess(self, s: str) -> str:
        return unicodedata.normalize(self.FORMS[self.form], s)

    def postprocess(self, s: str) -> str:
        return s

    def to_json(self):
        return {"type": "unicode", "form": self.form}


def _processor_from_json(obj):
    # untagged enum, src/processor.rs:13-18: Crlf is tried first, then Unicode
    if not isinstance(obj, dict):
        raise TokenGeeXError("data did not match any variant of untagged enum ProcessorWrapper", _lib.ERR_JSON)
    if obj.get("type") == "crlf" and "form" not in obj:
        return CrlfProcessor()
    if "form" in obj:
        return UnicodeProcessor(obj["form"])
    raise TokenGeeXError("data did not match any variant of untagged enum ProcessorWrapper", _lib.ERR_JSON)


# ---- special token splitter: src/tokenizer.rs:299-347 ------------------------------

def split_special_tokens(text: str, special_tokens: list[str]) -> list[tuple[str, bool]]:
    """Earliest char position wins; at one position the first special in list order
    wins (not the longest).  A regex alternation has exactly these semantics."""
    if not text:
        return []
    if not special_tokens:
        return [(text, False)]
    pat = re.compile("|".join(re.escape(t) for t in special_tokens))
    out, cursor = [], 0
    for mt in pat.finditer(text):
        if mt.start() > cursor:
            out.append((text[cursor:mt.start()], False))
        out.append((mt.group(0), True))
        cursor = mt.end()
    if cursor < len(text):
        out.append((text[cursor:], False))
    return out


def _split_rows(ids: np.ndarray, offs: np.ndarray) -> list[list[int]]:
    flat = ids.tolist()  # one conversion for the batch, then list slices
    o = offs.tolist()
    return [flat[o[i]:o[i + 1]] for i in range(len(o) - 1)]


# ---- Tokenizer ---------------------------------------------------------------------

class Tokenizer:
    """tokengeex.Tokenizer (bindings/python/src/lib.rs:11-224)."""

    def __init__(self, vocab: list[tuple[bytes, float, bool]] | None = None, processors=None,
                 special_tokens=None, device: int = 0):
        # Model: src/model.rs:8-12
        self._vocab: list[tuple[bytes, float, bool]] = [(bytes(v), float(s), bool(k)) for v, s, k in (vocab or [])]
        self._token_to_ids: dict[bytes, int] = {}
        for i, (v, _, _) in enumerate(self._vocab):
            self._token_to_ids[v] = i  # later duplicates overwrite, src/model.rs:21
        self._processors = list(processors or [])
        self._special_tokens: list[str] = []
        self._special_tokens_map: dict[str, int] = {}
        self._device = device
        self._native: _lib.NativeModel | None = None
        self.seed: int | None = None  # dropout seed; None = fresh random seed per call
        self.add_special_tokens(list(special_tokens or []))

    # -- native model (lazy: building it needs the GPU) --
    def _model(self) -> _lib.NativeModel:
        if self._native is None:
            self._native = _lib.NativeModel([v for v, _, _ in self._vocab],
                                            np.array([s for _, s, _ in self._vocab], dtype=np.float64),
                                            self._device)
        return self._native

    def _seed(self, dropout: float) -> int:
        if dropout <= 0.0:
            return 0
        return self.seed if self.seed is not None else random.getrandbits(64)

    def _preprocess(self, s: str) -> str:
        for p in self._processors:  # src/tokenizer.rs:79-82
            s = p.preprocess(s)
        return s

    # -- encode: src/tokenizer.rs:65-123 --
    def encode(self, text: str, dropout: float) -> list[int]:
        return self.encode_batch([text], dropout)[0]

    def encode_ordinary(self, text: str, dropout: float) -> list[int]:
        return self.encode_ordinary_batch([text], dropout)[0]

    def _native_front(self) -> bool:
        """The packed-buffer front end covers both processors of the reference (src/processor.rs): CRLF (csrc/frontback.cpp)
        and, from round 4, the Unicode normalisation forms (csrc/unicode_norm.cpp; `unicodedata` stays as the checker of the
        tests and in UnicodeProcessor.preprocess for single strings)."""
        return all(isinstance(p, (CrlfProcessor, UnicodeProcessor)) for p in self._processors)

    def _preprocess_flat(self, flat: np.ndarray, offs: np.ndarray):
        """The processors, in order (src/tokenizer.rs:79-82), over a packed batch of segments."""
        for p in self._processors:
            if isinstance(p, CrlfProcessor):
                flat, offs = _lib.pack_segments(flat, np.ascontiguousarray(offs[:-1]), np.ascontiguousarray(offs[1:]), None, True)
            else:
                flat, offs = _lib.normalize_flat(p.form, flat, offs)
        return flat, offs

    def _rows_native(self, texts: list[str], dropout: float, ordinary: bool) -> list[list[int]]:
        """list[str] -> list[list[int]] with both ends in native code (csrc/pyfast.c; bindings/python/src/lib.rs:51-69 builds
        the same shapes in Rust): the samples' UTF-8 packed by host threads straight from the strings, the rows built from
        shared int objects (one per id, made once per tokenizer) — no bytes object per sample, no int object per token."""
        text_b, offs_b = _fast.pack_strs(texts)
        flat = np.frombuffer(text_b, dtype=np.uint8)
        offs = np.frombuffer(offs_b, dtype=np.uint64)
        ids, o = self.encode_batch_flat(flat, offs, dropout, ordinary=ordinary)
        n_ids = self.vocab_size()
        if getattr(self, "_int_cache", None) is None or len(self._int_cache) != n_ids:
            self._int_cache = list(range(n_ids))
        return _fast.rows_from_flat(np.ascontiguousarray(ids, np.uint32), np.ascontiguousarray(o, np.uint64), self._int_cache)

    def encode_ordinary_batch(self, texts: list[str], dropout: float) -> list[list[int]]:
        if self._native_front():
            return self._rows_native(texts, dropout, True)
        segs = [self._preprocess(t).encode("utf-8") for t in texts]
        ids, offs = self._encode_segments(segs, dropout)
        return [ids[int(offs[i]):int(offs[i + 1])].tolist() for i in range(len(texts))]

    def encode_batch_flat(self, flat: np.ndarray, offs: np.ndarray, dropout: float = 0.0, ordinary: bool = False):
        """encode_batch / encode_ordinary_batch over a packed batch of UTF-8 samples (uint8 flat, uint64
        offsets[S+1]) -> (ids uint32[T], offsets uint64[S+1]): special-token split, CRLF processor, encode and
        the assembly of the ids all run on packed buffers in native code (src/tokenizer.rs:65-123)."""
        if not self._native_front():
            raise TokenGeeXError("encode_batch_flat: a processor without a packed-buffer form", _lib.ERR_UNSUPPORTED)
        flat = np.ascontiguousarray(flat, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        only_crlf = all(isinstance(p, CrlfProcessor) for p in self._processors)  # then the packing pass does it on the way
        crlf = only_crlf and len(self._processors) > 0
        n = offs.shape[0] - 1
        if ordinary or not self._special_tokens:
            if n and self._processors:
                flat, offs = self._preprocess_flat(flat, offs)
            return self.encode_ordinary_batch_flat(flat, offs, dropout) if n else (np.zeros(0, np.uint32), np.zeros(1, np.uint64))
        seg_offs, sb, se, ss = _lib.split_specials_flat(flat, offs, [t.encode("utf-8") for t in self._special_tokens])
        pflat, poffs = _lib.pack_segments(flat, sb, se, ss, crlf)
        if not only_crlf and poffs.shape[0] > 1:
            pflat, poffs = self._preprocess_flat(pflat, poffs)
        if poffs.shape[0] > 1:
            ids, id_offs = self.encode_ordinary_batch_flat(pflat, poffs, dropout)
        else:
            ids, id_offs = np.zeros(0, np.uint32), np.zeros(1, np.uint64)
        return _lib.assemble_ids(seg_offs, ss, ids, id_offs, self.base_vocab_size())

    def encode_batch(self, texts: list[str], dropout: float) -> list[list[int]]:
        if not self._special_tokens:
            return self.encode_ordinary_batch(texts, dropout)
        if self._native_front():
            return self._rows_native(texts, dropout, False)
        base = self.base_vocab_size()
        plan, segs = [], []  # per text: list of (special id | -1)
        for t in texts:
            row = []
            for sub, is_special in split_special_tokens(t, self._special_tokens):
                if is_special:
                    row.append(base + self._special_tokens_map[sub])  # src/tokenizer.rs:70-77
                else:
                    row.append(-1)
                    segs.append(self._preprocess(sub).encode("utf-8"))
            plan.append(row)
        ids, offs = self._encode_segments(segs, dropout)
        out, k = [], 0
        for row in plan:
            cur: list[int] = []
            for item in row:
                if item >= 0:
                    cur.append(item)
                else:
                    cur.extend(ids[int(offs[k]):int(offs[k + 1])].tolist())
                    k += 1
            out.append(cur)
        return out

    def _encode_segments(self, segs: list[bytes], dropout: float):
        if not segs:
            return np.zeros(0, np.uint32), np.zeros(1, np.uint64)
        flat, offs = _lib.pack(segs)
        return self.encode_ordinary_batch_flat(flat, offs, dropout)

    def encode_ordinary_batch_flat(self, flat: np.ndarray, offs: np.ndarray, dropout: float = 0.0):
        """Flat-buffer entry point the reference lacks: packed bytes + offsets in,
        (ids uint32[T], offsets uint64[S+1]) out, no per-sample Python objects."""
        res = self._model().encode_batch_flat(flat, offs, dropout, self._seed(dropout))
        try:
            return res.ids(), res.offsets()
        finally:
            res.free()

    # -- decode: src/tokenizer.rs:126-187, src/model.rs:146-160 --
    def _model_decode(self, ids) -> str:
        buf = bytearray()
        n = len(self._vocab)
        for i in ids:
            if i >= n:
                raise TokenGeeXError(f"token id {i} is out of bounds", _lib.ERR_TOKEN_ID_OOB)
            buf += self._vocab[i][0]
        return bytes(buf).decode("utf-8", "replace")  # String::from_utf8_lossy

    def _postprocess(self, s: str) -> str:
        for p in reversed(self._processors):
            s = p.postprocess(s)
        return s

    def decode(self, ids: list[int], include_special_tokens: bool) -> str:
        n = len(self._vocab)
        out, run = [], []
        for i in ids:
            if i >= n:
                out.append(self._postprocess(self._model_decode(run)))
                run = []
                k = i - n
                if k >= len(self._special_tokens):
                    raise TokenGeeXError(f"token id {i} is out of bounds", _lib.ERR_TOKEN_ID_OOB)
                if include_special_tokens:
                    out.append(self._special_tokens[k])
            else:
                run.append(i)
        out.append(self._postprocess(self._model_decode(run)))
        return "".join(out)

    def _vocab_packed(self):
        if getattr(self, "_packed", None) is None or self._packed[0] != (len(self._vocab), len(self._special_tokens)):
            vf, vo = _lib.pack([v for v, _, _ in self._vocab])
            sf, so = _lib.pack([t.encode("utf-8") for t in self._special_tokens])
            self._packed = ((len(self._vocab), len(self._special_tokens)), vf, vo, sf, so)
        return self._packed[1:]

    def decode_batch_flat(self, ids: np.ndarray, offs: np.ndarray, include_special_tokens: bool):
        """decode_batch over packed ids (uint32 ids, uint64 offsets[S+1]) -> (utf-8 bytes, offsets[S+1]), in
        native code: src/tokenizer.rs:126-187 incl. String::from_utf8_lossy per run of base ids."""
        vf, vo, sf, so = self._vocab_packed()
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        return _lib.decode_batch_flat(vf, vo, len(self._vocab), sf, so, len(self._special_tokens), ids, offs,
                                      include_special_tokens)

    def decode_batch(self, ids: list[list[int]], include_special_tokens: bool) -> list[str]:
        try:
            flat = np.fromiter((i for row in ids for i in row), dtype=np.uint32, count=sum(len(r) for r in ids))
        except (OverflowError, ValueError):  # negative or > 2^32 - 1: the per-sample path names the id
            return [self.decode(x, include_special_tokens) for x in ids]
        offs = np.zeros(len(ids) + 1, np.uint64)
        if ids:
            np.cumsum(np.fromiter((len(r) for r in ids), dtype=np.uint64, count=len(ids)), out=offs[1:])
        text, to = self.decode_batch_flat(flat, offs, include_special_tokens)
        raw = text.tobytes()
        return [raw[int(to[i]):int(to[i + 1])].decode("utf-8") for i in range(len(ids))]

    # -- id / token queries: src/tokenizer.rs:189-259 --
    def token_to_id(self, token: bytes) -> int | None:
        r = self.base_token_to_id(token)
        if r is not None:
            return r
        try:
            return self.special_token_to_id(bytes(token).decode("utf-8"))
        except UnicodeDecodeError:
            return None

    def base_token_to_id(self, token: bytes) -> int | None:
        return self._token_to_ids.get(bytes(token))

    def special_token_to_id(self, token: str) -> int | None:
        k = self._special_tokens_map.get(token)
        return None if k is None else k + len(self._vocab)

    def id_to_token(self, id: int) -> bytes | None:
        s = self.id_to_special_token(id)
        if s is not None:
            return s.encode("utf-8")
        b = self.id_to_base_token(id)
        return None if b is None else b[0]

    def id_to_base_token(self, id: int) -> tuple[bytes, float] | None:
        if 0 <= id < len(self._vocab):
            return self._vocab[id][0], self._vocab[id][1]
        return None

    def id_to_special_token(self, id: int) -> str | None:
        k = id - len(self._vocab)
        if 0 <= k < len(self._special_tokens):
            return self._special_tokens[k]
        return None

    def is_special(self, id: int) -> bool:
        return self.id_to_special_token(id) is not None

    def is_base(self, id: int) -> bool:
        return id < len(self._vocab)

    def add_special_tokens(self, tokens: list[str]) -> None:
        for t in tokens:  # src/tokenizer.rs:39-53: existing specials are ignored
            if t == "":
                raise ValueError("empty special token (the reference's splitter would never advance)")
            if t in self._special_tokens_map:
                continue
            self._special_tokens_map[t] = len(self._special_tokens)
            self._special_tokens.append(t)

    def add_base_tokens(self, tokens: list[tuple[bytes, float, bool]]) -> None:
        """Tokenizer::add_base_tokens -> Model::add_tokens, src/model.rs:184-194."""
        for v, s, k in tokens:
            self._token_to_ids[bytes(v)] = len(self._vocab)
            self._vocab.append((bytes(v), float(s), bool(k)))
        if self._native is not None:  # mutation = build a new device handle
            self._native.free()
            self._native = None

    def special_tokens(self) -> list[str]:
        return list(self._special_tokens)

    def vocab_size(self) -> int:
        return len(self._vocab) + len(self._special_tokens)

    def base_vocab_size(self) -> int:
        return len(self._vocab)

    def special_vocab_size(self) -> int:
        return len(self._special_tokens)

    def common_prefix_search(self, text: str) -> list[int]:
        return [i for i, _ in self._model().common_prefix_search(text.encode("utf-8"))]

    # -- serialisation: src/tokenizer.rs:261-297, 349-435, src/lib.rs:109-204 --
    def to_string(self) -> str:
        vocab = []
        for v, s, k in self._vocab:
            try:
                e = {"value": v.decode("utf-8"), "score": s}
            except UnicodeDecodeError:
                e = {"value": base64.b64encode(v).decode("ascii").rstrip("="), "score": s, "encoded": True}
            if k:
                e["keep"] = True
            vocab.append(e)
        return json.dumps({"version": SERIALIZATION_VERSION, "special_tokens": self._special_tokens,
                           "processors": [p.to_json() for p in self._processors], "vocab": vocab},
                          ensure_ascii=False, separators=(",", ":"))

    def save(self, filename: str) -> None:
        with open(filename, "w", encoding="utf-8") as f:
            f.write(self.to_string())

    @staticmethod
    def from_str(data: str, device: int = 0) -> "Tokenizer":
        try:
            obj = json.loads(data)
        except json.JSONDecodeError as e:
            raise TokenGeeXError(str(e), _lib.ERR_JSON) from None
        if not isinstance(obj, dict):
            raise TokenGeeXError("invalid type: expected struct Tokenizer", _lib.ERR_JSON)
        allowed = ("version", "special_tokens", "processors", "vocab")
        for key in obj:
            if key not in allowed:  # src/tokenizer.rs:414-419
                raise TokenGeeXError(f"unknown field `{key}`, expected one of `version`, `special_tokens`, "
                                     "`processors`, `vocab`", _lib.ERR_JSON)
        if "version" not in obj:
            raise TokenGeeXError("missing field `version`", _lib.ERR_JSON)
        if obj["version"] != SERIALIZATION_VERSION:  # src/tokenizer.rs:423-429
            raise TokenGeeXError(f"unsupported version: {obj['version']}", _lib.ERR_JSON)
        vocab = []
        for e in obj.get("vocab", []):
            for key in e:
                if key not in ("value", "score", "encoded", "keep"):  # src/lib.rs:173-175
                    raise TokenGeeXError(f"unknown field `{key}`, expected one of `value`, `score`, "
                                         "`encoded`, `keep`", _lib.ERR_JSON)
            if "value" not in e:
                raise TokenGeeXError("missing field `token`", _lib.ERR_JSON)  # sic, src/lib.rs:189
            if "score" not in e:
                raise TokenGeeXError("missing field `score`", _lib.ERR_JSON)
            if e.get("encoded", False):
                v = e["value"]
                raw = base64.b64decode(v + "=" * (-len(v) % 4))  # STANDARD_NO_PAD, src/lib.rs:8
            else:
                raw = e["value"].encode("utf-8")
            vocab.append((raw, float(e["score"]), bool(e.get("keep", False))))
        procs = [_processor_from_json(p) for p in obj.get("processors", [])]
        return Tokenizer(vocab, procs, obj.get("special_tokens", []), device=device)

    @staticmethod
    def from_file(filepath: str, device: int = 0) -> "Tokenizer":
        try:
            with open(filepath, encoding="utf-8") as f:
                data = f.read()
        except OSError as e:
            raise TokenGeeXError(str(e), _lib.ERR_IO) from None
        return Tokenizer.from_str(data, device=device)

    def __getstate__(self):
        return self.to_string().encode("utf-8")

    def __setstate__(self, state):
        other = Tokenizer.from_str(bytes(state).decode("utf-8"))
        self.__dict__.update(other.__dict__)

    # -- access for the training-loop entry points --
    def vocab(self) -> list[tuple[bytes, float, bool]]:
        return list(self._vocab)

    def native_model(self) -> _lib.NativeModel:
        return self._model()
