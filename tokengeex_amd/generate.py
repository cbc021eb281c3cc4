"""`generate` — the reference's VocabularyGenerator (src/generate.rs:12-243), the step that produces the
initial vocabulary `prune` starts from.

`feed` counts, for every char-aligned substring of at most max_token_length bytes, the samples it occurs in
(src/generate.rs:54-139).  With a device (the default) that is `tgx_substring_df` (csrc/generate.hip): one
lane per byte position hashes its windows, a radix sort groups equal substrings, a run pass counts distinct
samples; the split regex reaches the device as byte ranges, the allow regex — a pure function of the
candidate — is applied on the host to the DISTINCT substrings that come back, and the added / suggested
tokens are searched on the host as in the reference.  With `device=None` the same counts come from
`_feed_host`, the per-sample Python restatement of the reference's loops: the checker of the device path
(tests) and the path for max_token_length > 16.

Differences from the reference, both forced: the reference draws `rng.gen_range(0.0..1.0)` from an unseeded
thread RNG for `insert_probability` (src/generate.rs:88,112,126), here a counter hash of (seed, sample,
FNV-1a-64 of the candidate's bytes) decides — the same function on the device and on the host — so runs are
reproducible; tokens of equal frequency / score are ordered by their bytes (`sort_unstable_by` leaves their
order unspecified), so the device path and the host path give the same vocabulary.  The split regex
(fancy_regex) is taken as a compiled Python pattern.
"""
from __future__ import annotations

import math

import numpy as np

from .merge import compile_rust_regex

_M64 = (1 << 64) - 1


def _fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def _u01(seed: int, sample: int, token: str) -> float:
    """tgx_generate_u01(seed, sample, FNV-1a-64(token bytes)) — include/tgx.h, csrc/generate.hip."""
    h = _fnv1a64(token.encode("utf-8", "surrogatepass"))
    x = (seed ^ (sample * 0x9E3779B97F4A7C15) ^ (h * 0xC2B2AE3D27D4EB4F)) & _M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & _M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & _M64
    x ^= x >> 31
    return (x >> 11) * (1.0 / 9007199254740992.0)


class VocabularyGenerator:
    """VocabularyGenerator::new(max_token_length, insert_probability, split, allow, added, suggested)."""

    def __init__(self, max_token_length: int, insert_probability: float, split=None, allow=None,
                 added_tokens=(), suggested_tokens=(), seed: int = 0, device: int | None = 0):
        self.device = device  # None: the host restatement (checker); an int: substring counting on that GPU
        self.max_token_length = int(max_token_length)
        self.insert_probability = float(insert_probability)
        self.split = split
        self.allow = compile_rust_regex(allow) if isinstance(allow, str) else allow
        self.added_tokens, self.suggested_tokens = list(added_tokens), list(suggested_tokens)
        self.seed = seed
        self._fed = 0
        self.frequencies: dict[str, int] = {}
        for t in self.added_tokens + self.suggested_tokens:  # src/generate.rs:33-41
            self.frequencies[t] = self.frequencies.get(t, 0) + 1

    def _keep(self, sample_index: int, token: str) -> bool:
        return self.insert_probability >= 1.0 or _u01(self.seed, sample_index, token) < self.insert_probability

    def _candidates(self, part: str, sample_index: int, out: set):
        """src/generate.rs:72-96 / 99-120: every char-aligned substring of at most max_token_length BYTES."""
        n = len(part)
        blen = [len(c.encode("utf-8", "surrogatepass")) for c in part]
        for i in range(n):
            total = 0
            for j in range(i, n):
                total += blen[j]
                if total > self.max_token_length:
                    break
                cand = part[i:j + 1]
                if (self.allow is None or self.allow.search(cand)) and self._keep(sample_index, cand):
                    out.add(cand)

    def feed(self, samples: list[str]) -> None:
        """feed(&mut self, samples) — src/generate.rs:54-139: DOCUMENT frequencies (one count per sample)."""
        if self.device is None or self.max_token_length > 16:
            return self._feed_host(samples)
        from . import _lib
        first = self._fed
        self._fed += len(samples)
        enc = [s.encode("utf-8", "surrogatepass") for s in samples]
        flat, offs = _lib.pack(enc)
        if self.split is None:
            keep = np.flatnonzero(offs[1:] > offs[:-1])
            pb, pe, ps = offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32)
        else:  # the parts are the split regex's matches, as byte ranges (src/generate.rs:67-70)
            pb_l, pe_l, ps_l = [], [], []
            for i, (s, b) in enumerate(zip(samples, enc)):
                if len(b) == len(s):  # ASCII: char offsets are byte offsets
                    spans = [mt.span() for mt in self.split.finditer(s)]
                else:
                    cum = np.concatenate([[0], np.cumsum([len(c.encode("utf-8", "surrogatepass")) for c in s])])
                    spans = [(int(cum[a]), int(cum[z])) for a, z in (mt.span() for mt in self.split.finditer(s))]
                base = int(offs[i])
                last = 0
                for a, z in spans:
                    if z > a and a >= last:
                        pb_l.append(base + a)
                        pe_l.append(base + z)
                        ps_l.append(i)
                        last = z
            pb, pe, ps = np.array(pb_l, np.uint64), np.array(pe_l, np.uint64), np.array(ps_l, np.uint32)
        extra = self.added_tokens + self.suggested_tokens
        extra_set = set(extra)
        if pb.size:
            # the keep rule sees the global sample index (the device packs it into 27 bits)
            pos, ln, df, _ = _lib.substring_df(flat, pb, pe, (ps.astype(np.uint64) + first).astype(np.uint32),
                                               self.max_token_length, self.insert_probability, self.seed, self.device)
            raw = flat.tobytes()
            for p_, l_, d_ in zip(pos.tolist(), ln.tolist(), df.tolist()):
                cand = raw[p_:p_ + l_].decode("utf-8", "surrogatepass")
                if cand in extra_set:
                    continue  # counted below: a sample's set holds the token once, whichever rule put it there
                if self.allow is None or self.allow.search(cand):
                    self.frequencies[cand] = self.frequencies.get(cand, 0) + d_
        # added and suggested tokens: any occurrence in the sample counts (src/generate.rs:122-131).  The keep
        # rule is a function of (seed, sample, token), so a window of the same text made the same decision and
        # the union of both rules is this one.
        for t in dict.fromkeys(extra):
            if not t:
                continue
            n = sum(1 for i, sample in enumerate(samples) if t in sample and self._keep(first + i, t))
            if n:
                self.frequencies[t] = self.frequencies.get(t, 0) + n

    def _feed_host(self, samples: list[str]) -> None:
        """The reference's loops per sample, in Python: the checker of the device path."""
        for sample in samples:
            idx = self._fed
            self._fed += 1
            toks: set = set()
            if self.split is not None:
                for m in self.split.finditer(sample):
                    self._candidates(m.group(0), idx, toks)
            else:
                self._candidates(sample, idx, toks)
            for t in self.added_tokens + self.suggested_tokens:  # src/generate.rs:122-131
                if t and t in sample and self._keep(idx, t):
                    toks.add(t)
            for t in toks:
                self.frequencies[t] = self.frequencies.get(t, 0) + 1

    def current_size(self) -> int:
        return len(self.frequencies)

    def generate(self, size: int) -> list[tuple[bytes, float, bool]]:
        """generate(&mut self, size) — src/generate.rs:148-243 -> [(value, log-probability, keep)]."""
        # (ties: ascending token bytes — the reference's sort_unstable_by leaves their order unspecified; a fixed
        # rule makes the result independent of the order in which the counts arrived)
        frequent = sorted(self.frequencies.items(), key=lambda kv: (-kv[1], kv[0].encode("utf-8", "surrogatepass")))
        highest = frequent[0][1] if frequent else 1
        seen = {bytes([b]) for b in range(255)}
        vocab = [(bytes([b]), float(highest), True) for b in range(255)]  # bytes 0..254, src/generate.rs:164-169
        for tok, keep in [(t, True) for t in self.added_tokens] + [(t, False) for t in self.suggested_tokens]:
            if len(vocab) >= size:
                break
            b = tok.encode("utf-8")
            if b not in seen and len(b) > 1:
                seen.add(b)
                vocab.append((b, float(self.frequencies[tok] * len(b)), keep))
        for tok, freq in frequent:
            if len(vocab) >= size:
                break
            b = tok.encode("utf-8", "surrogatepass")
            if b not in seen and len(b) > 1:
                seen.add(b)
                vocab.append((b, float(freq * len(b)), False))
        vocab.sort(key=lambda t: (-t[1], t[0]))
        logsum = math.log(sum(t[1] for t in vocab))  # logprobs, src/generate.rs:245-251
        out = []
        for value, score, keep in vocab:
            lp = math.log(score) - logsum if score > 0 else float("nan")
            if lp != lp or lp in (float("inf"), float("-inf")) or lp == 0.0:
                raise ValueError(f"Vocabulary generation: invalid frequency for token {value!r}: {lp}")
            out.append((value, lp, keep))
        return out
