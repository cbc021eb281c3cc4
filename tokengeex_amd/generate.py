"""`generate` — the reference's VocabularyGenerator (src/generate.rs:12-243), the step that produces the
initial vocabulary `prune` starts from.

`feed` counts, for every char-aligned substring of at most max_token_length bytes, the samples it occurs in
(src/generate.rs:54-139).  Here that is `tgx_substring_df[_top]` (csrc/generate.hip): one lane per byte position
hashes its windows, a radix sort groups equal substrings, a run pass counts distinct samples; the split regex
reaches the device as byte ranges, the allow regex — a pure function of the candidate — is applied on the host to
the substrings that come back, and the added / suggested tokens are searched on the host as in the reference.
There is no host fallback: without a GPU (or for max_token_length > 32) the calls fail.  The checker of this path
is the CPU oracle's restatement of the generator (test infrastructure, never imported here).

Round 3: `top_k`.  `generate(size)` keeps the most frequent substrings (src/generate.rs:150-152, 199-213), so only
the top_k most frequent ones of a device pass leave the device (the pass over 64 MiB of text finds 254 M distinct
substrings; turning each into a Python string, an allow-regex call and a dictionary entry was what bounded `feed`:
0.4 MB/s).  Samples are buffered by `feed` and counted in as few device passes as fit (`generate` and
`frequencies` trigger them), because frequencies add over passes but a truncated pass only bounds what it cut off:
every pass reports the frequency of the first substring it cut, `generate` adds those bounds up and REFUSES
(TokenGeeXError) to return a vocabulary that a cut-off substring could have changed — a larger top_k, or fewer
passes, make it exact again.  With top_k = None everything is returned and nothing can be uncertain.

Differences from the reference, both forced: the reference draws `rng.gen_range(0.0..1.0)` from an unseeded
thread RNG for `insert_probability` (src/generate.rs:88,112,126), here a counter hash of (seed, sample, the
occurrence: byte offset in the sample << 8 | byte length) decides, one draw per occurrence as in the reference's
loops — tgx_generate_u01, the same function on the device and in the
CPU oracle — so runs are reproducible; tokens of equal frequency / score are ordered by their bytes
(`sort_unstable_by` leaves their order unspecified).  The split regex (fancy_regex) is taken as a compiled Python
pattern.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib
from .merge import compile_rust_regex

_M64 = (1 << 64) - 1


def pass_bytes(max_token_length: int) -> int:
    """UTF-8 bytes of text per device pass: tgx_substring_df takes < 2^32 kept windows per call, a byte position starts at
    most max_token_length of them (ASCII text, insert_probability 1; 10.5 per byte measured on the mixed bench corpus at 16),
    and a window costs 32 bytes of device scratch.  256 MiB at most."""
    return min(256 << 20, int(0.9 * (1 << 32)) // (int(max_token_length) + 2))



def _fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def _u01(seed: int, sample: int, offset: int, length: int, added: bool = False) -> float:
    """tgx_generate_u01(seed, sample, occurrence) — include/tgx.h, csrc/generate.hip: the draw of ONE occurrence (byte
    offset in the sample, byte length); `added`: the draws of the added / suggested tokens' matches (src/generate.rs:122-127)."""
    return _lib.generate_u01(seed, sample, (offset << 8) | length | ((1 << 63) if added else 0))


class VocabularyGenerator:
    """VocabularyGenerator::new(max_token_length, insert_probability, split, allow, added, suggested)."""

    def __init__(self, max_token_length: int, insert_probability: float, split=None, allow=None,
                 added_tokens=(), suggested_tokens=(), seed: int = 0, device: int = 0, top_k: int | None = None):
        if device is None:
            raise _lib.TokenGeeXError("VocabularyGenerator counts on a GPU: there is no host path")
        if int(max_token_length) > 32:  # (the reference has no limit, src/generate.rs:78-83; its CLI's default is 24, src/cli.rs:675)
            raise _lib.TokenGeeXError("VocabularyGenerator: max_token_length must be <= 32 (tgx_substring_df)")
        if _lib.device_count() <= int(device):
            raise _lib.TokenGeeXError("VocabularyGenerator: no usable HIP device (gfx950 required); there is no CPU fallback")
        self.device = int(device)
        self.max_token_length = int(max_token_length)
        self.insert_probability = float(insert_probability)
        self.split = split
        self.allow = compile_rust_regex(allow) if isinstance(allow, str) else allow
        self.added_tokens, self.suggested_tokens = list(added_tokens), list(suggested_tokens)
        self.seed = seed
        self.top_k = int(top_k) if top_k else 0
        self._fed = 0                       # samples seen (the keep rule hashes the global sample index)
        self._pending: list[str] = []       # fed, not yet counted ...
        self._pending_enc: list[bytes] = []  # ... and their UTF-8 (a pass is sized by ENCODED bytes)
        self._pending_bytes = 0
        self._pending_first = 0
        self._freq: dict[str, int] = {}
        self._bound_seen: dict[str, int] = {}  # per substring: the cut-off bounds of the passes that DID return it
        self._bound_total = 0               # sum over passes of the frequency of the first substring cut off
        self.passes = 0
        for t in self.added_tokens + self.suggested_tokens:  # src/generate.rs:33-41
            self._freq[t] = self._freq.get(t, 0) + 1

    def _keep(self, sample_index: int, offset: int, length: int, added: bool = False) -> bool:
        return self.insert_probability >= 1.0 or _u01(self.seed, sample_index, offset, length, added) < self.insert_probability

    def _extra_kept(self, sample_index: int, sample: str, b: bytes, parts, t: str, tb: bytes) -> bool:
        """Does sample's set hold the added / suggested token t (src/generate.rs:72-127)?  Either a window that spells it was
        kept (every char-aligned occurrence inside a part is a candidate with a draw of its own, if the token is short enough
        and allowed) or one of its successive non-overlapping matches was (str::match_indices, a draw per match)."""
        if self.insert_probability >= 1.0:
            return t in sample
        if len(tb) <= self.max_token_length and (self.allow is None or self.allow.search(t)):
            for a, z in parts:
                o = b.find(tb, a, z)
                while o >= 0:
                    if (b[o] & 0xC0) != 0x80 and self._keep(sample_index, o, len(tb)):
                        return True
                    o = b.find(tb, o + 1, z)
        o = b.find(tb)
        while o >= 0:
            if self._keep(sample_index, o, len(tb), added=True):
                return True
            o = b.find(tb, o + len(tb))
        return False

    def feed(self, samples: list[str]) -> None:
        """feed(&mut self, samples) — src/generate.rs:54-139: DOCUMENT frequencies (one count per sample).
        The samples are counted by the next device pass (as many samples as fit one pass at a time)."""
        if not self._pending:
            self._pending_first = self._fed
        enc = [s.encode("utf-8", "surrogatepass") for s in samples]
        self._pending.extend(samples)
        self._pending_enc.extend(enc)
        self._pending_bytes += sum(len(b) for b in enc)
        self._fed += len(samples)
        if self._pending_bytes >= pass_bytes(self.max_token_length):
            self._flush()

    def _flush(self) -> None:
        limit = pass_bytes(self.max_token_length)
        while self._pending:
            n, size = 0, 0
            while n < len(self._pending) and (n == 0 or size + len(self._pending_enc[n]) <= limit):
                size += len(self._pending_enc[n])
                n += 1
            self._count_split(self._pending[:n], self._pending_enc[:n], self._pending_first)
            self._pending, self._pending_enc = self._pending[n:], self._pending_enc[n:]
            self._pending_bytes -= size
            self._pending_first += n

    def _count_split(self, samples: list[str], enc: list[bytes], first: int) -> None:
        """One device pass; a batch the device refuses for its size (more than 2^32 kept windows) is halved and retried."""
        try:
            self._count(samples, enc, first)
        except _lib.TokenGeeXError as exc:
            if exc.status != _lib.ERR_UNSUPPORTED or "windows" not in str(exc) or len(samples) < 2:
                raise
            h = len(samples) // 2
            self._count_split(samples[:h], enc[:h], first)
            self._count_split(samples[h:], enc[h:], first + h)

    def _count(self, samples: list[str], enc: list[bytes], first: int) -> None:
        flat, offs = _lib.pack(enc)
        parts_of: list[list[tuple[int, int]]] = [[(0, len(b))] for b in enc]  # per sample: its parts as byte ranges in the sample
        if self.split is None:
            keep = np.flatnonzero(offs[1:] > offs[:-1])
            pb, pe, ps = offs[:-1][keep], offs[1:][keep], keep.astype(np.uint32)
            po = pb
        else:  # the parts are the split regex's matches, as byte ranges (src/generate.rs:67-70)
            pb_l, pe_l, ps_l = [], [], []
            parts_of = [[] for _ in enc]
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
                        parts_of[i].append((a, z))
                        last = z
            pb, pe, ps = np.array(pb_l, np.uint64), np.array(pe_l, np.uint64), np.array(ps_l, np.uint32)
            po = offs[:-1][ps.astype(np.int64)] if ps.size else pb  # where every part's sample begins in the pass's buffer
        extra = self.added_tokens + self.suggested_tokens
        extra_set = set(extra)
        if pb.size:
            # the keep rule sees the global sample index (the device packs it into 27 bits)
            pos, ln, df, _, _, cut = _lib.substring_df_top(flat, pb, pe, (ps.astype(np.uint64) + first).astype(np.uint32),
                                                           self.max_token_length, self.top_k, self.insert_probability, self.seed, self.device,
                                                           part_origin=po)
            self.passes += 1  # (counted once the device has answered: a refused batch is retried in halves)
            self._bound_total += cut
            raw = flat.tobytes()
            for p_, l_, d_ in zip(pos.tolist(), ln.tolist(), df.tolist()):
                cand = raw[p_:p_ + l_].decode("utf-8", "surrogatepass")
                if cand in extra_set:
                    continue  # counted below: a sample's set holds the token once, whichever rule put it there
                if self.allow is None or self.allow.search(cand):
                    self._freq[cand] = self._freq.get(cand, 0) + d_
                    if cut:
                        self._bound_seen[cand] = self._bound_seen.get(cand, 0) + cut
        # added and suggested tokens (src/generate.rs:117-127): counted here, sample by sample, with both rules that can put
        # them into a sample's set (the device's counts of these strings are skipped above)
        for t in dict.fromkeys(extra):
            if not t:
                continue
            tb = t.encode("utf-8", "surrogatepass")
            n = sum(1 for i, sample in enumerate(samples) if t in sample and self._extra_kept(first + i, sample, enc[i], parts_of[i], t, tb))
            if n:
                self._freq[t] = self._freq.get(t, 0) + n

    @property
    def frequencies(self) -> dict[str, int]:
        """The counts so far (of the substrings that left the device: all of them without top_k)."""
        self._flush()
        return self._freq

    def current_size(self) -> int:
        return len(self.frequencies)

    def generate(self, size: int) -> list[tuple[bytes, float, bool]]:
        """generate(&mut self, size) — src/generate.rs:148-243 -> [(value, log-probability, keep)]."""
        freqs = self.frequencies
        # (ties: ascending token bytes — the reference's sort_unstable_by leaves their order unspecified; a fixed
        # rule makes the result independent of the order in which the counts arrived)
        frequent = sorted(freqs.items(), key=lambda kv: (-kv[1], kv[0].encode("utf-8", "surrogatepass")))
        highest = frequent[0][1] if frequent else 1
        seen = {bytes([b]) for b in range(255)}
        vocab = [(bytes([b]), float(highest), True) for b in range(255)]  # bytes 0..254, src/generate.rs:164-169
        extra = set(self.added_tokens) | set(self.suggested_tokens)  # counted exactly on the host, whatever top_k
        for tok, keep in [(t, True) for t in self.added_tokens] + [(t, False) for t in self.suggested_tokens]:
            if len(vocab) >= size:
                break
            b = tok.encode("utf-8")
            if b not in seen and len(b) > 1:
                seen.add(b)
                vocab.append((b, float(freqs[tok] * len(b)), keep))
        last_freq, chosen = None, []
        for tok, freq in frequent:
            if len(vocab) >= size:
                break
            b = tok.encode("utf-8", "surrogatepass")
            if b not in seen and len(b) > 1:
                seen.add(b)
                vocab.append((b, float(freq * len(b)), False))
                if tok not in extra:
                    chosen.append(tok)
                    last_freq = freq
        if self._bound_total:
            # Some pass cut substrings off.  The selection is exact iff every chosen substring came back from every
            # pass that cut anything (its count is complete) and no substring outside the selection can reach the
            # frequency of the last one chosen: a cut-off substring occurs in at most `cutoff` samples of its pass.
            floor = last_freq if (last_freq is not None and len(vocab) >= size) else 0
            top = frequent[0][0] if frequent else None
            if highest < self._bound_total or (top is not None and top not in extra and self._bound_seen.get(top, 0) != self._bound_total):
                # (the most frequent substring's count is the score of every single-byte token: it must be complete too)
                raise _lib.TokenGeeXError("VocabularyGenerator: top_k too small: the most frequent substring's count is incomplete, or a substring "
                                          "that was cut off could be the most frequent one")
            for tok in chosen:
                if self._bound_seen.get(tok, 0) != self._bound_total:
                    raise _lib.TokenGeeXError(f"VocabularyGenerator: top_k too small for an exact vocabulary: {tok!r} was cut off in some pass "
                                              f"({self.passes} passes; feed fewer, larger batches or raise top_k)")
            chosen_set = set(chosen)
            if floor <= self._bound_total:  # a substring no pass returned could tie with or beat the last one chosen
                raise _lib.TokenGeeXError("VocabularyGenerator: top_k too small for an exact vocabulary: the selection reaches "
                                          f"frequency {floor}, substrings of up to {self._bound_total} samples were cut off")
            for tok, freq in freqs.items():
                if tok in chosen_set or tok in extra or len(tok.encode("utf-8", "surrogatepass")) <= 1:
                    continue
                missing = self._bound_total - self._bound_seen.get(tok, 0)  # what the passes that cut it off may have held
                if missing > 0 and freq + missing >= floor:  # (also a substring AT the floor: its true count may exceed it)
                    raise _lib.TokenGeeXError(f"VocabularyGenerator: top_k too small for an exact vocabulary: {tok!r} may belong to it")
        vocab.sort(key=lambda t: (-t[1], t[0]))
        logsum = math.log(sum(t[1] for t in vocab))  # logprobs, src/generate.rs:245-251
        out = []
        for value, score, keep in vocab:
            lp = math.log(score) - logsum if score > 0 else float("nan")
            if lp != lp or lp in (float("inf"), float("-inf")) or lp == 0.0:
                raise ValueError(f"Vocabulary generation: invalid frequency for token {value!r}: {lp}")
            out.append((value, lp, keep))
        return out
