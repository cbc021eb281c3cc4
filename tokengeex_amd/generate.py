"""`generate` — host mirror of the reference's VocabularyGenerator (src/generate.rs:12-243), the step that
produces the initial vocabulary `prune` starts from.  It is regex-bound candidate enumeration on the CPU in
the reference and stays host code here (not on the encode / E-step path); this module
exists so that the pipeline generate -> prune -> merge can be run end to end against the same semantics.
The benchmark vocabularies are built by synth.build_vocab (numpy, fixed pattern classes), which is much
faster on large corpora.

Differences from the reference, both forced: the reference draws `rng.gen_range(0.0..1.0)` from an unseeded
thread RNG for `insert_probability` (src/generate.rs:88,112,126), here a counter hash of (seed, sample,
candidate) decides, so runs are reproducible; tokens of equal frequency / score keep their first-seen order
(`sort_unstable_by` leaves it unspecified).  The split regex (fancy_regex) is taken as a compiled Python
pattern.
"""
from __future__ import annotations

import math
import zlib

from .merge import compile_rust_regex


def _u01(seed: int, sample: int, token: str) -> float:
    h = zlib.crc32(token.encode("utf-8", "surrogatepass"), (seed * 0x9E3779B1 + sample * 0x85EBCA6B) & 0xFFFFFFFF)
    return (h & 0xFFFFFFFF) / 4294967296.0


class VocabularyGenerator:
    """VocabularyGenerator::new(max_token_length, insert_probability, split, allow, added, suggested)."""

    def __init__(self, max_token_length: int, insert_probability: float, split=None, allow=None,
                 added_tokens=(), suggested_tokens=(), seed: int = 0):
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
        frequent = sorted(self.frequencies.items(), key=lambda kv: -kv[1])  # stable: first-seen order on ties
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
        vocab.sort(key=lambda t: -t[1])
        logsum = math.log(sum(t[1] for t in vocab))  # logprobs, src/generate.rs:245-251
        out = []
        for value, score, keep in vocab:
            lp = math.log(score) - logsum if score > 0 else float("nan")
            if lp != lp or lp in (float("inf"), float("-inf")) or lp == 0.0:
                raise ValueError(f"Vocabulary generation: invalid frequency for token {value!r}: {lp}")
            out.append((value, lp, keep))
        return out
