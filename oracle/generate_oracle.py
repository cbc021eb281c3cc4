"""CPU restatement of the reference's VocabularyGenerator (src/generate.rs:12-243) — TEST INFRASTRUCTURE: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/; the product
(tokengeex_amd/generate.py) counts on the device and fails without one.

Pinned by the reference's own test (src/generate.rs:258-284: the vocabulary generated from four samples contains
"string") and by hand-checkable document frequencies (tests/test_generate_cpu.py).  The reference cannot be built
or imported here (Rust / PyO3).

Two forced differences from the reference, shared with the product so that both can be compared exactly:
  * `rng.gen_range(0.0..1.0) < insert_probability` draws from an unseeded thread RNG (src/generate.rs:88, 112, 126);
    here a counter hash of (seed, sample index, the occurrence: byte offset in the sample << 8 | byte length) decides —
    the function include/tgx.h documents as tgx_generate_u01 — so runs are reproducible, with one draw per occurrence as
    in the reference (round 4; one draw per (sample, substring) before: a substring occurring k times in a sample was
    counted for it with probability p instead of 1 - (1 - p)^k);
  * `sort_unstable_by_key` / `sort_unstable_by` (src/generate.rs:152, 216-220) leave the order of equal frequencies /
    scores unspecified; ties are ordered by the token's bytes.
"""
from __future__ import annotations

import math

_M64 = (1 << 64) - 1


def fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & _M64
    return h


def keep_u01(seed: int, sample: int, occurrence: int) -> float:
    """The seeded stand-in for the reference's thread RNG (tgx_generate_u01 of include/tgx.h): one draw per OCCURRENCE,
    as the reference draws inside its loops over positions and lengths (src/generate.rs:84-89, 108-113, 122-127).
    occurrence = byte offset in the sample << 8 | byte length (| 1 << 63 for the draws of the added / suggested tokens)."""
    x = (seed ^ (sample * 0x9E3779B97F4A7C15) ^ ((occurrence & _M64) * 0xC2B2AE3D27D4EB4F)) & _M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & _M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & _M64
    x ^= x >> 31
    return (x >> 11) * (1.0 / 9007199254740992.0)


class OracleVocabularyGenerator:
    """VocabularyGenerator::new / feed / current_size / generate (src/generate.rs:22-243), sample by sample.
    `split` and `allow` are compiled Python patterns (or None) with the reference patterns' meaning."""

    def __init__(self, max_token_length: int, insert_probability: float, split=None, allow=None,
                 added_tokens=(), suggested_tokens=(), seed: int = 0):
        self.max_token_length = int(max_token_length)
        self.insert_probability = float(insert_probability)
        self.split, self.allow = split, allow
        self.added_tokens, self.suggested_tokens = list(added_tokens), list(suggested_tokens)
        self.seed = seed
        self._fed = 0
        self.frequencies: dict[str, int] = {}
        for t in self.added_tokens + self.suggested_tokens:  # src/generate.rs:33-41
            self.frequencies[t] = self.frequencies.get(t, 0) + 1

    def _keep(self, sample_index: int, offset: int, length: int, added: bool = False) -> bool:
        occ = (offset << 8) | length | ((1 << 63) if added else 0)
        return self.insert_probability >= 1.0 or keep_u01(self.seed, sample_index, occ) < self.insert_probability

    def _candidates(self, part: str, part_offset: int, sample_index: int, out: set):
        """src/generate.rs:72-96 / 99-120: every char-aligned substring of at most max_token_length BYTES, every
        occurrence with a draw of its own.  part_offset: byte offset of the part in its sample."""
        n = len(part)
        blen = [len(c.encode("utf-8", "surrogatepass")) for c in part]
        start = part_offset
        for i in range(n):
            total = 0
            for j in range(i, n):
                total += blen[j]
                if total > self.max_token_length:
                    break
                cand = part[i:j + 1]
                if (self.allow is None or self.allow.search(cand)) and self._keep(sample_index, start, total):
                    out.add(cand)
            start += blen[i]

    def feed(self, samples: list[str]) -> None:
        """src/generate.rs:54-139: DOCUMENT frequencies — a sample's set of candidates counts once each."""
        for sample in samples:
            idx = self._fed
            self._fed += 1
            toks: set = set()
            if self.split is not None:
                for m in self.split.finditer(sample):
                    self._candidates(m.group(0), len(sample[:m.start()].encode("utf-8", "surrogatepass")), idx, toks)
            else:
                self._candidates(sample, 0, idx, toks)
            for t in self.added_tokens + self.suggested_tokens:  # src/generate.rs:117-127: a draw per match, the first success inserts
                if not t:
                    continue
                pos = sample.find(t)
                while pos >= 0:  # str::match_indices: successive non-overlapping matches
                    if self._keep(idx, len(sample[:pos].encode("utf-8", "surrogatepass")), len(t.encode("utf-8", "surrogatepass")), added=True):
                        toks.add(t)
                        break
                    pos = sample.find(t, pos + len(t))
            for t in toks:
                self.frequencies[t] = self.frequencies.get(t, 0) + 1

    def current_size(self) -> int:
        return len(self.frequencies)

    def generate(self, size: int) -> list[tuple[bytes, float, bool]]:
        """src/generate.rs:148-243 -> [(value, log-probability, keep)]."""
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
            if lp != lp or lp in (float("inf"), float("-inf")) or lp == 0.0:  # !is_normal(): src/generate.rs:226-235
                raise ValueError(f"Vocabulary generation: invalid frequency for token {value!r}: {lp}")
            out.append((value, lp, keep))
        return out
