"""`merge` on the MI355X path — mirror of the reference's ModelVocabularyMerger (src/merge.rs:8-135):
each round runs the pair scan on the GPU (tgx_count_pairs: Viterbi encode + adjacent-pair histogram,
src/merge.rs:53-76), then the host picks up to `step` most frequent pairs whose concatenation is at most
`max_token_length` bytes and matches the allow-regex, scores them (a + b) * scale_factor and appends them
to the vocabulary (src/merge.rs:78-126).  The corpus stays in HBM across rounds.

Pairs of equal frequency are taken in ascending (a, b) order; the reference iterates a hash map there
(src/merge.rs:78-84), i.e. leaves it unspecified.  No CPU fallback for the corpus pass.
"""
from __future__ import annotations

import re

import numpy as np

from . import _lib
from . import dist as tdist

_POSIX_ASCII = {  # Rust regex: POSIX classes are ASCII only
    "alnum": "0-9A-Za-z", "alpha": "A-Za-z", "ascii": "\\x00-\\x7F", "blank": " \\t", "cntrl": "\\x00-\\x1F\\x7F",
    "digit": "0-9", "graph": "!-~", "lower": "a-z", "print": " -~", "punct": "!-/:-@\\[-`{-~",
    "space": " \\t\\n\\r\\x0B\\x0C", "upper": "A-Z", "word": "0-9A-Za-z_", "xdigit": "0-9A-Fa-f",
}


def compile_rust_regex(pattern: str):
    """Compiles the subset of Rust `regex` syntax the reference's allow-patterns use (src/regex.rs:82-425,
    data/exact.regex) with Python's `re`, keeping Rust's meaning where the two differ:
    `$` is the end of the text (Python's also matches before a final newline) -> `\\Z`; `\\z` -> `\\Z`;
    `[[:punct:]]` and friends are ASCII classes; `\\u{XXXX}` / `\\x{XX}` -> `\\uXXXX` / `\\UXXXXXXXX`.
    Unicode property classes (`\\p{..}`) are not used by those patterns and are rejected."""
    out, i, n, in_class = [], 0, len(pattern), 0
    while i < n:
        ch = pattern[i]
        if ch == "\\" and i + 1 < n:
            nxt = pattern[i + 1]
            if nxt in "pP":
                raise ValueError("\\p{..} classes are not supported by this translator")
            if nxt == "z":
                out.append("\\Z"); i += 2; continue
            if nxt in "ux" and i + 2 < n and pattern[i + 2] == "{":
                j = pattern.index("}", i + 3)
                out.append("\\U%08X" % int(pattern[i + 3:j], 16)); i = j + 1; continue
            out.append(pattern[i:i + 2]); i += 2; continue
        if ch == "[":
            if in_class and pattern.startswith("[:", i):
                j = pattern.index(":]", i + 2)
                name = pattern[i + 2:j]
                neg = name.startswith("^")
                if neg:
                    raise ValueError("negated POSIX classes are not supported by this translator")
                out.append(_POSIX_ASCII[name]); i = j + 2; continue
            if in_class:
                out.append("\\["); i += 1; continue
            in_class = 1; out.append("["); i += 1
            if i < n and pattern[i] == "^":
                out.append("^"); i += 1
            if i < n and pattern[i] == "]":
                out.append("\\]"); i += 1
            continue
        if ch == "]" and in_class:
            in_class = 0; out.append("]"); i += 1; continue
        if ch == "$" and not in_class:
            out.append("\\Z"); i += 1; continue
        out.append(ch); i += 1
    return re.compile("".join(out))


def load_regex(path: str):
    """load_regex — src/cli.rs:316-324: newlines removed, trimmed."""
    with open(path, encoding="utf-8") as f:
        return compile_rust_regex(f.read().replace("\n", "").replace("\r", "").strip())


class ModelVocabularyMerger:
    """ModelVocabularyMerger::new(allow, num_merges, step, scale_factor, max_token_length) — src/merge.rs:16-31."""

    def __init__(self, allow, num_merges: int, step: int, scale_factor: float, max_token_length: int,
                 device: int = 0, log=None, dist=None, reduce_device: str = "cpu"):
        self.allow = compile_rust_regex(allow) if isinstance(allow, str) else allow
        self.num_merges, self.step = int(num_merges), int(step)
        self.scale_factor, self.max_token_length = float(scale_factor), int(max_token_length)
        self.device = device
        self.log = log or (lambda *_: None)
        # multi-GPU: one process per GPU with its own shard; the per-rank pair tables are merged on every
        # rank (dist.allreduce_pairs), so all ranks pick the same merges
        self.dist, self.reduce_device = dist, reduce_device
        self.head_pairs = 1 << 18  # pairs fetched per round in single-process mode
        self.exchange_pairs = 1 << 12  # pairs every rank contributes to a round's exchange (multi-rank mode)
        self.exchanged_bytes = 0     # bytes this rank has gathered / reduced for the pair exchange so far
        self.rounds: list[dict] = []

    def select(self, vocab, keys: np.ndarray, counts: np.ndarray, budget: int, ignore: set):
        """src/merge.rs:84-126 for one round -> list of new (value, score, keep) tokens."""
        order = np.argsort(-counts.astype(np.int64), kind="stable")  # keys ascend already
        new = []
        for j in order:
            if budget == 0:
                break
            key = int(keys[j])
            if key in ignore:
                continue
            a, b = key >> 32, key & 0xFFFFFFFF
            value = vocab[a][0] + vocab[b][0]
            if len(value) > self.max_token_length or not self.allow.search(value.decode("utf-8", errors="replace")):
                ignore.add(key)  # src/merge.rs:105-118
                continue
            new.append((value, (vocab[a][1] + vocab[b][1]) * self.scale_factor, False))
            budget -= 1
        return new

    def merge(self, vocab, flat: np.ndarray, offs: np.ndarray):
        """merge(self, &mut model, samples) — src/merge.rs:33-134."""
        import time
        vocab = list(vocab)
        start = len(vocab)
        ignore: set = set()
        corpus = _lib.NativeCorpus(flat, offs, self.device)
        try:
            while len(vocab) < start + self.num_merges:
                model = _lib.NativeModel([t[0] for t in vocab], [t[1] for t in vocab], self.device)
                t0 = time.perf_counter()
                single = self.dist is None or not self.dist.is_initialized() or self.dist.get_world_size() == 1
                budget = min(self.step, self.num_merges - (len(vocab) - start))
                if single:
                    # the head of the table, ordered on the device (tgx_count_pairs_top): the loop below looks at a
                    # few hundred candidates of millions of pairs; the whole table only if the head runs dry
                    keys, counts, total = model.count_pairs_top(corpus, self.head_pairs)
                    if keys.size < total and len(self.select(vocab, keys, counts, budget, set(ignore))) < budget:
                        keys, counts = model.count_pairs(corpus)
                else:
                    # several ranks: only the head of the global table is ever looked at, so the ranks exchange their
                    # local heads and the exact counts of the union (dist.top_pairs_exchange: ~1 MB per round instead
                    # of every rank's whole table); candidates above `bound` are provably the global head.  Widened
                    # (x 4) when the selection would have to go below the bound — every rank takes the same decision.
                    lk, lc = model.count_pairs(corpus)
                    k = self.exchange_pairs
                    while True:
                        keys, counts, bound, nbytes = tdist.top_pairs_exchange(lk, lc, k, self.dist, self.reduce_device)
                        self.exchanged_bytes += nbytes
                        sure = counts.astype(np.int64) > bound
                        keys, counts = keys[sure], counts[sure]
                        # (bound 0: every rank contributed its whole table)
                        if bound == 0 or len(self.select(vocab, keys, counts, budget, set(ignore))) >= budget:
                            break
                        k *= 4
                t1 = time.perf_counter()
                model.free()
                new = self.select(vocab, keys, counts, budget, ignore)
                vocab.extend(new)  # model.add_tokens, src/merge.rs:121
                self.rounds.append({"vocab": len(vocab), "pairs": int(keys.size), "merged": len(new),
                                    "pair_scan_s": t1 - t0, "select_s": time.perf_counter() - t1})
                self.log(f"BPE merge {len(vocab) - start}/{self.num_merges}: {len(new)} merged of {keys.size} pairs")
                if budget - len(new) == self.step:  # `if merges == self.step` — src/merge.rs:128-134
                    self.log("no more merges possible")
                    break
                if not new:  # a last partial round without a candidate: the reference would spin here
                    self.log("no more merges possible (partial round)")
                    break
        finally:
            corpus.free()
        return vocab
