"""`prune` on the MI355X path — mirror of the reference's ModelVocabularyPruner
(src/prune.rs:6-57): EM sub-iterations (E-step on the GPU, M-step on the host), then one
loss-based shrink (frequency pass on the GPU, selection on the host), until the vocabulary is
at most `vocab_size`.  The corpus is uploaded once and stays in HBM across every pass.

The corpus passes run through tgx_estep / tgx_count_tokens; the O(V) host steps through
tgx_prune_m_step / tgx_prune_alternatives / tgx_prune_select.  No CPU fallback: without the
HIP library (or a GPU) this module raises.
"""
from __future__ import annotations

import time

import numpy as np

from . import _lib
from . import dist as tdist

Vocab = list  # list[tuple[bytes, float, bool]]  (value, score, keep) — ScoredToken, src/lib.rs:77-84


class ModelVocabularyPruner:
    """ModelVocabularyPruner::new(vocab_size, shrink_factor, em_subiters, dropout) — src/prune.rs:13-21."""

    def __init__(self, vocab_size: int, shrink_factor: float, em_subiters: int, dropout: float,
                 device: int = 0, seed: int = 0, log=None, dist=None, reduce_device: str = "cpu"):
        self.vocab_size = int(vocab_size)
        self.shrink_factor = float(shrink_factor)
        self.em_subiters = int(em_subiters)
        self.dropout = float(dropout)
        self.device = device
        self.seed = seed
        self.log = log or (lambda *_: None)
        # multi-GPU: one process per GPU, each with its own shard of the samples; `dist` is an initialised
        # torch.distributed module.  The only exchange is one vector per pass, summed in rank order, so every
        # rank derives the same vocabulary (no data-path collective).
        self.dist, self.reduce_device = dist, reduce_device
        self.timings: list[dict] = []

    # -- one pass each ------------------------------------------------------------------
    def _model(self, vocab: Vocab, for_estep: bool = False) -> _lib.NativeModel:
        return _lib.NativeModel([t[0] for t in vocab], [t[1] for t in vocab], self.device, for_estep=for_estep)

    def run_e_step(self, model: _lib.NativeModel, corpus: _lib.NativeCorpus) -> np.ndarray:
        """src/prune.rs:64-120 (81 920-byte snippets; z must be normal)."""
        self.seed += 1
        expected, _ = model.estep(corpus, _lib.ESTEP_SNIPPET_LEN, self.dropout, self.seed)
        return tdist.allreduce_vector(expected, self.dist, self.reduce_device)

    @staticmethod
    def run_m_step(vocab: Vocab, expected: np.ndarray) -> Vocab:
        """src/prune.rs:124-170."""
        keep = np.array([1 if t[2] else 0 for t in vocab], np.uint8)
        idx, scores = _lib.prune_m_step(expected, keep)
        return [(vocab[int(i)][0], float(s), vocab[int(i)][2]) for i, s in zip(idx, scores)]

    def prune_vocab(self, vocab: Vocab, model: _lib.NativeModel, corpus: _lib.NativeCorpus) -> Vocab:
        """src/prune.rs:173-319."""
        V = len(vocab)
        pruned_size = max(int(V * self.shrink_factor), self.vocab_size)
        tokens = [t[0] for t in vocab]
        scores = np.array([t[1] for t in vocab], np.float64)
        keep = np.array([1 if t[2] else 0 for t in vocab], np.uint8)
        trie = _lib.FlatTrie(tokens, scores)
        always_keep, alt_offs, alt_ids = trie.prune_alternatives(tokens, scores)
        freq = tdist.allreduce_vector(model.count_tokens(corpus), self.dist, self.reduce_device)
        n_samples = tdist.allreduce_scalar(corpus.num_samples, self.dist, self.reduce_device)
        out = _lib.prune_select(freq, keep, always_keep, alt_offs, alt_ids, scores, n_samples, pruned_size)
        return [vocab[int(i)] for i in out]

    # -- the loop -----------------------------------------------------------------------
    def prune(self, vocab: Vocab, flat: np.ndarray, offs: np.ndarray) -> Vocab:
        """prune(&mut model, samples) — src/prune.rs:23-57.  `flat`/`offs`: packed samples."""
        corpus = _lib.NativeCorpus(flat, offs, self.device)
        try:
            while len(vocab) > self.vocab_size:
                rec = {"from": len(vocab), "e_step_s": 0.0, "m_step_s": 0.0}
                for sub in range(self.em_subiters):
                    model = self._model(vocab, for_estep=True)
                    t0 = time.perf_counter()
                    expected = self.run_e_step(model, corpus)
                    t1 = time.perf_counter()
                    new_vocab = self.run_m_step(vocab, expected)
                    t2 = time.perf_counter()
                    rec["e_step_s"] += t1 - t0
                    rec["m_step_s"] += t2 - t1
                    self.log(f"EM subiter {sub + 1}/{self.em_subiters} vocab_size={len(vocab)} "
                             f"alternative_vocab_size={len(new_vocab)}")
                    model.free()
                    vocab = new_vocab
                model = self._model(vocab)
                t0 = time.perf_counter()
                vocab = self.prune_vocab(vocab, model, corpus)
                rec["prune_vocab_s"] = time.perf_counter() - t0
                rec["to"] = len(vocab)
                model.free()
                self.timings.append(rec)
                self.log(f"pruned vocabulary from={rec['from']} to={rec['to']}")
        finally:
            corpus.free()
        return vocab
