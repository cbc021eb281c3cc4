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
    # The public methods take and return ScoredToken lists as the reference's do; the loop below carries the
    # vocabulary as (Packed token bytes, scores f64[V], keep u8[V]) so that nothing iterates over V in Python.
    @staticmethod
    def _arrays(vocab: Vocab):
        return (_lib.Packed.of([t[0] for t in vocab]), np.array([t[1] for t in vocab], np.float64),
                np.array([1 if t[2] else 0 for t in vocab], np.uint8))

    @staticmethod
    def _vocab(toks: "_lib.Packed", scores: np.ndarray, keep: np.ndarray) -> Vocab:
        return list(zip(toks.tolist(), scores.tolist(), (keep != 0).tolist()))

    def _model(self, vocab, for_estep: bool = False) -> _lib.NativeModel:
        toks, scores = (vocab[0], vocab[1]) if isinstance(vocab, tuple) else ([t[0] for t in vocab], [t[1] for t in vocab])
        return _lib.NativeModel(toks, scores, self.device, for_estep=for_estep)

    def run_e_step(self, model: _lib.NativeModel, corpus: _lib.NativeCorpus) -> np.ndarray:
        """src/prune.rs:64-120 (81 920-byte snippets; z must be normal)."""
        self.seed += 1
        expected, _ = model.estep(corpus, _lib.ESTEP_SNIPPET_LEN, self.dropout, self.seed)
        return tdist.allreduce_vector(expected, self.dist, self.reduce_device)

    @staticmethod
    def _m_step(arrays, expected: np.ndarray, with_index: bool = False):
        toks, _, keep = arrays
        idx, scores = _lib.prune_m_step(expected, keep)
        out = (toks.take(idx), np.asarray(scores, np.float64), keep[np.asarray(idx, np.int64)])
        return (out, np.asarray(idx, np.uint32)) if with_index else out

    def _derived(self, parent: "_lib.NativeModel", idx: np.ndarray, arrays, for_estep: bool) -> "_lib.NativeModel":
        """The model of a SUBSET of `parent`'s vocabulary (round 3): on the parent's tries where the library can
        (tgx_model_create_derived: no double-array is rebuilt), from scratch otherwise (duplicate tokens)."""
        try:
            return parent.derive(idx, arrays[1], for_estep=for_estep)
        except _lib.TokenGeeXError as exc:
            # only "the parent has duplicate tokens" (TGX_ERR_UNSUPPORTED) means "build it from scratch": a device failure or a
            # bad index list must surface (and every rank of a multi-rank run takes the same path)
            if exc.status != _lib.ERR_UNSUPPORTED:
                raise
            return self._model(arrays, for_estep=for_estep)

    @staticmethod
    def run_m_step(vocab: Vocab, expected: np.ndarray) -> Vocab:
        """src/prune.rs:124-170."""
        return ModelVocabularyPruner._vocab(*ModelVocabularyPruner._m_step(ModelVocabularyPruner._arrays(vocab), expected))

    def _prune_arrays(self, arrays, model: _lib.NativeModel, corpus: _lib.NativeCorpus):
        toks, scores, keep = arrays
        V = len(toks)
        pruned_size = max(int(V * self.shrink_factor), self.vocab_size)
        t0 = time.perf_counter()
        always_keep, alt_offs, alt_ids = model.prune_alternatives()  # over the model's own table
        t1 = time.perf_counter()
        freq = tdist.allreduce_vector(model.count_tokens(corpus), self.dist, self.reduce_device)
        n_samples = tdist.allreduce_scalar(corpus.num_samples, self.dist, self.reduce_device)
        t2 = time.perf_counter()
        out = np.asarray(_lib.prune_select(freq, keep, always_keep, alt_offs, alt_ids, scores, n_samples, pruned_size), np.int64)
        t3 = time.perf_counter()
        res = toks.take(out), scores[out], keep[out]
        self.last_prune_phases = {"alternatives_s": t1 - t0, "count_tokens_s": t2 - t1, "select_s": t3 - t2, "take_s": time.perf_counter() - t3}
        return res

    def prune_vocab(self, vocab: Vocab, model: _lib.NativeModel, corpus: _lib.NativeCorpus) -> Vocab:
        """src/prune.rs:173-319."""
        return self._vocab(*self._prune_arrays(self._arrays(vocab), model, corpus))

    # -- the loop -----------------------------------------------------------------------
    def prune(self, vocab: Vocab, flat: np.ndarray, offs: np.ndarray) -> Vocab:
        """prune(&mut model, samples) — src/prune.rs:23-57.  `flat`/`offs`: packed samples."""
        corpus = _lib.NativeCorpus(flat, offs, self.device)
        arrays = self._arrays(vocab)
        try:
            while len(arrays[0]) > self.vocab_size:
                rec = {"from": len(arrays[0]), "model_s": 0.0, "e_step_s": 0.0, "m_step_s": 0.0, "derive_s": 0.0}
                t_iter = time.perf_counter()
                # one double-array build per iteration: the models of the later sub-iterations and of the pruning step
                # are subsets of the first one's vocabulary and live on its tables (tgx_model_create_derived)
                model = self._model(arrays, for_estep=True)
                rec["model_s"] = time.perf_counter() - t_iter
                for sub in range(self.em_subiters):
                    t0 = time.perf_counter()
                    expected = self.run_e_step(model, corpus)
                    t1 = time.perf_counter()
                    new_arrays, idx = self._m_step(arrays, expected, with_index=True)
                    t2 = time.perf_counter()
                    rec["e_step_s"] += t1 - t0
                    rec["m_step_s"] += t2 - t1
                    self.log(f"EM subiter {sub + 1}/{self.em_subiters} vocab_size={len(arrays[0])} "
                             f"alternative_vocab_size={len(new_arrays[0])}")
                    nxt = self._derived(model, idx, new_arrays, for_estep=sub + 1 < self.em_subiters)
                    model.free()
                    model = nxt
                    arrays = new_arrays
                    rec["derive_s"] += time.perf_counter() - t2
                t0 = time.perf_counter()
                arrays = self._prune_arrays(arrays, model, corpus)
                rec["prune_vocab_s"] = time.perf_counter() - t0
                rec["prune_vocab_phases"] = getattr(self, "last_prune_phases", None)
                rec["to"] = len(arrays[0])
                model.free()
                rec["wall_s"] = time.perf_counter() - t_iter
                self.timings.append(rec)
                self.log(f"pruned vocabulary from={rec['from']} to={rec['to']}")
            vocab = self._vocab(*arrays)
        finally:
            corpus.free()
        return vocab
