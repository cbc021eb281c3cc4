"""Shared helpers for the parity tests."""
from __future__ import annotations

import numpy as np

from oracle import oracle as orc
from tokengeex_amd import synth


def corpus_and_vocab(n_bytes=1 << 20, kind="mixed", vocab_size=4000, max_token_length=16, seed_offset=0,
                     max_len=65536):
    flat, offs = synth.make_corpus(n_bytes, kind, max_len=max_len, seed_offset=seed_offset)
    toks, scores = synth.build_vocab(flat[: min(flat.size, 1 << 20)], vocab_size, max_token_length)
    return flat, offs, toks, scores


def assert_same_encoding(native_model, oracle_model, flat, offs, dropout=0.0, seed=0, threads=8):
    want_ids, want_offs = oracle_model.encode_batch_flat(flat, offs, dropout, seed, threads=threads)
    res = native_model.encode_batch_flat(flat, offs, dropout, seed)
    got_ids, got_offs = res.ids(), res.offsets()
    res.free()
    np.testing.assert_array_equal(got_offs, want_offs)
    np.testing.assert_array_equal(got_ids, want_ids)
    return got_ids, got_offs
