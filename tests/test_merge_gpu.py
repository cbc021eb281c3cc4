"""End-to-end `merge` (src/merge.rs:33-134) on the GPU path against the same loop over the oracle's pair scan."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc
from tokengeex_amd.merge import ModelVocabularyMerger

from util import AllowByHand, corpus_and_vocab
from test_merge_cpu import ALLOW


def oracle_merge(vocab, flat, offs, allow, num_merges, step, scale_factor, max_token_length):
    """The reference loop restated over oracle functions only (checker); ties: ascending (a, b).  The allow
    pattern is evaluated by hand (util.AllowByHand), not through the product's pattern translation."""
    assert allow == ALLOW
    rx = AllowByHand()
    vocab, start, ignore = list(vocab), len(vocab), set()
    while len(vocab) < start + num_merges:
        m = orc.OracleModel([t[0] for t in vocab], [t[1] for t in vocab])
        keys, counts = m.count_pairs_flat(flat, offs, threads=8)
        pairs = sorted(zip(keys.tolist(), counts.tolist()), key=lambda kc: (-kc[1], kc[0]))
        merges = budget = min(step, num_merges - (len(vocab) - start))
        for k, _ in pairs:
            if merges == 0:
                break
            a, b = k >> 32, k & 0xFFFFFFFF
            value = vocab[a][0] + vocab[b][0]
            if len(value) > max_token_length or not rx.search(value.decode("utf-8", errors="replace")):
                ignore.add(k)
                continue
            vocab.append((value, (vocab[a][1] + vocab[b][1]) * scale_factor, False))
            merges -= 1
        if merges == step or merges == budget:
            break
    return vocab


def test_merge_end_to_end_matches_oracle_loop():
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 3000, 12, max_len=16384)
    vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
    merger = ModelVocabularyMerger(ALLOW, num_merges=150, step=40, scale_factor=0.9, max_token_length=20)
    got = merger.merge(vocab, flat, offs)
    want = oracle_merge(vocab, flat, offs, ALLOW, 150, 40, 0.9, 20)
    assert len(got) == len(vocab) + 150 and len(merger.rounds) == 4
    assert got == want                      # same tokens, same order, bit-identical scores
    assert max(len(t[0]) for t in got) > 12  # merged tokens outgrow the 12-byte start; past 16 bytes the model
    # moves from the four-samples-per-wave kernels to the one-sample-per-wave kernel, both are exercised
