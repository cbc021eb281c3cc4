"""generate -> prune -> merge -> encode/decode, the reference's recipe (README.md:150-260 there), end to end on
this build: host generate, GPU passes for prune and merge, the Tokenizer mirror for the round trip."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import tokengeex_amd as tgx
from tokengeex_amd import synth
from tokengeex_amd.generate import VocabularyGenerator
from tokengeex_amd.merge import ModelVocabularyMerger
from tokengeex_amd.prune import ModelVocabularyPruner

from test_merge_cpu import ALLOW


def test_recipe_end_to_end():
    flat, offs = synth.make_corpus(192 << 10, "mixed", max_len=2048)
    o = offs.astype(np.int64)
    samples = [flat[o[i]:o[i + 1]].tobytes().decode("utf-8") for i in range(o.size - 1)]
    gen = VocabularyGenerator(12, 1.0, None, ALLOW)
    gen.feed(samples[:120])
    vocab = gen.generate(1500)
    assert len(vocab) == 1500 and all(len(t[0]) <= 12 for t in vocab)
    pruned = ModelVocabularyPruner(900, 0.75, 2, 0.0).prune(vocab, flat, offs)
    # (the M-step drops tokens whose expected count falls below 0.5, so one round may land below the target)
    assert 255 < len(pruned) <= 900 and {bytes([b]) for b in range(255)} <= {t[0] for t in pruned}   # keep=true survives
    merged = ModelVocabularyMerger(ALLOW, 40, 20, 0.9, 16).merge(pruned, flat, offs)
    assert len(merged) == len(pruned) + 40
    tok = tgx.Tokenizer(merged)
    texts = samples[120:160]
    ids = tok.encode_batch(texts, 0.0)
    assert tok.decode_batch(ids, False) == texts
    base = tgx.Tokenizer(pruned).encode_batch(texts, 0.0)
    assert sum(map(len, ids)) < sum(map(len, base))      # the merges shorten the encoding
