"""Builds the benchmark vocabularies SURVEY.md section 8(d) prescribes — the generate stand-in (synth.build_vocab:
255 single bytes, then the most frequent allowed substrings of at most 16 bytes, score = ln(freq * len) - ln(sum))
over a fixed 64 MB (64 MiB) slice of the mixed synthetic corpus — and stores them compressed under tests/golden/
(vocab_32000.npz, vocab_65536.npz; minutes of numpy each, so they are committed like vocab_500000.npz and loaded by
bench.py and the tests: synth.load_spec_vocab).
usage: python tools/make_spec_vocab.py [SIZE ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tokengeex_amd import synth

SLICE_MIB = 64
for V in [int(a) for a in sys.argv[1:]] or [32000, 65536]:
    cached = os.path.join(ROOT, "cache", f"vocab_{V}.npz")   # tools/make_vocab_cache.py V 64: the same build
    if os.path.exists(cached):
        z = np.load(cached)
        fb, o, scores = z["flat"].tobytes(), z["offs"].astype(np.int64), z["scores"].astype(np.float64)
        toks = [fb[o[i]:o[i + 1]] for i in range(o.size - 1)]
    else:
        vflat, _ = synth.make_corpus(SLICE_MIB << 20, "mixed", seed_offset=0)
        toks, scores = synth.build_vocab(vflat, V, 16)
        scores = np.asarray(scores, np.float64)
    assert len(toks) == V
    us, inv = np.unique(scores, return_inverse=True)
    assert us.size < 65536 and np.array_equal(us[inv], scores)
    out = os.path.join(ROOT, "tests", "golden", f"vocab_{V}.npz")
    np.savez_compressed(out, flat=np.frombuffer(b"".join(toks), np.uint8), lens=np.array([len(t) for t in toks], np.uint8),
                        uscores=us, inv=inv.astype(np.uint16), slice_mib=np.array([SLICE_MIB]))
    print(out, V, "tokens,", us.size, "distinct score values,", os.path.getsize(out), "bytes")
