"""One `prune` run (src/prune.rs:23-57) on the GPU path with per-phase wall times — recorded in
profiles/, not part of bench.py's line.  usage: prune_bench.py [corpus MiB] [vocab] [target]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tokengeex_amd import synth
from tokengeex_amd.prune import ModelVocabularyPruner
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
V = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
target = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
if V == 500000:  # the committed vocabulary of BASELINE.json configs[3] (tests/golden/vocab_500000.npz)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from util import load_vocab_500k
    toks, scores = load_vocab_500k()
else:
    vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
    toks, scores = synth.build_vocab(vflat[: 2 << 20], V, 16)
assert len(toks) == V, (len(toks), V)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
p = ModelVocabularyPruner(target, 0.75, 2, 0.01, log=lambda m: print(m, file=sys.stderr, flush=True))
import tokengeex_amd as tgx
tgx.NativeModel(toks[:300], scores[:300]).free()  # (the device context, the first hipMalloc: not part of a prune run's rate)
t0 = time.perf_counter()
out = p.prune(vocab, flat, offs)
wall = time.perf_counter() - t0
print(json.dumps({"corpus_bytes": int(flat.size), "samples": int(offs.size - 1), "vocab_from": len(vocab),
                  "vocab_to": len(out), "shrink_factor": 0.75, "em_subiters": 2, "dropout": 0.01,
                  "wall_s": wall, "iterations": p.timings}))
