"""A few `merge` rounds (src/merge.rs:33-134) on the GPU path with per-round times — recorded in profiles/.
usage: merge_bench.py [corpus MiB] [vocab] [num_merges] [step] [max_token_length]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tokengeex_amd import synth
from tokengeex_amd.merge import ModelVocabularyMerger
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
V = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
num = int(sys.argv[3]) if len(sys.argv) > 3 else 300
step = int(sys.argv[4]) if len(sys.argv) > 4 else 100
mtl = int(sys.argv[5]) if len(sys.argv) > 5 else 16
ALLOW = r"^(?:.)$|^(?:[a-z]+)$|^(?:[A-Z]+)$|^(?:[A-Z][a-z]+)$|^(?:[㐀-䶿一-鿿]+)$|^(?:(?:[ ]+)|[\t]+)$|^(?: ?[[:punct:]] ?)$"
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], V, 16)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
m = ModelVocabularyMerger(ALLOW, num, step, 0.9, mtl, log=lambda s: print(s, file=sys.stderr, flush=True))
t0 = time.perf_counter()
out = m.merge(vocab, flat, offs)
print(json.dumps({"corpus_bytes": int(flat.size), "samples": int(offs.size - 1), "vocab_from": len(vocab), "vocab_to": len(out),
                  "num_merges": num, "step": step, "max_token_length": mtl, "longest_token": max(len(t[0]) for t in out),
                  "wall_s": time.perf_counter() - t0, "rounds": m.rounds}))
