"""Builds the large synthetic vocabularies once (python, minutes) and stores them under cache/
(git-ignored, shipped to the GPU box) so GPU time is not spent on host-side vocabulary building.
usage: make_vocab_cache.py SIZE [slice MiB]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tokengeex_amd import synth
V = int(sys.argv[1]); sl = int(sys.argv[2]) if len(sys.argv) > 2 else 64
vflat, _ = synth.make_corpus(sl << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat, V, 16)
flat = np.frombuffer(b"".join(toks), np.uint8)
offs = np.zeros(len(toks) + 1, np.uint64); offs[1:] = np.cumsum([len(t) for t in toks])
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cache", f"vocab_{V}.npz")
np.savez(out, flat=flat, offs=offs, scores=np.asarray(scores, np.float64))
print(out, len(toks))
