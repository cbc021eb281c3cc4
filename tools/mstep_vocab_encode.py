"""Encode with the vocabulary an M-step leaves (src/prune.rs:124-170 scores: digamma of expected counts — kept single-byte
tokens get tiny scores and still match at every position) against the bench's --distinct-scores stand-in.
usage: mstep_vocab_encode.py [corpus MiB]"""
import os, sys, time
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth, _lib
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
toks, scores, _ = synth.load_spec_vocab(32000)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
def enc(model, tag):
    best = None
    for _ in range(4):
        r = model.encode_corpus(c); n = r.num_tokens; r.free()
        kt = model.last_kernel_times()
        if best is None or sum(kt.values()) < sum(best.values()): best = dict(kt)
    print(f"{tag:44s} kernels={ {k: round(v, 3) for k, v in best.items()} } sum={sum(best.values()):.2f} ms  {flat.size / sum(best.values()) / 1e6:.1f} GB/s  values={model.score_values()} in LDS={model.last_encode_hot_values()} tokens={n}", flush=True)
m = tgx.NativeModel(toks, scores)
enc(m, "spec vocabulary (9 652 values)")
sc2 = np.asarray(scores) + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks))
enc(tgx.NativeModel(toks, sc2), "spec scores + noise (--distinct-scores)")
os.environ["TGX_VALUE_RANK"] = "model"
enc(tgx.NativeModel(toks, sc2), "  ... values in build_trie8's order")
del os.environ["TGX_VALUE_RANK"]
me = tgx.NativeModel(toks, scores, for_estep=True)
exp, _ = me.estep(c)
keep = np.array([1 if len(t) == 1 else 0 for t in toks], np.uint8)
idx, sc3 = _lib.prune_m_step(exp, keep)
idx = np.asarray(idx, np.int64); sc3 = np.asarray(sc3, np.float64)
toks3 = [toks[i] for i in idx]
print("M-step:", len(toks3), "tokens, single-byte scores: median", float(np.median(sc3[[len(t) == 1 for t in toks3]])), "all: median", float(np.median(sc3)))
enc(tgx.NativeModel(toks3, sc3), "after an M-step (digamma scores)")
os.environ["TGX_VALUE_RANK"] = "model"
enc(tgx.NativeModel(toks3, sc3), "  ... values in build_trie8's order")
