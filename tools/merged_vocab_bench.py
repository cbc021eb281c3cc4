"""Encode throughput with a REALISTIC merged vocabulary: the 32 000-entry vocabulary (tokens <= 16 bytes) plus
a few hundred tokens of up to 24 bytes produced by `merge` (README.md:248 of the reference), on the 16-lane rows
with overflow list (encode4l_kernel, default) and on two samples per wave (TGX_PATH=rows2)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
from tokengeex_amd.merge import ModelVocabularyMerger
ALLOW = r"^(?:.)$|^(?:[a-z]+)$|^(?:[A-Z]+)$|^(?:[A-Z][a-z]+)$|^(?:[㐀-䶿一-鿿]+)$|^(?:(?:[ ]+)|[\t]+)$|^(?: ?[[:punct:]] ?)$"
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
mflat, moffs = synth.make_corpus(64 << 20, "mixed", seed_offset=500)
vocab = [(t, float(s), len(t) == 1) for t, s in zip(toks, scores)]
vocab = ModelVocabularyMerger(ALLOW, 600, 100, 0.9, 24).merge(vocab, mflat, moffs)
lens = np.array([len(t[0]) for t in vocab])
flat, offs = synth.make_corpus(1024 << 20, "mixed", seed_offset=1000)
m = tgx.NativeModel([t[0] for t in vocab], [t[1] for t in vocab])
c = tgx.NativeCorpus(flat, offs)
out = {"vocab": len(vocab), "longest_token": int(lens.max()), "tokens_over_16_bytes": int((lens > 16).sum())}
ref = None
for path in ("default", "rows2", "default"):
    if path == "rows2": os.environ["TGX_PATH"] = "rows2"
    else: os.environ.pop("TGX_PATH", None)
    r = m.encode_corpus(c); ids = r.ids(); r.free()
    if ref is None: ref = ids
    kt = m.last_kernel_times()
    out[path] = {"kernel_ms": kt, "GB_per_s": flat.size / sum(kt.values()) / 1e6, "same_ids": bool(np.array_equal(ids, ref))}
print(json.dumps(out))
