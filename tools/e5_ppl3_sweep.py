import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
import small_batch_sweep as sb
flat, offs = synth.make_corpus(1024 << 20, "mixed", seed_offset=1000)
corpus = tgx.NativeCorpus(flat, offs)
for size in (32000, 65536):
    toks, scores, _ = synth.load_spec_vocab(size)
    m = tgx.NativeModel(toks, scores)
    print("== spec", size, flush=True)
    cfgs = [dict(), dict(TGX_PPL="3", TGX_WAVES="13"), dict(TGX_PPL="3", TGX_WAVES="12"), dict(TGX_PPL="3", TGX_WAVES="14"), dict(TGX_PPL="3", TGX_WAVES="16"), dict(TGX_PPL="4", TGX_WAVES="13")]
    sb.run(m, corpus, cfgs)
    m.free()
toks, scores, _ = synth.load_spec_vocab(32000)
m = tgx.NativeModel(toks, scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks)))
print("== distinct", flush=True)
sb.run(m, corpus, [dict(), dict(TGX_PPL="3", TGX_WAVES="16"), dict(TGX_PPL="3", TGX_WAVES="14")])
