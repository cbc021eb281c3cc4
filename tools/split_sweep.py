import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
import small_batch_sweep as sb
toks, scores, _ = synth.load_spec_vocab(32000)
models = {"spec32k": tgx.NativeModel(toks, scores), "distinct32k": tgx.NativeModel(toks, scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks)))}
cfgs = [dict(), dict(TGX_LONG_THRESHOLD="0"), dict(TGX_LONG_THRESHOLD="8192"), dict(TGX_LONG_THRESHOLD="16384"), dict(TGX_LONG_THRESHOLD="32768"), dict(TGX_LONG_THRESHOLD="49152"),
        dict(TGX_LONG_THRESHOLD="0", TGX_PPL="4", TGX_WAVES="8"), dict(TGX_LONG_THRESHOLD="0", TGX_PPL="3", TGX_WAVES="8"), dict(TGX_LONG_THRESHOLD="0", TGX_PPL="3", TGX_WAVES="10")]
for mib in (128, 256, 384, 512):
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    corpus = tgx.NativeCorpus(flat, offs)
    for name, m in models.items():
        print(f"== {mib} MiB {name}", flush=True)
        sb.run(m, corpus, cfgs)
    corpus.free()
