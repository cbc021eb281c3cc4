import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
flat, offs = synth.make_corpus(1024 << 20, "mixed", seed_offset=1000)
corpus = tgx.NativeCorpus(flat, offs)
def run(m, cfgs, name):
    print("==", name, flush=True)
    for cfg in cfgs:
        for k in ("TGX_PATH", "TGX_WAVES", "TGX_E5_HOT", "TGX_PPL", "TGX_BPC"):
            os.environ.pop(k, None)
        os.environ.update(cfg); os.environ["TGX_LONG_THRESHOLD"] = "0"
        best = None
        for _ in range(3):
            r = m.encode_corpus(corpus); r.free()
            kt = m.last_kernel_times(); t = sum(v for k, v in kt.items() if k.startswith("encode"))
            if best is None or t < best: best = t
        print(f"   {str(cfg):60s} hot={m.last_encode_hot_values():5d}/{m.score_values()} waves/CU={m.last_encode_waves_per_cu():2d} encode {best:7.3f} ms", flush=True)
vflat, _ = synth.make_corpus(8 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat, 32000, 16)
m = tgx.NativeModel(toks, scores)
run(m, [dict(TGX_WAVES="16"), dict(TGX_WAVES="16", TGX_PPL="2", TGX_BPC="1")], "8 MiB-slice vocabulary (all values hot)")
m.free()
toks, scores, _ = synth.load_spec_vocab(32000)
m = tgx.NativeModel(toks, scores)
cf = [dict(TGX_WAVES="16", TGX_PPL="2", TGX_BPC="1"), dict(TGX_WAVES="12", TGX_PPL="2", TGX_BPC="1"), dict(TGX_WAVES="16", TGX_PPL="2", TGX_BPC="1", TGX_E5_HOT="8191")]
run(m, cf, "spec32k")
m.free()
m = tgx.NativeModel(toks, scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks)))
run(m, cf, "distinct32k")
