"""encode5_kernel by waves per block and size of the LDS copy of the score-value table (round 3: values outside
the copy are read from L2 by the relaxing lane), for the spec vocabulary (9 652 distinct values),
the same with every token its own score, and the 65 536-entry vocabulary.
usage: TGX_KNOBS=1 python tools/e5_hot_sweep.py [size_mb] [vocab ...]"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vocabs = sys.argv[2:] or ["spec32k", "distinct32k", "spec64k"]
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
corpus = tgx.NativeCorpus(flat, offs)
for vname in vocabs:
    toks, scores, _ = synth.load_spec_vocab(65536 if vname.endswith("64k") else 32000)
    if vname.startswith("distinct"):
        scores = scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks))
    m = tgx.NativeModel(toks, scores)
    print(f"== {vname}: {len(toks)} tokens, {np.unique(scores).size} distinct values, {size} MiB", flush=True)
    cfgs = [dict(TGX_WAVES=str(w)) for w in (16, 15, 14)] + \
           [dict(TGX_WAVES="16", TGX_E5_POOL=p) for p in ("24", "32", "64")] + \
           [dict(TGX_WAVES="16", TGX_E5_HOT=h) for h in ("2047", "0")] + \
           [dict(TGX_WAVES="16", TGX_PPL="2", TGX_BPC="1")]
    for cfg in cfgs:
        for k in ("TGX_PATH", "TGX_WAVES", "TGX_E5_HOT", "TGX_E5_POOL", "TGX_PPL", "TGX_BPC"):
            os.environ.pop(k, None)
        os.environ.update(cfg)
        os.environ.setdefault("TGX_LONG_THRESHOLD", "0")
        best = None
        for _ in range(3):
            r = m.encode_corpus(corpus)
            r.free()
            kt = m.last_kernel_times()
            t = sum(v for k, v in kt.items() if k.startswith("encode"))
            if best is None or t < best[0]:
                best = (t, kt)
        print(f"   {str(cfg):70s} hot={m.last_encode_hot_values():5d} waves/CU={m.last_encode_waves_per_cu():2d}  encode {best[0]:7.3f} ms   " +
              " ".join(f"{k}={v:.3f}" for k, v in best[1].items()), flush=True)
    m.free()
