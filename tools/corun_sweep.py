"""Co-run of the long-sample kernel and encode5_kernel on disjoint CUs (tgx_api.cpp): pass time by length threshold and
by the CUs given to the long-sample kernel, against the default decision.  usage: python tools/corun_sweep.py [MiB ...]"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
import small_batch_sweep as sb

toks, scores, _ = synth.load_spec_vocab(32000)
models = {"spec32k": tgx.NativeModel(toks, scores), "distinct32k": tgx.NativeModel(toks, scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks)))}
cfgs = [dict(), dict(TGX_CORUN="0")]
for thr in ("16384", "24576", "32768", "40960", "49152"):
    for x in ("64", "96", "128", "160", "192"):
        cfgs.append(dict(TGX_LONG_THRESHOLD=thr, TGX_CORUN=x))
for mib in [int(a) for a in sys.argv[1:]] or [256]:
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    corpus = tgx.NativeCorpus(flat, offs)
    for name, m in models.items():
        print(f"== {mib} MiB {name}", flush=True)
        sb.KNOBS = sb.KNOBS + ("TGX_CORUN",) if "TGX_CORUN" not in sb.KNOBS else sb.KNOBS
        sb.run(m, corpus, cfgs)
    corpus.free()
