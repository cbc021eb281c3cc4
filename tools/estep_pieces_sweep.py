"""E-step on pieces (csrc/cuts.hip): window size and the default decision, by shard size.
usage: python tools/estep_pieces_sweep.py [sizes MiB ...]"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth

toks, scores, _ = synth.load_spec_vocab(32000)
m = tgx.NativeModel(toks, scores, for_estep=True)
for mib in [int(a) for a in sys.argv[1:]] or [64, 256, 512]:
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    print(f"== {mib} MiB, {offs.size - 1} samples", flush=True)
    for cfg in [dict(), dict(TGX_ESTEP_PIECES="0"), dict(TGX_ESTEP_PIECES="1", TGX_ESTEP_WINDOW="512"), dict(TGX_ESTEP_PIECES="1", TGX_ESTEP_WINDOW="1024"),
                dict(TGX_ESTEP_PIECES="1", TGX_ESTEP_WINDOW="2048"), dict(TGX_ESTEP_PIECES="1", TGX_ESTEP_WINDOW="4096"), dict(TGX_ESTEP_PIECES="1", TGX_ESTEP_WINDOW="8192")]:
        for k in ("TGX_ESTEP_PIECES", "TGX_ESTEP_WINDOW"):
            os.environ.pop(k, None)
        os.environ.update(cfg)
        best = None
        for _ in range(3):
            _, z = m.estep(c)
            kt = m.last_kernel_times()
            t = sum(kt.values())
            if best is None or t < best[0]:
                best = (t, kt)
        print(f"   {str(cfg):60s} pieces={m.last_estep_pieces():7d} kernels {best[0]:7.3f} ms  " + " ".join(f"{k.replace('_kernel','')}={v:.2f}" for k, v in best[1].items()) + f"  logz={z:.6f}", flush=True)
    c.free()
