"""E-step backward kernel with the staggered walk over 8-byte records (W8 builds): positions per lane x groups per block,
against the walk over the 16-byte records.  usage: python tools/bwd8_sweep.py [MiB ...]"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
toks, scores, _ = synth.load_spec_vocab(32000)
m = tgx.NativeModel(toks, scores, for_estep=True)
for mib in [int(a) for a in sys.argv[1:]] or [256]:
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    ref = None
    cfgs = [dict(TGX_ESTEP_BWD="rows4")] + [dict(TGX_EPPL=str(p), TGX_BWD_GROUPS=str(g)) for p in (1, 2, 3, 4) for g in (12, 16, 20, 24, 28) if g // p <= 16 and g // p >= 3]
    for cfg in cfgs:
        for k in ("TGX_ESTEP_BWD", "TGX_EPPL", "TGX_BWD_GROUPS"):
            os.environ.pop(k, None)
        os.environ.update(cfg)
        best = None
        for _ in range(2):
            exp, z = m.estep(c)
            kt = m.last_kernel_times()
            if best is None or kt["estep4l_bwd_kernel"] < best["estep4l_bwd_kernel"]: best = kt
        if ref is None: ref = exp
        err = float(np.max(np.abs(exp - ref) / np.maximum(np.abs(ref), 1e-300)))
        print(f"{mib} MiB {str(cfg):55s} fwd={list(best.items())[-3][1] if False else [v for k,v in best.items() if 'fwd' in k][0]:6.2f} bwd={best['estep4l_bwd_kernel']:6.2f} ms  rel.diff vs rows4 {err:.1e}", flush=True)
    c.free()
