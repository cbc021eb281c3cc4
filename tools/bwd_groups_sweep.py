"""E-step backward kernel: groups of 16 positions per block (TGX_BWD_GROUPS) against hot slots in LDS.
Fewer groups = fewer waves, but more expected counts summed in LDS instead of memory-side f64 atomics."""
import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores, for_estep=True)
for max_len in (65536, 4096):
    flat, offs = synth.make_corpus(size << 20, "mixed", max_len=max_len, seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    ref = None
    for eppl in ("", "1", "2", "4"):
        for g in (12, 10, 8, 6, 4):
            os.environ["TGX_BWD_GROUPS"] = str(g)
            if eppl: os.environ["TGX_EPPL"] = eppl
            else: os.environ.pop("TGX_EPPL", None)
            m.estep(c)
            exp, z = m.estep(c)
            kt = m.last_kernel_times()
            if ref is None: ref = exp
            err = float(np.max(np.abs(exp - ref) / np.maximum(np.abs(ref), 1e-300)))
            print(f"max_len={max_len:6d} eppl={eppl or 'auto':4s} groups={g:2d} hot={(160*1024 - g*12288)//8:5d} fwd={kt.get('estep4l_fwd_kernel',0):7.3f} bwd={kt.get('estep4l_bwd_kernel',0):7.3f} ms  rel.diff {err:.1e}", flush=True)
    c.free()
