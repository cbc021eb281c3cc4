import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
toks, scores, _ = synth.load_spec_vocab(32000)
m = tgx.NativeModel(toks, scores, for_estep=True)
for mib in (256, 1024):
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    for pieces in ("1",):
        for eppl in ("1", "2"):
            for g in (16, 14, 12, 10, 8):
                os.environ["TGX_ESTEP_PIECES"] = pieces; os.environ["TGX_BWD_GROUPS"] = str(g); os.environ["TGX_EPPL"] = eppl
                best = None
                for _ in range(2):
                    m.estep(c); kt = m.last_kernel_times()
                    if best is None or kt["estep4l_bwd_kernel"] < best["estep4l_bwd_kernel"]: best = kt
                print(f"{mib} MiB pieces={pieces} eppl={eppl} groups={g:2d} fwd={best['estep4l_fwd_kernel']:.2f} bwd={best['estep4l_bwd_kernel']:.2f}", flush=True)
    c.free()
