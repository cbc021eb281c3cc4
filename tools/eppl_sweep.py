"""Linear-domain E-step kernel times vs positions per lane (TGX_EPPL) on corpora with 64 KiB and with
<= 4 KiB samples: the chain of the longest snippet against waves per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
for max_len in (65536, 4096):
    flat, offs = synth.make_corpus(size << 20, "mixed", max_len=max_len, seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    ref = None
    for ppl in (1, 2, 4, 1):
        os.environ["TGX_EPPL"] = str(ppl)
        exp, z = m.estep(c)
        if ref is None: ref = exp
        rel = float(np.max(np.abs(exp - ref) / np.maximum(np.abs(ref), 1e-300) * (np.abs(ref) > 1e-9)))
        print(f"size={size} max_len={max_len} eppl={ppl} {m.last_kernel_times()} max_rel_vs_eppl1={rel:.2e}", flush=True)
    c.free()
