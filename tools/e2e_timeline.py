import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth, _lib
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
flat, offs = synth.make_corpus(1024 << 20, "mixed", seed_offset=1000)
ids = np.empty(flat.size // 3, np.uint32)
for i in range(3):
    if i == 2: os.environ["TGX_DEBUG"] = "1"
    t0 = time.perf_counter(); m.encode_batch_host(flat, offs, ids_out=ids); print("total ms", (time.perf_counter() - t0) * 1e3, flush=True)
