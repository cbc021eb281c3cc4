"""Host time of model creation and of the first E-step (which builds the backward tables) with and without\nTGX_MODEL_FOR_ESTEP, at 32 K and 500 K tokens (needs cache/vocab_500000.npz: tools/make_vocab_cache.py)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
z = np.load("/root/repo/cache/vocab_500000.npz"); o = z["offs"].astype(np.int64); fb = z["flat"].tobytes()
toks = [fb[o[i]:o[i + 1]] for i in range(o.size - 1)]; scores = z["scores"]
flat, offs = synth.make_corpus(64 << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
for n in (32000, 500000):
    for fe in (False, True):
      t = time.perf_counter(); m = tgx.NativeModel(toks[:n], scores[:n], for_estep=fe); t1 = time.perf_counter()
      m.estep(c); t2 = time.perf_counter(); m.estep(c); t3 = time.perf_counter()
      print(f"V={n} for_estep={fe}: model create {t1 - t:.3f} s, first estep {t2 - t1:.3f} s, second estep {t3 - t2:.3f} s  {m.last_kernel_times()}", flush=True)
