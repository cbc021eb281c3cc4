"""Host time of model creation at 32 000 and 500 000 tokens (tests/golden/vocab_500000.npz), for encode and for
E-step passes, and of the first / second E-step (the first uploads the backward tables)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth, _lib
z = np.load(os.path.join(ROOT, "tests", "golden", "vocab_500000.npz"))
fb = z["flat"].tobytes(); o = np.concatenate([[0], np.cumsum(z["lens"].astype(np.int64))])
toks = [fb[o[i]:o[i + 1]] for i in range(o.size - 1)]; scores = z["uscores"][z["inv"]].astype(np.float64)
flat, offs = synth.make_corpus(64 << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
for n in (32000, 500000):
    packed = _lib.Packed.of(toks[:n])
    for fe in (False, True):
        best = None
        for _ in range(3):
            t = time.perf_counter(); m = tgx.NativeModel(packed, scores[:n], for_estep=fe); t1 = time.perf_counter()
            if fe:
                m.estep(c); t2 = time.perf_counter(); m.estep(c); t3 = time.perf_counter()
                rec = (t1 - t, t2 - t1, t3 - t2)
            else:
                r = m.encode_corpus(c); r.free(); t2 = time.perf_counter(); r = m.encode_corpus(c); r.free(); t3 = time.perf_counter()
                rec = (t1 - t, t2 - t1, t3 - t2)
            m.free()
            best = rec if best is None or rec[0] < best[0] else best
        print(f"V={n} for_estep={fe}: model create {best[0]:.3f} s, first pass {best[1]:.3f} s, second pass {best[2]:.3f} s", flush=True)
