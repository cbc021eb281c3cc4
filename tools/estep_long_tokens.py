"""E-step with a vocabulary whose tokens reach 24 bytes (as after `merge`): the long-token builds of the
linear-domain kernels against the generic kernel (TGX_PATH=fused) on the same corpus."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 24)
print("longest token", max(len(t) for t in toks), "tokens > 16 bytes:", sum(len(t) > 16 for t in toks), flush=True)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
m = tgx.NativeModel(toks, scores)
c = tgx.NativeCorpus(flat, offs)
for path in (None, "fused"):
    if path: os.environ["TGX_PATH"] = path
    best = None
    for _ in range(2):
        exp, z = m.estep(c, 81920)
        kt = m.last_kernel_times()
        tot = sum(kt.values())
        if best is None or tot < best[0]: best = (tot, kt, exp, z)
    print(path or "default", {k: round(v, 3) for k, v in best[1].items()}, f"{flat.size / best[0] / 1e6:.2f} GB/s", flush=True)
    if path is None: ref = best
print("max rel diff linear vs generic", float(np.max(np.abs(best[2] - ref[2]) / np.maximum(np.abs(ref[2]), 1e-300) * (np.abs(ref[2]) > 1e-9))), "dz", abs(best[3] - ref[3]) / abs(ref[3]))
