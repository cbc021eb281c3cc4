"""estep7_kernel at its defaults: kernel times at a few sizes (spec vocabulary).  usage: python tools/e7_quick.py [vocab] [MiB ...]"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
which = sys.argv[1] if len(sys.argv) > 1 else "32000"
if which == "500000":
    from util import load_vocab_500k
    toks, scores = load_vocab_500k()
elif which == "distinct":
    toks, scores, _ = synth.load_spec_vocab(32000)
    scores = np.asarray(scores) + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks))
else:
    toks, scores, _ = synth.load_spec_vocab(int(which))
m = tgx.NativeModel(toks, scores, for_estep=True)
for mib in [int(a) for a in sys.argv[2:]] or [256, 1024]:
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    best = None
    for _ in range(4):
        _, z = m.estep(c)
        kt = m.last_kernel_times()
        t = sum(kt.values())
        if best is None or t < best[0]:
            best = (t, kt)
    print(f"{which} {mib} MiB: pieces={m.last_estep_pieces()} redo={m.last_estep_redo()} kernels {best[0]:.3f} ms = {flat.size / best[0] / 1e6:.1f} GB/s  " +
          " ".join(f"{k.replace('_kernel','')}={v:.2f}" for k, v in best[1].items()), flush=True)
    c.free()
