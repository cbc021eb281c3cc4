"""estep7_kernel (csrc/estep7.hip): waves per block x positions per lane x pieces, by shard size, against the chained kernels.
usage: python tools/e7_sweep.py <vocab: 32000 | 65536 | 500000 | distinct> [sizes MiB ...]"""
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
KEYS = ("TGX_ESTEP", "TGX_EPPL", "TGX_E7_WAVES", "TGX_E7_HOT", "TGX_ESTEP_PIECES", "TGX_ESTEP_WINDOW")
for mib in [int(a) for a in sys.argv[2:]] or [256, 1024]:
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    print(f"== {which}: {mib} MiB, {offs.size - 1} samples", flush=True)
    cfgs = [dict(TGX_ESTEP="chain"), dict()]
    geoms = ((8, 4), (6, 4), (10, 4), (12, 4), (8, 3), (10, 3), (12, 3), (12, 2), (8, 2))
    if len(toks) > 65535:  # 32-bit entries: 4 KiB per wave and 16 positions per lane
        geoms = ((5, 4), (6, 4), (7, 4), (8, 4), (6, 3), (7, 3), (8, 3), (9, 3), (10, 3), (12, 2))
    for w, p in geoms:
        cfgs.append(dict(TGX_E7_WAVES=str(w), TGX_EPPL=str(p)))
    cfgs += [dict(TGX_ESTEP_PIECES="0"), dict(TGX_ESTEP_PIECES="1", TGX_ESTEP_WINDOW="1024"), dict(TGX_ESTEP_PIECES="1", TGX_ESTEP_WINDOW="4096"),
             dict(TGX_E7_HOT="2048"), dict(TGX_E7_HOT="4096")]
    for cfg in cfgs:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(cfg)
        best = None
        for _ in range(3):
            _, z = m.estep(c)
            kt = m.last_kernel_times()
            t = sum(kt.values())
            if best is None or t < best[0]:
                best = (t, kt)
        print(f"   {str(cfg):60s} pieces={m.last_estep_pieces():7d} redo={m.last_estep_redo():5d} kernels {best[0]:7.3f} ms = {flat.size / best[0] / 1e6:6.1f} GB/s  " +
              " ".join(f"{k.replace('_kernel','')}={v:.2f}" for k, v in best[1].items()) + f"  logz={z:.6f}", flush=True)
    c.free()
