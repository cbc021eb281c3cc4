"""E-step kernels at 1 GiB / 32 K with parts switched off (TGX_DEBUG=1 TGX_FLAGS=8: no cold atomics — results WRONG,
times only) and by TGX_BWD_GROUPS; one process per configuration."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import numpy as np
    import tokengeex_amd as tgx
    from tokengeex_amd import synth
    vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
    toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
    scores = scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks))
    flat, offs = synth.make_corpus(int(sys.argv[2]) << 20, "mixed", seed_offset=1000)
    m = tgx.NativeModel(toks, scores)
    c = tgx.NativeCorpus(flat, offs)
    best = None
    for _ in range(3):
        m.estep(c, 65536)
        kt = m.last_kernel_times()
        tot = sum(kt.values())
        if best is None or tot < best[0]: best = (tot, kt)
    print("   ", {k: round(v, 3) for k, v in best[1].items()}, flush=True)
    sys.exit(0)
size = sys.argv[1] if len(sys.argv) > 1 else "1024"
for cfg in ({}, {"TGX_BWD_GROUPS": "14"}, {"TGX_BWD_GROUPS": "12"}):
    print(cfg, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one", size], env=dict(os.environ, **cfg), timeout=300)
