"""encode of vocabularies whose score values do NOT all fit encode5_kernel's LDS table: the 32 000-entry bench
vocabulary after an M-step (every token its own score) and the 500 000-entry vocabulary of configs[3], by table
size (TGX_E5_MAX_HOT), positions per lane, waves and pool size, against encode4_kernel (TGX_PATH=rows4).
One process per configuration (the table size is fixed when the model is built).
usage: python tools/cold_bench.py [size_mb]            (driver)   |   python tools/cold_bench.py --one <vocab> <size_mb>"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(vocab: str, size_mb: int):
    import numpy as np
    import tokengeex_amd as tgx
    from tokengeex_amd import synth
    if vocab == "500k":  # the committed 500 000-entry vocabulary of configs[3] (tests/golden/vocab_500000.npz)
        z = np.load(os.path.join(ROOT, "tests", "golden", "vocab_500000.npz"))
        fb = z["flat"].tobytes()
        o = np.concatenate([[0], np.cumsum(z["lens"].astype(np.int64))])
        toks = [fb[o[i]:o[i + 1]] for i in range(o.size - 1)]
        scores = z["uscores"][z["inv"]].astype(np.float64)
    else:
        vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
        toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
        rng = np.random.default_rng(5)
        scores = scores + rng.uniform(-0.4, 0.4, scores.size)  # as after an M-step: all values distinct
    flat, offs = synth.make_corpus(size_mb << 20, "mixed", seed_offset=1000)
    m = tgx.NativeModel(toks, scores)
    c = tgx.NativeCorpus(flat, offs)
    best = None
    for _ in range(3):
        r = m.encode_corpus(c)
        n_tok = r.num_tokens
        r.free()
        kt = m.last_kernel_times()
        tot = sum(v for k, v in kt.items() if k.startswith("encode"))
        if best is None or tot < best[0]:
            best = (tot, kt)
    print(f"    tokens={n_tok} redo={m.last_encode_redo_samples()} waves/CU={m.last_encode_waves_per_cu()} encode kernels {best[0]:.3f} ms: "
          + " ".join(f"{k}={v:.3f}" for k, v in best[1].items()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cfgs = [dict(TGX_PATH="rows4"), dict(TGX_PATH="rows5"), dict(TGX_PATH="rows5", TGX_LONG_THRESHOLD="0"),
            dict(TGX_PATH="rows5", TGX_PPL="2"), dict(TGX_PATH="rows5", TGX_PPL="2", TGX_LONG_THRESHOLD="0"),
            dict(TGX_PATH="rows5", TGX_PPL="2", TGX_E5_POOL="128", TGX_LONG_THRESHOLD="0"),
            dict(TGX_PATH="rows5", TGX_PPL="4", TGX_E5_POOL="128", TGX_LONG_THRESHOLD="0"),
            dict(TGX_PATH="rows5", TGX_E5_MAX_HOT="6600", TGX_PPL="2", TGX_E5_POOL="96", TGX_LONG_THRESHOLD="0")]
    for vocab in ("32k-distinct", "500k"):
        for cfg in cfgs:
            print(f"{vocab} {size} MiB {cfg}", flush=True)
            env = dict(os.environ, **cfg)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", vocab, str(size)], env=env, timeout=300)
