"""Re-runs one case of tests/measure/fuzz_gpu.py (seed, case) and prints the E-step's expected count of a token under
several kernel choices next to the oracle's.  usage: fuzz_case.py SEED CASE TOKEN_ID"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth
seed0, case, tok = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
base_flat, _ = synth.make_corpus(2 << 20, "mixed", seed_offset=77)
base = bytes(base_flat)
rng = np.random.default_rng(seed0 * 100003 + case)
max_len = int(rng.choice([2, 3, 5, 8, 12, 15, 16, 17, 20, 24, 31, 32, 33, 40]))
all_bytes = bool(rng.random() < 0.8)
toks, scores = synth.random_vocab(rng, base[: 64 << 10], n_multi=int(rng.integers(50, 3000)), max_len=max_len,
                                  all_bytes=all_bytes, tie_fraction=float(rng.choice([0.0, 0.2, 0.6])))
if rng.random() < 0.3:
    k = int(rng.integers(1, 20))
    idx = rng.integers(0, len(toks), k)
    toks = toks + [toks[i] for i in idx]
    scores = np.concatenate([scores, -rng.random(k) * 5])
lens = []
for _ in range(int(rng.integers(1, 400))):
    r = rng.random()
    if r < 0.1: lens.append(int(rng.choice([0, 1, 2])))
    elif r < 0.4: lens.append(int(rng.choice([15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129])))
    elif r < 0.97: lens.append(int(rng.integers(3, 3000)))
    else: lens.append(int(rng.integers(20000, 90000)))
texts = []
for n in lens:
    if rng.random() < 0.05:
        texts.append(bytes(rng.integers(0, 256, n).astype(np.uint8)))
    else:
        o = int(rng.integers(0, len(base) - n - 1))
        texts.append(base[o:o + n])
flat, offs = tgx.pack(texts)
dropout = float(rng.choice([0.0, 0.0, 0.1, 0.5, 1.0]))
sd = int(rng.integers(0, 1 << 62))
# (the draws of the switches follow in the fuzzer; the snippet length is drawn after them: try all)
d = dropout if dropout < 1.0 else 0.3
ora = orc.OracleModel(toks, scores)
corpus = tgx.NativeCorpus(flat, offs)
print("token", tok, toks[tok], scores[tok], "dropout", d, "samples", len(texts), "longest", max(lens))
for snip in (81920, 4096):
    st, want, wz, _ = ora.estep_flat(flat, offs, snip, d, sd, threads=8)
    print(f"snip={snip} oracle           {want[tok]:.12f}  logz {wz:.6f}")
    for env in ({}, {"TGX_ESTEP_PIECES": "1", "TGX_ESTEP_WINDOW": "512"}, {"TGX_ESTEP_PIECES": "0"}, {"TGX_ESTEP": "log"}, {"TGX_PATH": "fused"}):
        for k in ("TGX_ESTEP_PIECES", "TGX_ESTEP_WINDOW", "TGX_ESTEP", "TGX_PATH"):
            os.environ.pop(k, None)
        os.environ.update(env)
        nat = tgx.NativeModel(toks, scores, for_estep=True)
        got, gz = nat.estep(corpus, snip, d, sd)
        bad = np.nonzero(~np.isclose(got, want, rtol=1.2e-8, atol=1e-12))[0]
        print(f"snip={snip} {str(env):52s} {got[tok]:.12f}  logz {gz:.6f}  kernels {sorted(nat.last_kernel_times())}  tokens beyond rtol 1.2e-8: {bad.size}")
