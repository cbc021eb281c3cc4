"""Locate the first mismatch between the HIP path and the oracle (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth
max_len = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(1000 + max_len)
flat, offs = synth.make_corpus(256 << 10, "mixed", seed_offset=max_len)
toks, scores = synth.random_vocab(rng, bytes(flat[: 64 << 10]), n_multi=3000, max_len=max_len)
nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
print("max token len", nat.max_token_len, flush=True)
nbad = 0
for i in range(len(offs) - 1):
    t = bytes(flat[int(offs[i]):int(offs[i + 1])])
    want = ora.encode(t)
    try:
        f1, o1 = tgx.pack([t]); r = nat.encode_batch_flat(f1, o1); got = r.ids().tolist()
    except tgx.TokenGeeXError as e:
        got = None; print("sample", i, "len", len(t), "ERROR", e, flush=True)
    if got != want:
        nbad += 1
        if got is not None:
            k = next(j for j in range(min(len(got), len(want))) if got[j] != want[j]) if got[:len(want)] != want[:len(got)] else min(len(got), len(want))
            pos = sum(len(toks[x]) for x in want[:k])
            print("sample", i, "len", len(t), "first diff token", k, "byte pos", pos, "pos%64", pos % 64, "got", got[k:k+3], "want", want[k:k+3],
                  [toks[x] for x in want[k:k+3]], [toks[x] for x in got[k:k+3]], flush=True)
        if nbad > 12: break
print("bad samples", nbad)
