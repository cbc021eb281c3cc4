"""Rate of the Python Tokenizer surface (list[str] in, list[list[int]] out: bindings/python/src/lib.rs:51-59
upstream) next to the flat-buffer entry point, on the same samples."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
toks, scores, _ = synth.load_spec_vocab(32000)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000, max_len=8192)
o = offs.astype(np.int64)
texts = [flat[o[i]:o[i + 1]].tobytes().decode("utf-8") for i in range(o.size - 1)]
tok = tgx.Tokenizer([(t, float(s), len(t) == 1) for t, s in zip(toks, scores)])
out = {"bytes": int(flat.size), "samples": len(texts)}
for name, fn in (("encode_batch(list[str]) -> list[list[int]]", lambda: tok.encode_batch(texts, 0.0)),
                 ("encode_ordinary_batch_flat(u8, u64) -> (u32, u64)", lambda: tok.encode_ordinary_batch_flat(flat, offs))):
    fn()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t)
    out[name] = {"ms": best * 1e3, "GB_per_s": flat.size / best / 1e9}
# where the time of the list surface goes: its two native ends alone (csrc/pyfast.c), and what one Python int object per
# token costs however it is made (numpy's C loop: the floor of any binding that returns fresh ints)
from tokengeex_amd import _tgxfast
ids, oo = tok.encode_ordinary_batch_flat(flat, offs)
t = time.perf_counter(); tb, ob = _tgxfast.pack_strs(texts); out["pack_strs_ms"] = (time.perf_counter() - t) * 1e3
t = time.perf_counter(); rows = _tgxfast.rows_from_flat(ids, oo, list(range(len(toks)))); out["rows_from_flat_shared_ints_ms"] = (time.perf_counter() - t) * 1e3
t = time.perf_counter(); lst = ids.tolist(); out["ids_tolist_fresh_ints_ms"] = (time.perf_counter() - t) * 1e3
out["tokens"] = int(ids.size)
print(json.dumps(out))
