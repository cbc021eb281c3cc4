"""Staged GPU bring-up: each stage appends a line to gpurun_out/debug.log before it
starts, so a device fault can be attributed to a stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.makedirs("gpurun_out", exist_ok=True)
LOG = open("gpurun_out/debug.log", "a")
def log(*a):
    print(*a, file=LOG, flush=True); print(*a, flush=True)
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
log("devices", tgx.device_count())
toks = [b"a", b"b", b"c", b"ab"]; scores = [-3.0, -3.0, -3.0, -4.0]
log("stage 1: model create"); m = tgx.NativeModel(toks, scores); log("  ok lm", m.max_token_len, "trie bytes", m.trie_bytes)
log("stage 2: corpus upload"); flat, offs = tgx.pack([b"abc"]); c = tgx.NativeCorpus(flat, offs); log("  ok")
log("stage 3: encode 'abc'"); r = m.encode_corpus(c); log("  ok tokens", r.num_tokens, r.ids().tolist(), r.offsets().tolist())
log("stage 4: batch of small strings"); texts = [b"", b"a", b"abcabc", b"c" * 70, b"ab" * 100]
flat, offs = tgx.pack(texts); r = m.encode_batch_flat(flat, offs); ids, oo = r.ids(), r.offsets()
got = [ids[int(oo[i]):int(oo[i+1])].tolist() for i in range(len(texts))]
want = orc.OracleModel(toks, scores).encode_batch(texts); log("  ok equal:", got == want)
from tokengeex_amd import synth
for size in (4 << 10, 64 << 10, 1 << 20):
    log("stage 5: synthetic", size); flat, offs = synth.make_corpus(size, "mixed")
    vt, vs = synth.build_vocab(flat, 2000, 16)
    nm = tgx.NativeModel(vt, vs); log("  model ok, trie bytes", nm.trie_bytes)
    r = nm.encode_batch_flat(flat, offs); ids, oo = r.ids(), r.offsets()
    w_ids, w_oo = orc.OracleModel(vt, vs).encode_batch_flat(flat, offs, threads=4)
    log("  ok equal:", bool(np.array_equal(ids, w_ids) and np.array_equal(oo, w_oo)), "tokens", ids.size, nm.last_kernel_times())
log("all stages done")
