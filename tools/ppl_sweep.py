"""encode4_kernel time vs positions per lane (TGX_PPL) and corpus size: the serial chain of the longest
sample bounds small corpora, more positions per lane shorten it at the price of fewer waves per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
for size in (1024, 256, 64):
    flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    ref = None
    for ppl, waves, bpc in ((1, 10, 2), (2, 5, 2), (2, 10, 1), (4, 5, 1), (4, 2, 2), (1, 10, 2)):
        os.environ["TGX_PPL"], os.environ["TGX_WAVES"], os.environ["TGX_BPC"] = str(ppl), str(waves), str(bpc)
        r = m.encode_corpus(c); ids = r.ids(); r.free()
        if ref is None: ref = ids
        kt = m.last_kernel_times()
        print(f"size={size:5d} MiB ppl={ppl} waves={waves:2d} bpc={bpc} encode4={kt.get('encode4_kernel', 0):8.3f} ms trace={kt.get('trace_kernel', 0):7.3f} ms same_ids={bool(np.array_equal(ids, ref))}", flush=True)
    c.free()
