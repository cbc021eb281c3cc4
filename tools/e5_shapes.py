"""encode5_kernel by corpus shape and positions per lane: throughput on large corpora of long / short samples,
and the serial chain of long samples (a corpus of a few 64 KiB samples: each alone on its row)."""
import os, sys
os.environ.setdefault("TGX_LONG_THRESHOLD", "0")  # encode5_kernel alone: no long samples to encode6_kernel
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
cfgs = ((1, 16, 2), (2, 10, 2), (2, 16, 1), (4, 16, 1), (4, 8, 2))
def run(tag, flat, offs):
    c = tgx.NativeCorpus(flat, offs)
    ref = None
    for ppl, waves, bpc in cfgs:
        os.environ["TGX_PPL"], os.environ["TGX_WAVES"], os.environ["TGX_BPC"] = str(ppl), str(waves), str(bpc)
        best = None
        for _ in range(3):
            r = m.encode_corpus(c); ids = r.ids(); r.free()
            kt = m.last_kernel_times()
            best = kt if best is None or kt["encode5_kernel"] < best["encode5_kernel"] else best
        if ref is None: ref = ids
        print(f"{tag:28s} ppl={ppl} waves={waves:2d} bpc={bpc} encode5={best.get('encode5_kernel', 0):8.3f} ms trace={best.get('trace_kernel', 0):7.3f} ms "
              f"GB/s(encode5)={flat.size / best['encode5_kernel'] / 1e6:6.1f} same_ids={bool(np.array_equal(ids, ref))}", flush=True)
    c.free()
# chain: 64 samples of exactly 64 KiB, 1 sample of 256 KiB
flat, _ = synth.make_corpus(8 << 20, "mixed", seed_offset=7)
run("chain 64 x 64 KiB", flat[: 64 * 65536], np.arange(65, dtype=np.uint64) * 65536)
run("chain 1 x 256 KiB", flat[: 262144], np.array([0, 262144], dtype=np.uint64))
for size, ml in ((10, 65536), (64, 65536), (256, 65536), (1024, 65536), (1024, 4096), (1024, 1024), (256, 256)):
    flat, offs = synth.make_corpus(size << 20, "mixed", max_len=ml, seed_offset=1000)
    run(f"{size} MiB, samples <= {ml}", flat, offs)
