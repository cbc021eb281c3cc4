"""Long-sample kernel (encode6_kernel) against encode5_kernel alone (TGX_LONG_THRESHOLD=0) by corpus shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
def run(tag, flat, offs):
    c = tgx.NativeCorpus(flat, offs)
    ref = None
    for thr in ("0", None, "2048", "8192", "32768"):
        if thr is None: os.environ.pop("TGX_LONG_THRESHOLD", None)
        else: os.environ["TGX_LONG_THRESHOLD"] = thr
        best = None
        for _ in range(3):
            r = m.encode_corpus(c); ids = r.ids(); r.free()
            kt = m.last_kernel_times()
            tot = kt.get("encode5_kernel", 0) + kt.get("encode6_kernel", 0)
            if best is None or tot < best[0]: best = (tot, dict(kt))
        if ref is None: ref = ids
        print(f"{tag:26s} thr={str(thr):7s} long={m.last_encode_long_samples():6d} e6={best[1].get('encode6_kernel', 0):7.3f} e5={best[1].get('encode5_kernel', 0):7.3f} "
              f"trace={best[1].get('trace_kernel', 0):6.3f} ms  GB/s(e5+e6)={flat.size / best[0] / 1e6:6.1f} same={bool(np.array_equal(ids, ref))}", flush=True)
    c.free()
flat, _ = synth.make_corpus(8 << 20, "mixed", seed_offset=7)
run("chain 64 x 64 KiB", flat[: 64 * 65536], np.arange(65, dtype=np.uint64) * 65536)
run("chain 1 x 256 KiB", flat[: 262144], np.array([0, 262144], dtype=np.uint64))
for size, ml in ((10, 65536), (64, 65536), (256, 65536), (512, 65536), (1024, 65536)):
    flat, offs = synth.make_corpus(size << 20, "mixed", max_len=ml, seed_offset=1000)
    run(f"{size} MiB, samples <= {ml}", flat, offs)
