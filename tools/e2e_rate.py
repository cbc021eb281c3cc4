"""PCIe-inclusive rate of the C ABI's host-buffer entry point: tgx_encode_batch (text and offsets from host
memory, H2D, the kernels) and the copy of ids + offsets back to the host.  Never bench.py's `value`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
best = None
for _ in range(4):
    t0 = time.perf_counter()
    res = m.encode_batch_flat(flat, offs)
    t1 = time.perf_counter()
    ids, oo = res.ids(), res.offsets()
    t2 = time.perf_counter()
    n_tok = int(ids.size)
    res.free()
    cur = {"upload_and_encode_ms": (t1 - t0) * 1e3, "ids_to_host_ms": (t2 - t1) * 1e3, "total_ms": (t2 - t0) * 1e3}
    if best is None or cur["total_ms"] < best["total_ms"]:
        best = cur
best.update({"bytes": int(flat.size), "tokens": n_tok, "GB_per_s_host_to_host": flat.size / best["total_ms"] / 1e6,
             "GB_per_s_upload_and_encode": flat.size / best["upload_and_encode_ms"] / 1e6,
             "pcie_bytes_per_text_byte": (flat.size + offs.nbytes + 4 * n_tok + offs.nbytes) / flat.size})
print(json.dumps(best))
