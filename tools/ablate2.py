"""Where does encode4_kernel's time go?  Full / no-walk / no-relax timings (TGX_FLAGS 0/1/2; results are
WRONG when flags != 0) on the benchmark corpus (samples up to 64 KiB) and on one with short samples only
(max 4 KiB: no long serial chains)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TGX_DEBUG"] = "1"
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
for max_len in (65536, 4096, 1024):
    flat, offs = synth.make_corpus(size << 20, "mixed", max_len=max_len, seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    for fl in (0, 1, 2, 0):
        os.environ["TGX_FLAGS"] = str(fl)
        try:
            r = m.encode_corpus(c); r.free()
        except tgx.TokenGeeXError:
            pass
        kt = m.last_kernel_times()
        print(f"max_len={max_len:6d} samples={offs.size - 1:8d} flags={fl} encode4={kt.get('encode4_kernel', 0):8.3f} ms trace={kt.get('trace_kernel', 0):7.3f} ms", flush=True)
    c.free()
