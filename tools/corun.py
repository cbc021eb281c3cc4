"""Can trace_kernel hide under encode4_kernel?  TGX_FLAGS=64 launches the trace of the previous pass's
back-pointers (same corpus, same bytes) on a second stream while encode4_kernel runs; the "encode4_kernel"
time then covers both.  Compare with encode4 alone and with encode4 + trace back to back."""
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
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
for fl in (0, 0, 64, 64, 64, 0):
    os.environ["TGX_FLAGS"] = str(fl)
    r = m.encode_corpus(c); r.free()
    kt = m.last_kernel_times()
    print(f"flags={fl:3d} encode4(+co-run)={kt.get('encode4_kernel', 0):8.3f} ms trace={kt.get('trace_kernel', 0):7.3f} ms", flush=True)
