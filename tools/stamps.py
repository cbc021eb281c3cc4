"""Diagnostic: per-phase s_memtime stamps of the rows4 encode kernel (TGX_STAMPS=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TGX_STAMPS", "1")
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
m = tgx.NativeModel(toks, scores); c = tgx.NativeCorpus(flat, offs)
for _ in range(2):
    r = m.encode_corpus(c); r.free()
    print(m.last_kernel_times(), flush=True)
