import os, sys
sys.path.insert(0, "/root/repo")
os.environ["TGX_DEBUG"] = "1"
import tokengeex_amd as tgx
from tokengeex_amd import synth
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
flat, offs = synth.make_corpus(1024 << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
for fl in (0, 16, 4, 0):
    os.environ["TGX_FLAGS"] = str(fl)
    try:
        r = m.encode_corpus(c); r.free()
    except tgx.TokenGeeXError:
        pass
    print("flags", fl, m.last_kernel_times(), flush=True)
