"""Timing ablations of the encode kernel (results are WRONG when TGX_FLAGS != 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TGX_DEBUG"] = "1"  # TGX_FLAGS is only honoured in debug mode
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
m = tgx.NativeModel(toks, scores); c = tgx.NativeCorpus(flat, offs)
names = {0: "full"}
paths = ["rows4:1:4:5", "rows4:1:5:4", "rows4:1:2:8"]
for rnd in range(2):
  for path in paths:
    os.environ["TGX_PATH"] = path.split(":")[0]
    if ":" in path:
        os.environ["TGX_PPL"] = path.split(":")[1]
        os.environ["TGX_WAVES"] = path.split(":")[2]
        os.environ["TGX_BPC"] = path.split(":")[3]
    for fl, nm in names.items():
        nm = path + " " + nm
        os.environ["TGX_FLAGS"] = str(fl)
        try:
            r = m.encode_corpus(c); r.free()
        except tgx.TokenGeeXError as e:
            pass
        kt = m.last_kernel_times()
        t = sum(v for k, v in kt.items() if k.startswith(("encode", "trace")))
        print(f"round {rnd} flags={fl} {nm:34s} {t:8.3f} ms  {flat.size / t / 1e6:8.2f} GB/s  {kt}", flush=True)
