"""E-step kernel times on the benchmark corpus (samples up to 64 KiB) and on short samples only (<= 4 KiB):
the second shows the throughput limit, the first is bound by the serial chain of the longest snippets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
m = tgx.NativeModel(toks, scores)
for max_len in (65536, 4096):
    flat, offs = synth.make_corpus(size << 20, "mixed", max_len=max_len, seed_offset=1000)
    c = tgx.NativeCorpus(flat, offs)
    for _ in range(2):
        m.estep(c)
        print(f"max_len={max_len} samples={offs.size - 1} {m.last_kernel_times()}", flush=True)
    c.free()
