"""Phase stamps of estep7_kernel (TGX_DEBUG=1 TGX_STAMPS=7).  usage: python tools/e7_stamps.py [MiB] [waves] [ppl]"""
import os, sys
os.environ["TGX_KNOBS"] = "1"; os.environ["TGX_DEBUG"] = "1"; os.environ["TGX_STAMPS"] = "7"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tokengeex_amd as tgx
from tokengeex_amd import synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2: os.environ["TGX_E7_WAVES"] = sys.argv[2]
if len(sys.argv) > 3: os.environ["TGX_EPPL"] = sys.argv[3]
toks, scores, _ = synth.load_spec_vocab(32000)
m = tgx.NativeModel(toks, scores, for_estep=True)
flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
for _ in range(2):
    m.estep(c)
    print(m.last_kernel_times(), flush=True)
