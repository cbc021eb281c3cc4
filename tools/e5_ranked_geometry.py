"""encode5_kernel on the spec vocabulary with its values re-ranked by match counts: does a COLD build with four positions
per lane (a smaller LDS copy) beat the default (three positions per lane, every value in LDS)?"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
toks, scores, _ = synth.load_spec_vocab(32000)
ref = None
for rank in ("model", "counts"):
    os.environ["TGX_VALUE_RANK"] = rank
    m = tgx.NativeModel(toks, scores)
    for cfg in ({}, {"TGX_PPL": "4", "TGX_WAVES": "16"}, {"TGX_PPL": "4", "TGX_WAVES": "15"}, {"TGX_PPL": "4", "TGX_WAVES": "14"}, {"TGX_PPL": "4", "TGX_WAVES": "13"},
                {"TGX_PPL": "3", "TGX_WAVES": "14"}, {"TGX_PPL": "3", "TGX_WAVES": "15"}, {"TGX_PPL": "3", "TGX_WAVES": "16"}):
        for k in ("TGX_PPL", "TGX_WAVES"):
            os.environ.pop(k, None)
        os.environ.update(cfg)
        best = None
        for _ in range(3):
            r = m.encode_corpus(c); ids = r.ids() if ref is None else None; r.free()
            if ids is not None: ref = ids
            kt = m.last_kernel_times()
            if best is None or kt["encode5_kernel"] < best: best = kt["encode5_kernel"]
        print(f"rank={rank:6s} {str(cfg):44s} encode5 {best:7.3f} ms  values in LDS {m.last_encode_hot_values():5d} of {m.score_values()}  waves/CU {m.last_encode_waves_per_cu()}", flush=True)
    r = m.encode_corpus(c); same = bool(np.array_equal(r.ids(), ref)); r.free()
    print("   ids equal to the first run:", same)
