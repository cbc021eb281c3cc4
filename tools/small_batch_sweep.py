"""Batches smaller than the chip (10 / 64 / 256 MiB of the mixed corpus, samples up to 64 KiB): the pass is bounded
by the serial chain of the longest samples, not by throughput.  Sweeps the encode5_kernel geometry (waves per block,
positions per lane) and the encode6_kernel split on the spec vocabulary and on its distinct-scores variant.
usage: python tools/small_batch_sweep.py [sizes MiB ...]   (TGX_KNOBS is set here)"""
import os, sys
os.environ["TGX_KNOBS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth

KNOBS = ("TGX_PATH", "TGX_WAVES", "TGX_E5_HOT", "TGX_PPL", "TGX_BPC", "TGX_LONG_THRESHOLD", "TGX_E6_BPC")


def run(m, corpus, cfgs):
    for cfg in cfgs:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(cfg)
        best, best_kt = None, None
        for _ in range(4):
            r = m.encode_corpus(corpus)
            r.free()
            kt = m.last_kernel_times()
            t = sum(kt.values())
            if best is None or t < best:
                best, best_kt = t, kt
        enc = {k.replace("_kernel", ""): round(v, 3) for k, v in best_kt.items() if k.startswith("encode")}
        print(f"   {str(cfg):75s} hot={m.last_encode_hot_values():5d}/{m.score_values()} pass {best:7.3f} ms  {enc}", flush=True)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [10, 64, 256]
    toks, scores, _ = synth.load_spec_vocab(32000)
    models = {"spec32k": tgx.NativeModel(toks, scores),
              "distinct32k": tgx.NativeModel(toks, scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks)))}
    cfgs = [dict()]
    cfgs += [dict(TGX_LONG_THRESHOLD="0")]
    for ppl in ("2", "4"):
        for w in ("4", "6", "8", "12"):
            cfgs.append(dict(TGX_LONG_THRESHOLD="0", TGX_PPL=ppl, TGX_BPC="1", TGX_WAVES=w))
    cfgs += [dict(TGX_E6_BPC="1"), dict(TGX_E6_BPC="1", TGX_LONG_THRESHOLD="1")]
    for thr in ("8192", "16384", "32768"):
        cfgs.append(dict(TGX_LONG_THRESHOLD=thr))
        cfgs.append(dict(TGX_LONG_THRESHOLD=thr, TGX_PPL="4", TGX_BPC="1", TGX_WAVES="8"))
    for mib in sizes:
        flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
        corpus = tgx.NativeCorpus(flat, offs)
        for name, m in models.items():
            print(f"== {mib} MiB, {offs.size - 1} samples, longest {int(np.diff(offs).max())} B, {name}", flush=True)
            run(m, corpus, cfgs)
        corpus.free()


if __name__ == "__main__":
    main()
