"""Timings of the prune / merge corpus passes (E-step, frequency pass, pair scan) next to
the CPU oracle on a bounded sample — recorded in profiles/, not part of bench.py's line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
V = int(sys.argv[2]) if len(sys.argv) > 2 else 32000
cpu_mib = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if V == 500000:  # the committed vocabulary of BASELINE.json configs[3] (tests/golden/vocab_500000.npz)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from util import load_vocab_500k
    toks, scores = load_vocab_500k()
elif V in (32000, 65536):  # the committed 64 MiB-slice vocabularies (bench.py's)
    toks, scores, _ = synth.load_spec_vocab(V)
else:
    vflat, _ = synth.make_corpus(8 << 20, "mixed", seed_offset=0)
    toks, scores = synth.build_vocab(vflat, V, 16)
if len(toks) != V:  # a slice too small for V tokens once produced a "500 K" record of 190 730 tokens
    raise SystemExit(f"vocabulary has {len(toks)} tokens, {V} were asked for")
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
m = tgx.NativeModel(toks, scores, for_estep=True); c = tgx.NativeCorpus(flat, offs)
cores = len(os.sched_getaffinity(0))
out = {"corpus_bytes": int(flat.size), "samples": int(offs.size - 1), "vocab": len(toks), "host_cores": cores}
def timed(fn, reps=3):
    fn(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
    return min(ts), r
# E-step
dt, (exp, z) = timed(lambda: m.estep(c))
kt = m.last_kernel_times()
out["estep"] = {"wall_ms": dt * 1e3, "kernel_ms": kt, "GB_per_s": flat.size / dt / 1e9, "logz_sum": z}
# frequency pass
dt, freq = timed(lambda: m.count_tokens(c))
out["count_tokens"] = {"wall_ms": dt * 1e3, "kernel_ms": m.last_kernel_times(), "GB_per_s": flat.size / dt / 1e9}
# pair scan
dt, (keys, counts) = timed(lambda: m.count_pairs(c))
out["count_pairs"] = {"wall_ms": dt * 1e3, "kernel_ms": m.last_kernel_times(), "GB_per_s": flat.size / dt / 1e9, "distinct_pairs": int(keys.size)}
# the pair scan as the merge driver asks for it: the 2^18 most frequent pairs only (tgx_count_pairs_top)
dt, (tk, tc, total) = timed(lambda: m.count_pairs_top(c, 1 << 18))
out["count_pairs_top"] = {"wall_ms": dt * 1e3, "kernel_ms": m.last_kernel_times(), "GB_per_s": flat.size / dt / 1e9, "pairs_returned": int(tk.size), "distinct_pairs": int(total)}
# CPU oracle on a bounded prefix (all host cores) + parity
k = int(np.searchsorted(offs, min(flat.size, cpu_mib << 20))); sf, so = flat[: int(offs[k])], offs[: k + 1]
ora = orc.OracleModel(toks, scores)
t = time.perf_counter(); st, wexp, wz, _ = ora.estep_flat(sf, so, threads=cores); dt = time.perf_counter() - t
cs = tgx.NativeCorpus(sf, so); gexp, gz = m.estep(cs)
rel = np.abs(gexp - wexp) / np.maximum(np.abs(wexp), 1e-300); rel[(wexp == 0) & (gexp == 0)] = 0
out["estep_cpu"] = {"sample_bytes": int(sf.size), "threads": cores, "MB_per_s": sf.size / dt / 1e6, "max_rel_err_vs_cpu": float(rel[np.abs(wexp) > 1e-9].max()), "logz_rel_diff": abs(gz - wz) / abs(wz)}
t = time.perf_counter(); wk, wc = ora.count_pairs_flat(sf, so, threads=cores); dt = time.perf_counter() - t
gk, gc = m.count_pairs(cs)
out["pairs_cpu"] = {"sample_bytes": int(sf.size), "threads": cores, "MB_per_s": sf.size / dt / 1e6, "equal": bool(np.array_equal(gk, wk) and np.array_equal(gc, wc))}
print(json.dumps(out))
