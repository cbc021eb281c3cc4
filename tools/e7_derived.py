"""The E-step of prune's second sub-iteration at 500 000 entries (a derived model of ~197 K tokens with M-step scores):
kernel times, pieces, redo stretches, hot entries.  usage: e7_derived.py [corpus MiB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import tokengeex_amd as tgx
from tokengeex_amd import synth, _lib
from util import load_vocab_500k
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
toks, scores = load_vocab_500k()
flat, offs = synth.make_corpus(size << 20, "mixed", seed_offset=1000)
c = tgx.NativeCorpus(flat, offs)
m = tgx.NativeModel(toks, scores, for_estep=True)
keep = np.array([1 if len(t) == 1 else 0 for t in toks], np.uint8)
def run(model, tag, dropout=0.01, seed=1):
    for rep in range(2):
        t = time.perf_counter(); exp, z = model.estep(c, _lib.ESTEP_SNIPPET_LEN, dropout, seed); dt = time.perf_counter() - t
        print(f"{tag:34s} rep={rep} wall={dt * 1e3:7.2f} ms kernels={model.last_kernel_times()} pieces={model.last_estep_pieces()} redo={model.last_estep_redo()} V={model.vocab_size}", flush=True)
    return exp
exp = run(m, "500K spec scores")
idx, sc2 = _lib.prune_m_step(exp, keep)
idx = np.asarray(idx, np.uint32); sc2 = np.asarray(sc2, np.float64)
d = m.derive(idx, sc2, for_estep=True)
exp2 = run(d, "derived 197K, M-step scores", seed=2)
toks2 = [toks[i] for i in idx]
f = tgx.NativeModel(toks2, sc2, for_estep=True)
exp3 = run(f, "from scratch 197K, M-step scores", seed=2)
print("derived == scratch:", float(np.max(np.abs(exp2 - exp3) / np.maximum(np.abs(exp3), 1e-300))))
run(d, "derived, no dropout", dropout=0.0)
print("score range", float(sc2.min()), float(sc2.max()), "n", len(sc2))
if len(sys.argv) > 2 and sys.argv[2] == "short":
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "matches":
    # how often does every token of the derived vocabulary MATCH (lattice edges = adds to its expected count)?
    n = 1 << 20
    text = bytes(flat[:n])
    cnt = np.zeros(len(toks2), np.int64)
    for p in range(n):
        for tid, ln in d.common_prefix_search(text[p:p + 16]):
            cnt[tid] += 1
    lens = np.array([max(1, len(t)) for t in toks2], np.float64)
    r_old = np.empty(len(toks2), np.int64); r_old[np.argsort(-(np.exp(sc2) / lens), kind="stable")] = np.arange(len(toks2))
    top = np.argsort(-cnt)[:40]
    print("matches in 1 MiB: total", int(cnt.sum()), "tokens matched", int((cnt > 0).sum()))
    for i in top:
        print(f"  id={i:7d} matches={cnt[i]:8d} marginal_sum_256MiB={exp2[i]:12.1f} score={sc2[i]:8.3f} rank_by_exp_over_len={r_old[i]:7d} token={toks2[i]!r}")
    order = np.argsort(r_old)
    c_sorted = cnt[order]
    for nh in (2000, 3500, 8000, 16384):
        print("cold matches beyond rank", nh, int(c_sorted[nh:].sum()), "max single", int(c_sorted[nh:].max()))
    sys.exit(0)
def cold_share(tk, sc, ex, n_hot):
    lens = np.array([max(1, len(t)) for t in tk], np.float64)
    order = np.argsort(-(np.exp(sc) / lens), kind="stable")
    cold = ex[order[n_hot:]]
    top = np.sort(cold)[::-1][:8]
    return float(cold.sum() / ex.sum()), [float(x) for x in top], float(ex.sum())
for nh in (2000, 3500, 5000):
    print("n_hot", nh, "spec 500K: cold share, top cold counts, total", cold_share(toks, np.asarray(scores), exp, nh))
    print("n_hot", nh, "M-step 197K:", cold_share(toks2, sc2, exp2, nh))
for hot in ("4000", "2000", "500"):
    os.environ["TGX_E7_HOT"] = hot
    run(d, f"derived, TGX_E7_HOT={hot}", seed=2)
    run(m, f"500K spec, TGX_E7_HOT={hot}", seed=2)
del os.environ["TGX_E7_HOT"]
for w, pp in ((12, 1), (8, 2), (8, 4), (6, 4)):
    os.environ["TGX_E7_WAVES"] = str(w); os.environ["TGX_EPPL"] = str(pp)
    run(d, f"derived, waves={w} ppl={pp}", seed=2)
