"""Why encode cannot be cut into pieces the way the E-step is: Viterbi decisions between candidates whose f64 sums are equal
or within rounding of each other (they depend on the running sum from the sample start, src/model.rs:96-108), counted
on the CPU.  usage: python tools/viterbi_ties.py <spec32k | distinct | 2mib> [MiB]"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
from tokengeex_amd import synth, _lib
import hot_coverage as hc
name = sys.argv[1]; mib = float(sys.argv[2]) if len(sys.argv) > 2 else 1
toks, scores = hc.load_vocab(name)
scores = np.asarray(scores, np.float64)
flat, offs = synth.make_corpus(int(mib * (1 << 20)), "mixed", seed_offset=1000)
ft = _lib.FlatTrie(toks, scores)
check, basef, tokid = ft.table()
base = basef & 0x7FFFFFFF; term = (basef >> 31).astype(bool)
N = flat.size
text = np.concatenate([flat, np.zeros(32, np.uint8)]).astype(np.uint32)
o = offs.astype(np.int64); ends = np.repeat(o[1:], np.diff(o)); pos = np.arange(N, dtype=np.int64)
cur = np.zeros(N, np.uint32); b = np.full(N, base[0], np.uint32); alive = np.ones(N, bool)
# matches[d][p] = score or nan
M = np.full((16, N), np.nan)
for d in range(16):
    alive &= (pos + d) < ends
    idx = np.nonzero(alive)[0]
    if idx.size == 0: break
    t = b[idx] ^ text[idx + d]
    ok = check[t] == cur[idx]
    tm = ok & term[t]
    M[d, idx[tm]] = scores[tokid[t[tm]]]
    alive[idx[~ok]] = False
    good = idx[ok]; cur[good] = t[ok]; b[good] = base[t[ok]]
exact = 0; near = 0; decisions = 0
Ml = [M[d].tolist() for d in range(16)]
ninf = float("-inf")
for s in range(offs.size - 1):
    a, e = int(o[s]), int(o[s + 1])
    n = e - a
    best = [ninf] * (n + 1); best[0] = 0.0
    for x in range(1, n + 1):
        top = ninf; second = ninf
        for d in range(min(16, x)):
            sc = Ml[d][a + x - d - 1]
            if sc == sc:
                c = best[x - d - 1] + sc
                if c > top: second = top; top = c
                elif c > second: second = c
        best[x] = top
        if second > ninf:
            decisions += 1
            m = top - second
            if m == 0.0: exact += 1
            elif m < 1e-6: near += 1
print(name, "bytes", N, "decisions with >=2 candidates", decisions, "exact ties", exact, "near ties (<1e-6)", near)
