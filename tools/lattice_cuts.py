"""How often the lattice of the bench corpus has a position no token match crosses (the cut points of csrc/cuts.hip):
segment lengths between consecutive cuts, on the CPU.  usage: python tools/lattice_cuts.py <spec32k | 500k | 2mib | distinct>"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np
from tokengeex_amd import synth, _lib
import hot_coverage as hc
name = sys.argv[1]
toks, scores = hc.load_vocab(name)
flat, offs = synth.make_corpus(4 << 20, "mixed", seed_offset=1000)
ft = _lib.FlatTrie(toks, scores)
check, basef, tokid = ft.table()
base = basef & 0x7FFFFFFF; term = (basef >> 31).astype(bool)
N = flat.size
text = np.concatenate([flat, np.zeros(32, np.uint8)]).astype(np.uint32)
o = offs.astype(np.int64); ends = np.repeat(o[1:], np.diff(o)); pos = np.arange(N, dtype=np.int64)
cur = np.zeros(N, np.uint32); b = np.full(N, base[0], np.uint32); alive = np.ones(N, bool)
reach = pos.copy()
for d in range(32):
    alive &= (pos + d) < ends
    idx = np.nonzero(alive)[0]
    if idx.size == 0: break
    t = b[idx] ^ text[idx + d]
    ok = check[t] == cur[idx]
    tm = ok & term[t]
    reach[idx[tm]] = idx[tm] + d + 1
    alive[idx[~ok]] = False
    good = idx[ok]; cur[good] = t[ok]; b[good] = base[t[ok]]
# cut at q (strictly inside a sample) iff max(reach[start..q-1]) <= q
gaps = []
for s in range(offs.size - 1):
    a, e = int(o[s]), int(o[s + 1])
    if e - a < 2: continue
    m = np.maximum.accumulate(reach[a:e])
    cuts = np.nonzero(m[:-1] <= np.arange(a + 1, e))[0] + 1  # relative positions q
    pts = np.concatenate([[0], cuts, [e - a]])
    gaps.append(np.diff(pts))
g = np.concatenate(gaps)
print(name, "segments", g.size, "mean", g.mean(), "median", np.median(g), "p99", np.percentile(g, 99), "max", g.max())
# longest run if we cut greedily at first cut after 2 KiB
