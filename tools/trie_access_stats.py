"""CPU analysis: how the trie walk's gathers distribute over slots and depths
(what an LDS-resident or better-packed hot set could absorb)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tokengeex_amd import synth, _lib
vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
flat, offs = synth.make_corpus(16 << 20, "mixed", seed_offset=1000)
ft = _lib.FlatTrie(toks, scores)
check, basef, tokid = ft.table()
n_slots = check.size
base = basef & 0x7FFFFFFF
term = (basef >> 31).astype(bool)
print("slots", n_slots, "nodes", ft.stats()["n_nodes"], "table bytes", n_slots * 16)
text = np.concatenate([flat, np.zeros(32, np.uint8)]).astype(np.uint32)
N = flat.size
# sample ends: limit walks at sample boundaries
ends = np.zeros(N, dtype=np.int64)
o = offs.astype(np.int64)
ends = np.repeat(o[1:], np.diff(o))
pos = np.arange(N, dtype=np.int64)
cur = np.zeros(N, np.uint32); b = np.full(N, base[0], np.uint32); alive = np.ones(N, bool)
hits = np.zeros(n_slots, np.int64)
per_depth = []; matches = 0; fails = 0
for d in range(16):
    alive &= (pos + d) < ends
    idx = np.nonzero(alive)[0]
    if idx.size == 0: break
    t = b[idx] ^ text[idx + d]
    np.add.at(hits, t, 1) if idx.size < 1 else None
    hits += np.bincount(t, minlength=n_slots)
    ok = check[t] == cur[idx]
    fails += int((~ok).sum())
    per_depth.append((d, idx.size, int(ok.sum()), int((ok & term[t]).sum())))
    matches += int((ok & term[t]).sum())
    alive[idx[~ok]] = False
    good = idx[ok]
    cur[good] = t[ok]; b[good] = base[t[ok]]
tot = hits.sum()
print("gathers/pos %.3f  matches/pos %.3f  failing probes/pos %.3f" % (tot / N, matches / N, fails / N))
for d, n, ok, tm in per_depth:
    print(f"  depth {d:2d}: gathers/pos {n/N:.3f}  hit {ok/N:.3f}  terminal {tm/N:.3f}")
cum = np.cumsum(hits) / tot
for K in (256, 1024, 2048, 3840, 4096, 8192, 16384, 32768):
    if K <= n_slots: print(f"  first {K:6d} slots (BFS order): {cum[K-1]*100:5.1f}% of gathers")
srt = np.sort(hits)[::-1]; cs = np.cumsum(srt) / tot
for K in (256, 1024, 2048, 3840, 4096, 8192, 16384, 32768):
    if K <= n_slots: print(f"  hottest {K:6d} slots (ideal):   {cs[K-1]*100:5.1f}% of gathers")
lines = np.add.reduceat(hits, np.arange(0, n_slots, 8)); ls = np.sort(lines)[::-1]; cl = np.cumsum(ls) / tot
for K in (256, 512, 1024, 2048): print(f"  hottest {K:5d} 128-B lines: {cl[K-1]*100:5.1f}% of gathers (L1 = 256 lines)")
# chain statistics: nodes with exactly one child and subtree being a single chain

# ---- tail-compression potential: nodes whose whole subtree is a single chain
parent = check.copy()
used = (check != 0xFFFFFFFF)
used[0] = True
nchild = np.bincount(parent[used & (np.arange(n_slots) != 0)], minlength=n_slots)
# chain[node] = True if subtree is a path: nchild <= 1 and child (if any) is chain.  process by depth (leaves first)
depth = np.zeros(n_slots, np.int32)
order = [np.array([0])]
# BFS levels via parent pointers
lvl = np.full(n_slots, -1, np.int32); lvl[0] = 0
cand = np.nonzero(used)[0]
changed = True
while changed:
    changed = False
    unk = cand[lvl[cand] < 0]
    if unk.size == 0: break
    ok = lvl[parent[unk]] >= 0
    if ok.any():
        lvl[unk[ok]] = lvl[parent[unk[ok]]] + 1; changed = True
maxl = lvl.max()
chain = np.zeros(n_slots, bool)
only_child = np.zeros(n_slots, np.int64)  # child slot if exactly one child
ch_idx = cand[cand != 0]
only_child[parent[ch_idx]] = ch_idx  # last writer wins; fine when nchild == 1
for L in range(maxl, -1, -1):
    nodes = cand[lvl[cand] == L]
    leaf = nchild[nodes] == 0
    one = nchild[nodes] == 1
    chain[nodes[leaf]] = True
    chain[nodes[one]] = chain[only_child[nodes[one]]]
print("nodes", used.sum(), "chain-subtree nodes", chain[used].sum(), "(%.1f%%)" % (100 * chain[used].sum() / used.sum()))
# walk simulation with tails: count branch gathers until entering a chain node (then one tail access)
cur = np.zeros(N, np.uint32); b = np.full(N, base[0], np.uint32); alive = np.ones(N, bool)
branch_g = 0; tail_acc = 0; steps_hist = np.zeros(20, np.int64); nsteps = np.zeros(N, np.int32)
for d in range(16):
    alive &= (pos + d) < ends
    idx = np.nonzero(alive)[0]
    if idx.size == 0: break
    t = b[idx] ^ text[idx + d]
    branch_g += idx.size
    nsteps[idx] += 1
    ok = check[t] == cur[idx]
    alive[idx[~ok]] = False
    good = idx[ok]; tg = t[ok]
    cur[good] = tg; b[good] = base[tg]
    into_tail = chain[tg] & (nchild[tg] > 0)   # entered a chain head with something below: finish via tail compare
    tail_acc += int(into_tail.sum())
    nsteps[good[into_tail]] += 1
    alive[good[into_tail]] = False
    alive[good[nchild[tg] == 0]] = False       # leaf: nothing more to probe
print("with tails: branch gathers/pos %.3f + tail accesses/pos %.3f (each 2 x 16 B, one line)" % (branch_g / N, tail_acc / N))
h = np.bincount(nsteps, minlength=20)
print("dependent steps per walk (incl. tail step): " + " ".join(f"{i}:{h[i]/N:.3f}" for i in range(1, 12)))
blk = nsteps[: (N // 64) * 64].reshape(-1, 64).max(axis=1)
print("max steps over 64 consecutive positions: mean %.2f, hist %s" % (blk.mean(), np.bincount(blk, minlength=18)[:18] / blk.size))
