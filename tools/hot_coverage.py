"""CPU analysis for encode5_kernel's LDS score table: which share of the trie matches of a corpus have their score
VALUE among the K values of the table, (a) with the builder's ranking (weight = sum over the value's tokens of
exp(score) / len, trie_build.cpp: build_trie8) and (b) with the values ranked by their measured match counts.
usage: python tools/hot_coverage.py <vocab: 2mib | spec32k | spec64k | distinct | 500k> [corpus MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tokengeex_amd import synth, _lib


def load_vocab(name):
    if name in ("2mib", "distinct"):
        vflat, _ = synth.make_corpus(4 << 20, "mixed", seed_offset=0)
        toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
        scores = np.asarray(scores, np.float64)
        if name == "distinct":
            scores = scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks))
        return toks, scores
    if name == "500k":
        z = np.load(os.path.join(ROOT, "tests", "golden", "vocab_500000.npz"))
        fb = z["flat"].tobytes()
        o = np.concatenate([[0], np.cumsum(z["lens"].astype(np.int64))])
        return [fb[o[i]:o[i + 1]] for i in range(o.size - 1)], z["uscores"][z["inv"]].astype(np.float64)
    toks, scores, _ = synth.load_spec_vocab({"spec32k": 32000, "spec64k": 65536}[name])
    return toks, scores


def match_counts(toks, scores, flat, offs):
    ft = _lib.FlatTrie(toks, scores)
    check, basef, tokid = ft.table()
    base = basef & 0x7FFFFFFF
    term = (basef >> 31).astype(bool)
    N = flat.size
    text = np.concatenate([flat, np.zeros(32, np.uint8)]).astype(np.uint32)
    o = offs.astype(np.int64)
    ends = np.repeat(o[1:], np.diff(o))
    pos = np.arange(N, dtype=np.int64)
    cur = np.zeros(N, np.uint32)
    b = np.full(N, base[0], np.uint32)
    alive = np.ones(N, bool)
    cnt = np.zeros(len(toks), np.int64)
    gathers = 0
    leaf_end = 0
    for d in range(64):
        alive &= (pos + d) < ends
        idx = np.nonzero(alive)[0]
        if idx.size == 0:
            break
        t = b[idx] ^ text[idx + d]
        gathers += idx.size
        ok = check[t] == cur[idx]
        tm = ok & term[t]
        cnt += np.bincount(tokid[t[tm]], minlength=len(toks))
        alive[idx[~ok]] = False
        good = idx[ok]
        cur[good] = t[ok]
        b[good] = base[t[ok]]
    return cnt, gathers / N


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "distinct"
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    toks, scores = load_vocab(name)
    flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
    cnt, gpp = match_counts(toks, scores, flat, offs)
    total = cnt.sum()
    vals, inv = np.unique(scores, return_inverse=True)
    print(f"{name}: {len(toks)} tokens, {vals.size} distinct score values, {total / flat.size:.3f} matches/position, {gpp:.3f} gathers/position")
    lens = np.array([max(1, len(t)) for t in toks], np.float64)
    w_build = np.zeros(vals.size)
    np.add.at(w_build, inv, np.exp(scores) / lens)
    c_val = np.zeros(vals.size, np.int64)
    np.add.at(c_val, inv, cnt)
    for label, order in (("builder ranking", np.argsort(-w_build, kind="stable")), ("measured ranking", np.argsort(-c_val, kind="stable"))):
        cs = np.cumsum(c_val[order]) / total
        print(f"  {label}: " + "  ".join(f"K={K}: {100 * (1 - cs[min(K, vals.size) - 1]):.2f}% cold" for K in (2047, 4095, 6143, 8191, 10239, 12287, 14335, 16383)))


if __name__ == "__main__":
    main()
