"""Development check of estep7_kernel against the oracle on a few shapes (GPU box).  usage: python tests/measure/e7_check.py"""
import os, sys, time
os.environ.setdefault("TGX_KNOBS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth
from util import corpus_and_vocab


def check(name, nat, ora, flat, offs, snippet_len=81920, dropout=0.0, seed=0, env=None):
    env = env or {}
    for k, v in env.items():
        os.environ[k] = v
    try:
        corpus = tgx.NativeCorpus(flat, offs)
        t = time.perf_counter()
        got, gz = nat.estep(corpus, snippet_len, dropout, seed)
        dt = time.perf_counter() - t
        kt = nat.last_kernel_times()
    finally:
        for k in env:
            del os.environ[k]
    st, want, wz, _ = ora.estep_flat(flat, offs, snippet_len, dropout, seed, threads=8)
    assert st == orc.OK
    big = np.abs(want) > 1e-9
    rel = float((np.abs(got - want)[big] / np.abs(want)[big]).max()) if big.any() else 0.0
    absd = float(np.abs(got - want).max())
    same = bool(np.array_equal(got != 0, want != 0))
    zrel = abs(gz - wz) / max(abs(wz), 1e-300)
    print(f"{name}: rel {rel:.2e} abs {absd:.2e} same-support {same} zrel {zrel:.1e} pieces {nat.last_estep_pieces()} redo {nat.last_estep_redo()} kernels {kt} wall {dt*1e3:.1f} ms", flush=True)


def main():
    toks = [b"a", b"b", b"ab", b"ba", b"aba", b"c", b"bcb"]
    scores = [-1.0, -1.5, -1.7, -2.0, -2.2, -3.0, -0.5]
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    texts = [b"abaabab", b"a", b"abcba", b"cbcb", b"ab" * 100, b"aba" * 43 + b"c", b"bcbcb"]
    flat, offs = tgx.pack(texts)
    check("tiny", nat, ora, flat, offs)
    check("tiny snip3", nat, ora, flat, offs, snippet_len=3)
    check("tiny snip64", nat, ora, flat, offs, snippet_len=64)
    toks = [b"a", b"aa", b"aaaa", b"b", b"ab"]
    scores = [-1.0, -1.6, -2.5, -2.0, -2.2]
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    texts = [b"a" * 5000, b"a" * 300 + b"b" + b"a" * 2000, (b"a" * 999 + b"b") * 8, b"ab" * 700, b"a" * 16, b"a" * 17, b"a" * 15, b"a" * 64, b"a" * 65]
    flat, offs = tgx.pack(texts)
    check("no-cut runs", nat, ora, flat, offs)
    check("no-cut runs snip1000", nat, ora, flat, offs, snippet_len=1000)
    check("no-cut runs dropout", nat, ora, flat, offs, dropout=0.3, seed=3)
    flat, offs, toks, scores = corpus_and_vocab(2 << 20, "mixed", 8000, 16)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    check("2MiB 8000", nat, ora, flat, offs)
    for ppl in "1234":
        check(f"2MiB ppl{ppl}", nat, ora, flat, offs, env={"TGX_EPPL": ppl})
    check("2MiB hot300", nat, ora, flat, offs, env={"TGX_E7_HOT": "300"})
    check("2MiB hot0", nat, ora, flat, offs, env={"TGX_E7_HOT": "0", "TGX_EPPL": "2"})
    check("2MiB dropout", nat, ora, flat, offs, dropout=0.2, seed=5)
    check("2MiB pieces512", nat, ora, flat, offs, env={"TGX_ESTEP_PIECES": "1", "TGX_ESTEP_WINDOW": "512"})
    check("2MiB pieces dropout", nat, ora, flat, offs, dropout=0.3, seed=11, env={"TGX_ESTEP_PIECES": "1", "TGX_ESTEP_WINDOW": "512"})
    check("2MiB chain", nat, ora, flat, offs, env={"TGX_ESTEP": "chain"})
    rng = np.random.default_rng(31)
    sc2 = np.asarray(scores) + rng.uniform(-0.3, 0.3, len(toks))
    nat2, ora2 = tgx.NativeModel(toks, sc2), orc.OracleModel(toks, sc2)
    check("2MiB distinct", nat2, ora2, flat, offs)
    # more than 65 535 tokens: 32-bit entries
    from util import load_vocab_500k
    t5, s5 = load_vocab_500k()
    nat5, ora5 = tgx.NativeModel(t5, s5), orc.OracleModel(t5, s5)
    f5, o5 = synth.make_corpus(2 << 20, "mixed", seed_offset=77)
    check("2MiB 500k", nat5, ora5, f5, o5)
    check("2MiB 500k ppl4", nat5, ora5, f5, o5, env={"TGX_EPPL": "4", "TGX_E7_WAVES": "6"})


if __name__ == "__main__":
    main()
