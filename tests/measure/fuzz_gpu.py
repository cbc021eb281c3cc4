"""Randomised differential run of the GPU paths against the CPU oracle (not part of the pytest suite: run it
on the GPU box, `python tests/measure/fuzz_gpu.py [cases] [seed]`).  Every case draws a vocabulary shape (max
token length 1..40, with or without full byte cover, duplicate and tied scores), a batch shape (empty and
1-byte samples, lengths around the 16/32/64 block boundaries, a few long ones), dropout, and the kernel
variant knobs (positions per lane); encode ids must be bit-identical, E-step counts within the tolerance."""
import os, sys, time
os.environ["TGX_KNOBS"] = "1"  # the library honours its kernel switches only in processes that opt in
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base_flat, _ = synth.make_corpus(2 << 20, "mixed", seed_offset=77)
base = bytes(base_flat)
t0 = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    max_len = int(rng.choice([2, 3, 5, 8, 12, 15, 16, 17, 20, 24, 31, 32, 33, 40]))
    all_bytes = bool(rng.random() < 0.8)
    toks, scores = synth.random_vocab(rng, base[: 64 << 10], n_multi=int(rng.integers(50, 3000)), max_len=max_len,
                                      all_bytes=all_bytes, tie_fraction=float(rng.choice([0.0, 0.2, 0.6])))
    if rng.random() < 0.3:  # duplicates: the later id must win
        k = int(rng.integers(1, 20))
        idx = rng.integers(0, len(toks), k)
        toks = toks + [toks[i] for i in idx]
        scores = np.concatenate([scores, -rng.random(k) * 5])
    lens = []
    for _ in range(int(rng.integers(1, 400))):
        r = rng.random()
        if r < 0.1: lens.append(int(rng.choice([0, 1, 2])))
        elif r < 0.4: lens.append(int(rng.choice([15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129])))
        elif r < 0.97: lens.append(int(rng.integers(3, 3000)))
        else: lens.append(int(rng.integers(20000, 90000)))
    texts = []
    for n in lens:
        if rng.random() < 0.05:
            texts.append(bytes(rng.integers(0, 256, n).astype(np.uint8)))
        else:
            o = int(rng.integers(0, len(base) - n - 1))
            texts.append(base[o:o + n])
    flat, offs = tgx.pack(texts)
    dropout = float(rng.choice([0.0, 0.0, 0.1, 0.5, 1.0]))
    sd = int(rng.integers(0, 1 << 62))
    for k in ("TGX_PPL", "TGX_EPPL", "TGX_PATH", "TGX_LONG_THRESHOLD", "TGX_E5_HOT", "TGX_E6_POOL", "TGX_E2E_CHUNK_MB", "TGX_ESTEP_PIECES", "TGX_ESTEP_WINDOW", "TGX_CORUN",
              "TGX_TRACE", "TGX_TRACE_CARRY", "TGX_E7_HOT", "TGX_E7_WAVES", "TGX_E7_RANK", "TGX_E7_OVF_AT", "TGX_ESTEP", "TGX_VALUE_RANK"):
        os.environ.pop(k, None)
    # round 4: the trace's two modes and the mask pipeline; estep7_kernel's table size, waves, rank order, overflow build;
    # the chained kernels now and then
    if rng.random() < 0.5: os.environ["TGX_TRACE_CARRY"] = str(int(rng.choice([0, 1])))
    if rng.random() < 0.2: os.environ["TGX_TRACE"] = "mask"
    if rng.random() < 0.4: os.environ["TGX_E7_HOT"] = str(int(rng.choice([0, 5, 60, 700])))
    if rng.random() < 0.3: os.environ["TGX_E7_WAVES"] = str(int(rng.choice([1, 3, 8])))
    if rng.random() < 0.3: os.environ["TGX_E7_RANK"] = "model"
    if rng.random() < 0.3: os.environ["TGX_E7_OVF_AT"] = str(int(rng.choice([3, 50, 400, 2000])))
    if rng.random() < 0.15: os.environ["TGX_ESTEP"] = "chain"
    if rng.random() < 0.5: os.environ["TGX_VALUE_RANK"] = str(rng.choice(["counts", "model"]))  # encode5's values re-ranked by match counts
    if rng.random() < 0.6: os.environ["TGX_PPL"] = str(int(rng.choice([1, 2, 4])))
    if rng.random() < 0.6: os.environ["TGX_EPPL"] = str(int(rng.choice([1, 2, 4])))
    # round 2: kernel choice (encode5 / encode4), long-sample kernel threshold, score table size (cold values
    # through the pools), pool size (overflow -> redo pass)
    if rng.random() < 0.5: os.environ["TGX_PATH"] = str(rng.choice(["rows4", "rows5"]))
    if rng.random() < 0.5: os.environ["TGX_LONG_THRESHOLD"] = str(int(rng.choice([0, 1, 100, 1000, 30000])))
    # round 3: both encode kernels at once (needs a threshold that leaves samples on both sides)
    if os.environ.get("TGX_LONG_THRESHOLD") in ("100", "1000") and rng.random() < 0.6: os.environ["TGX_CORUN"] = str(int(rng.choice([16, 96, 200])))
    if rng.random() < 0.4: os.environ["TGX_E5_HOT"] = str(int(rng.choice([0, 3, 40, 500])))
    if rng.random() < 0.3: os.environ["TGX_E6_POOL"] = str(int(rng.choice([0, 4, 16, 128])))
    # round 3: the E-step on pieces (snippets cut where no match crosses), small windows
    if rng.random() < 0.5:
        os.environ["TGX_ESTEP_PIECES"] = "1"
        os.environ["TGX_ESTEP_WINDOW"] = str(int(rng.choice([256, 512, 2048])))
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    tag = f"case {case} max_len={max_len} all_bytes={all_bytes} V={len(toks)} S={len(texts)} N={flat.size} dropout={dropout} env={os.environ.get('TGX_PPL')}/{os.environ.get('TGX_EPPL')}/{os.environ.get('TGX_PATH')}/{os.environ.get('TGX_LONG_THRESHOLD')}/{os.environ.get('TGX_E5_HOT')}/{os.environ.get('TGX_E6_POOL')}/{os.environ.get('TGX_ESTEP_PIECES')}/{os.environ.get('TGX_ESTEP_WINDOW')}/{os.environ.get('TGX_CORUN')} r4={os.environ.get('TGX_TRACE_CARRY')}/{os.environ.get('TGX_TRACE')}/{os.environ.get('TGX_E7_HOT')}/{os.environ.get('TGX_E7_WAVES')}/{os.environ.get('TGX_E7_RANK')}/{os.environ.get('TGX_E7_OVF_AT')}/{os.environ.get('TGX_ESTEP')}"
    try:
        want_ids, want_offs = ora.encode_batch_flat(flat, offs, dropout, sd, threads=8)
        want_err = None
    except orc.NoPath as e:
        want_err = e
    try:
        res = nat.encode_batch_flat(flat, offs, dropout, sd)
        got_ids, got_offs = res.ids(), res.offsets(); res.free()
        got_err = None
    except tgx.TokenGeeXError as e:
        got_err = e
    if (want_err is None) != (got_err is None):
        print("MISMATCH (error)", tag, want_err, got_err); sys.exit(1)
    if want_err is None and not (np.array_equal(got_ids, want_ids) and np.array_equal(got_offs, want_offs)):
        print("MISMATCH (ids)", tag, nat.last_kernel_times()); sys.exit(1)
    if want_err is None and dropout == 0.0 and case % 4 == 0:  # the host-to-host entry point, in small chunks
        os.environ["TGX_E2E_CHUNK_MB"] = "1"
        hi, ho = nat.encode_batch_host(flat, offs)
        if not (np.array_equal(hi, want_ids) and np.array_equal(ho, want_offs)):
            print("MISMATCH (host-to-host)", tag); sys.exit(1)
    # E-step on the same batch (every byte must be coverable for z to be normal: skip otherwise)
    if all_bytes and flat.size:
        snip = int(rng.choice([48, 1000, 4096, 81920]))
        corpus = tgx.NativeCorpus(flat, offs)
        got, gz = nat.estep(corpus, snip, dropout if dropout < 1.0 else 0.3, sd)
        st, want, wz, _ = ora.estep_flat(flat, offs, snip, dropout if dropout < 1.0 else 0.3, sd, threads=8)
        longest = min(snip, max(lens))
        rtol = 1.2e-8 * max(1.0, longest / 4096.0)  # the oracle's own rounding: tests/test_estep_pairs_gpu.py
        ok = st == orc.OK and np.allclose(got, want, rtol=rtol, atol=1e-12) and np.array_equal(got != 0, want != 0) and abs(gz - wz) <= 1e-12 * abs(wz) + 1e-9
        if not ok:
            bad = np.nonzero(~np.isclose(got, want, rtol=rtol, atol=1e-12))[0][:5]
            print("MISMATCH (estep)", tag, "snip", snip, nat.last_kernel_times(), bad, got[bad], want[bad], gz, wz); sys.exit(1)
        corpus.free()
    if case % 10 == 9:
        print(f"{case + 1} cases ok, {time.time() - t0:.0f} s", flush=True)
print("all", cases, "cases ok")
