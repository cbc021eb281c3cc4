"""tgx_substring_df (the device part of VocabularyGenerator::feed): windows per second on a mixed corpus, whole
samples as parts, max token length 16, and the whole feed() with its host side (allow regex on the distinct
substrings) on a smaller slice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tokengeex_amd import _lib, synth
from tokengeex_amd.generate import VocabularyGenerator
for mb in (16, 64):
    flat, offs = synth.make_corpus(mb << 20, "mixed", max_len=4096, seed_offset=3000)
    pb, pe, ps = offs[:-1], offs[1:], np.arange(offs.size - 1, dtype=np.uint32)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        pos, ln, df, nw = _lib.substring_df(flat, pb, pe, ps, 16, 1.0, 0, 0)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{mb} MiB, {offs.size - 1} samples: {nw} windows -> {pos.size} distinct substrings in {best * 1e3:.1f} ms "
          f"({nw / best / 1e9:.2f} G windows/s, {flat.size / best / 1e6:.0f} MB/s of text)", flush=True)
# feed() + generate() end to end with top_k (only the most frequent substrings leave the device), 64 MiB of text
flat, offs = synth.make_corpus(64 << 20, "mixed", max_len=4096, seed_offset=3000)
o = offs.astype(np.int64)
raw = flat.tobytes()
samples = [raw[o[i]:o[i + 1]].decode("utf-8") for i in range(o.size - 1)]
for top_k in (1 << 19, 1 << 21):
    g = VocabularyGenerator(16, 1.0, None, open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "exact.regex")).read().strip()
                            if os.path.exists(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "exact.regex")) else None, [], [], top_k=top_k)
    t0 = time.perf_counter(); g.feed(samples); n = len(g.frequencies); dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    try:
        v = g.generate(32000); ok = f"{len(v)} tokens"
    except Exception as e:
        ok = f"refused: {e}"
    dt2 = time.perf_counter() - t1
    print(f"feed() with top_k={top_k}: {flat.size / 1e6:.1f} MB in {dt:.2f} s = {flat.size / dt / 1e6:.1f} MB/s ({n} substrings on the host), generate(32000) {dt2:.2f} s: {ok}", flush=True)
