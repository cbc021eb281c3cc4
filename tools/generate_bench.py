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
flat, offs = synth.make_corpus(4 << 20, "mixed", max_len=2000, seed_offset=3001)
o = offs.astype(np.int64)
samples = [flat[o[i]:o[i + 1]].tobytes().decode("utf-8") for i in range(o.size - 1)]
for dev in (0, None):
    g = VocabularyGenerator(16, 1.0, None, None, [], [], device=dev)
    sub = samples if dev is not None else samples[: len(samples) // 16]
    t0 = time.perf_counter(); g.feed(sub); dt = time.perf_counter() - t0
    nb = sum(len(s.encode()) for s in sub)
    print(f"feed() {'device' if dev is not None else 'host restatement (Python)'}: {nb / 1e6:.1f} MB in {dt:.2f} s = {nb / dt / 1e6:.2f} MB/s, {len(g.frequencies)} substrings", flush=True)
