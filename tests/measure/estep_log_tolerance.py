"""Deviation of the E-step kernels from the CPU oracle by snippet length, log-domain (TGX_ESTEP=log: the literal
restatement of src/lattice.rs:245-333) and linear-domain (default) — the figures behind the tolerances in
BASELINE.md section 3.  Uses oracle/ as the checker: lives under tests/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth
flat, offs = synth.make_corpus(4 << 20, "mixed", seed_offset=5)
toks, scores = synth.build_vocab(flat[: 2 << 20], 8000, 16)
ora = orc.OracleModel(toks, scores)
for mode in ("log", "linear"):
    if mode == "log": os.environ["TGX_ESTEP"] = "log"
    else: os.environ.pop("TGX_ESTEP", None)
    nat = tgx.NativeModel(toks, scores)
    for snip in (1024, 4096, 16384, 65536, 81920):
        got, gz = nat.estep(tgx.NativeCorpus(flat, offs), snip)
        st, want, wz, _ = ora.estep_flat(flat, offs, snip, threads=8)
        big = np.abs(want) > 1e-9
        rel = float((np.abs(got - want)[big] / np.abs(want)[big]).max())
        print(f"{mode:6s} snippet {snip:6d}: max rel diff vs oracle {rel:.3e}  |dz|/|z| {abs(gz - wz) / abs(wz):.2e}  kernels {sorted(nat.last_kernel_times())}", flush=True)
