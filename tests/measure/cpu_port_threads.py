"""the CPU port (oracle) on config 0 — 10 MB synthetic ASCII code, 32 000-entry vocabulary — with 1
thread and with more host cores, and the GPU path on the same batch (ids compared).  Test infrastructure: uses oracle/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth
vflat, _ = synth.make_corpus(4 << 20, "ascii", seed_offset=0)
toks, scores = synth.build_vocab(vflat[: 2 << 20], 32000, 16)
flat, offs = synth.make_corpus(10 << 20, "ascii", seed_offset=1000)
ora = orc.OracleModel(toks, scores)
out = {"bytes": int(flat.size), "samples": int(offs.size - 1), "host_cores": len(os.sched_getaffinity(0)), "cpu_port_MB_per_s": {}}
want = None
for th in (1, 8, 64, 256):
    if th > out["host_cores"]: continue
    best = 1e9
    for _ in range(2):
        t = time.perf_counter(); ids, oo = ora.encode_batch_flat(flat, offs, threads=th); best = min(best, time.perf_counter() - t)
    want = ids
    out["cpu_port_MB_per_s"][str(th)] = flat.size / best / 1e6
if tgx.device_count() > 0:
    nat = tgx.NativeModel(toks, scores)
    c = tgx.NativeCorpus(flat, offs)
    r = nat.encode_corpus(c); r.free()
    t = time.perf_counter(); r = nat.encode_corpus(c); dt = time.perf_counter() - t
    out["gpu_MB_per_s_resident"] = flat.size / dt / 1e6
    out["gpu_ids_equal"] = bool(np.array_equal(r.ids(), want))
    r.free()
print(json.dumps(out))
