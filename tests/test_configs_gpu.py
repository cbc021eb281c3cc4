"""BASELINE.json's configurations that the other GPU tests do not run at their own parameters:

  configs[0]  encode_batch on 10 MB of synthetic ASCII with the 32 000-entry vocabulary — the whole batch against
              the CPU oracle (reference src/tokenizer.rs:102-123 over src/model.rs:59-129);
  configs[3]  the prune passes at the 500 000-entry vocabulary (E-step src/prune.rs:64-120, frequency pass
              src/prune.rs:205-244) — a bounded prefix of the corpus against the oracle;
  the driver's multi-GPU command: `python bench.py --gpus N` starts its own ranks (two of them here, both on
  the one GPU of the test box).
"""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import tokengeex_amd as tgx
from oracle import oracle as orc
from tokengeex_amd import synth

from util import assert_same_encoding, load_vocab_500k

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config0_10mb_ascii_32k_vocab():
    toks, scores, _ = synth.load_spec_vocab(32000)   # the committed 64 MiB-slice vocabulary
    assert len(toks) == 32000
    flat, offs = synth.make_corpus(10_000_000, "ascii", seed_offset=1000)
    assert abs(int(flat.size) - 10_000_000) < 70_000 and int(flat.max()) < 128
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    ids, oo = assert_same_encoding(nat, ora, flat, offs)
    assert "encode5_kernel" in nat.last_kernel_times()
    # and sample by sample through the reference's single-threaded path (configs[0]: "single thread")
    want_ids, want_oo = ora.encode_batch_flat(flat, offs, threads=1)
    np.testing.assert_array_equal(ids, want_ids)
    np.testing.assert_array_equal(oo, want_oo)
    freq = nat.count_tokens(tgx.NativeCorpus(flat, offs))
    np.testing.assert_array_equal(freq, np.bincount(want_ids, minlength=len(toks)).astype(np.uint64))


def test_config3_prune_passes_at_500k_vocab():
    toks, scores = load_vocab_500k()
    flat, offs = synth.make_corpus(24 << 20, "mixed", seed_offset=1000)
    k = int(np.searchsorted(offs, 6 << 20))          # ~6 MiB: seconds of oracle time at this vocabulary
    sf, so = flat[: int(offs[k])], offs[: k + 1]
    nat, ora = tgx.NativeModel(toks, scores, for_estep=True), orc.OracleModel(toks, scores)
    assert nat.vocab_size == 500000
    # encode + frequency pass: bit-exact
    ids, _ = assert_same_encoding(nat, ora, sf, so)
    corpus = tgx.NativeCorpus(sf, so)
    freq = nat.count_tokens(corpus)
    np.testing.assert_array_equal(freq, ora.count_tokens_flat(sf, so, threads=8))
    np.testing.assert_array_equal(freq, np.bincount(ids, minlength=len(toks)).astype(np.uint64))
    # E-step: expected counts to the documented tolerance (tests/test_estep_pairs_gpu.py), same support, same z
    got, gz = nat.estep(corpus)
    st, want, wz, _ = ora.estep_flat(sf, so, threads=8)
    assert st == orc.OK
    longest = int(np.diff(so.astype(np.int64)).max())
    np.testing.assert_allclose(got, want, rtol=1.2e-8 * max(1.0, longest / 4096.0), atol=1e-12)
    assert np.array_equal(got != 0, want != 0)
    assert abs(gz - wz) <= 1e-12 * abs(wz) + 1e-9
    mass = float(np.dot(got, np.array([len(t) for t in toks], np.float64)))
    assert abs(mass - sf.size) < 1e-6 * sf.size       # every byte covered with total mass 1
    assert "estep7_kernel" in nat.last_kernel_times()   # round 4: 32-bit match entries for more than 65 535 tokens


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent launches two ranks before it touches the
    GPU and rank 0 prints the one line with n_gpus = 2 (whole-job throughput, a CPU-port leg on rank 0)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--size-mb", "64",
                        "--steps", "3", "--warmup", "1", "--cpu-seconds", "2", "--master-port", "29547"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["bytes_per_gpu"] >= 60 << 20 and out["value"] > 0
    assert out["roofline"]["achieved"] > 0 and out["cpu_baseline"]["cores"] >= 1 and out["cpu_baseline"]["value_1thread"] > 0


def test_two_models_share_one_resident_corpus_from_two_threads():
    """prune and merge keep the corpus in HBM while models come and go (src/prune.rs:48); two models encoding
    and scanning the SAME corpus handle from two host threads must not see each other's scratch."""
    flat, offs = synth.make_corpus(6 << 20, "mixed", seed_offset=21, max_len=20000)
    toks, scores = synth.build_vocab(flat[: 2 << 20], 6000, 16)
    toks2 = toks[:2500] + [bytes([b]) for b in range(255) if bytes([b]) not in set(toks[:2500])]
    scores2 = np.concatenate([np.asarray(scores[:2500]) * 1.03, np.full(len(toks2) - 2500, -11.0)])
    models = [(tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)),
              (tgx.NativeModel(toks2, scores2), orc.OracleModel(toks2, scores2))]
    want = [(o.encode_batch_flat(flat, offs, threads=8), o.count_tokens_flat(flat, offs, threads=8)) for _, o in models]
    corpus = tgx.NativeCorpus(flat, offs)
    got, errors = [None, None], []

    def work(i):
        try:
            out = []
            for _ in range(6):
                res = models[i][0].encode_corpus(corpus)
                out.append((res.ids().copy(), res.offsets().copy(), models[i][0].count_tokens(corpus)))
                res.free()
            got[i] = out
        except Exception as e:  # surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for ids, oo, freq in got[i]:
            np.testing.assert_array_equal(ids, want[i][0][0])
            np.testing.assert_array_equal(oo, want[i][0][1])
            np.testing.assert_array_equal(freq, want[i][1])


def test_default_decisions_by_batch_size_on_the_spec_vocabulary():
    """The launch decisions of run_encode_kernel (tgx_api.cpp) at the sizes they were measured on, with the committed
    32 000-entry spec vocabulary and no switches: 64 MiB -> the long-sample kernel takes the long samples; 256 MiB -> both
    encode kernels at once on CUs of their own; 512 MiB -> encode5_kernel alone on rows = bytes / longest sample.  Ids
    bit-exact against the oracle at every size (the oracle on all host threads)."""
    import os
    from util import assert_same_encoding
    toks, scores, _ = synth.load_spec_vocab(32000)
    nat, ora = tgx.NativeModel(toks, scores), orc.OracleModel(toks, scores)
    threads = max(8, min(64, os.cpu_count() or 8))
    for mib, want in ((64, "long"), (256, "corun"), (512, "rows")):
        flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
        assert_same_encoding(nat, ora, flat, offs, threads=threads)
        kt = nat.last_kernel_times()
        if want == "long":
            assert "encode6_kernel" in kt and nat.last_encode_long_samples() > 0 and nat.last_encode_corun_cus() == 0
        elif want == "corun":
            assert "encode6_kernel" in kt and "encode5_kernel" in kt and nat.last_encode_corun_cus() > 0
            assert nat.encode_corun_timeouts() == 0  # the host saw encode5_kernel's blocks report themselves within the limit
            assert 0 < nat.last_encode_long_samples() < offs.size - 1
        else:
            assert "encode6_kernel" not in kt and nat.last_encode_corun_cus() == 0
            assert nat.last_encode_waves_per_cu() <= 10 and nat.last_encode_hot_values() == nat.score_values()


def test_the_cost_model_picks_a_near_best_launch(monkeypatch):
    """run_encode_kernel (tgx_api.cpp) chooses between the long-sample kernel, both encode kernels at once and
    encode5_kernel alone from constants measured once (bytes per second of each kernel, seconds per byte of a chain).
    This is the check that they still hold on the box at hand: at every size the default must be within 15 % of the best
    of the forced alternatives (no long-sample kernel; the long-sample kernel without co-run), best of five passes each —
    a drifted constant shows up as a default that loses to an alternative."""
    import time
    toks, scores, _ = synth.load_spec_vocab(32000)
    nat = tgx.NativeModel(toks, scores)

    def best_ms(corpus, reps=5):
        nat.encode_corpus(corpus).free()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            nat.encode_corpus(corpus).free()
            ts.append(time.perf_counter() - t)
        return min(ts) * 1e3

    for mib in (64, 256, 512):
        flat, offs = synth.make_corpus(mib << 20, "mixed", seed_offset=1000)
        corpus = tgx.NativeCorpus(flat, offs)
        t_default = best_ms(corpus)
        alts = {}
        monkeypatch.setenv("TGX_LONG_THRESHOLD", str(1 << 30))  # no sample is "long": encode5_kernel alone
        alts["encode5 alone"] = best_ms(corpus)
        monkeypatch.delenv("TGX_LONG_THRESHOLD")
        monkeypatch.setenv("TGX_CORUN", "0")                    # the long-sample kernel first, then encode5_kernel
        alts["no co-run"] = best_ms(corpus)
        monkeypatch.delenv("TGX_CORUN")
        corpus.free()
        best = min(alts.values())
        assert t_default <= 1.15 * best, (mib, t_default, alts)
