#!/usr/bin/env python3
"""Headline benchmark: MB/s of raw bytes encoded by the Unigram encode hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of Tokenizer::encode_ordinary_batch (reference
src/tokenizer.rs:114-123 over src/model.rs:59-129) over one resident batch of
synthetic text: BASELINE.json configs[1] — 32K vocab, 1 GB mixed code+Chinese
corpus per GPU (weak scaling: every rank encodes its own shard, no data-path
collective).  Inputs are already in HBM when the timed region starts; ids stay
in HBM.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size-mb", type=int, default=1024, help="corpus bytes per GPU, in MiB")
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--max-token-length", type=int, default=16)
    ap.add_argument("--kind", default="mixed", choices=["mixed", "ascii"])
    ap.add_argument("--max-sample-len", type=int, default=65536, help="longest sample of the synthetic corpus, bytes")
    ap.add_argument("--vocab-slice-mb", type=int, default=64,
                    help="slice of the mixed corpus the vocabulary is built over: 64 = SURVEY.md section 8(d)'s (the committed "
                         "32 000- and 65 536-entry vocabularies); anything else is built here over at most 8 MiB")
    ap.add_argument("--distinct-scores", action="store_true",
                    help="give every token its own score (as after an M-step or merge: any trained vocabulary; the "
                         "generate-style default scores tokens by integer counts: 9 652 distinct values at 32 000 entries)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive tgx_encode_batch measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline leg")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0")
    ap.add_argument("--collective", default="gloo", choices=["gloo", "nccl"],
                    help="backend of the barrier and of the three timing scalars at N > 1: gloo on CPU tensors (the default — the "
                         "path shards with no data-path collective, north_star: no RCCL needed, and it is what the tests run) "
                         "or nccl (RCCL over xGMI)")
    ap.add_argument("--no-estep", action="store_true", help="skip the E-step sub-record (prune's pass over the same corpus)")
    ap.add_argument("--estep-steps", type=int, default=3)
    ap.add_argument("--no-distinct", action="store_true", help="skip the encode passes with every token its own score")
    ap.add_argument("--master-port", type=int, default=29533)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Launched bare (`python bench.py --gpus N`): start the N ranks ourselves, one fresh process per
        # GPU under torch.distributed.run, BEFORE this process imports torch or touches HIP (a process that
        # has initialised the GPU must never exec or fork workers), relay their output and exit with their code.
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")

    import numpy as np
    import torch

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()  # no-op when the in-tree .so files are current
    dist = None
    if world > 1 or os.environ.get("TGX_BENCH_FORCE_DIST") == "1":  # the env var rehearses the N > 1 path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.collective == "gloo" or args.single_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        dist.barrier()

    import tokengeex_amd as tgx
    from tokengeex_amd import synth

    dev = 0 if args.single_device else local_rank
    coll_dev = "cpu" if (args.collective == "gloo" or args.single_device) else f"cuda:{dev}"  # where the timing collectives' tensors live
    if tgx.device_count() <= dev:
        raise SystemExit("bench.py needs a GPU (no usable HIP device); there is no CPU fallback")

    # ---- workload: same vocabulary on every rank, a different corpus shard per rank
    t0 = time.time()
    # vocabulary: SURVEY.md section 8(d)'s — the generate stand-in over a fixed 64 MiB slice of the mixed corpus,
    # committed (tests/golden/vocab_32000.npz, vocab_65536.npz) — or, for other shapes, built here over a small slice
    if args.vocab_slice_mb == 64 and args.vocab in (32000, 65536) and args.max_token_length == 16:
        toks, scores, slice_mib = synth.load_spec_vocab(args.vocab)
    else:
        slice_mib = min(args.vocab_slice_mb, 8)
        vflat, _ = synth.make_corpus(max(4, slice_mib) << 20, "mixed", seed_offset=0)
        toks, scores = synth.build_vocab(vflat[: slice_mib << 20], args.vocab, args.max_token_length)
    scores = np.asarray(scores, np.float64)
    if args.distinct_scores:
        # as after an M-step (reference src/prune.rs:124-151: score = digamma(max(freq, 0.5)) - digamma(sum)): every
        # token that occurs gets a score of its own; the single-byte tokens that never occur in the corpus (control
        # characters, bytes no UTF-8 text contains) all sit on the floor value, as they do there
        scores = scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks))
        seen = np.zeros(256, bool)
        seen[np.unique(synth.make_corpus(8 << 20, args.kind, seed_offset=1000)[0])] = True
        floor = float(scores.min()) - 1.0
        for i, t in enumerate(toks):
            if len(t) == 1 and not seen[t[0]]:
                scores[i] = floor
    n_values = int(np.unique(scores).size)
    flat, offs = synth.make_corpus(args.size_mb << 20, args.kind, max_len=args.max_sample_len, seed_offset=1000 + rank)
    n_bytes, n_samples = int(flat.size), int(offs.size - 1)
    model = tgx.NativeModel(toks, scores, device=dev)
    corpus = tgx.NativeCorpus(flat, offs, device=dev)
    setup_s = time.time() - t0

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def step():
        res = model.encode_corpus(corpus)  # synchronous: returns when ids are complete in HBM
        n_tok = res.num_tokens
        res.free()
        return n_tok

    n_tokens = 0
    for _ in range(args.warmup):
        n_tokens = step()
    kernel_ms: dict[str, float] = {}
    sync_all()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        n_tokens = step()
        for k, v in model.last_kernel_times().items():  # hipEvents on the model's stream
            kernel_ms[k] = kernel_ms.get(k, 0.0) + v
    sync_all()
    elapsed = time.perf_counter() - t_start
    alg_bytes = model.last_algorithmic_bytes()
    corun_cus = model.last_encode_corun_cus()  # (of the timed passes: the host-to-host leg below encodes 256 MiB chunks)

    from tokengeex_amd import dist as tdist
    max_elapsed, tot_bytes, tot_tokens = tdist.aggregate_timing(elapsed, n_bytes, n_tokens, dist, coll_dev)

    # ---- the same encode with every token its own score (what an M-step or merge leaves: any trained vocabulary);
    # reported as `distinct_scores_mb_s`, never as `value`
    distinct = None
    if not args.no_distinct and not args.distinct_scores:
        sc2 = scores + np.random.default_rng(5).uniform(-0.4, 0.4, len(toks))
        m2 = tgx.NativeModel(toks, sc2, device=dev)
        for _ in range(max(1, args.warmup)):
            m2.encode_corpus(corpus).free()
        sync_all()
        t2 = time.perf_counter()
        k2: dict[str, float] = {}
        for _ in range(args.steps):
            m2.encode_corpus(corpus).free()
            for k, v in m2.last_kernel_times().items():
                k2[k] = k2.get(k, 0.0) + v
        sync_all()
        e2 = time.perf_counter() - t2
        mx2, tb2, _ = tdist.aggregate_timing(e2, n_bytes, 0, dist, coll_dev)
        distinct = {"distinct_scores_mb_s": round(tb2 * max(1, args.steps) / mx2 / 1e6, 2),
                    "distinct_scores_kernel_ms_per_step": {k: round(v / max(1, args.steps), 3) for k, v in k2.items()},
                    "distinct_scores_lds_values": m2.last_encode_hot_values()}
        m2.free()

    # ---- E-step sub-record (src/prune.rs:64-120 over the same resident corpus: BASELINE.json configs[3]'s pass)
    estep = None
    if not args.no_estep and args.max_token_length <= 16:
        me = tgx.NativeModel(toks, scores, device=dev, for_estep=True)
        expected, logz = me.estep(corpus)  # warm-up (builds the E-step tables)
        ek: dict[str, float] = {}
        sync_all()
        t3 = time.perf_counter()
        for _ in range(args.estep_steps):
            expected, logz = me.estep(corpus)
            for k, v in me.last_kernel_times().items():
                ek[k] = ek.get(k, 0.0) + v
        sync_all()
        e3 = time.perf_counter() - t3
        mx3, tb3, _ = tdist.aggregate_timing(e3, n_bytes, 0, dist, coll_dev)
        es = max(1, args.estep_steps)
        e_alg = me.last_algorithmic_bytes()  # N + 8 (S + 1) + 8 V
        eper = {k: v / es for k, v in ek.items()}
        e_pass = sum(eper.values())
        e_dom = max(eper, key=eper.get) if eper else "estep7_kernel"
        e_ach = e_alg / (e_pass * 1e-3) / 1e9 if e_pass > 0 else 0.0
        estep = {"value": round(tb3 * es / mx3 / 1e6, 2), "unit": "MB/s", "ms_per_step": round(mx3 / es * 1e3, 3), "steps": es,
                 "kernel_ms_per_step": {k: round(v, 3) for k, v in eper.items()},
                 "pieces": me.last_estep_pieces(), "redo_stretches": me.last_estep_redo(), "logz_sum": logz,
                 "roofline": {"bound": "hbm", "kernel": e_dom, "kernel_ms": round(eper.get(e_dom, 0.0), 3), "pass_kernels_ms": round(e_pass, 3),
                              "achieved": round(e_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(e_ach / HBM_PEAK_GBS, 5),
                              "traffic": None, "algorithmic_bytes_per_launch": int(e_alg)}}
        if rank == 0 and not args.no_cpu_baseline:
            estep["cpu_baseline"] = cpu_baseline_estep(toks, scores, flat, offs, me, min(args.cpu_seconds, 10.0))
        me.free()
        # ---- encode with the vocabulary an M-step makes of these expected counts (src/prune.rs:124-170: one score per token,
        # kept single-byte tokens with tiny scores): what `prune` encodes with; reported as `mstep_vocab_mb_s`, never as `value`
        if not args.no_distinct:
            from tokengeex_amd import _lib as _tl
            keep = np.array([1 if len(t) == 1 else 0 for t in toks], np.uint8)
            idx, sc3 = _tl.prune_m_step(expected, keep)
            idx, sc3 = np.asarray(idx, np.int64), np.asarray(sc3, np.float64)
            m3 = tgx.NativeModel([toks[i] for i in idx], sc3, device=dev)
            for _ in range(max(1, args.warmup)):
                m3.encode_corpus(corpus).free()
            sync_all()
            t4 = time.perf_counter()
            k4: dict[str, float] = {}
            for _ in range(args.steps):
                m3.encode_corpus(corpus).free()
                for k, v in m3.last_kernel_times().items():
                    k4[k] = k4.get(k, 0.0) + v
            sync_all()
            e4 = time.perf_counter() - t4
            mx4, tb4, _ = tdist.aggregate_timing(e4, n_bytes, 0, dist, coll_dev)
            estep["mstep_vocab_mb_s"] = round(tb4 * max(1, args.steps) / mx4 / 1e6, 2)
            estep["mstep_vocab"] = {"tokens": int(idx.size), "score_values": m3.score_values(), "lds_score_values": m3.last_encode_hot_values(),
                                    "kernel_ms_per_step": {k: round(v / max(1, args.steps), 3) for k, v in k4.items()}}
            m3.free()

    # ---- PCIe-inclusive rate of the host-buffer entry point (tgx_encode_batch + copies of ids and offsets
    # into caller memory), rank 0 only, after the timed region.  Reported as `e2e_mb_s`, never as `value`.
    e2e = None
    if rank == 0 and not args.no_e2e:
        def best_of(f, o, ids_buf, reps=4):
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                model.encode_batch_host(f, o, ids_out=ids_buf)  # chunks: upload | kernels | download overlapped
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return best
        # caller buffers in ordinary (pageable) memory, as a caller of the reference has them ...
        t_page = best_of(flat, offs, np.empty(int(n_tokens) + 16, np.uint32))
        # ... and in page-locked memory from tgx_host_alloc (DMA at the link's rate, no driver staging)
        from tokengeex_amd import _lib
        try:
            pf, po, pi = _lib.pinned_empty(flat.shape, np.uint8), _lib.pinned_empty(offs.shape, np.uint64), _lib.pinned_empty(int(n_tokens) + 16, np.uint32)
            pf[:] = flat
            po[:] = offs
            t_pin = best_of(pf, po, pi)
            del pf, po, pi
            e2e = {"e2e_mb_s": round(n_bytes / t_pin / 1e6, 2), "e2e_ms": round(t_pin * 1e3, 3),
                   "e2e_buffers": "caller buffers from tgx_host_alloc (page-locked)",
                   "e2e_pageable_mb_s": round(n_bytes / t_page / 1e6, 2), "e2e_pageable_ms": round(t_page * 1e3, 3)}
        except tgx.TokenGeeXError as exc:  # no page-locked memory to be had on this host: the pageable figure alone
            e2e = {"e2e_mb_s": round(n_bytes / t_page / 1e6, 2), "e2e_ms": round(t_page * 1e3, 3),
                   "e2e_buffers": f"pageable caller buffers (tgx_host_alloc failed: {exc})"}

    if dist is not None:
        dist.barrier()  # every rank is done with its device work; only rank 0 goes on (CPU leg, the line)

    if rank == 0:
        steps = max(1, args.steps)
        ms_per_step = max_elapsed / steps * 1e3
        mb_s = tot_bytes * steps / max_elapsed / 1e6
        per_step = {k: v / steps for k, v in kernel_ms.items()}
        dom = max(per_step, key=per_step.get) if per_step else "encode_kernel"
        dom_ms = per_step.get(dom, 0.0)
        pass_ms = sum(per_step.values())
        if corun_cus:  # the two encode kernels ran at once (mid-size batches): their times overlap
            pass_ms -= min(per_step.get("encode5_kernel", 0.0), per_step.get("encode6_kernel", 0.0))
        # roofline: the pass's algorithmic bytes (N + 4 T + 16 (S + 1)) belong to the
        # whole kernel sequence of a pass, so they are divided by the SUM of its kernels' times (HIP events on
        # the library's stream); the dominant kernel and its own time are named beside it.
        achieved = alg_bytes / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        # HBM bytes per pass from the rocprofv3 PMC passes of this workload (tools/pmc_traffic.sh; counters
        # cannot be read inside this process): the committed figure of the profiled build, with its source
        # named, and only for the workload it was taken on; null otherwise
        traffic, traffic_source = None, None
        tpath = _traffic_file()
        if tpath and (args.size_mb, args.vocab, args.kind, args.max_token_length, args.max_sample_len, args.distinct_scores, args.vocab_slice_mb) == (1024, 32000, "mixed", 16, 65536, False, 64):
            with open(tpath) as f:
                tj = json.load(f)
            enc = lambda names: {k.split("<")[0] for k in names if k.startswith(("encode", "estep"))}
            if enc(tj.get("kernels", {})) == enc(per_step):  # same encode kernels as the profiled build
                traffic = tj.get("hbm_bytes_per_pass_corrected")
                traffic_source = f"{os.path.relpath(tpath, ROOT)} ({tj.get('commit', '?')})"
            if estep and enc(tj.get("estep_kernels", {})) == {k.split("<")[0] for k in estep["kernel_ms_per_step"] if k.startswith("estep")} and tj.get("estep_hbm_bytes_per_pass_corrected"):
                estep["roofline"]["traffic"] = tj["estep_hbm_bytes_per_pass_corrected"]
                estep["roofline"]["traffic_source"] = traffic_source
        out = {
            "metric": "MB/s raw bytes encoded (and tokens/s) at 32K/64K vocab, 1/2/4/8 GPUs",
            "value": round(mb_s, 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"encode_ordinary_batch, {args.vocab} vocab (max token {args.max_token_length} B; generate stand-in over a "
                            f"{slice_mib} MiB slice of the mixed corpus; {n_values} distinct score values"
                            f"{': every token its own, as after an M-step' if args.distinct_scores else ''}; "
                            f"{model.last_encode_hot_values()} of them in the kernel's LDS copy), "
                            f"{args.size_mb} MiB {args.kind} corpus per GPU (samples <= {args.max_sample_len} B), "
                            f"ids bit-exact vs CPU oracle",
                "vocab_slice_mib": slice_mib, "distinct_score_values": n_values, "lds_score_values": model.last_encode_hot_values(),
                "bytes_per_gpu": n_bytes, "samples_per_gpu": n_samples, "tokens_per_gpu": int(n_tokens),
                "parallelism": f"dp{world} (samples sharded, no collective)",
            },
            "mib_per_s": round(tot_bytes * steps / max_elapsed / 1048576.0, 2),
            "tokens_per_s": round(tot_tokens * steps / max_elapsed, 1),
            "kernel_ms_per_step": {k: round(v, 3) for k, v in per_step.items()},
            "setup_s": round(setup_s, 1),
            "roofline": {"bound": "hbm", "kernel": dom, "kernel_ms": round(dom_ms, 3), "pass_kernels_ms": round(pass_ms, 3),
                         "achieved": round(achieved, 2),
                         "achieved_over_dominant_kernel": round(alg_bytes / (dom_ms * 1e-3) / 1e9, 2) if dom_ms > 0 else 0.0,
                         "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
        }
        if e2e:
            out.update(e2e)
        if distinct:
            out.update(distinct)
        if estep:
            out["estep"] = estep
        if not args.no_cpu_baseline:  # rank 0 only; the other ranks have left the timed region
            out["cpu_baseline"] = cpu_baseline(toks, scores, flat, offs, model, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(toks, scores, flat, offs, model, target_seconds: float) -> dict:
    """Times the CPU oracle ("port" of the reference's Rust path) on a bounded
    prefix of the same corpus with all host cores, and checks the GPU ids
    against it on that prefix (the parity gate of SURVEY.md §8d)."""
    import numpy as np

    from oracle import oracle as orc
    cores = max(1, min(os.cpu_count() or 1, 64))
    try:
        cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    ora = orc.OracleModel(toks, scores)
    # calibrate on 8 MiB, then size the sample for ~target_seconds
    k = int(np.searchsorted(offs, 8 << 20))
    k = max(1, min(k, offs.size - 1))
    t = time.perf_counter()
    ora.encode_batch_flat(flat[: int(offs[k])], offs[: k + 1], threads=cores)
    rate = float(offs[k]) / (time.perf_counter() - t)
    want = int(min(float(flat.size), rate * target_seconds))
    k = int(np.searchsorted(offs, want))
    k = max(1, min(k, offs.size - 1))
    sub_flat, sub_offs = flat[: int(offs[k])], offs[: k + 1]
    t = time.perf_counter()
    ids, oo = ora.encode_batch_flat(sub_flat, sub_offs, threads=cores)
    dt = time.perf_counter() - t
    res = model.encode_batch_flat(sub_flat, sub_offs)
    same = bool(np.array_equal(res.offsets(), oo) and np.array_equal(res.ids(), ids))
    res.free()
    if not same:
        raise SystemExit("PARITY FAILURE: GPU token ids differ from the CPU oracle on the baseline sample")
    # the same port on ONE thread (configs[0]'s "single thread" reading), on ~1/6 of the time budget
    k1 = int(np.searchsorted(offs, max(1 << 20, int(rate / cores * target_seconds / 6))))
    k1 = max(1, min(k1, k))
    t = time.perf_counter()
    ora.encode_batch_flat(flat[: int(offs[k1])], offs[: k1 + 1], threads=1)
    dt1 = time.perf_counter() - t
    return {"value": round(float(sub_flat.size) / dt / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": "port",
            "sample": f"first {k} samples ({sub_flat.size} bytes) of the same corpus, {cores} threads, "
                      f"{dt:.1f} s; GPU ids on this sample bit-exact: {same}",
            "tokens_per_s": round(float(ids.size) / dt, 1),
            "value_1thread": round(float(offs[k1]) / dt1 / 1e6, 2),
            "sample_1thread": f"first {k1} samples ({int(offs[k1])} bytes), 1 thread, {dt1:.1f} s"}


def _traffic_file():
    """The newest committed PMC traffic record (tools/pmc_traffic.sh): profiles/rNN/pmc_traffic.json."""
    for rnd in ("r04", "r03"):
        p = os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")
        if os.path.exists(p):
            return p
    return None


def cpu_baseline_estep(toks, scores, flat, offs, model, target_seconds: float) -> dict:
    """The CPU oracle's E-step (the reference's f64 log-domain arithmetic, src/lattice.rs:245-333) on a bounded prefix of
    the same corpus with all host cores, and the GPU's expected counts against it on that prefix."""
    import numpy as np

    import tokengeex_amd as tgx
    from oracle import oracle as orc
    cores = max(1, min(os.cpu_count() or 1, 64))
    try:
        cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    except AttributeError:
        pass
    ora = orc.OracleModel(toks, scores)
    k = max(1, min(int(np.searchsorted(offs, 1 << 20)), offs.size - 1))
    t = time.perf_counter()
    ora.estep_flat(flat[: int(offs[k])], offs[: k + 1], threads=cores)
    rate = float(offs[k]) / (time.perf_counter() - t)
    k = max(1, min(int(np.searchsorted(offs, int(min(float(flat.size), rate * target_seconds)))), offs.size - 1))
    sf, so = flat[: int(offs[k])], offs[: k + 1]
    t = time.perf_counter()
    st, want, wz, _ = ora.estep_flat(sf, so, threads=cores)
    dt = time.perf_counter() - t
    cs = tgx.NativeCorpus(sf, so, device=model.device)
    got, gz = model.estep(cs)
    cs.free()
    longest = int(np.diff(so.astype(np.int64)).max())
    rtol = 1.2e-8 * max(1.0, min(81920, longest) / 4096.0)  # the oracle's own rounding: tests/test_estep_pairs_gpu.py
    ok = bool(st == orc.OK and np.allclose(got, want, rtol=rtol, atol=1e-12) and np.array_equal(got != 0, want != 0)
              and abs(gz - wz) <= 1e-12 * abs(wz) + 1e-9)
    if not ok:
        raise SystemExit("PARITY FAILURE: GPU expected counts differ from the CPU oracle on the E-step baseline sample")
    return {"value": round(float(sf.size) / dt / 1e6, 2), "unit": "MB/s", "cores": cores, "kind": "port",
            "sample": f"first {k} samples ({sf.size} bytes) of the same corpus, {cores} threads, {dt:.1f} s; GPU expected counts on "
                      f"this sample within rtol {rtol:.1e} / atol 1e-12 of the oracle's, same support, log Z to 1e-12: {ok}"}


if __name__ == "__main__":
    main()
