"""Builds tokengeex_amd/libtgx.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python tokengeex_amd/build.py [--force]      (run as a script: importing the package needs the .so)

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtgx.so")
SOURCES = ["kernels.hip", "estep.hip", "encode2.hip", "encode4l.hip", "encode5.hip", "estep4.hip", "estep4l.hip", "estep7.hip", "trace2.hip", "cuts.hip", "pairs.hip", "generate.hip", "tgx_api.cpp", "trie_build.cpp", "prune_host.cpp", "frontback.cpp", "unicode_norm.cpp"]
HEADERS = ["kernels.h", "device_common.h", "trace_body.h", "trie_build.h", "unicode_tables.h", os.path.join("..", "..", "include", "tgx.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


SYNTH_OUT = os.path.join(HERE, "libtgx_synth.so")


def build_synth(force: bool = False) -> str:
    """Host-only corpus generator used by bench.py / tests (gcc, no GPU code)."""
    src = os.path.join(CSRC, "synth_gen.c")
    if not force and os.path.exists(SYNTH_OUT) and os.path.getmtime(SYNTH_OUT) >= os.path.getmtime(src):
        return SYNTH_OUT
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-o", SYNTH_OUT + ".tmp", src, "-lm"])
    os.replace(SYNTH_OUT + ".tmp", SYNTH_OUT)
    return SYNTH_OUT


FAST_OUT = os.path.join(HERE, "_tgxfast.so")


def build_pyfast(force: bool = False) -> str:
    """The native ends of the list[str] -> list[list[int]] surface (csrc/pyfast.c: CPython C API, gcc, no GPU code)."""
    import sysconfig
    src = os.path.join(CSRC, "pyfast.c")
    if not force and os.path.exists(FAST_OUT) and os.path.getmtime(FAST_OUT) >= os.path.getmtime(src):
        return FAST_OUT
    inc = sysconfig.get_paths()["include"]
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-pthread", "-I", inc, "-o", FAST_OUT + ".tmp", src])
    os.replace(FAST_OUT + ".tmp", FAST_OUT)
    return FAST_OUT


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles every source that is newer than its object (or whose headers are) with hipcc, in parallel, into
    csrc/build/ (git-ignored), then links tokengeex_amd/libtgx.so."""
    from concurrent.futures import ThreadPoolExecutor
    build_synth(force)
    build_pyfast(force)
    if not force and not needs_build():
        return OUT
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    hdr_time = max(hdr_time, os.path.getmtime(os.path.abspath(__file__)))
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
             "-ffp-contract=off",  # the DP's f64 adds must not be fused or reassociated
             "-munsafe-fp-atomics",  # f64 atomicAdd as one hardware atomic (E-step counts)
             "-Wall", "-Wno-unused-result", "-Wno-pass-failed", "-x", "hip"]
    flags += os.environ.get("TGX_EXTRA_HIPCC_FLAGS", "").split()  # experiments only (e.g. -DTGX_E7_STAMPS)
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([_hipcc()] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    link = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-o", OUT + ".tmp", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
