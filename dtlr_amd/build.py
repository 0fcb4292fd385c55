"""Build libdtlr_hip.so (gfx950 only) in-tree with hipcc.  `python -m dtlr_amd.build [--force]`.

No torch headers are involved: the library is a pure C-ABI shared object taking raw device
pointers and a hipStream_t (include/dtlr_hip.h), so it is independent of the torch/ROCm pairing.
"""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdtlr_hip.so")
STAMP = os.path.join(HERE, ".libdtlr_hip.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _fingerprint() -> str:
    h = hashlib.sha256()
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "dtlr_hip.h")]:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())   # path-independent: the tree moves on the GPU box
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


LIB_F16 = os.path.join(HERE, "libdtlr_hip_f16.so")


def _src_hash(src, defs) -> str:
    h = hashlib.sha256()
    for p in [src] + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "dtlr_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(defs).encode())
    return h.hexdigest()


def _compile_all(defs, objdir, out, verbose, force=False):
    """every csrc/*.hip -> objdir/*.o (in parallel: one hipcc per source; an object whose source, headers and flags are unchanged
    is kept), linked into `out`."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        jobs.append(([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + defs + ["-c", src, "-o", obj], obj,
                     _src_hash(src, defs)))

    def run(job):
        cmd, obj, hsh = job
        tag = obj + ".hash"
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read() == hsh:
            return obj
        if verbose:
            print("[dtlr build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(tag, "w") as f:
            f.write(hsh)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(run, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
    if verbose:
        print("[dtlr build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = True) -> str:
    """libdtlr_hip.so (16-bit format = bf16) and libdtlr_hip_f16.so (the same sources with -DDTLR_HALF_IS_F16: 16-bit format = IEEE
    fp16, the parity build -- csrc/dtlr_common.h).  Returns the path of the first; both are rebuilt when any source changes."""
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(LIB_F16) and os.path.exists(STAMP) and open(STAMP).read().strip() == fp:
        return LIB
    _compile_all([], CSRC, LIB, verbose, force)
    _compile_all(["-DDTLR_HALF_IS_F16"], os.path.join(CSRC, "f16"), LIB_F16, verbose, force)
    with open(STAMP, "w") as f:
        f.write(fp)
    return LIB


def build_instrumented(verbose: bool = True, trace: bool = False) -> str:
    """Profiling-only variants: same sources with -DDTLR_EXPERIMENT (the env A/B switches of csrc/dtlr_common.h), -DDTLR_GEMM_ABLATION (phase switches in the GEMM; env
    DTLR_GEMM_ABLATE) and, with trace, -DDTLR_GEMM_TRACE (per-role cycle attribution; perturbs the kernel).
    Never loaded by the product; select with DTLR_HIP_LIB."""
    out = os.path.join(HERE, "libdtlr_hip_trace.so" if trace else "libdtlr_hip_instr.so")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-DDTLR_GEMM_ABLATION", "-DDTLR_EXPERIMENT"] + (["-DDTLR_GEMM_TRACE"] if trace else []) + _sources() + ["-o", out]
    if verbose:
        print("[dtlr build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--instr" in sys.argv or "--trace" in sys.argv:
        print(build_instrumented(trace="--trace" in sys.argv))
    else:
        print(build(force="--force" in sys.argv))
