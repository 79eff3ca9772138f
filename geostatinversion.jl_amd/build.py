"""Build libgsi_hip.so (gfx950) in-tree with hipcc.  `python build.py [--force]`.

The product library is compiled for MI355X only (--offload-arch=gfx950); hipcc cross-
compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libgsi_hip.so")
SOURCES = ["gemm_f64.hip", "gemm_f64_gen1.hip", "syrk_f64.hip", "panel_lu.hip", "panel_lu_leaf.hip", "panel_qr.hip", "cholqr.hip", "jacobi_svd.hip", "misc.hip", "lsqr_dev.hip", "pointcov.hip", "pointcov_gemm.hip", "fft_cov.hip",
           "hip_backend.hip", "pipeline.cpp", "api.cpp"]
HEADERS = ["backend.hpp", "hip_common.hpp", "pipeline.hpp", "lsqr_state.hpp", "pointcov.hpp", "pointcov_gen.hpp", "host_staging.hpp", "gemm_f64_kernel.inc.hpp", "../../include/gsi_hip.h"]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
            "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libgsi_hip.so cannot be built (there is no CPU fallback)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".", "_") + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [hipcc] + CXXFLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn:
                    print(warn)
    if jobs or force or _newer(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,-Bsymbolic", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
