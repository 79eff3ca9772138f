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
SOURCES = ["gemm_f64.hip", "gemm_f64_gen1.hip", "syrk_f64.hip", "panel_lu_leaf.hip", "panel_qr.hip", "cholqr.hip", "jacobi_svd.hip", "misc.hip", "lsqr_dev.hip", "pointcov_gemm.hip", "fft_cov.hip",
           "hip_backend.hip", "pipeline.cpp", "api.cpp"]
HEADERS = ["backend.hpp", "hip_common.hpp", "pipeline.hpp", "lsqr_state.hpp", "pointcov.hpp", "pointcov_gen.hpp", "host_staging.hpp", "gemm_f64_kernel.inc.hpp", "../../include/gsi_hip.h"]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
            "-Wno-unused-result", "-Wno-unused-value"]
# Every kernel file is compiled with the backend's resource-usage remarks on; what hipcc made of each kernel (VGPRs, AGPRs,
# SGPRs, scratch, spills, occupancy, LDS) is kept beside the object as build/<source>.resources.json.  The contraction and
# panel kernels sit on register-allocation cliffs (a zero-offset pointer cost 24 %): tests/test_isa_resources.py
# compares the hot instantiations with the committed profiles/isa_resources.json and fails when one moves.
RESOURCE_FLAG = "-Rpass-analysis=kernel-resource-usage"
_RES_KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
             "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
             "LDS Size [bytes/block]": "lds_bytes_per_block", "Dynamic Stack": "dynamic_stack"}


def parse_resource_remarks(text):
    """{mangled kernel name: {vgprs, agprs, sgprs, scratch_bytes_per_lane, ...}} from hipcc's kernel-resource-usage remarks."""
    out, cur = {}, None
    for line in text.splitlines():
        if "remark:" not in line or "[-Rpass-analysis=kernel-resource-usage]" not in line:
            continue
        body = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if ":" not in body:
            continue
        key, val = [x.strip() for x in body.rsplit(":", 1)]
        if key == "Function Name":
            cur = out.setdefault(val, {})
        elif cur is not None and key in _RES_KEYS:
            cur[_RES_KEYS[key]] = (val == "True") if key == "Dynamic Stack" else int(val)
    return out


def hipcc_version():
    r = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True)
    lines = [ln.strip() for ln in r.stdout.splitlines() if ln.strip()]
    return "; ".join(lines[:2])


def kernel_resources():
    """Resource usage of every kernel of the library as the LAST compile of each source reported it (build() first)."""
    import json
    res = {}
    for src in SOURCES:
        if not src.endswith(".hip"):
            continue
        path = os.path.join(OBJ, src.replace(".", "_") + ".resources.json")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run build() (it recompiles sources whose resource record is absent)")
        for name, r in json.load(open(path)).items():
            res[name] = dict(r, source=src)
    return res


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
        is_hip = src.endswith(".hip")
        resj = os.path.join(OBJ, src.replace(".", "_") + ".resources.json")
        if force or _newer(o, [s] + hdrs) or (is_hip and not os.path.exists(resj)):
            cmd = [hipcc] + CXXFLAGS + (["-x", "hip", RESOURCE_FLAG] if is_hip else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        import json
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if RESOURCE_FLAG in cmd:
            obj = cmd[cmd.index("-o") + 1]
            with open(obj[:-2] + ".resources.json", "w") as f:
                json.dump(parse_resource_remarks(r.stderr), f, indent=0, sort_keys=True)
            # the remarks are not warnings: keep only what is left for the verbose log
            return "\n".join(ln for ln in r.stderr.splitlines() if "kernel-resource-usage" not in ln and not ln.startswith(" ") and "remark" not in ln)
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn:
                    print(warn)
    if jobs or force or _newer(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-Wl,-Bsymbolic", "-o", LIB] + objs + ["-ldl"])
    # what compiled the library, as a build-time fact beside it (bench.py prints it; it must not start a compiler at run time:
    # under `rocprofv3 --pmc` every child inherits the profiler's preload, and hipcc's own exec of its helper is then an exec
    # from a process that has initialised the GPU -- the GPU boxes refuse that)
    vfile = os.path.join(OBJ, "hipcc_version.txt")
    if jobs or force or not os.path.exists(vfile):
        with open(vfile, "w") as f:
            f.write(hipcc_version() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
