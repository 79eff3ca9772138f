#!/usr/bin/env python3
"""What hipcc made of the hot kernels: VGPRs / AGPRs / SGPRs, scratch, spills, occupancy, LDS per instantiation, as the
compiler's own kernel-resource-usage remarks report them at build time (geostatinversion.jl_amd/build.py keeps them beside
every object).  `python3 tools/isa_resources.py` prints the pinned set; `--update` rewrites profiles/isa_resources.json (do
that ONLY together with an A/B timing of the kernels that moved: DESIGN.md 4.1); `--all` lists every kernel of the library.

The pinned set = the instantiations that carry a step (profiles/r04_bench_kernel_stats.csv, top rows) plus the generated-
operand forms: the contraction kernels sit at 256 VGPRs, where an edit that does not change a single instruction of the
source's meaning moved S'X by 24 % (a derived LDS pointer with offset zero).  tests/test_isa_resources.py compares.
"""
import importlib.util
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "profiles", "isa_resources.json")
# demangled-name patterns (regular expressions, anchored at the start of the demangled name without its return type)
PINNED = [
    r"gsi::hipk::gemm_f64_kernel<10, false, 0, 0>",      # S T / dense A X                 (RandMatFact.jl:55,70)
    r"gsi::hipk::gemm_f64_kernel<10, true, 0, 0>",       # S'X / dense A'X                 (RandMatFact.jl:67,85)
    r"gsi::hipk::gemm_f64_kernel<8, false, 0, 0>",       # l = 256 (p = 0)
    r"gsi::hipk::gemm_f64_kernel<8, true, 0, 0>",
    r"gsi::hipk::gemm_f64_kernel<10, false, 1, [012]>",  # table-generated operand (the implicit 10^6 x 10^6 covariance)
    r"gsi::hipk::gemm_f64_kernel<10, false, 2, [012]>",  # scattered-point operand, 128 x 160 form
    r"gsi::hipk::\(anonymous namespace\)::pointcov_wide_kernel<",   # scattered-point operand: 96 x 320 and 192 x 160 forms
    r"gsi::hipk::lu_leaf_kernel<512, 8, false, false>",  # lu(Y).L leaves                   (RandMatFact.jl:60-61,68-69,72-73)
    r"gsi::hipk::lu_rankk_kernel<64, 1, (64|128)>",   # (64-column chunks at three waves per SIMD: the default since round 5)
    r"gsi::hipk::sy_kernel<20, true>",                   # CholeskyQR Gram                  (RandMatFact.jl:75-76)
    r"gsi::hipk::tr_kernel<20, true>",
    r"gsi::hipk::jacobi_block_kernel<16>",               # svd(B)                           (RandMatFact.jl:86)
    r"gsi::hipk::fft_pass_kernel<",                      # FFT covariance passes            (FFTRF.jl:83-90)
]
FIELDS = ["vgprs", "agprs", "sgprs", "scratch_bytes_per_lane", "vgpr_spills", "sgpr_spills", "occupancy_waves_per_simd",
          "lds_bytes_per_block"]


def _build_module():
    spec = importlib.util.spec_from_file_location("gsi_build", os.path.join(ROOT, "geostatinversion.jl_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def demangle(names):
    filt = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(filt):
        filt = "c++filt"
    r = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.splitlines()


def short(dem):
    """'void ns::kernel<args>(params)' -> 'ns::kernel<args>' (the parameter list is noise; template arguments are the identity)."""
    d = dem[5:] if dem.startswith("void ") else dem
    depth = 0
    for i, ch in enumerate(d):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return d[:i]
    return d


def current(all_kernels=False):
    mod = _build_module()
    mod.build()
    res = mod.kernel_resources()
    names = sorted(res)
    dem = [short(d) for d in demangle(names)]
    out = {}
    for m, d in zip(names, dem):
        if all_kernels or any(re.match(pat, d) for pat in PINNED):
            out[d] = {k: res[m].get(k) for k in FIELDS}
            out[d]["source"] = res[m]["source"]
    return {"hipcc_version": mod.hipcc_version(), "kernels": out}


def diff(old, new):
    lines = []
    for k in sorted(set(old["kernels"]) | set(new["kernels"])):
        a, b = old["kernels"].get(k), new["kernels"].get(k)
        if a is None:
            lines.append(f"  + {k}: new instantiation {b}")
        elif b is None:
            lines.append(f"  - {k}: no longer compiled")
        else:
            ch = [f"{f} {a.get(f)} -> {b.get(f)}" for f in FIELDS if a.get(f) != b.get(f)]
            if ch:
                lines.append(f"  * {k}: " + ", ".join(ch))
    return lines


if __name__ == "__main__":
    cur = current("--all" in sys.argv)
    if "--update" in sys.argv:
        with open(RECORD, "w") as f:
            json.dump(cur, f, indent=1, sort_keys=True)
        print("wrote", RECORD)
    print(cur["hipcc_version"])
    w = max(len(k) for k in cur["kernels"])
    print(f"{'kernel':{w}s}  vgpr agpr sgpr scratch vspill sspill occ    lds")
    for k, r in sorted(cur["kernels"].items()):
        print(f"{k:{w}s}  {r['vgprs']:4d} {r['agprs']:4d} {r['sgprs']:4d} {r['scratch_bytes_per_lane']:7d} {r['vgpr_spills']:6d} "
              f"{r['sgpr_spills']:6d} {r['occupancy_waves_per_simd']:3d} {r['lds_bytes_per_block']:6d}")
    if os.path.exists(RECORD) and "--all" not in sys.argv:
        d = diff(json.load(open(RECORD)), cur)
        print("\nagainst profiles/isa_resources.json:", "identical" if not d else "\n" + "\n".join(d))
