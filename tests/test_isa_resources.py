"""A guard on what hipcc makes of the hot kernels.

The contraction kernels (RandMatFact.jl:55,67,70,85 -- 70 % of a step) are compiled at exactly 256 VGPRs, where the register
allocator, not the source, decides the speed: a derived LDS pointer with offset ZERO once cost S'X 24 % and was found only
because "a slow box kept being slow".  build.py keeps the compiler's kernel-resource-usage remarks of every kernel; this test
compares the instantiations that carry a step with the committed record profiles/isa_resources.json and fails, printing
old -> new, when VGPR / AGPR / SGPR counts, scratch bytes, spill counts, occupancy or LDS of any of them move.  A move is not
necessarily a regression -- it is the signal to A/B the kernel against the previous build in one process
(tools/bench_lrcm_products.py with GSI_HIP_LIB) before `python3 tools/isa_resources.py --update` records the new numbers.
hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU suite."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_resources  # noqa: E402


def test_hot_kernel_resources_match_the_record():
    rec = json.load(open(isa_resources.RECORD))
    cur = isa_resources.current()
    moved = isa_resources.diff(rec, cur)
    note = "" if rec["hipcc_version"] == cur["hipcc_version"] else \
        f"\n(the compiler changed: recorded with '{rec['hipcc_version']}', now '{cur['hipcc_version']}')"
    assert not moved, "hipcc allocated hot kernels differently from profiles/isa_resources.json -- A/B them before updating the " \
                      "record (python3 tools/isa_resources.py --update):\n" + "\n".join(moved) + note


def test_the_record_covers_the_kernels_that_carry_a_step():
    """The six kernels on top of the last profiled bench run (profiles/r04_bench_kernel_stats.csv) + the generated forms."""
    rec = json.load(open(isa_resources.RECORD))["kernels"]
    for k in ("gsi::hipk::gemm_f64_kernel<10, false, 0, 0>", "gsi::hipk::gemm_f64_kernel<10, true, 0, 0>",
              "gsi::hipk::lu_leaf_kernel<512, 8, false, false>", "gsi::hipk::lu_rankk_kernel<64, 1, 64>",
              "gsi::hipk::sy_kernel<20, true>", "gsi::hipk::tr_kernel<20, true>", "gsi::hipk::gemm_f64_kernel<10, false, 1, 0>",
              "gsi::hipk::gemm_f64_kernel<10, false, 1, 1>", "gsi::hipk::gemm_f64_kernel<10, false, 1, 2>"):
        assert k in rec, k
        assert rec[k]["occupancy_waves_per_simd"] >= 1 and rec[k]["vgprs"] + rec[k]["agprs"] <= 512
    # the contraction forms are the ones on the cliff: all VGPRs in use, accumulators NOT moved to AGPRs
    for k in ("gsi::hipk::gemm_f64_kernel<10, false, 0, 0>", "gsi::hipk::gemm_f64_kernel<10, true, 0, 0>"):
        assert rec[k]["vgprs"] == 256 and rec[k]["agprs"] == 0
