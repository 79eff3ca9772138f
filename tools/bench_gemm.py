#!/usr/bin/env python3
"""Time the operator products alone (A*X and A'*X, A and X resident in HBM):
    python tools/bench_gemm.py [--grid 256] [--l 160] [--reps 5]
Prints avg ms and TFLOP/s per kernel.  With GSI_GEMM_ABLATE set the results are garbage (timing only)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=256)
ap.add_argument("--l", type=int, default=160)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = gsi.Context(0)
n = a.grid * a.grid
op = gsi.gridcov_operator(ctx, a.grid, a.grid, 16.0, 0)
X = gsi.DeviceMatrix(ctx, n, a.l).randn(1)
Y = gsi.DeviceMatrix(ctx, n, a.l)
lib = ctx.lib
for trans in (0, 1):
    gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, X.h, Y.h), lib)
    ctx.sync()
    ctx.profile(True); ctx.phase_reset()
    for _ in range(a.reps):
        gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, X.h, Y.h), lib)
    ph = ctx.phase_times(); ctx.profile(False)
    key = "gemm_t" if trans else "gemm_n"
    ms = ph[key][0] / ph[key][1]
    print(f"ablate={os.environ.get('GSI_GEMM_ABLATE','0')} trans={trans} n={n} l={a.l}: {ms:.3f} ms  {2.0*n*n*a.l/ms/1e9:.2f} TFLOP/s", flush=True)
