#!/usr/bin/env python3
"""Time the two contractions of a LowRankCovMatrix product alone (S'X: TN, N_s x l, K = n; S T: NN, n x l, K = N_s):
    python tools/bench_lrcm_products.py [--n 1000000] [--samples 1024] [--l 320] [--reps 5]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--samples", type=int, default=1024)
ap.add_argument("--l", type=int, default=320)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
ctx = gsi.Context(0)
op = gsi.lowrank_synthetic_operator(ctx, a.n, a.samples, seed=0, decay=0.75)
X = gsi.DeviceMatrix(ctx, a.n, a.l).randn(1)
Y = gsi.DeviceMatrix(ctx, a.n, a.l)
lib = ctx.lib
gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib)
ctx.sync()
ctx.profile(True); ctx.phase_reset()
for _ in range(a.reps):
    gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib)
ph = ctx.phase_times(); ctx.profile(False)
fl = 2.0 * a.n * a.samples * a.l
for key, name in (("gemm_t", "S'X (TN)"), ("gemm_n", "S T (NN)")):
    ms = ph[key][0] / ph[key][1]
    print(f"{name}: n={a.n} N_s={a.samples} l={a.l}: {ms:.3f} ms  {fl/ms/1e9:.2f} TFLOP/s", flush=True)
