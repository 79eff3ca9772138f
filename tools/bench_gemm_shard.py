#!/usr/bin/env python3
"""Operator products at the per-rank shapes of a G-GPU C2 run (row shard mloc x n, mloc = n/G), on one GPU:
    python tools/bench_gemm_shard.py [--n 65536] [--G 8] [--l 160]
A*X is an (mloc x n)(n x l) product (few row blocks -> split-K), A'*Y an (n x mloc)(mloc x l) one."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--G", type=int, default=8)
ap.add_argument("--l", type=int, default=160)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
ctx = gsi.Context(0)
lib = ctx.lib
for G in ([a.G] if a.G > 0 else [2, 4, 8]):
    mloc = a.n // G
    rng = np.random.default_rng(0)
    A = np.asfortranarray(rng.standard_normal((mloc, 1024))[:, np.arange(a.n) % 1024])
    op = gsi.dense_operator(ctx, A)
    del A
    X = gsi.DeviceMatrix(ctx, a.n, a.l).randn(1)
    Y = gsi.DeviceMatrix(ctx, mloc, a.l)
    Z = gsi.DeviceMatrix(ctx, a.n, a.l)
    for trans, src, dst in ((0, X, Y), (1, Y, Z)):
        gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, src.h, dst.h), lib)
        ctx.sync(); ctx.profile(True); ctx.phase_reset()
        for _ in range(a.reps):
            gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, trans, src.h, dst.h), lib)
        ph = ctx.phase_times(); ctx.profile(False)
        key = "gemm_t" if trans else "gemm_n"
        ms = ph[key][0] / ph[key][1]
        print(f"G={G} mloc={mloc} trans={trans}: {ms:.3f} ms  {2.0*mloc*a.n*a.l/ms/1e9:.2f} TFLOP/s", flush=True)
    for h in (X, Y, Z, op):
        h.close()
