#!/usr/bin/env python3
"""One randsvd through the implicit grid covariance at a size that takes seconds: python tools/implicit_ab.py [grid] [kind]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
g = int(sys.argv[1]) if len(sys.argv) > 1 else 500
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, K, p, q = g * g, 256, 64, 2
ctx = gsi.Context(0)
op = gsi.gridcov_implicit_operator(ctx, g, g, 50.0 if kind == 0 else 100.0, kind=kind)
Om = gsi.DeviceMatrix(ctx, n, K + p).randn(1); Z = gsi.DeviceMatrix(ctx, n, K + p)
for it in range(2):
    ctx.profile(True); ctx.phase_reset(); ctx.sync(); t0 = time.time()
    gsi._lib.check(ctx.lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, None), ctx.lib); ctx.sync()
    dt = time.time() - t0; ph = ctx.phase_times(); ctx.profile(False)
    gm = ph["gemm_n"][0] + ph["gemm_t"][0]
    print(f"kind {kind} n={n}: {dt*1e3:.1f} ms, products {gm:.1f} ms = {6*2.0*n*n*(K+p)/gm/1e9:.1f} TFLOP/s", flush=True)
