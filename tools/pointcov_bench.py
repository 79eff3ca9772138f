#!/usr/bin/env python3
"""Row-streamed scattered-point covariance (gsi_op_pointcov_implicit) against the table-based implicit grid operator at the
same size: one product A*X each, device-resident X.   python tools/pointcov_bench.py [grid side, default 450] [l, default 320]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
g = int(sys.argv[1]) if len(sys.argv) > 1 else 450
l = int(sys.argv[2]) if len(sys.argv) > 2 else 320
n = g * g
ctx = gsi.Context(0)
lib = ctx.lib
xs, ys = np.meshgrid(np.arange(g, dtype=np.float64), np.arange(g, dtype=np.float64), indexing="ij")
P = np.stack([xs.ravel(), ys.ravel()])                       # the same points as the grid operator, as scattered coordinates
ops = {"table-based implicit grid operator (exp(-d/ell))": gsi.gridcov_implicit_operator(ctx, g, g, 45.0, kind=1),
       "row-streamed scattered-point operator (exponential)": gsi.pointcov_implicit_operator(ctx, P, "exponential", ell=45.0),
       "row-streamed scattered-point operator (matern52)": gsi.pointcov_implicit_operator(ctx, P, "matern52", ell=45.0)}
X = gsi.DeviceMatrix(ctx, n, l).randn(1)
Y = gsi.DeviceMatrix(ctx, n, l)
ref = None
for name, op in ops.items():
    gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2):
        gsi._lib.check(lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), lib)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 2
    col = Y.to_host()[:, 0]
    if ref is None:
        ref = col
    err = np.abs(col - ref).max() / np.abs(ref).max() if "exponential" in name or "table" in name else float("nan")
    print(f"{name}: n = {n}, l = {l}: {dt*1e3:.1f} ms per product, {2.0*n*n*l/dt/1e12:.1f} TFLOP/s of contraction, "
          f"max rel diff vs table operator {err:.2e}", flush=True)
