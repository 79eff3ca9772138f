#!/usr/bin/env python3
"""randsvd wall time over a grid of problem shapes (dense Gaussian covariance generated on the device), to spot
shape-dependent performance cliffs:  python tools/shape_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
ctx = gsi.Context(0)
lib = ctx.lib
q = 2
print(f"{'n':>7s} {'l':>4s} {'ms':>9s} {'TF(all)':>8s}  phases(ms)")
for (nx, ny) in [(100, 100), (141, 142), (173, 174), (224, 223), (250, 201)]:
    n = nx * ny
    op = gsi.gridcov_operator(ctx, nx, ny, 12.0, 0)
    for l in (32, 50, 100, 130, 160, 200, 300):
        K, p = l - l // 5, l // 5
        Om = gsi.DeviceMatrix(ctx, n, l).randn(3)
        Z = gsi.DeviceMatrix(ctx, n, l)
        S = gsi.DeviceMatrix(ctx, l, 1)
        gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, S.h), lib)
        ctx.sync(); ctx.profile(True); ctx.phase_reset()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, S.h), lib)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        ph = ctx.phase_times(); ctx.profile(False)
        flops = (2 * q + 2) * 2.0 * n * n * l
        print(f"{n:7d} {l:4d} {dt*1e3:9.2f} {flops/dt/1e12:8.1f}  " +
              " ".join(f"{k}={v[0]/reps:.1f}" for k, v in ph.items() if v[0] > 0), flush=True)
        for h in (Om, Z, S):
            h.close()
    op.close()
