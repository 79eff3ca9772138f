#!/usr/bin/env python3
"""ONE product of the implicit grid covariance (table-based generator) with device-resident X: python tools/implicit_product.py [grid side] [l]
(the program a rocprofv3 --pmc pass is pointed at: how many bytes of X does a product really fetch?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
l = int(sys.argv[2]) if len(sys.argv) > 2 else 320
n = g * g
ctx = gsi.Context(0)
op = gsi.gridcov_implicit_operator(ctx, g, g, 100.0, kind=1)
X = gsi.DeviceMatrix(ctx, n, l).randn(1)
Y = gsi.DeviceMatrix(ctx, n, l)
ctx.sync(); t0 = time.perf_counter()
gsi._lib.check(ctx.lib.gsi_op_mul_dev(ctx.h, op.h, 0, X.h, Y.h), ctx.lib)
ctx.sync(); dt = time.perf_counter() - t0
print(f"n = {n}, l = {l}: {dt*1e3:.1f} ms, {2.0*n*n*l/dt/1e12:.2f} TFLOP/s; X panel {8.0*n*l/1e9:.2f} GB", flush=True)
