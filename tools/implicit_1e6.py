#!/usr/bin/env python3
"""BASELINE.json north_star size, implicit form: randsvd of the (GRID^2) x (GRID^2) Gaussian grid covariance at
rank 256 with the operator NEVER stored (gsi_op_gridcov_implicit: entries regenerated inside the MFMA kernel on
each of the 2q+2 passes).  At GRID = 1000 the matrix would be 8 TB.  The Gaussian kernel on a regular grid is a
Kronecker product Ax (x) Ay, so its exact spectrum is the set of products of two 1-D Toeplitz spectra -- used
here as the size-independent check of the singular values."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsi_amd as gsi

grid = int(os.environ.get("GRID", 1000))
ell = float(os.environ.get("ELL", 50.0))
K, p, q = int(os.environ.get("K", 256)), int(os.environ.get("P", 64)), int(os.environ.get("Q", 2))
n, l = grid * grid, K + p
ctx = gsi.Context(0)
lib = ctx.lib
op = gsi.gridcov_implicit_operator(ctx, grid, grid, ell)
Om = gsi.DeviceMatrix(ctx, n, l).randn(5)
Z = gsi.DeviceMatrix(ctx, n, l)
Sv = gsi.DeviceMatrix(ctx, l, 1)
print(f"n = {n}, l = {l}, q = {q}; operator would be {8*n*n/1e12:.2f} TB, device bytes in use {ctx.device_bytes()/1e9:.2f} GB", flush=True)
ctx.profile(True); ctx.phase_reset(); ctx.sync()
t0 = time.time()
gsi._lib.check(lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, Sv.h), lib)
ctx.sync()
dt = time.time() - t0
ph = ctx.phase_times(); ctx.profile(False)
passes = 2 * q + 2
flops = passes * 2.0 * n * n * l
gemm_ms = ph["gemm_n"][0] + ph["gemm_t"][0]
print(f"randsvd: {dt:.2f} s", {k: round(v[0], 1) for k, v in ph.items()}, flush=True)
print(f"generated-operand products: {flops/ (gemm_ms*1e-3) / 1e12:.1f} TFLOP/s fp64 over {passes} passes; "
      f"equivalent stored-matrix stream {passes*8.0*n*n/dt/1e9:.0f} GB/s", flush=True)
Sh = Sv.to_host()[:, 0]
d = np.arange(grid)
T = np.exp(-(d[:, None] - d[None, :]) ** 2 / (2 * ell * ell))
ev1 = np.linalg.eigvalsh(T)[::-1]
ev = np.sort(np.outer(ev1[:200], ev1[:200]).ravel())[::-1][:K]
rel = np.abs(Sh[:K] - ev) / ev
print("sv rel err vs exact Kronecker spectrum: max over top 8 / 64 / K:", rel[:8].max(), rel[:64].max(), rel.max(), flush=True)
Zh = Z.to_host()
G = Zh[:, :K].T @ Zh[:, :K]
print("max |Z'Z - diag(S)| / S1 =", np.abs(G - np.diag(Sh[:K])).max() / Sh[0], " trailing zero:", bool(np.all(Zh[:, K:] == 0)))
