#!/usr/bin/env python3
"""Host-boundary probe at C2 (65536^2 host matrix): whole upload, randsvd on the resident operator with host Omega / Z, and
gsi_randsvd_dense_host for several row-block heights (GSI_STAGE_BLOCK_ROWS is read per call; GSI_STAGE_THREADS /
GSI_STAGE_CHUNK_MB per process).  usage: python3 tools/boundary_probe.py [n_grid=256] [blocks=4096,16384,32768]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsi_amd as gsi  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 256
blocks = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "4096,16384,32768").split(",")]
K, p, q = 128, 32, 2
n, l = g * g, K + p
L = gsi._lib
ctx = gsi.default_context()
lib = ctx.lib
d = np.arange(g, dtype=np.float64)
k1 = np.exp(-((d[:, None] - d[None, :]) ** 2) / (2.0 * 16.0 ** 2))
A = np.kron(k1, k1).T
Om = np.asfortranarray(np.random.default_rng(2).standard_normal((n, l)))
Z = np.zeros((n, l), order="F")
S = np.zeros(l)
print(f"threads {os.environ.get('GSI_STAGE_THREADS', '4')} chunk {os.environ.get('GSI_STAGE_CHUNK_MB', '16')} MiB; matrix {A.nbytes / 1e9:.1f} GB", flush=True)
for rep in range(2):
    t0 = time.perf_counter()
    op = gsi.dense_operator(ctx, A)
    t1 = time.perf_counter()
    L.check(lib.gsi_randsvd(ctx.h, op.h, L.dptr(Om), K, p, q, L.dptr(Z), S.ctypes.data_as(L.c_dp)), lib)
    t2 = time.perf_counter()
    op.close()
    print(f"whole upload {t1 - t0:.3f} s ({A.nbytes / (t1 - t0) / 1e9:.1f} GB/s) + randsvd host Omega/Z {1e3 * (t2 - t1):.1f} ms = {t2 - t0:.3f} s", flush=True)
S0 = S.copy()
for b in blocks:
    os.environ["GSI_STAGE_BLOCK_ROWS"] = str(b)
    for rep in range(2):
        t0 = time.perf_counter()
        L.check(lib.gsi_randsvd_dense_host(ctx.h, L.dptr(A), n, n, n, L.dptr(Om), K, p, q, L.dptr(Z), S.ctypes.data_as(L.c_dp), None), lib)
        t1 = time.perf_counter()
        print(f"dense_host block rows {b:6d}: {t1 - t0:.3f} s   same S: {np.array_equal(S, S0)}", flush=True)

# upload-dominated: rangefinder with a 16-column sketch, q = 0 (one tiny product + QR): the row-block upload alone
Om16 = np.asfortranarray(Om[:, :16])
Q = np.zeros((n, 16), order="F")
for b in blocks + [10 ** 9]:
    os.environ["GSI_STAGE_BLOCK_ROWS"] = str(b)
    for rep in range(2):
        t0 = time.perf_counter()
        L.check(lib.gsi_rangefinder_dense_host(ctx.h, L.dptr(A), n, n, n, L.dptr(Om16), 16, 0, L.dptr(Q), None), lib)
        t1 = time.perf_counter()
        print(f"rangefinder l=16 q=0 block rows {b:10d}: {t1 - t0:.3f} s ({A.nbytes / (t1 - t0) / 1e9:.1f} GB/s)", flush=True)
