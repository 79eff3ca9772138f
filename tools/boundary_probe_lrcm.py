#!/usr/bin/env python3
"""Host boundary at the headline: gsi_randsvd with host Omega in / host Z out on the resident LowRankCovMatrix (n = 1e6, N_s =
1024, l = 320), against its parts timed one by one (upload of Omega, device-resident randsvd, download of Z).  GSI_STAGE_TRACE=1
prints what every staging worker did.   usage: python3 tools/boundary_probe_lrcm.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gsi_amd as gsi  # noqa: E402

n, Ns, K, p, q = 1000000, 1024, 256, 64, 2
l = K + p
L = gsi._lib
ctx = gsi.default_context()
lib = ctx.lib
op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=0, decay=0.75)
Om = np.asfortranarray(np.random.default_rng(0).standard_normal((n, l)))
Z = np.zeros((n, l), order="F")
S = np.zeros(l)
for rep in range(3):
    t0 = time.perf_counter()
    L.check(lib.gsi_randsvd(ctx.h, op.h, L.dptr(Om), K, p, q, L.dptr(Z), S.ctypes.data_as(L.c_dp)), lib)
    print(f"gsi_randsvd host Omega in / Z out: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    Omd = gsi.DeviceMatrix.from_host(ctx, Om)
    t1 = time.perf_counter()
    Zd = gsi.DeviceMatrix(ctx, n, l)
    Sd = gsi.DeviceMatrix(ctx, l, 1)
    L.check(lib.gsi_randsvd_dev(ctx.h, op.h, Omd.h, K, p, q, Zd.h, Sd.h), lib)
    ctx.sync()
    t2 = time.perf_counter()
    L.check(lib.gsi_mat_download(ctx.h, Zd.h, L.dptr(Z), n), lib)
    t3 = time.perf_counter()
    for h in (Omd, Zd, Sd):
        h.close()
    print(f"parts: upload {1e3 * (t1 - t0):.1f} ms ({Om.nbytes / (t1 - t0) / 1e9:.1f} GB/s), randsvd_dev {1e3 * (t2 - t1):.1f} ms, "
          f"download {1e3 * (t3 - t2):.1f} ms ({Z.nbytes / (t3 - t2) / 1e9:.1f} GB/s), sum {1e3 * (t3 - t0):.1f} ms", flush=True)
