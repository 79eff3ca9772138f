#!/usr/bin/env python3
"""What a FRESH result array costs at the host boundary.  The headline's gsi_randsvd with host Omega in / host Z out (n = 1e6,
N_s = 1024, l = 320: 2.56 GB each way) into (a) an array that has been written before, (b) a fresh np.empty -- what the Julia
shim's Matrix{Float64}(undef, n, l) is -- and the product-only entry point gsi_op_mul and the plain gsi_mat_download the same
way; allocation and release of the array stay outside the timed call.  profiles/r05_prefault_probe.log holds three runs of this
script against a build that faulted the result's pages in behind the computation (GSI_PREFAULT on / off / with MADV_HUGEPAGE):
a fresh array costs 2-5 ms, the prefault cost 65-75 -- not kept (tools/rejected_kernels/host_prefault.hpp.txt).
usage: python3 tools/prefault_probe.py"""
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
for f in ("enabled", "defrag", "shmem_enabled"):
    try:
        print(f"transparent_hugepage/{f}: {open('/sys/kernel/mm/transparent_hugepage/' + f).read().strip()}")
    except OSError:
        pass
op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=0, decay=0.75)
Om = np.asfortranarray(np.random.default_rng(0).standard_normal((n, l)))
S = np.zeros(l)
Zold = np.zeros((n, l), order="F")


def randsvd_into(Z):
    L.check(lib.gsi_randsvd(ctx.h, op.h, L.dptr(Om), K, p, q, L.dptr(Z), S.ctypes.data_as(L.c_dp)), lib)


def mul_into(Y):
    L.check(lib.gsi_op_mul(ctx.h, op.h, 0, L.dptr(Om), n, l, L.dptr(Y), n), lib)


def timed(f, fresh):
    Z = np.empty((n, l), order="F") if fresh else Zold
    t0 = time.perf_counter()
    f(Z)
    return 1e3 * (time.perf_counter() - t0), Z


randsvd_into(Zold)                                               # warm: workspaces, staging ring
ref = Zold.copy()
for name, f in (("gsi_randsvd (host Omega in, Z out)", randsvd_into), ("gsi_op_mul (host X in, Y out)", mul_into)):
    f(Zold)
    want = Zold.copy()
    for fresh in (False, True, False, True, True):
        ms, Z = timed(f, fresh)
        same = bool(np.array_equal(Z, want))
        print(f"{name}: {'fresh np.empty' if fresh else 'written before'}: {ms:.1f} ms, result identical to the first call: {same}", flush=True)
        assert same
        del Z
Zd = gsi.DeviceMatrix.from_host(ctx, ref)
for fresh in (False, True, False, True, True):
    Z = np.empty((n, l), order="F") if fresh else Zold
    t0 = time.perf_counter()
    L.check(lib.gsi_mat_download(ctx.h, Zd.h, L.dptr(Z), n), lib)
    ms = 1e3 * (time.perf_counter() - t0)
    print(f"gsi_mat_download 2.56 GB: {'fresh np.empty' if fresh else 'written before'}: {ms:.1f} ms ({Z.nbytes / ms / 1e6:.1f} GB/s), equal: {bool(np.array_equal(Z, ref))}", flush=True)
    del Z
