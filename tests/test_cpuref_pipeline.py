"""pipeline.cpp + api.cpp (the code that ships in libgsi_hip.so) driven through the C ABI on the
CPU reference backend, against the scipy oracle and the golden vectors.  This checks the host-side
order of operations, argument checking and error mapping without a GPU."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from helpers import gaussian_cov, exact_rank_matrix, powerlaw_fields, rel_sv_err
import cpuref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def cx(gsi):
    lib = cpuref.load_cpuref()
    assert lib.gsi_backend_name().startswith(b"cpu-reference")
    c = gsi.Context(0, lib=lib)
    yield c
    c.close()


def test_golden_dense(gsi, cx):
    g = np.load(os.path.join(GOLD, "dense_gauss_n192.npz"))
    A = gaussian_cov(int(g["grid"][0]), int(g["grid"][1]), float(g["ell"]))
    K, p, q = int(g["K"]), int(g["p"]), int(g["q"])
    Z, S = gsi.randsvd(A, K, p, q, Omega=g["Omega"], return_S=True, ctx=cx)
    assert rel_sv_err(S, g["S"], K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, g["Z"], K) < 1e-6
    assert np.all(Z[:, K:] == 0)
    L, piv = gsi.lu_L(A @ g["Omega"], return_pivots=True, ctx=cx)
    assert np.array_equal(piv, g["lu_pivots"])
    assert np.abs(L - g["lu_L"]).max() < 1e-11


@pytest.mark.parametrize("q", [0, 3])
def test_golden_exp(gsi, cx, q):
    from helpers import exponential_cov
    g = np.load(os.path.join(GOLD, "dense_exp_n144.npz"))
    A = exponential_cov(12, 12, float(g["ell"]))
    Z, S = gsi.randsvd(A, 7, 3, q, Omega=g["Omega"], return_S=True, ctx=cx)
    assert rel_sv_err(S, g[f"S_q{q}"], 7) < 1e-9
    assert orc.xis_error_up_to_sign(Z, g[f"Z_q{q}"], 7) < 1e-6


def test_golden_lowrank(gsi, cx):
    g = np.load(os.path.join(GOLD, "lowrank_n100_N24.npz"))
    lrcm = gsi.LowRankCovMatrix(g["fields"], ctx=cx)
    assert np.abs(lrcm.todense() - g["dense"]).max() < 1e-12 * np.abs(g["dense"]).max()
    Z = gsi.randsvd(lrcm, int(g["K"]), int(g["p"]), int(g["q"]), Omega=g["Omega"])
    assert orc.xis_error_up_to_sign(Z, g["xis"].T, int(g["K"])) < 1e-6
    lrcm.close()


def test_lowrank_samples_field(gsi, cx):
    """`A.samples` (lowrank.jl:14-16, 25-27) comes back mean-removed, sample i = row i; the oracle's GEMM-form products
    (bench.py's full-size parity leg) agree with the reference's ger! loop."""
    rng = np.random.default_rng(5)
    fields = rng.standard_normal((9, 37)) + 3.0
    lrcm = gsi.LowRankCovMatrix(fields, ctx=cx)
    ref = orc.LowRankCovMatrix(fields)
    assert np.abs(lrcm.samples - ref.samples).max() < 1e-14
    with pytest.raises(gsi.GsiError):
        gsi.device_samples(gsi.dense_operator(cx, np.eye(4)), 3)
    X = rng.standard_normal((37, 5))
    assert np.abs(orc.LowRankCovMatrix(fields, gemm_form=True).matmul(X) - ref.matmul(X)).max() < 1e-13
    lrcm.close()


@pytest.mark.parametrize("n,m", [(10, 2), (100, 10), (100, 25)])
def test_rangefinders_exact_rank(gsi, cx, n, m):
    """test/testrmf.jl:11-19 through the C ABI."""
    rng = np.random.default_rng(n * m)
    A = exact_rank_matrix(rng, n, m)
    gsi.RandMatFact.seed(1)
    Q = gsi.rangefinder(A, ctx=cx)
    assert abs(Q.shape[1] - m) <= 1 and np.linalg.norm(A - Q @ Q.T @ A) < 1e-8
    Q = gsi.rangefinder(A, m, 2, Omega=rng.standard_normal((n, m)), ctx=cx)
    assert np.linalg.norm(A - Q @ Q.T @ A) < 1e-8


def test_nystrom_kat(gsi, cx):
    g = np.load(os.path.join(GOLD, "nystrom_kat.npz"))
    gsi.RandMatFact.seed(2)
    Q = gsi.rangefinder(g["A"], ctx=cx)
    U, Sig = gsi.eig_nystrom(g["A"], Q, ctx=cx)
    assert np.linalg.norm(Sig ** 2 - g["eigenvalues"]) < 1e-8


def test_error_mapping(gsi, cx):
    with pytest.raises(gsi.GsiError) as ei:
        gsi.rangefinder(np.eye(6), 2, -3, Omega=np.ones((6, 2)), ctx=cx)
    assert ei.value.code == 2 and "numiterations=-3" in str(ei.value)
    with pytest.raises(gsi.GsiError) as ei:
        gsi.lu_L(np.zeros((5, 2)), ctx=cx)
    assert ei.value.code == 3
    with pytest.raises(gsi.GsiError) as ei:
        gsi.rangefinder(np.eye(6), 9, 1, Omega=np.ones((6, 9)), ctx=cx)
    assert ei.value.code == 1
    with pytest.raises(gsi.GsiError) as ei:
        gsi.eig_nystrom(-np.eye(4), np.eye(4)[:, :2], ctx=cx)       # not positive definite
    assert ei.value.code == 7
    with pytest.raises(ValueError):
        gsi.randsvd(np.eye(6), 2, 1, 1, Omega=np.ones((5, 3)), ctx=cx)


def test_operator_products_and_sizes(gsi, cx):
    rng = np.random.default_rng(8)
    A = rng.standard_normal((30, 20))
    op = gsi.dense_operator(cx, A)
    assert op.shape == (30, 20) and op.size(1) == 30 and op.size(2) == 20
    X = rng.standard_normal((20, 4))
    assert np.allclose(op.matmul(X), A @ X)
    Y = rng.standard_normal((30, 3))
    assert np.allclose(op.rmatmul_t(Y), A.T @ Y)
    with pytest.raises(IndexError):
        op.size(3)
    op.close()


def test_pcga_consumers(gsi, cx):
    import scipy.sparse as sp
    rng = np.random.default_rng(31)
    M, N, mu = 8, 64, 10.0
    x = rng.standard_normal(N)
    Q0 = rng.standard_normal((M, N))
    Qc = Q0.T @ Q0
    w, V = np.linalg.eigh(Qc)
    truep = (V * np.sqrt(np.clip(w, 0, None))) @ V.T @ rng.standard_normal(N) + mu
    forward = lambda pv: pv * x
    Om = rng.standard_normal((N, M + 1))
    Z = gsi.randsvd(Qc, M, 1, 3, Omega=Om, ctx=cx)
    xis = [np.ascontiguousarray(Z[:, i]) for i in range(M)]
    X = np.full(N, mu)
    R = 1e-8 * sp.identity(N, format="csc")
    y = forward(truep) + 1e-4 * rng.standard_normal(N)
    got = gsi.pcgadirect(forward, X.copy(), X, xis, R, y, ctx=cx)
    ref = orc.pcgadirect(forward, X.copy(), X, xis, R, y)
    assert np.linalg.norm(got - truep) / np.linalg.norm(truep) < 2e-2
    assert np.linalg.norm(got - ref) < 1e-6 * np.linalg.norm(ref)
    got = gsi.pcgalsqr(forward, X.copy(), X, xis, R, y, ctx=cx)
    assert np.linalg.norm(got - truep) / np.linalg.norm(truep) < 2e-2


def test_device_resident_basis(gsi, cx):
    """the xi-basis stays on the device between getxis and the PCGA iterations."""
    import scipy.sparse as sp
    rng = np.random.default_rng(41)
    M, N, mu = 8, 96, 3.0
    x = rng.standard_normal(N)
    Q0 = rng.standard_normal((M, N))
    Qc = Q0.T @ Q0
    w, V = np.linalg.eigh(Qc)
    truep = (V * np.sqrt(np.clip(w, 0, None))) @ V.T @ rng.standard_normal(N) + mu
    forward = lambda pv: pv * x
    Om = rng.standard_normal((N, M + 1))
    basis = gsi.getxis_device(Qc, M, 1, 3, Omega=Om, ctx=cx)
    xis_host = gsi.getxis(Qc, M, 1, 3, Omega=Om) if False else [basis[i] for i in range(M)]
    Zref = orc.randsvd(Qc, M, 1, 3, Om)
    assert orc.xis_error_up_to_sign(np.stack(xis_host, axis=1), Zref, M) < 1e-6
    X = np.full(N, mu)
    R = 1e-8 * sp.identity(N, format="csc")
    y = forward(truep) + 1e-4 * rng.standard_normal(N)
    got_dev = gsi.pcgadirect(forward, X.copy(), X, basis, R, y)
    got_host = gsi.pcgadirect(forward, X.copy(), X, xis_host, R, y, ctx=cx)
    assert np.linalg.norm(got_dev - got_host) < 1e-9 * np.linalg.norm(got_host)
    assert np.linalg.norm(got_dev - truep) / np.linalg.norm(truep) < 2e-2
    got_lsqr = gsi.pcgalsqr(forward, X.copy(), X, basis, R, y)
    assert np.linalg.norm(got_lsqr - truep) / np.linalg.norm(truep) < 2e-2
    S = rng.standard_normal((48, N)) / np.sqrt(N)
    got_rga = gsi.rga(forward, X.copy(), X, basis, R, y, S)
    assert got_rga.shape == (N,)


def test_implicit_gridcov_operator(gsi, cx):
    """gsi_op_gridcov_implicit: generated products equal the stored Gaussian covariance (implicit)."""
    nx, ny, ell = 11, 6, 2.0
    G = gaussian_cov(nx, ny, ell)
    op = gsi.gridcov_implicit_operator(cx, nx, ny, ell)
    assert op.shape == (nx * ny, nx * ny)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((nx * ny, 7))
    assert np.abs(op.matmul(X) - G @ X).max() < 1e-12
    assert np.abs(op.rmatmul_t(X) - G.T @ X).max() < 1e-12
    Om = rng.standard_normal((nx * ny, 12))
    Z, S = gsi.randsvd(op, 8, 4, 2, Omega=Om, return_S=True)
    Zr, Sr, _ = orc.randsvd_full(G, 8, 4, 2, Om)
    assert rel_sv_err(S, Sr, 8) < 1e-9
    assert orc.xis_error_up_to_sign(Z, Zr, 8) < 1e-6
    op.close()
    with pytest.raises(gsi.GsiError):
        gsi.gridcov_implicit_operator(cx, 4, 4, -1.0)


def test_randsvd_shape_sweep_cpuref(gsi, cx):
    """The same seeded shape sweep as the GPU suite, through pipeline.cpp on the CPU reference backend."""
    from helpers import random_shape_cases, decaying_matrix
    rng = np.random.default_rng(2024)
    for (m, n, K, p, q, decay) in random_shape_cases(11, 60, 60):
        A = decaying_matrix(rng, m, n, decay)
        Om = rng.standard_normal((n, K + p))
        Z, S = gsi.randsvd(A, K, p, q, Omega=Om, return_S=True, ctx=cx)
        Zr, Sr, _ = orc.randsvd_full(A, K, p, q, Om)
        assert np.all(Z[:, K:] == 0)
        assert np.abs(S - Sr).max() / Sr[0] < 1e-10, (m, n, K, p, q)
        assert np.abs(Z @ Z.T - Zr @ Zr.T).max() / Sr[0] < 1e-8, (m, n, K, p, q)


@pytest.mark.parametrize("Ns,beta", [((12,), -2.0), ((6, 5), -3.5), ((4, 3, 5), -3.0)])
def test_fft_powerlaw_operator_cpuref(gsi, cx, Ns, beta):
    """gsi_op_fft_powerlaw through the shipped pipeline (naive DFT backend) against the numpy-FFT oracle: products,
    symmetry, unit diagonal, and a randsvd through the matrix-free operator."""
    n = int(np.prod(Ns))
    op = gsi.fft_powerlaw_operator(cx, Ns, beta)
    assert op.shape == (n, n)
    A = op.matmul(np.eye(n))
    Aref = orc.fft_powerlaw_apply(np.eye(n), list(Ns), beta)
    assert np.abs(A - Aref).max() < 1e-12
    assert np.abs(A - A.T).max() < 1e-12 and np.abs(np.diag(A) - 1.0).max() < 1e-12
    assert np.linalg.eigvalsh(0.5 * (A + A.T)).min() > -1e-12
    rng = np.random.default_rng(n)
    K, p = min(6, n - 3), 2
    Om = rng.standard_normal((n, K + p))
    Z, S = gsi.randsvd(op, K, p, 2, Omega=Om, return_S=True)
    Zr, Sr, _ = orc.randsvd_full(Aref, K, p, 2, Om)
    assert rel_sv_err(S, Sr, K) < 1e-9
    assert np.abs(Z @ Z.T - Zr @ Zr.T).max() < 1e-8 * Sr[0]
    op.close()


# ---- LSQR consumers, PCGALowRankMatrix, fp32 basis through the shipped pipeline ----------
def _pcga_problem(rng, nobs, K):
    etas = [rng.standard_normal(nobs) for _ in range(K)]
    HX = rng.standard_normal(nobs)
    import scipy.sparse as sp
    R = 1e-2 * sp.identity(nobs, format="csc")
    return etas, HX, R


def test_lowrank_solve_cpuref(gsi, cx):
    """`\\(A::LowRankCovMatrix, b)` (lowrank.jl:141-144) = lsqr with maxiter = N: same iterate as the oracle's."""
    rng = np.random.default_rng(11)
    fields = powerlaw_fields(rng, (7, 6), 12)
    lr = gsi.LowRankCovMatrix(fields, ctx=cx)
    ref = orc.LowRankCovMatrix(fields)
    b = ref.matmul(rng.standard_normal(42))            # in range(A): a consistent system
    x, it = lr.solve(b, return_iterations=True)
    xr = ref.solve(b)
    assert it <= 12
    # the Krylov iterates of the two products (two GEMMs here, N rank-1 updates in the oracle) agree to rounding
    # amplified by the operator's condition number; the iteration stops on maxiter, not on convergence
    assert np.linalg.norm(x - xr) < 1e-3 * np.linalg.norm(xr)
    assert np.linalg.norm(ref.matmul(x) - b) < 1e-3 * np.linalg.norm(b)
    lr.close()


def test_pcga_lowrank_matrix_cpuref(gsi, cx):
    rng = np.random.default_rng(12)
    nobs, K = 23, 5
    etas, HX, R = _pcga_problem(rng, nobs, K)
    A = gsi.PCGALowRankMatrix(etas, HX, R, ctx=cx)
    ref = orc.PCGALowRankMatrix(etas, HX, R)
    assert A.shape == (nobs + 1, nobs + 1) and A.size(1) == nobs + 1
    with pytest.raises(IndexError):
        A.size(3)
    x = rng.standard_normal(nobs + 1)
    assert np.abs(A.matvec(x) - ref.matvec(x)).max() < 1e-12
    # dense R takes the other code path
    Rd = R.toarray() + 1e-3 * np.diag(rng.random(nobs))
    A2 = gsi.PCGALowRankMatrix(etas, HX, Rd, ctx=cx)
    assert np.abs(A2.matvec(x) - orc.PCGALowRankMatrix(etas, HX, Rd).matvec(x)).max() < 1e-12
    b = np.concatenate([rng.standard_normal(nobs), [0.0]])
    sol, it = A.lsqr(b, return_iterations=True)
    solr, itr = orc.lsqr(ref.matvec, ref.matvec, b, nobs + 1)
    assert it == itr
    assert np.linalg.norm(sol - solr) < 1e-7 * np.linalg.norm(solr)     # both stop at atol = btol = sqrt(eps)
    A.close(); A2.close()


def test_fp32_basis_cpuref(gsi, cx):
    """fp32-stored xi-basis: params/update equal the fp64 formulas on the ROUNDED basis to fp64 accuracy, and the
    fp64 results to fp32 accuracy."""
    rng = np.random.default_rng(13)
    n, K, nobs = 97, 6, 15
    Zh = rng.standard_normal((n, K + 2))
    Zd = gsi.DeviceMatrix.from_host(cx, Zh)
    b64 = gsi.DeviceBasis(Zd, K)
    b32 = gsi.DeviceBasis(Zd, K, precision=32)
    s, X = rng.standard_normal(n), rng.standard_normal(n)
    Z32 = Zh[:, :K].astype(np.float32).astype(np.float64)
    P32 = b32.params(s, X, 1e-3)
    assert np.abs(P32[:, :K] - (s[:, None] + 1e-3 * Z32)).max() < 1e-15
    assert np.abs(P32 - b64.params(s, X, 1e-3)).max() < 1e-3 * 1e-6
    etas = [rng.standard_normal(nobs) for _ in range(K)]
    xb = rng.standard_normal(nobs)
    w = np.array([e @ xb for e in etas])
    u32 = b32.update(X, 0.7, etas, xb)
    assert np.abs(u32 - (0.7 * X + Z32 @ w)).max() < 1e-12
    assert np.abs(u32 - b64.update(X, 0.7, etas, xb)).max() < 1e-5
    assert np.abs(b32[2] - Z32[:, 2]).max() == 0.0 and np.abs(b64[2] - Zh[:, 2]).max() == 0.0
    b32.close(); b64.close(); Zd.close()


def test_implicit_exponential_and_table_operator(gsi, cx):
    """gsi_op_gridcov_implicit_kind(kind=1) / _table: the generated operator equals the stored exponential covariance
     and an arbitrary stationary kernel given as a table over grid offsets."""
    from helpers import exponential_cov, grid_points
    nx, ny, ell = 9, 7, 3.0
    G = exponential_cov(nx, ny, ell)
    op = gsi.gridcov_implicit_operator(cx, nx, ny, ell, kind=1)
    rng = np.random.default_rng(6)
    X = rng.standard_normal((nx * ny, 5))
    assert np.abs(op.matmul(X) - G @ X).max() < 1e-12
    assert np.abs(op.rmatmul_t(X) - G @ X).max() < 1e-12
    op.close()
    # anisotropic Matern-3/2-like table
    dx = np.arange(nx)[:, None]; dy = np.arange(ny)[None, :]
    r = np.sqrt((dx / 2.0) ** 2 + (dy / 5.0) ** 2)
    T = (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
    P = grid_points(nx, ny)
    D = np.abs(P[:, None, :] - P[None, :, :]).astype(int)
    Gt = T[D[:, :, 0], D[:, :, 1]]
    op = gsi.gridcov_implicit_operator(cx, nx, ny, 1.0, table=T)
    assert np.abs(op.matmul(X) - Gt @ X).max() < 1e-12
    Om = rng.standard_normal((nx * ny, 10))
    Z, S = gsi.randsvd(op, 7, 3, 2, Omega=Om, return_S=True)
    Zr, Sr, _ = orc.randsvd_full(Gt, 7, 3, 2, Om)
    assert rel_sv_err(S, Sr, 7) < 1e-9
    op.close()
    with pytest.raises(gsi.GsiError):
        gsi.gridcov_implicit_operator(cx, 4, 4, 1.0, kind=2)


@pytest.mark.parametrize("m,l", [(10, 2), (40, 7), (100, 25), (625, 50), (300, 70), (97, 33)])
def test_lu_sharded_single_rank_equals_lu(gsi, cx, m, l):
    """The row-sharded LU (pipeline.cpp:lu_panel_sharded) on one rank: same pivots as dgetrf, L equal to the single-rank
    factorization bit for bit."""
    import scipy.linalg as sl
    rng = np.random.default_rng(m * l)
    Y = rng.standard_normal((m, l))
    Ls, ps = gsi.lu_L_sharded(Y, return_pivots=True, ctx=cx)
    L1, p1 = gsi.lu_L(Y, return_pivots=True, ctx=cx)
    assert np.array_equal(ps, p1)
    assert np.array_equal(ps, orc.lu_pivots(Y))
    assert np.array_equal(Ls, L1)


@pytest.mark.parametrize("m,l,G", [(100, 25, 2), (625, 50, 3), (3000, 136, 4), (9, 2, 4), (1000, 100, 7)])
def test_lu_sharded_virtual_ranks(gsi, cx, m, l, G):
    """gsi_lu_L_sharded_virtual (pipeline.cpp:lu_panel_sharded_virtual) on the CPU reference backend: G virtual ranks with
    real row offsets give dgetrf's pivots and the single-rank L exactly."""
    rng = np.random.default_rng(m + l + G)
    Y = rng.standard_normal((m, l))
    pad = -(-m // G)
    if m == 3000:
        Y[pad + 10:pad + 110] = Y[100:200]
    Lv, pv = gsi.lu_L_sharded_virtual(Y, G, return_pivots=True, ctx=cx)
    L1, p1 = gsi.lu_L(Y, return_pivots=True, ctx=cx)
    assert np.array_equal(pv, orc.lu_pivots(Y)) and np.array_equal(pv, p1)
    assert np.array_equal(Lv, L1)


@pytest.mark.parametrize("d,kind", [(1, "gaussian"), (2, "exponential"), (3, "matern32"), (2, "matern52")])
def test_pointcov_implicit_cpuref(gsi, cx, d, kind):
    """gsi_op_pointcov_implicit through api.cpp / pipeline.cpp on the CPU reference backend (pointcov.hpp's kernel
    definitions, shared with the device generator) against the dense kernel matrix numpy builds from the coordinates."""
    rng = np.random.default_rng(17 + d)
    n = 90
    P = rng.uniform(0.0, 10.0, size=(d, n))
    r = np.sqrt(((P[:, :, None] - P[:, None, :]) ** 2).sum(axis=0)) / 3.0
    Kd = {"gaussian": np.exp(-0.5 * r * r), "exponential": np.exp(-r),
          "matern32": (1 + np.sqrt(3) * r) * np.exp(-np.sqrt(3) * r),
          "matern52": (1 + np.sqrt(5) * r + 5 * r * r / 3) * np.exp(-np.sqrt(5) * r)}[kind]
    A = 1.5 * Kd + 0.2 * np.eye(n)
    op = gsi.pointcov_implicit_operator(cx, P, kind, ell=3.0, sigma2=1.5, nugget=0.2)
    X = rng.standard_normal((n, 6))
    assert np.abs(op.matmul(X) - A @ X).max() < 1e-12 * np.abs(A @ X).max()
    assert np.abs(op.rmatmul_t(X) - A @ X).max() < 1e-12 * np.abs(A @ X).max()
    Om = rng.standard_normal((n, 14))
    Z, S = gsi.randsvd(op, 10, 4, 2, Omega=Om, return_S=True)
    Zr, Sr, _ = orc.randsvd_full(A, 10, 4, 2, Om)
    assert rel_sv_err(S, Sr, 10) < 1e-9 and orc.xis_error_up_to_sign(Z, Zr, 10) < 1e-6
    op.close()
    with pytest.raises(gsi.GsiError):
        gsi.pointcov_implicit_operator(cx, rng.uniform(size=(4, 10)).reshape(4, 10)[:3], 9)


def test_basis_outlives_its_matrix(gsi, cx):
    """a 64-bit gsi_basis points into the gsi_mat's buffer -- it shares ownership, so destroying the matrix
    first is safe (under `make -C oracle asan` + tools/asan_cpu_suite.sh this is a use-after-free check)."""
    rng = np.random.default_rng(3)
    Zh = np.asfortranarray(rng.standard_normal((50, 6)))
    Zm = gsi.DeviceMatrix.from_host(cx, Zh)
    basis = gsi.DeviceBasis(Zm, 4)
    Zm.close()                                        # gsi_mat_destroy before the basis is used
    s0, X0 = rng.standard_normal(50), rng.standard_normal(50)
    P = basis.params(s0, X0, 0.5)
    assert np.abs(P[:, :4] - (s0[:, None] + 0.5 * Zh[:, :4])).max() < 1e-14
    assert np.abs(basis[2] - Zh[:, 2]).max() == 0.0
    basis.close()
