"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (fp64 everywhere; BASELINE.json north_star: top-k singular values within 1e-5
relative of the reference, xis up to sign as test/testrpcga.jl:100 with 1e-6):
  * products / panel factors vs LAPACK:     <= 1e-12 relative (rounding-order differences only)
  * LU pivot sequence:                       identical (integer)
  * singular values:                         <= 1e-9 relative asserted (bar: 1e-5)
  * xis up to sign:                          <= 1e-6 absolute (the reference's own bar)
"""
import numpy as np
import pytest

from oracle import oracle as orc
from helpers import exact_rank_matrix, gaussian_cov, exponential_cov, powerlaw_fields, rel_sv_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gsi):
    c = gsi.default_context()
    yield c


# ---- MFMA products ----------------------------------------------------------------------
@pytest.mark.parametrize("m,k,l", [(64, 32, 16), (100, 77, 5), (257, 130, 48), (1000, 999, 160),
                                   (2000, 2000, 161), (33, 4000, 320), (5000, 64, 33)])
def test_gemm_nn_tn(gsi, ctx, m, k, l):
    rng = np.random.default_rng(m * 7 + k)
    A = rng.standard_normal((m, k))
    B = rng.standard_normal((k, l))
    C = gsi.gemm(A, B)
    ref = A @ B
    assert np.abs(C - ref).max() <= 1e-12 * np.abs(A).sum(axis=1).max() * np.abs(B).max()
    At = np.asfortranarray(A.T)                         # k x m stored, compute At' * B
    Ct = gsi.gemm(At, B, trans=True)
    assert np.abs(Ct - ref).max() <= 1e-12 * np.abs(A).sum(axis=1).max() * np.abs(B).max()


def test_gemm_exact_integer_layout(gsi, ctx):
    """asymmetric small-integer operands: any fragment-layout mistake shows up exactly."""
    m, k, l = 70, 37, 21
    A = (np.arange(m * k).reshape(m, k) % 7 - 3).astype(float)
    B = (np.arange(k * l).reshape(k, l) % 5 - 1).astype(float)
    assert np.array_equal(gsi.gemm(A, B), A @ B)
    assert np.array_equal(gsi.gemm(np.asfortranarray(A.T), B, trans=True), A @ B)


# ---- lu(Y).L --------------------------------------------------------------------------------
@pytest.mark.parametrize("m,l", [(10, 2), (40, 7), (100, 25), (625, 50), (2000, 48), (5000, 160), (3000, 33)])
def test_lu_L_matches_lapack(gsi, ctx, m, l):
    rng = np.random.default_rng(m + l)
    Y = rng.standard_normal((m, l))
    L, piv = gsi.lu_L(Y, return_pivots=True)
    assert np.array_equal(piv, orc.lu_pivots(Y)), "pivot sequence differs from dgetrf"
    Lref = orc.lu_L(Y)
    assert np.abs(L - Lref).max() < 1e-11


def test_lu_singular_raises(gsi, ctx):
    Y = np.zeros((20, 3))
    Y[:, 0] = 1.0
    with pytest.raises(gsi.GsiError) as ei:
        gsi.lu_L(Y)
    assert ei.value.code == 3           # Julia: SingularException


# ---- qr -> thin Q ------------------------------------------------------------------------------
@pytest.mark.parametrize("m,l", [(10, 2), (40, 7), (100, 25), (625, 50), (2000, 48), (5000, 160), (3000, 33)])
def test_qr_thinQ(gsi, ctx, m, l):
    rng = np.random.default_rng(3 * m + l)
    Y = rng.standard_normal((m, l)) @ np.diag(np.logspace(0, -8, l))
    Q, R = gsi.qr_thinQ(Y, return_R=True)
    assert np.abs(Q.T @ Q - np.eye(l)).max() < 1e-13
    assert np.abs(Q @ R - Y).max() < 1e-13 * np.abs(Y).max() * l
    assert np.abs(np.tril(R, -1)).max() == 0.0
    Qref = orc.qr_thinQ(Y)
    assert orc.subspace_sin(Qref, Q) < 1e-7   # same range (conditioning of Y limits the angle)


def test_qr_rank_deficient(gsi, ctx):
    rng = np.random.default_rng(9)
    Y = exact_rank_matrix(rng, 300, 5)[:, :12]        # rank 5, 12 columns
    Q = gsi.qr_thinQ(Y)
    assert np.abs(Q.T @ Q - np.eye(12)).max() < 1e-13
    assert np.linalg.norm(Y - Q @ (Q.T @ Y)) < 1e-11 * np.linalg.norm(Y)


# ---- svd(B) ------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,l", [(50, 3), (200, 16), (625, 50), (2000, 48), (4000, 160), (1500, 100)])
def test_svd_tall(gsi, ctx, n, l):
    rng = np.random.default_rng(n - l)
    W = rng.standard_normal((n, l)) @ np.diag(np.logspace(0, -6, l)) @ rng.standard_normal((l, l))
    S, V = gsi.svd_tall(W)
    Uref, Sref, _ = np.linalg.svd(W, full_matrices=False)
    assert np.all(np.diff(S) <= 0)
    assert np.abs(S - Sref).max() <= 1e-12 * Sref[0]
    assert np.abs(V.T @ V - np.eye(l)).max() < 1e-12
    for i in range(l):
        if i + 1 < l and (Sref[i] - Sref[i + 1]) < 1e-6 * Sref[0]:
            continue
        if i > 0 and (Sref[i - 1] - Sref[i]) < 1e-6 * Sref[0]:
            continue
        assert min(np.linalg.norm(V[:, i] - Uref[:, i]), np.linalg.norm(V[:, i] + Uref[:, i])) < 1e-6


# ---- rangefinder: the reference's own property tests (test/testrmf.jl:11-19) + oracle parity ----
@pytest.mark.parametrize("n,m", [(10, 2), (10, 5), (100, 5), (100, 10), (100, 25)])
def test_rangefinder_exact_rank(gsi, ctx, n, m):
    rng = np.random.default_rng(100 * n + m)
    A = exact_rank_matrix(rng, n, m)
    gsi.RandMatFact.seed(n + m)
    Q = gsi.rangefinder(A)                              # adaptive, Alg 4.2
    assert abs(Q.shape[1] - m) <= 1
    assert np.linalg.norm(A - Q @ Q.T @ A) < 1e-8
    Omega = rng.standard_normal((n, m))
    Q = gsi.rangefinder(A, m, 2, Omega=Omega)
    assert abs(Q.shape[1] - m) <= 1
    assert np.linalg.norm(A - Q @ Q.T @ A) < 1e-8
    Qref = orc.rangefinder(A, m, 2, Omega)
    assert orc.subspace_sin(Qref, Q) < 1e-9


def test_rangefinder_negative_iterations(gsi, ctx):
    A = np.eye(8)
    with pytest.raises(gsi.GsiError) as ei:
        gsi.rangefinder(A, 2, -1, Omega=np.ones((8, 2)))
    assert ei.value.code == 2
    assert "numiterations should be positive" in str(ei.value)     # RandMatFact.jl:63


@pytest.mark.parametrize("q", [0, 1, 2, 3])
def test_rangefinder_oracle_parity(gsi, ctx, q):
    A = gaussian_cov(25, 20, 4.0)                       # n = 500
    rng = np.random.default_rng(q)
    Omega = rng.standard_normal((500, 24))
    Q = gsi.rangefinder(A, 24, q, Omega=Omega)
    Qref = orc.rangefinder(A, 24, q, Omega)
    assert np.abs(Q.T @ Q - np.eye(24)).max() < 1e-12
    # compare what downstream sees: singular values of Q'A and the dominant subspace
    s = np.linalg.svd(Q.T @ A, compute_uv=False)
    sref = np.linalg.svd(Qref.T @ A, compute_uv=False)
    assert rel_sv_err(s, sref, 16) < 1e-9


# ---- randsvd / getxis ---------------------------------------------------------------------------
def test_randsvd_C1_parity(gsi, ctx):
    """BASELINE.json configs[0]: n = 2000 Gaussian covariance, K = 32, p = 16, q = 1."""
    A = gaussian_cov(50, 40, 5.0)
    rng = np.random.default_rng(0)
    K, p, q = 32, 16, 1
    Omega = rng.standard_normal((2000, K + p))
    Z, S = gsi.randsvd(A, K, p, q, Omega=Omega, return_S=True)
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Omega)
    assert rel_sv_err(S, Sref, K) < 1e-9                  # bar: 1e-5
    assert np.all(Z[:, K:] == 0.0)                        # RandMatFact.jl:87: last p columns zero
    assert orc.xis_error_up_to_sign(Z, Zref, K) < 1e-6    # test/testrpcga.jl:100
    assert np.linalg.norm(Z @ Z.T - Zref @ Zref.T) < 1e-8 * np.linalg.norm(Zref @ Zref.T)


@pytest.mark.parametrize("kind,q", [("exp", 2), ("gauss", 3)])
def test_randsvd_other_spectra(gsi, ctx, kind, q):
    A = exponential_cov(30, 30, 8.0) if kind == "exp" else gaussian_cov(30, 30, 3.0)
    rng = np.random.default_rng(5)
    K, p = 20, 10
    Omega = rng.standard_normal((900, K + p))
    Z, S = gsi.randsvd(A, K, p, q, Omega=Omega, return_S=True)
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Omega)
    assert rel_sv_err(S, Sref, K) < 1e-8
    assert orc.xis_error_up_to_sign(Z, Zref, K) < 1e-6


def test_randsvd_rank_deficient_sketch(gsi, ctx):
    """l = K + p > rank(A), as test/testrpcga.jl:107,114 produces."""
    rng = np.random.default_rng(11)
    Q0 = rng.standard_normal((8, 64))
    A = Q0.T @ Q0                                       # rank 8
    Omega = rng.standard_normal((64, 9))
    Z, S = gsi.randsvd(A, 8, 1, 3, Omega=Omega, return_S=True)
    Zref, Sref, _ = orc.randsvd_full(A, 8, 1, 3, Omega)
    assert rel_sv_err(S, Sref, 8) < 1e-9
    assert np.linalg.norm(Z @ Z.T - A) < 1e-8 * np.linalg.norm(A)


def test_eig_nystrom_kat(gsi, ctx):
    """test/testrmf.jl:21-29."""
    A = np.array([[2.0, -1, 0], [-1, 2, -1], [0, -1, 2]])
    gsi.RandMatFact.seed(3)
    Q = gsi.rangefinder(A)
    U, Sigmavec = gsi.eig_nystrom(A, Q)
    lam = Sigmavec ** 2
    assert np.linalg.norm(np.array([2 + np.sqrt(2), 2.0, 2 - np.sqrt(2)]) - lam) < 1e-8


# ---- LowRankCovMatrix ---------------------------------------------------------------------------
def test_lowrankcov_kat(gsi, ctx):
    """test/testrpcga.jl:46-58."""
    samples = [[-.5, 0., .5], [1., -1., 0.], [-.5, 1., -.5]]
    lrcm = gsi.LowRankCovMatrix(samples)
    fullcm = np.eye(3) @ lrcm
    assert np.allclose(fullcm, lrcm @ np.eye(3))
    assert np.allclose(fullcm, [[.75, -.75, 0], [-.75, 1, -.25], [0, -.25, .25]])
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = rng.standard_normal((3, 3))
        assert np.allclose(fullcm @ x, lrcm @ x)
        assert np.allclose(fullcm.T @ x, lrcm.T @ x)
    v = rng.standard_normal(3)
    assert np.allclose(fullcm @ v, lrcm @ v)
    with pytest.raises(IndexError):
        lrcm.size(3)


def test_lowrankcov_uncentred_samples_and_consistency(gsi, ctx):
    """test/testrpcga.jl:60-81 at reduced N."""
    rng = np.random.default_rng(2017)
    N, M = 2000, 100
    sqrtcov = rng.standard_normal((M, M))
    samples = (sqrtcov @ rng.standard_normal((M, N))).T + 3.0      # non-zero mean: exercises centring
    lrcm = gsi.LowRankCovMatrix(samples)
    ref = orc.LowRankCovMatrix(samples)
    full = ref.samples.T @ ref.samples / (N - 1)
    X = rng.standard_normal((M, 7))
    assert np.abs(lrcm @ X - full @ X).max() < 1e-10 * np.abs(full @ X).max()
    assert np.linalg.norm(full - sqrtcov @ sqrtcov.T, 2) < M ** 2 / np.sqrt(N) + 10 * np.sqrt(10000 / N) * 10


def test_getxis_lrcm_vs_dense_same_omega(gsi, ctx):
    """test/testrpcga.jl:83-102 on the GPU, plus parity with the oracle."""
    rng = np.random.default_rng(0)
    numfields, numxis, p, q = 100, 30, 20, 3
    fields = powerlaw_fields(rng, (25, 25), numfields)
    Omega = rng.standard_normal((625, numxis + p))
    it = iter(fields)
    lrcmxis, got_fields = gsi.getxis_iwantfields(lambda: next(it), numfields, numxis, p, q, None, Omega=Omega)
    lrcm = gsi.LowRankCovMatrix(got_fields)
    fullcm = np.eye(625) @ lrcm
    fullxis = gsi.getxis(fullcm, numxis, p, q, None, Omega=Omega)
    refxis = orc.getxis_dense(fullcm, numxis, p, q, Omega)
    for a, b, c in zip(fullxis, lrcmxis, refxis):
        assert min(np.linalg.norm(a - b), np.linalg.norm(a + b)) < 1e-6
        assert min(np.linalg.norm(a - c), np.linalg.norm(a + c)) < 1e-6


# ---- consumers -----------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,mu", [(2, 16, 10.0), (8, 64, 0.0), (16, 256, 10.0)])
def test_pcga_end_to_end(gsi, ctx, M, N, mu):
    """test/testrpcga.jl:104-131 (reduced sweep)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2017 + M + N)
    x = rng.standard_normal(N)
    Q0 = rng.standard_normal((M, N))
    Q = Q0.T @ Q0
    w, V = np.linalg.eigh(Q)
    truep = (V * np.sqrt(np.clip(w, 0, None))) @ V.T @ rng.standard_normal(N) + mu
    forward = lambda pv: pv * x
    gsi.RandMatFact.seed(M * N)
    xis = gsi.getxis(Q, M, int(round(0.1 * M)))
    X = np.full(N, float(mu))
    noise = 1e-4
    R = noise ** 2 * sp.identity(N, format="csc")
    yobs = forward(truep) + noise * rng.standard_normal(N)
    popt = gsi.pcgadirect(forward, X.copy(), X, xis, R, yobs)
    assert np.linalg.norm(popt - truep) / np.linalg.norm(truep) < 2e-2
    if M < N / 6:
        popt = gsi.pcgalsqr(forward, X.copy(), X, xis, R, yobs)
        assert np.linalg.norm(popt - truep) / np.linalg.norm(truep) < 2e-2
