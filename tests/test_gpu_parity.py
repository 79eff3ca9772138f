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
                                   (2000, 2000, 161), (33, 4000, 320), (5000, 64, 33),
                                   (1025, 777, 150), (513, 2049, 100), (4099, 1283, 37)])
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


@pytest.mark.parametrize("m,k,l", [(70001, 96, 320), (66000, 130, 160), (131072, 33, 320)])
def test_gemm_persistent_mode(gsi, ctx, m, k, l):
    """Short reductions with more than two rounds of output tiles take the contraction kernel's PERSISTENT mode (one workgroup
    per CU walking the tiles, the next tile's operands requested before the result stores): a ragged last row block, one and
    two column chunks, K not a multiple of the tile depth -- both forms, against numpy."""
    rng = np.random.default_rng(m + k)
    A = np.asfortranarray(rng.standard_normal((m, k)))
    B = np.asfortranarray(rng.standard_normal((k, l)))
    ref = A @ B
    bound = 1e-12 * np.abs(A).sum(axis=1).max() * np.abs(B).max()
    assert np.abs(gsi.gemm(A, B) - ref).max() <= bound
    At = np.asfortranarray(A.T)                                   # k x m stored: the transposed-operand instantiation
    assert np.abs(gsi.gemm(At, B, trans=True) - ref).max() <= bound


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


@pytest.mark.parametrize("m", [40, 300, 1000, 4100, 70001])
def test_lu_pivot_ties_exact(gsi, ctx, m):
    """Dyadic entries (+-1, +-1/2), two columns: both elimination steps are exact in any operation order and both
    columns are full of equal-magnitude candidates spread over all workgroups; the pivot must be the FIRST
    maximal row, as idamax/dgetrf picks it.  Bit-exact pivots and bit-exact L."""
    rng = np.random.default_rng(m)
    Y = rng.choice([-1.0, 1.0], (m, 2)) * 2.0 ** -rng.integers(0, 2, (m, 2))
    Y[: m // 3, 0] *= 0.5                      # the first maximal row of column 0 is not row 0
    L, got = gsi.lu_L(Y, return_pivots=True)
    assert np.array_equal(got, orc.lu_pivots(Y))
    assert np.array_equal(L, orc.lu_L(Y))


@pytest.mark.parametrize("h,l", [(20, 7), (333, 40), (2500, 160), (1111, 33)])
def test_lu_pivot_ties_duplicate_rows(gsi, ctx, h, l):
    """Y = [R; R]: at every step the maximum is attained by two bitwise-identical rows (whatever the rounding of
    the previous updates), so every pivot decision is a tie between a row and its copy: lowest index wins."""
    rng = np.random.default_rng(h + l)
    R = rng.standard_normal((h, l))
    Y = np.vstack([R, R])
    L, got = gsi.lu_L(Y, return_pivots=True)
    assert np.array_equal(got, orc.lu_pivots(Y))
    assert np.abs(L - orc.lu_L(Y)).max() < 1e-11


def test_lu_singular_raises(gsi, ctx):
    Y = np.zeros((20, 3))
    Y[:, 0] = 1.0
    with pytest.raises(gsi.GsiError) as ei:
        gsi.lu_L(Y)
    assert ei.value.code == 3           # Julia: SingularException


# ---- qr -> thin Q ------------------------------------------------------------------------------
@pytest.mark.parametrize("m,l", [(10, 2), (40, 7), (100, 25), (625, 50), (2000, 48), (5000, 160), (3000, 33),
                                 (20000, 320), (9001, 250), (4100, 300), (70001, 130), (4096, 17)])   # >= 4096 rows: syrk_f64.hip
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


@pytest.mark.parametrize("n,l", [(4000, 160), (6000, 320)])
def test_svd_tall_clustered_spectrum(gsi, ctx, n, l):
    """Clusters of (nearly) equal singular values -- where the activity-driven sweeps' flag threshold (4 x the rotation
    threshold: pairs coupled below 4 sqrt(l) eps may stay unrotated) matters most for the VECTORS: the singular values must
    still be dgesdd's, V orthonormal, every cluster's subspace the right one, and the factorization must reconstruct W."""
    rng = np.random.default_rng(l)
    U0, _ = np.linalg.qr(rng.standard_normal((n, l)))
    V0, _ = np.linalg.qr(rng.standard_normal((l, l)))
    sv = np.repeat(np.logspace(0, -5, l // 16), 16) * (1.0 + 1e-13 * rng.standard_normal(l))   # clusters of 16, equal to 1e-13
    sv = np.sort(sv)[::-1]
    W = (U0 * sv) @ V0.T
    S, V = gsi.svd_tall(W)
    Sref = np.linalg.svd(W, compute_uv=False)
    assert np.abs(S - Sref).max() <= 1e-12 * Sref[0]
    assert np.abs(V.T @ V - np.eye(l)).max() < 1e-11
    for c in range(l // 16):                                  # each cluster's left subspace against the construction's
        Vc, Uc = V[:, 16 * c:16 * c + 16], U0[:, 16 * c:16 * c + 16]
        assert np.linalg.norm(Vc - Uc @ (Uc.T @ Vc)) < 1e-6, c
    R = V.T @ W                                               # rows of V' W have norms S: W = V diag(S) (right vectors)'
    assert np.abs(np.linalg.norm(R, axis=1) - S).max() <= 1e-11 * S[0]
    assert np.linalg.norm(W - V @ R) <= 1e-11 * np.linalg.norm(W)


def test_svd_activity_driven_and_plain_sweeps_agree(gsi):
    """The small SVD's default form (activity flags after every sweep, host-built schedule of the active block pairs) against the
    plain full sweeps (GSI_SVD_PLAIN=1): the same singular values to rounding, each in a process of its own (the switch is read
    once)."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gsi_amd as gsi
ctx = gsi.Context(0)
out = {}
for l, decay in [(320, 1.0), (160, 2.0), (96, 0.5), (33, 1.0)]:
    rng = np.random.default_rng(l)
    n = 3 * l + 11
    W = (rng.standard_normal((n, l)) * (np.arange(1, l + 1.0) ** -decay)) @ rng.standard_normal((l, l))
    S, V = gsi.svd_tall(W, ctx=ctx)
    ref = np.linalg.svd(W, compute_uv=False)
    assert np.max(np.abs(S - ref)) < 1e-12 * ref[0], (l, np.max(np.abs(S - ref)) / ref[0])
    assert np.abs(V.T @ V - np.eye(l)).max() < 1e-11
    out["S%d" % l] = S
np.savez(sys.argv[1], **out)
print("svd-ok")
"""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("activity", {}), ("plain", {"GSI_SVD_PLAIN": "1"})):
            env = dict(os.environ)
            env.update(extra)
            path = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=env, cwd=here)
            assert r.returncode == 0 and "svd-ok" in r.stdout, tag + ": " + r.stdout[-2000:] + r.stderr[-4000:]
            res[tag] = dict(np.load(path))
    for k in res["activity"]:
        assert np.max(np.abs(res["plain"][k] - res["activity"][k])) < 1e-12 * res["activity"][k][0], k


@pytest.mark.parametrize("n,l", [(1400, 1300), (2800, 2600)])
def test_svd_tall_very_wide(gsi, ctx, n, l):
    """Sketch widths beyond the 16- and 8-column LDS blockings of the Jacobi kernel (4 / 2 columns per block)."""
    rng = np.random.default_rng(l)
    W = rng.standard_normal((n, l)) * np.logspace(0, -3, l)[None, :]
    S, V = gsi.svd_tall(W)
    Sref = np.linalg.svd(W, compute_uv=False)
    assert np.all(np.diff(S) <= 0)
    assert np.abs(S - Sref).max() <= 1e-12 * Sref[0]
    assert np.abs(V.T @ V - np.eye(l)).max() < 1e-11


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


# ---- committed golden vectors (tests/golden/, generated by make_golden.py from the scipy oracle) ----
import os as _os
_GOLD = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


def test_golden_dense_gauss(gsi, ctx):
    g = np.load(_os.path.join(_GOLD, "dense_gauss_n192.npz"))
    A = gaussian_cov(int(g["grid"][0]), int(g["grid"][1]), float(g["ell"]))
    K, p, q = int(g["K"]), int(g["p"]), int(g["q"])
    Z, S = gsi.randsvd(A, K, p, q, Omega=g["Omega"], return_S=True)
    assert rel_sv_err(S, g["S"], K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, g["Z"], K) < 1e-6
    L, piv = gsi.lu_L(A @ g["Omega"], return_pivots=True)
    assert np.array_equal(piv, g["lu_pivots"])
    assert np.abs(L - g["lu_L"]).max() < 1e-11


@pytest.mark.parametrize("q", [0, 3])
def test_golden_dense_exp(gsi, ctx, q):
    g = np.load(_os.path.join(_GOLD, "dense_exp_n144.npz"))
    A = exponential_cov(12, 12, float(g["ell"]))
    Z, S = gsi.randsvd(A, 7, 3, q, Omega=g["Omega"], return_S=True)
    assert rel_sv_err(S, g[f"S_q{q}"], 7) < 1e-9
    assert orc.xis_error_up_to_sign(Z, g[f"Z_q{q}"], 7) < 1e-6


def test_golden_lowrank(gsi, ctx):
    g = np.load(_os.path.join(_GOLD, "lowrank_n100_N24.npz"))
    lrcm = gsi.LowRankCovMatrix(g["fields"])
    assert np.abs(lrcm.todense() - g["dense"]).max() < 1e-12 * np.abs(g["dense"]).max()
    Z = gsi.randsvd(lrcm, int(g["K"]), int(g["p"]), int(g["q"]), Omega=g["Omega"])
    assert orc.xis_error_up_to_sign(Z, g["xis"].T, int(g["K"])) < 1e-6


# ---- BASELINE.json configs[1] at full size: size-independent properties -----------------------------
def test_full_size_C2_properties(gsi, ctx):
    """n = 65536 dense fp64 covariance resident in HBM (34 GB), K = 128, p = 32, q = 2.  The oracle
    cannot run at this size in seconds, so check what must hold at any size:
      Z'Z = diag(S[:K]) (Z = V sqrt(S) with orthonormal V), trailing p columns zero, S descending,
      A v_i = s_i v_i for the leading vectors (A symmetric PSD), and device-generated A rows match
      the closed form."""
    grid, ell, K, p, q = 256, 16.0, 128, 32, 2
    n, l = grid * grid, K + p
    op = gsi.gridcov_operator(ctx, grid, grid, ell, 0)
    Om = gsi.DeviceMatrix(ctx, n, l).randn(99)
    Z = gsi.DeviceMatrix(ctx, n, l)
    S = gsi.DeviceMatrix(ctx, l, 1)
    gsi._lib.check(ctx.lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, S.h), ctx.lib)
    Zh, Sh = Z.to_host(), S.to_host()[:, 0]
    assert np.all(np.diff(Sh) <= 0) and Sh[K - 1] > 0
    assert np.all(Zh[:, K:] == 0.0)
    G = Zh[:, :K].T @ Zh[:, :K]
    assert np.abs(G - np.diag(Sh[:K])).max() < 1e-10 * Sh[0]
    V = Zh[:, :8] / np.sqrt(Sh[:8])
    AV = op.matmul(V)
    resid = np.linalg.norm(AV - V * Sh[:8], axis=0) / Sh[:8]
    assert resid.max() < 1e-4, resid      # randsvd vectors are approximate eigenvectors (q = 2)
    # a few entries of A against the closed form exp(-d^2 / (2 ell^2)) through e_i probes
    E = np.zeros((n, 2))
    E[12345, 0] = 1.0
    E[65535, 1] = 1.0
    cols = op.matmul(E)
    for c, i in enumerate((12345, 65535)):
        xi, yi = divmod(i, grid)
        jj = np.arange(n)
        d2 = (jj // grid - xi) ** 2 + (jj % grid - yi) ** 2
        assert np.abs(cols[:, c] - np.exp(-d2 / (2 * ell * ell))).max() < 1e-14
    for h in (Om, Z, S, op):
        h.close()


def test_rccl_single_rank_communicator(gsi):
    """RCCL path on one GPU: a 1-rank communicator (GSI_FORCE_COMM=1) makes every collective of the
    sharded pipeline run through librccl on the library's stream; results must equal the plain path."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gsi_amd as gsi
from helpers import gaussian_cov, powerlaw_fields
os.environ["GSI_FORCE_COMM"] = "1"
ctx = gsi.Context(0)
ctx.comm_init(1, 0, ctx.unique_id())
A = gaussian_cov(20, 15, 3.0); rng = np.random.default_rng(1); Om = rng.standard_normal((300, 20))
Z1, S1 = gsi.randsvd(A, 14, 6, 2, Omega=Om, return_S=True, ctx=ctx)
fields = powerlaw_fields(rng, (12, 12), 30); Om2 = rng.standard_normal((144, 12))
lr = gsi.LowRankCovMatrix(fields, ctx=ctx); Z3 = gsi.randsvd(lr, 8, 4, 3, Omega=Om2)
gi = gsi.gridcov_implicit_operator(ctx, 20, 15, 3.0); Z5 = gsi.randsvd(gi, 14, 6, 2, Omega=Om)
# the FFT covariance behind a communicator: rows <-> columns all-to-alls (ncclSend / ncclRecv groups), row-sharded LU, TSQR;
# and the row-sharded entry point (Omega rows in, Z rows out: with one rank the shard is the whole panel)
fo = gsi.fft_powerlaw_operator(ctx, [25, 12], -3.5, fftrf=True)
Z6, S6 = gsi.randsvd(fo, 14, 6, 2, Omega=Om, return_S=True)
Z7m, S7 = gsi.randsvd_rows(fo, 14, 6, 2, Om, return_S=True); Z7 = Z7m.to_host()
Xf = rng.standard_normal((300, 5)); Yf = fo.matmul(Xf); Yft = fo.rmatmul_t(Xf)
del os.environ["GSI_FORCE_COMM"]
ctx2 = gsi.Context(0)
fo2 = gsi.fft_powerlaw_operator(ctx2, [25, 12], -3.5, fftrf=True)
Z8, S8 = gsi.randsvd(fo2, 14, 6, 2, Omega=Om, return_S=True)
assert np.abs(S6 - S8).max() < 1e-12 * S8[0] and np.abs(S7 - S8).max() < 1e-12 * S8[0]
assert np.abs(Z6 @ Z6.T - Z8 @ Z8.T).max() < 1e-9 and np.abs(Z7 @ Z7.T - Z8 @ Z8.T).max() < 1e-9
assert np.abs(Yf - fo2.matmul(Xf)).max() < 1e-12 and np.abs(Yft - Yf).max() < 1e-12
Z2, S2 = gsi.randsvd(A, 14, 6, 2, Omega=Om, return_S=True, ctx=ctx2)
lr2 = gsi.LowRankCovMatrix(fields, ctx=ctx2); Z4 = gsi.randsvd(lr2, 8, 4, 3, Omega=Om2)
assert np.abs(S1 - S2).max() < 1e-12 * S2[0], np.abs(S1 - S2).max()
assert np.abs(Z1 - Z2).max() < 1e-9
assert np.abs(Z3 - Z4).max() < 1e-9
assert np.abs(Z5 @ Z5.T - Z2 @ Z2.T).max() < 1e-9
print("rccl-1rank-ok")
'''
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "rccl-1rank-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- CholeskyQR2 fast path and its Householder fall-back -------------------------------------------------
def test_qr_paths(gsi):
    """CholeskyQR2 for well conditioned panels, shifted CholeskyQR3 beyond (fast-decaying spectra), Householder as
    the last resort; all three give an orthonormal Q and Y = Q R to rounding.  A panel that needed the second tier
    makes the following factorizations start there (own context: the hint is per-context state)."""
    c = gsi.Context(0)
    m, l = 4096, 96
    seq = [(1e0, "cholqr2"), (1e4, "cholqr2"), (1e10, "scholqr3"), (1e12, "scholqr3"), (1e15, "householder"),
           (1e0, "cholqr2")]
    for cond, path in seq:
        rng = np.random.default_rng(int(np.log10(cond)) + 3)
        U, _ = np.linalg.qr(rng.standard_normal((m, l)))
        V, _ = np.linalg.qr(rng.standard_normal((l, l)))
        Y = (U * np.logspace(0, -np.log10(cond), l)) @ V.T
        before = c.counters()
        Q, R = gsi.qr_thinQ(Y, return_R=True, ctx=c)
        after = c.counters()
        for k in ("cholqr2", "scholqr3", "householder"):
            assert after[k] - before[k] == (1 if k == path else 0), (cond, k, before, after)
        assert np.abs(Q.T @ Q - np.eye(l)).max() < 1e-13
        assert np.abs(Q @ R - Y).max() < 1e-13 * l
        assert np.abs(np.tril(R, -1)).max() == 0.0
        s = np.linalg.svd(R, compute_uv=False)
        sref = np.linalg.svd(Y, compute_uv=False)
        assert np.abs(s - sref).max() < 1e-13 * sref[0]       # absolute O(eps*sigma_1), like Householder / dgesdd
    c.close()


def test_qr_rank_deficient_falls_back(gsi, ctx):
    rng = np.random.default_rng(12)
    Y = exact_rank_matrix(rng, 2000, 7)[:, :40]
    before = ctx.counters()
    Q = gsi.qr_thinQ(Y)
    after = ctx.counters()
    assert after["householder"] - before["householder"] == 1
    assert np.abs(Q.T @ Q - np.eye(40)).max() < 1e-13
    assert np.linalg.norm(Y - Q @ (Q.T @ Y)) < 1e-11 * np.linalg.norm(Y)


# ---- wide sketches: column chunking of the contraction kernel (l > 160) and the larger Jacobi problem --------
@pytest.mark.parametrize("n,K,p,q", [(1500, 200, 120, 1), (1200, 300, 212, 2)])
def test_randsvd_wide_sketch(gsi, ctx, n, K, p, q):
    rng = np.random.default_rng(n + K)
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    A = B @ np.diag(np.logspace(0, -3, n)) @ B.T          # SPD, slowly decaying spectrum
    Omega = rng.standard_normal((n, K + p))
    Z, S = gsi.randsvd(A, K, p, q, Omega=Omega, return_S=True)
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Omega)
    assert rel_sv_err(S, Sref, K) < 1e-9
    assert np.linalg.norm(Z @ Z.T - Zref @ Zref.T) < 1e-8 * np.linalg.norm(Zref @ Zref.T)


def test_unaligned_sizes(gsi, ctx):
    """odd n / odd l: every load goes through the element-wise predicated path."""
    rng = np.random.default_rng(77)
    n, K, p, q = 1237, 37, 12, 2
    B = rng.standard_normal((n, 300))
    A = B @ B.T / 300
    Omega = rng.standard_normal((n, K + p))
    Z, S = gsi.randsvd(A, K, p, q, Omega=Omega, return_S=True)
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Omega)
    assert rel_sv_err(S, Sref, K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, Zref, K) < 1e-6


def test_device_resident_basis_gpu(gsi, ctx):
    """SURVEY.md 8f (f1) on the GPU: getxis_device -> DeviceBasis -> pcgadirect / pcgalsqr."""
    import scipy.sparse as sp
    rng = np.random.default_rng(43)
    M, N, mu = 16, 512, 10.0
    x = rng.standard_normal(N)
    Q0 = rng.standard_normal((M, N))
    Qc = Q0.T @ Q0
    w, V = np.linalg.eigh(Qc)
    truep = (V * np.sqrt(np.clip(w, 0, None))) @ V.T @ rng.standard_normal(N) + mu
    forward = lambda pv: pv * x
    Om = rng.standard_normal((N, M + 2))
    basis = gsi.getxis_device(Qc, M, 2, 3, Omega=Om)
    xis_ref = orc.getxis_dense(Qc, M, 2, 3, Om)
    assert orc.xis_error_up_to_sign(np.stack([basis[i] for i in range(M)], axis=1), np.stack(xis_ref, axis=1), M) < 1e-6
    X = np.full(N, mu)
    R = 1e-8 * sp.identity(N, format="csc")
    y = forward(truep) + 1e-4 * rng.standard_normal(N)
    got = gsi.pcgadirect(forward, X.copy(), X, basis, R, y)
    ref = orc.pcgadirect(forward, X.copy(), X, xis_ref, R, y)
    # finite differences with delta = sqrt(eps) amplify rounding-level differences of the basis by 1/delta:
    # two correct runs agree to ~1e-5, far inside the reference's own 2e-2 bar
    assert np.linalg.norm(got - ref) < 1e-3 * np.linalg.norm(ref)
    assert np.linalg.norm(got - truep) / np.linalg.norm(truep) < 2e-2
    got = gsi.pcgalsqr(forward, X.copy(), X, basis, R, y)
    assert np.linalg.norm(got - truep) / np.linalg.norm(truep) < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,l", [(40, 33, 5), (40, 33, 160), (64, 50, 200), (130, 7, 48)])
def test_implicit_gridcov_products(gsi, ctx, nx, ny, l):
    """gsi_op_gridcov_implicit (entries generated inside the MFMA kernel, nothing stored) against the stored
    operator of gsi_op_dense_gridcov on the same grid: A*X and A'*X, ragged row / reduction edges included."""
    ell = 3.5
    n = nx * ny
    rng = np.random.default_rng(nx * ny + l)
    X = rng.standard_normal((n, l))
    dense = gsi.gridcov_operator(ctx, nx, ny, ell, 0)
    impl = gsi.gridcov_implicit_operator(ctx, nx, ny, ell)
    Yd, Yi = dense.matmul(X), impl.matmul(X)
    scale = np.abs(Yd).max()
    assert np.abs(Yi - Yd).max() < 1e-13 * scale
    assert np.abs(impl.rmatmul_t(X) - dense.rmatmul_t(X)).max() < 1e-13 * scale
    dense.close()
    impl.close()


@pytest.mark.gpu
def test_implicit_gridcov_at_the_headline_size(gsi, ctx):
    """The 10^6 x 10^6 implicit covariance of the bench (1000 x 1000 grid, exp(-d / 100)) cannot be compared with a stored
    matrix; rows of it can: A*X and A'*X for 16 columns against exp(-d(i, .) / 100) . X computed on the host for rows at both
    ends, around the 32-bit row * column boundaries and at random."""
    g, ell, l = 1000, 100.0, 16
    n = g * g
    rng = np.random.default_rng(5)
    X = np.asfortranarray(rng.standard_normal((n, l)))
    op = gsi.gridcov_implicit_operator(ctx, g, g, ell, kind=1)
    Y = op.matmul(X)
    Yt = op.rmatmul_t(X)
    op.close()
    px, py = np.divmod(np.arange(n), g)                        # point = x * ny + y (tests/helpers.py:grid_points)
    rows = np.concatenate([[0, 1, g - 1, g, n // 2, n - g, n - 1, 4295, 65535, 65536, 262143, 262144],
                           rng.integers(0, n, size=40)])
    scale = np.abs(Y).max()
    for i in rows:
        a = np.exp(-np.sqrt((px - px[i]) ** 2.0 + (py - py[i]) ** 2.0) / ell)
        ref = a @ X
        assert np.abs(Y[i] - ref).max() < 1e-12 * scale, i
        assert np.abs(Yt[i] - ref).max() < 1e-12 * scale, i


@pytest.mark.gpu
def test_implicit_gridcov_randsvd(gsi, ctx):
    """randsvd through the implicit operator = randsvd of the stored matrix (same Omega), and = the oracle."""
    nx, ny, ell, K, p, q = 48, 40, 4.0, 20, 12, 2
    n = nx * ny
    Om = np.random.default_rng(3).standard_normal((n, K + p))
    impl = gsi.gridcov_implicit_operator(ctx, nx, ny, ell)
    Z, S = gsi.randsvd(impl, K, p, q, Omega=Om, return_S=True)
    A = gaussian_cov(nx, ny, ell)
    Zr, Sr, _ = orc.randsvd_full(A, K, p, q, Om)
    assert rel_sv_err(S, Sr, K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, Zr, K) < 1e-6
    assert np.all(Z[:, K:] == 0)
    impl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,K,p", [(33, 35, 40, 10), (47, 21, 100, 20), (64, 33, 23, 0)])
def test_randsvd_odd_n_ragged_l(gsi, ctx, nx, ny, K, p):
    """n odd (X panels only 8-byte aligned) and K + p not a multiple of 16: the irregular-X instantiation of the
    contraction kernel carries every operator product; result against the oracle."""
    n = nx * ny
    A = gaussian_cov(nx, ny, 2.5) + 0.05 * exponential_cov(nx, ny, 6.0)
    Om = np.random.default_rng(n).standard_normal((n, K + p))
    Z, S = gsi.randsvd(A, K, p, 2, Omega=Om, return_S=True, ctx=ctx)
    Zr, Sr, _ = orc.randsvd_full(A, K, p, 2, Om)
    assert np.abs(S[:K] - Sr[:K]).max() < 1e-10 * Sr[0]
    assert np.abs(Z @ Z.T - Zr @ Zr.T).max() < 1e-8 * Sr[0]


@pytest.mark.gpu
def test_randsvd_shape_sweep(gsi, ctx):
    """60 seeded random (m, n, K, p, q) configurations -- 1 x 1, ragged, rectangular, l = min(m, n), p = 0 --
    against the oracle: svd(Q'A).S and the basis-independent product Z Z' (= V_K S_K V_K')."""
    from helpers import random_shape_cases, decaying_matrix
    rng = np.random.default_rng(2024)
    worst = 0.0
    for (m, n, K, p, q, decay) in random_shape_cases(11, 60, 150):
        A = decaying_matrix(rng, m, n, decay)
        Om = rng.standard_normal((n, K + p))
        Z, S = gsi.randsvd(A, K, p, q, Omega=Om, return_S=True, ctx=ctx)
        Zr, Sr, _ = orc.randsvd_full(A, K, p, q, Om)
        assert Z.shape == (n, K + p) and np.all(Z[:, K:] == 0)
        e_s = np.abs(S - Sr).max() / Sr[0]
        e_z = np.abs(Z @ Z.T - Zr @ Zr.T).max() / Sr[0]
        worst = max(worst, e_s, e_z)
        assert e_s < 1e-10 and e_z < 1e-8, (m, n, K, p, q, e_s, e_z)


@pytest.mark.gpu
@pytest.mark.parametrize("Ns,beta,l", [((50,), -2.0, 7), ((24, 17), -3.5, 10), ((40, 64), -2.5, 33), ((9, 6, 11), -3.0, 4),
                                        ((300, 200), -3.5, 5),
                                        ((4096,), -2.0, 3), ((3000,), -2.5, 2), ((2048, 4), -3.0, 3), ((3, 1100), -2.0, 2),   # 8192- / 4096-point lines
                                        ((2, 3), -2.0, 2), ((5, 2, 3), -3.0, 3)])                                            # lines shorter than 16 points
def test_fft_powerlaw_operator(gsi, ctx, Ns, beta, l):
    """The hand-written LDS FFT passes of the matrix-free covariance (gsi_op_fft_powerlaw) against numpy's FFT:
    odd/even column counts (two real columns per complex transform), 1/2/3 axes, non-power-of-two grids."""
    n = int(np.prod(Ns))
    rng = np.random.default_rng(n + l)
    X = rng.standard_normal((n, l))
    op = gsi.fft_powerlaw_operator(ctx, Ns, beta)
    Y = op.matmul(X)
    Yref = orc.fft_powerlaw_apply(X, list(Ns), beta)
    assert np.abs(Y - Yref).max() < 1e-12 * np.abs(Yref).max()
    assert np.abs(op.rmatmul_t(X) - Yref).max() < 1e-12 * np.abs(Yref).max()
    e = np.zeros((n, 2)); e[0, 0] = 1.0; e[n // 2, 1] = 1.0
    d = op.matmul(e)
    assert abs(d[0, 0] - 1.0) < 1e-12 and abs(d[n // 2, 1] - 1.0) < 1e-12          # unit diagonal
    op.close()


@pytest.mark.gpu
def test_fft_grid_limit_is_refused_at_creation(gsi, ctx):
    """An embedding of 2^31 points (32-bit offsets inside a column pair's array): GSI_ERR_ARG when the operator is created,
    before spectrum and work array are allocated -- not at the first product."""
    before = ctx.device_bytes()
    with pytest.raises(gsi.GsiError, match=r"fewer than 2\^31"):
        gsi.fft_powerlaw_operator(ctx, [1024, 512, 512], -3.5)
    assert ctx.device_bytes() == before


@pytest.mark.gpu
def test_fft_powerlaw_randsvd(gsi, ctx):
    """randsvd through the matrix-free FFT operator = randsvd of the same covariance stored densely (oracle)."""
    Ns, beta, K, p, q = (20, 16), -3.5, 12, 6, 2
    n = int(np.prod(Ns))
    A = orc.fft_powerlaw_apply(np.eye(n), list(Ns), beta)
    Om = np.random.default_rng(8).standard_normal((n, K + p))
    op = gsi.fft_powerlaw_operator(ctx, Ns, beta)
    Z, S = gsi.randsvd(op, K, p, q, Omega=Om, return_S=True)
    Zr, Sr, _ = orc.randsvd_full(A, K, p, q, Om)
    assert rel_sv_err(S, Sr, K) < 1e-9
    assert np.abs(Z @ Z.T - Zr @ Zr.T).max() < 1e-8 * Sr[0]
    op.close()


@pytest.mark.gpu
def test_panels_beyond_32bit_tile_offsets(gsi, ctx):
    """Panels of more than ~3.36 million rows: 160 columns * ld * 8 B no longer fits the contraction kernel's 32-bit
    per-thread offsets, so the 64-bit-offset instantiation carries the panel products inside LU (trailing updates),
    CholeskyQR (Gram matrices, Y R^-1) and the tall SVD.  Checked against LAPACK on the same panel."""
    m, l = 3_400_000, 24
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((m, l)) * np.logspace(0, -3, l)[None, :]
    L, piv = gsi.lu_L(Y, return_pivots=True)
    assert np.array_equal(piv, orc.lu_pivots(Y))
    assert np.abs(L - orc.lu_L(Y)).max() < 1e-10
    del L
    Q, R = gsi.qr_thinQ(Y, return_R=True)
    assert np.abs(Q.T @ Q - np.eye(l)).max() < 1e-12
    assert np.abs(Q @ R - Y).max() < 1e-11
    del Q
    S, V = gsi.svd_tall(Y)
    Sref = np.linalg.svd(Y, compute_uv=False)
    assert np.abs(S - Sref).max() < 1e-11 * Sref[0]
    assert np.abs(V.T @ V - np.eye(l)).max() < 1e-11


# ---- a15: the adaptive range finder against the oracle on the SAME Gaussian stream (RandMatFact.jl:15-48) ----------
@pytest.mark.gpu
@pytest.mark.parametrize("n,m,r", [(60, 8, 10), (200, 21, 10), (150, 12, 4)])
def test_rangefinder_adaptive_oracle_parity(gsi, ctx, n, m, r):
    rng = np.random.default_rng(5 * n + m)
    A = exact_rank_matrix(rng, n, m)
    gsi.RandMatFact.seed(77)                     # the library pulls its Gaussian stream through the randn callback
    Q = gsi.rangefinder(A, r=r, ctx=ctx)
    ref_rng = np.random.default_rng(77)          # the oracle consumes the same stream, in the reference's order,
    Qref = orc.rangefinder_adaptive(A, lambda shape: ref_rng.standard_normal(int(np.prod(shape))).reshape(shape, order="F"),
                                    r=r)         # filled column-major like Julia's randn(n, r)
    assert Q.shape == Qref.shape                 # same number of columns j
    assert abs(Q.shape[1] - m) <= 1
    # columns spanning the numerical range agree one by one; a column past the rank (if any) is normalised noise
    assert np.abs(Q[:, :m - 1] - Qref[:, :m - 1]).max() < 1e-9
    assert np.linalg.norm(A - Q @ Q.T @ A) < 1e-8


# ---- f4: `\(A::LowRankCovMatrix, b)` = lsqr(A, b; maxiter = N)  (lowrank.jl:141-144) --------------------------------
@pytest.mark.gpu
def test_lowrank_solve_gpu(gsi, ctx):
    rng = np.random.default_rng(21)
    # well-conditioned on its range (iid Gaussian samples): LSQR converges inside maxiter = N, the iterates agree tightly
    fields = list(rng.standard_normal((40, 625)))
    lr = gsi.LowRankCovMatrix(fields, ctx=ctx)
    ref = orc.LowRankCovMatrix(fields)
    b = ref.matmul(rng.standard_normal(625))
    x, it = lr.solve(b, return_iterations=True)
    xr, itr = orc.lsqr(ref.matmul, ref.matmul, b, 625, maxiter=40)
    assert it == itr and it < 40
    assert np.linalg.norm(x - xr) < 1e-7 * np.linalg.norm(xr)
    assert np.linalg.norm(ref.matmul(x) - b) < 1e-6 * np.linalg.norm(b)
    lr.close()
    # FFTRF-like fields (fast spectral decay): the iteration stops on maxiter; iterates of the two product
    # formulations differ by rounding x condition number, the residual they reach is the same
    fields = powerlaw_fields(rng, (25, 25), 40)
    lr = gsi.LowRankCovMatrix(fields, ctx=ctx)
    ref = orc.LowRankCovMatrix(fields)
    b = ref.matmul(rng.standard_normal(625))
    x, it = lr.solve(b, return_iterations=True)
    xr = ref.solve(b)
    assert 1 <= it <= 40
    assert np.linalg.norm(x - xr) < 2e-2 * np.linalg.norm(xr)
    rg, rr = np.linalg.norm(ref.matmul(x) - b), np.linalg.norm(ref.matmul(xr) - b)
    assert rg < 1e-2 * np.linalg.norm(b) and abs(rg - rr) < 0.5 * max(rg, rr) + 1e-12 * np.linalg.norm(b)
    lr.close()


# ---- f1: PCGALowRankMatrix product and IterativeSolvers-style LSQR on the device (lowrank.jl:83-97, lsqr.jl:53-54) ----
@pytest.mark.gpu
@pytest.mark.parametrize("nobs,K", [(23, 5), (300, 40), (2049, 64)])
def test_pcga_lowrank_matrix_gpu(gsi, ctx, nobs, K):
    import scipy.sparse as sp
    rng = np.random.default_rng(nobs + K)
    etas = [rng.standard_normal(nobs) for _ in range(K)]
    HX = rng.standard_normal(nobs)
    R = 1e-2 * sp.identity(nobs, format="csc")
    A = gsi.PCGALowRankMatrix(etas, HX, R, ctx=ctx)
    ref = orc.PCGALowRankMatrix(etas, HX, R)
    x = rng.standard_normal(nobs + 1)
    yr = ref.matvec(x)
    assert np.abs(A.matvec(x) - yr).max() < 1e-11 * np.abs(yr).max()
    if nobs <= 300:
        Rd = R.toarray() + 1e-3 * np.diag(rng.random(nobs))
        A2 = gsi.PCGALowRankMatrix(etas, HX, Rd, ctx=ctx)
        y2 = orc.PCGALowRankMatrix(etas, HX, Rd).matvec(x)
        assert np.abs(A2.matvec(x) - y2).max() < 1e-11 * np.abs(y2).max()
        A2.close()
    b = np.concatenate([rng.standard_normal(nobs), [0.0]])
    sol, it = A.lsqr(b, return_iterations=True)
    solr, itr = orc.lsqr(ref.matvec, ref.matvec, b, nobs + 1)
    # The stopping rules fire on sqrt(eps)-sized quantities: rounding-level differences (summation order of the norms)
    # can move the stop by an iteration, and LSQR iterates of this saddle-point system carry 1e-16 noise amplified to
    # ~1e-5 (the oracle itself with 1e-16 noise injected into its products moves by 6e-6).  So: the same iteration count
    # up to 2, the iterate at that count, and a residual as small as the oracle's.
    assert abs(it - itr) <= 2
    sol_at, _ = orc.lsqr(ref.matvec, ref.matvec, b, nobs + 1, maxiter=it, atol=0.0, btol=0.0, conlim=0.0)
    assert np.linalg.norm(sol - sol_at) < 1e-4 * np.linalg.norm(sol_at)
    assert np.linalg.norm(sol - solr) < 1e-4 * np.linalg.norm(solr)
    rg, rr = np.linalg.norm(ref.matvec(sol) - b), np.linalg.norm(ref.matvec(solr) - b)
    assert rg < 2.0 * rr + 1e-6 * np.linalg.norm(b)
    A.close()


@pytest.mark.gpu
def test_fp32_basis_gpu(gsi, ctx):
    rng = np.random.default_rng(31)
    n, K, nobs = 5003, 33, 70
    Zh = rng.standard_normal((n, K + 7))
    Zd = gsi.DeviceMatrix.from_host(ctx, Zh)
    b64 = gsi.DeviceBasis(Zd, K)
    b32 = gsi.DeviceBasis(Zd, K, precision=32)
    s, X = rng.standard_normal(n), rng.standard_normal(n)
    Z32 = Zh[:, :K].astype(np.float32).astype(np.float64)
    P32 = b32.params(s, X, 1e-3)
    assert np.abs(P32[:, :K] - (s[:, None] + 1e-3 * Z32)).max() < 1e-15
    assert np.array_equal(P32[:, K:], b64.params(s, X, 1e-3)[:, K:])
    etas = [rng.standard_normal(nobs) for _ in range(K)]
    xb = rng.standard_normal(nobs)
    w = np.array([e @ xb for e in etas])
    u32 = b32.update(X, 0.7, etas, xb)
    assert np.abs(u32 - (0.7 * X + Z32 @ w)).max() < 1e-11 * np.abs(u32).max()
    u64 = b64.update(X, 0.7, etas, xb)
    assert np.abs(u64 - (0.7 * X + Zh[:, :K] @ w)).max() < 1e-11 * np.abs(u64).max()
    assert np.abs(u32 - u64).max() < 1e-5 * np.abs(u64).max()           # fp32 storage: 6e-8 relative per entry
    assert np.array_equal(b32[2], Z32[:, 2]) and np.array_equal(b64[2], Zh[:, 2])
    b32.close(); b64.close(); Zd.close()


# ---- C5 (BASELINE configs[4]): pcgalsqr at n = 1e6, K = 256, nobs = 4096 with the xi-basis resident in HBM, fp32-stored
#      basis against the fp64 one.  Stated tolerance: the two inversions agree to 1e-5 relative (fp32 rounds each basis
#      entry to 6e-8 relative; the update is a sum of K = 256 such columns), both reach the same data misfit. ----------
@pytest.mark.gpu
def test_pcgalsqr_c5_fp32_vs_fp64(gsi, ctx):
    import scipy.sparse as sp
    n, Ns, K, p, q, nobs = 1000000, 256, 256, 64, 1, 4096
    op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=3, decay=0.75)
    gsi.RandMatFact.seed(4)
    Om = gsi.DeviceMatrix(ctx, n, K + p).randn(9)
    Z = gsi.DeviceMatrix(ctx, n, K + p)
    gsi._lib.check(ctx.lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, None), ctx.lib)
    Om.close(); op.close()
    b64 = gsi.DeviceBasis(Z, K)
    b32 = gsi.DeviceBasis(Z, K, precision=32)
    rng = np.random.default_rng(8)
    idx = np.arange(nobs) * (n // nobs) + 17
    xw = 1.0 + 0.1 * rng.standard_normal(n)

    def forward(s):                                  # h(s) = (s .* x)[observed points]   (test/testrpcga.jl:110-112)
        return (s * xw)[idx]

    mu = 2.0
    X = np.full(n, mu)
    coef = rng.standard_normal(6) * 3.0
    truth = X + sum(c * b64[i] for i, c in enumerate(coef))          # a field in the span of the leading xis
    noise = 1e-4
    y = forward(truth) + noise * rng.standard_normal(nobs)
    R = noise ** 2 * sp.identity(nobs, format="csc")
    s64 = gsi.pcgalsqr(forward, X.copy(), X, b64, R, y, maxiters=2)
    s32 = gsi.pcgalsqr(forward, X.copy(), X, b32, R, y, maxiters=2)
    mis0 = np.linalg.norm(forward(X) - y)
    mis64 = np.linalg.norm(forward(s64) - y)
    mis32 = np.linalg.norm(forward(s32) - y)
    assert mis64 < 1e-2 * mis0 and mis32 < 1e-2 * mis0               # both fit the data
    assert np.linalg.norm(s32 - s64) < 1e-5 * np.linalg.norm(s64)    # stated fp32-vs-fp64 tolerance
    assert np.linalg.norm(s64 - truth) < 0.2 * np.linalg.norm(truth - X)   # and recover the observed part of the field
    if not __import__("os").environ.get("GSI_SKIP_C5_ORACLE"):
        # configs[4] at FULL size against the oracle: the K = 256 xi-vectors come off the device and
        # orc.pcgalsqr (lsqr.jl:35-63 restated; lowrank.jl:83-97 products, Paige-Saunders LSQR) runs the same two iterations on
        # the host with the same forward model, R and y.  Bar = test_device_resident_basis_gpu's: finite differences with
        # delta = sqrt(eps) amplify rounding-level differences by 1/delta, two correct runs agree to ~1e-5.
        xis_host = [b64[i] for i in range(K)]
        s_ref = orc.pcgalsqr(forward, X.copy(), X, xis_host, R, y, maxiters=2)
        d64 = np.linalg.norm(s64 - s_ref) / np.linalg.norm(s_ref)
        d32 = np.linalg.norm(s32 - s_ref) / np.linalg.norm(s_ref)
        dfit = abs(np.linalg.norm(forward(s_ref) - y) - mis64) / mis0
        print(f"C5 at n = 1e6 vs orc.pcgalsqr: fp64 basis {d64:.2e}, fp32 basis {d32:.2e}, misfit difference {dfit:.2e} of the initial misfit")
        assert d64 < 1e-3 and d32 < 1e-3, (d64, d32)
        assert dfit < 1e-3
    b32.close(); b64.close(); Z.close()


# ---- f2 in FFTRF's own convention (2N embedding, integer wavenumbers: the covariance of FFTRF's fields, see
#      tests/test_fftrf_covariance.py for the statistical tie to the restated FFTRF.jl) -------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("Ns,beta,l", [((64,), -2.0, 5), ((32, 8), -3.5, 9), ((16, 64), -2.5, 33), ((8, 4, 16), -3.0, 6),
                                       ((256, 128), -3.5, 16),
                                       # grids whose 2N embedding is NOT a power of two (FFTRF.jl:83-90 takes any N): the
                                       # reference's own 25 x 25 (test/testrpcga.jl:84-88), ragged 2-D, 3-D, 1-D, a prime
                                       ((25, 25), -3.5, 50), ((50, 40), -3.5, 12), ((9, 6, 11), -3.0, 7), ((100,), -2.0, 3),
                                       ((7,), -2.5, 4), ((24, 16), -3.5, 5), ((300, 200), -3.5, 6)])
def test_fft_powerlaw_fftrf_convention(gsi, ctx, Ns, beta, l):
    n = int(np.prod(Ns))
    rng = np.random.default_rng(n + l)
    X = rng.standard_normal((n, l))
    op = gsi.fft_powerlaw_operator(ctx, Ns, beta, fftrf=True)
    Y = op.matmul(X)
    Yref = orc.fft_powerlaw_apply(X, list(Ns), beta, fftrf=True)
    assert np.abs(Y - Yref).max() < 1e-11 * np.abs(Yref).max()
    Yiso = orc.fft_powerlaw_apply(X, list(Ns), beta, fftrf=False)
    if len(set(Ns)) > 1:
        assert np.abs(Yiso - Yref).max() > 1e-3 * np.abs(Yref).max()       # a different operator on unequal axes
    op.close()


# ---- the FFT operator at FULL size against a host FFT (scipy.fft on all cores) of the oracle's own spectrum: the small-grid
#      parity tests cannot see an index that overflows at 10^9 embedding points.  512^3 is BASELINE configs[2]'s own grid
#      (a 1024^3 host transform: ~35 GB of host memory, about a minute); GSI_SKIP_FFT_FULLSIZE=1 skips the three large grids,
#      GSI_SKIP_FFT_512CUBE=1 only the largest ------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("Ns,beta,fftrf", [((700, 900), -2.5, False), ((3000, 3000), -3.5, True), ((300, 300, 300), -3.5, True),
                                           ((512, 512, 512), -3.5, True)])
def test_fft_operator_full_size_vs_host_fft(gsi, ctx, Ns, beta, fftrf):
    import os
    import scipy.fft as sfft
    n = int(np.prod(Ns))
    if n > 1000000 and os.environ.get("GSI_SKIP_FFT_FULLSIZE"):
        pytest.skip("GSI_SKIP_FFT_FULLSIZE set")
    if n > 100000000 and os.environ.get("GSI_SKIP_FFT_512CUBE"):
        pytest.skip("GSI_SKIP_FFT_512CUBE set")
    rng = np.random.default_rng(n)
    X = np.asfortranarray(rng.standard_normal((n, 2)))
    ctx.release_cache()
    op = gsi.fft_powerlaw_operator(ctx, list(Ns), beta, fftrf=fftrf)
    Y = op.matmul(X)
    op.close()
    ctx.release_cache()
    lam, Ms = orc.fft_powerlaw_spectrum(list(Ns), beta, fftrf)
    box = tuple(slice(0, N) for N in Ns)
    for c in range(2):
        w = np.zeros(Ms)
        w[box] = X[:, c].reshape(Ns, order="F")
        f = sfft.fftn(w, workers=-1)
        del w
        f *= lam
        y = sfft.ifftn(f, workers=-1, overwrite_x=True).real[box].reshape(-1, order="F")
        del f
        assert np.abs(Y[:, c] - y).max() < 1e-12 * np.abs(y).max(), (Ns, c)


@pytest.mark.gpu
def test_fftrf_covariance_on_the_reference_grid(gsi, ctx):
    """`FFTRFCovariance([25, 25], -3.5)`: the grid and beta of the reference's own getxis test (test/testrpcga.jl:84-88).
    (i) the HIP operator is the oracle's matrix, symmetric with a unit diagonal; (ii) randsvd through it (K = 30, p = 20,
    q = 3 as at :85-86,91) equals the oracle's on the materialised matrix with the same Omega; (iii) the sample covariance
    of restated FFTRF.jl fields on a non-power-of-two grid converges to the HIP operator (the statistical tie of
    tests/test_fftrf_covariance.py, now through the GPU on a grid the power-of-two transforms do not embed exactly)."""
    Ns, beta = [25, 25], -3.5
    n = 625
    op = gsi.fft_powerlaw_operator(ctx, Ns, beta, fftrf=True)
    A = op.matmul(np.eye(n))
    Aref = orc.fft_powerlaw_apply(np.eye(n), Ns, beta, fftrf=True)
    assert np.abs(A - Aref).max() < 1e-12
    assert np.abs(A - A.T).max() < 1e-12 and np.abs(np.diag(A) - 1.0).max() < 1e-12
    rng = np.random.default_rng(0)
    K, p, q = 30, 20, 3
    Om = rng.standard_normal((n, K + p))
    Z, S = gsi.randsvd(op, K, p, q, Omega=Om, return_S=True)
    Zref, Sref, _ = orc.randsvd_full(Aref, K, p, q, Om)
    assert rel_sv_err(S, Sref, K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, Zref, K) < 1e-6
    op.close()
    Ns2 = [6, 5]
    n2 = 30
    op2 = gsi.fft_powerlaw_operator(ctx, Ns2, -2.5, fftrf=True)
    A2 = op2.matmul(np.eye(n2))
    op2.close()
    rng = np.random.default_rng(1234)
    C = np.zeros((n2, n2))
    nsamples = 6000
    for _ in range(nsamples):
        f = orc.fftrf_powerlaw_structuredgrid(Ns2, 0.0, 1.0, -2.5, rng, raw=True).reshape(-1, order="F")
        C += np.outer(f, f)
    C /= nsamples
    scale = np.trace(C) / n2
    assert np.linalg.norm(C / scale - A2) / np.linalg.norm(A2) < 0.09


# ---- implicit operator, exponential kernel and a caller-supplied stationary kernel table, against the
#      STORED gsi_op_dense_gridcov(kind=1) on the same grid ------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nx,ny,l", [(40, 33, 5), (64, 50, 160), (130, 7, 48)])
def test_implicit_exponential_vs_stored(gsi, ctx, nx, ny, l):
    n = nx * ny
    rng = np.random.default_rng(n + l)
    X = rng.standard_normal((n, l))
    stored = gsi.gridcov_operator(ctx, nx, ny, 6.0, 1)               # exp(-d / ell), materialised in HBM
    implicit = gsi.gridcov_implicit_operator(ctx, nx, ny, 6.0, kind=1)
    Ys, Yi = stored.matmul(X), implicit.matmul(X)
    assert np.abs(Yi - Ys).max() < 1e-11 * np.abs(Ys).max()
    assert np.abs(implicit.rmatmul_t(X) - Ys).max() < 1e-11 * np.abs(Ys).max()
    if l <= 48:
        Om = rng.standard_normal((n, l))
        K, p = l - 3, 3
        Zi, Si = gsi.randsvd(implicit, K, p, 2, Omega=Om, return_S=True)
        Zs, Ss = gsi.randsvd(stored, K, p, 2, Omega=Om, return_S=True)
        assert rel_sv_err(Si, Ss, K) < 1e-9
    stored.close(); implicit.close()


@pytest.mark.gpu
def test_implicit_table_operator_gpu(gsi, ctx):
    from helpers import grid_points
    nx, ny = 37, 21
    dx = np.arange(nx)[:, None]; dy = np.arange(ny)[None, :]
    r = np.sqrt((dx / 4.0) ** 2 + (dy / 9.0) ** 2)
    T = (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)        # anisotropic Matern 3/2
    P = grid_points(nx, ny)
    D = np.abs(P[:, None, :] - P[None, :, :]).astype(int)
    G = T[D[:, :, 0], D[:, :, 1]]
    op = gsi.gridcov_implicit_operator(ctx, nx, ny, 1.0, table=T)
    rng = np.random.default_rng(2)
    X = rng.standard_normal((nx * ny, 40))
    assert np.abs(op.matmul(X) - G @ X).max() < 1e-11 * np.abs(G @ X).max()
    op.close()


# ---- row-sharded LU on the GPU (one rank: no collective, every kernel of the sharded path): bit-identical to the
#      register-resident single-rank factorization, dgetrf's pivots ---------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("m,l", [(10, 2), (100, 25), (625, 50), (5000, 160), (3000, 33), (70001, 72), (300000, 96)])
def test_lu_sharded_bit_identical(gsi, ctx, m, l):
    rng = np.random.default_rng(m + l)
    Y = rng.standard_normal((m, l))
    if m == 3000:
        Y[1500:1600] = Y[100:200]                                    # exact ties
    Ls, ps = gsi.lu_L_sharded(Y, return_pivots=True, ctx=ctx)
    L1, p1 = gsi.lu_L(Y, return_pivots=True, ctx=ctx)
    assert np.array_equal(ps, p1)
    assert np.array_equal(Ls, L1)
    if m <= 5000:
        assert np.array_equal(ps, orc.lu_pivots(Y))


@pytest.mark.gpu
def test_lu_sharded_through_rccl_single_rank(gsi):
    """The row-sharded LU with its collectives going through librccl (1-rank communicator, GSI_FORCE_COMM=1) and forced
    on for every operator (GSI_LU_SHARDED=1): pivots and L equal to the plain single-rank factorization exactly; a
    randsvd through the sharded range finder equals the replicated one to rounding."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gsi_amd as gsi
from helpers import gaussian_cov, powerlaw_fields
os.environ["GSI_FORCE_COMM"] = "1"; os.environ["GSI_LU_SHARDED"] = "1"
ctx = gsi.Context(0)
ctx.comm_init(1, 0, ctx.unique_id())
rng = np.random.default_rng(4)
Y = rng.standard_normal((20000, 72))
Ls, ps = gsi.lu_L_sharded(Y, return_pivots=True, ctx=ctx)
A = gaussian_cov(20, 15, 3.0); Om = rng.standard_normal((300, 20))
Z1, S1 = gsi.randsvd(A, 14, 6, 2, Omega=Om, return_S=True, ctx=ctx)
fields = powerlaw_fields(rng, (12, 12), 30); Om2 = rng.standard_normal((144, 12))
lr = gsi.LowRankCovMatrix(fields, ctx=ctx); Z3 = gsi.randsvd(lr, 8, 4, 3, Omega=Om2)
del os.environ["GSI_FORCE_COMM"]
ctx2 = gsi.Context(0)
L1, p1 = gsi.lu_L(Y, return_pivots=True, ctx=ctx2)
assert np.array_equal(ps, p1) and np.array_equal(Ls, L1)
Z2, S2 = gsi.randsvd(A, 14, 6, 2, Omega=Om, return_S=True, ctx=ctx2)
lr2 = gsi.LowRankCovMatrix(fields, ctx=ctx2); Z4 = gsi.randsvd(lr2, 8, 4, 3, Omega=Om2)
assert np.abs(S1 - S2).max() < 1e-12 * S2[0]
assert np.abs(Z1 - Z2).max() < 1e-9 and np.abs(Z3 - Z4).max() < 1e-9
print("rccl-sharded-lu-ok")
'''
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "rccl-sharded-lu-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- the bench's own workload at BASELINE.json's full size (n = 1e6, N_s = 1024, K = 256, p = 64, q = 2): properties
#      that do not depend on the size -- Z[:, K:] == 0, S positive and descending, V = Z S^-1/2 orthonormal, a second run
#      bit-identical (every reduction in the path has a fixed order), V spans an invariant subspace of A (A V = V (V'AV)
#      for the leading vectors), and the build with the general contraction kernel in place of the dedicated
#      Gram / triangular-product / small-Cholesky kernels (GSI_NO_SYRK_KERNEL, GSI_NO_TRMM_KERNEL) gives the same singular
#      values.  Separate processes: the switches are read once per process. --------------------------------------------
@pytest.mark.gpu
def test_headline_size_properties(gsi):
    import os, subprocess, sys, tempfile
    code = r'''
import os, sys, numpy as np
import gsi_amd as gsi
ctx = gsi.Context(0)
n, Ns, K, p, q = 1000000, 1024, 256, 64, 2
op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=0, decay=0.75)
Om = gsi.DeviceMatrix(ctx, n, K + p).randn(1)
Z = gsi.DeviceMatrix(ctx, n, K + p); S = gsi.DeviceMatrix(ctx, K + p, 1)
def run():
    gsi._lib.check(ctx.lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, S.h), ctx.lib)
    return Z.to_host(), S.to_host()[:, 0]
Zh, Sh = run()
assert np.all(Zh[:, K:] == 0.0)
assert np.all(Sh[:K] > 0) and np.all(np.diff(Sh) <= 0)
if os.environ.get("GSI_TEST_FULL"):
    Z2, S2 = run()
    assert np.array_equal(Zh, Z2) and np.array_equal(Sh, S2)            # deterministic
    V = Zh[:, :K] / np.sqrt(Sh[:K])
    G = V.T @ V
    assert np.abs(G - np.eye(K)).max() < 1e-10, np.abs(G - np.eye(K)).max()
    j = 12
    AV = op.matmul(np.asfortranarray(V[:, :j]))
    T = V[:, :j].T @ AV
    assert np.abs(T - np.diag(Sh[:j])).max() < 1e-6 * Sh[0]             # V'AV = diag(S) on the leading vectors (q = 2)
    assert np.linalg.norm(AV - V[:, :j] * Sh[:j]) < 1e-5 * Sh[0] * np.sqrt(j)
np.save(sys.argv[1], Sh)
print("headline-ok")
'''
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for tag, extra in (("dedicated", {"GSI_TEST_FULL": "1"}), ("general", {"GSI_NO_SYRK_KERNEL": "1", "GSI_NO_TRMM_KERNEL": "1"})):
            env = dict(os.environ)
            env.update(extra)
            path = os.path.join(td, tag + ".npy")
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=900, env=env, cwd=here)
            assert r.returncode == 0 and "headline-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
            out[tag] = np.load(path)
    S1, S2 = out["dedicated"], out["general"]
    assert np.abs(S1 - S2).max() < 1e-11 * S1[0], np.abs(S1 - S2).max() / S1[0]


# ---- the metric's rel-err at the metric's size (BASELINE.json "top-k singular-value rel-err, n=1e6 rank=256"): the oracle
#      on the operator the HIP path ran on (samples, Omega downloaded), products in GEMM form.  Runs by default (VERDICT r4
#      item 3: the metric's own parity belongs in the driver-run suite; ~130 s of host LAPACK and ~40 GB of host memory on the
#      GPU box, the suite stays under its 900 s limit); GSI_SKIP_HEADLINE_PARITY=1 opts out, as for the C2 test.  bench.py runs
#      the same comparison in every default run and records it in the driver's line. -----------------------------------
@pytest.mark.gpu
@pytest.mark.skipif(bool(__import__("os").environ.get("GSI_SKIP_HEADLINE_PARITY")),
                    reason="opted out (GSI_SKIP_HEADLINE_PARITY=1): ~2 minutes of host LAPACK at n = 1e6")
def test_headline_parity_vs_oracle(gsi):
    ctx = gsi.default_context()
    n, Ns, K, p, q = 1000000, 1024, 256, 64, 2
    op = gsi.lowrank_synthetic_operator(ctx, n, Ns, seed=0, decay=0.75)
    Om = gsi.DeviceMatrix(ctx, n, K + p).randn(1234)
    Z = gsi.DeviceMatrix(ctx, n, K + p)
    S = gsi.DeviceMatrix(ctx, K + p, 1)
    gsi._lib.check(ctx.lib.gsi_randsvd_dev(ctx.h, op.h, Om.h, K, p, q, Z.h, S.h), ctx.lib)
    samples, Omh, Zh, Sh = gsi.device_samples(op, Ns), Om.to_host(), Z.to_host(), S.to_host()[:, 0]
    for h in (op, Om, Z, S):
        h.close()
    A = orc.LowRankCovMatrix(samples, gemm_form=True)
    del samples
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Omh)
    assert rel_sv_err(Sh, Sref, K) < 1e-9                      # bar 1e-5 (north_star)
    assert orc.xis_error_up_to_sign(Zh, Zref, K) < 1e-6        # test/testrpcga.jl:100
    assert np.all(Zh[:, K:] == 0.0)


# ---- lu_leaf_kernel<512, 8>: the instantiation the headline panel (m = 1e6) runs -- 524288 < m <= 1048576 rows -- against
#      dgetrf for pivots and L, a tie case included ---------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("m,l,ties", [(600001, 24, False), (1000000, 16, True)])
def test_lu_headline_height_matches_lapack(gsi, ctx, m, l, ties):
    rng = np.random.default_rng(m + l)
    Y = rng.standard_normal((m, l))
    if ties:                       # equal-magnitude maxima far apart, across workgroups: the lowest row must win
        for j in range(0, l, 3):
            r = rng.choice(m, size=3, replace=False)
            Y[r, j] = [7.5, -7.5, 7.5]
    L, piv = gsi.lu_L(Y, return_pivots=True, ctx=ctx)
    assert np.array_equal(piv, orc.lu_pivots(Y))
    Lref = orc.lu_L(Y)
    assert np.abs(L - Lref).max() < 1e-11 * max(1.0, np.abs(Lref).max())


# ---- panels taller than the register-resident path holds (> 4096 rows per CU = 1 048 576 rows): streamed leaves with lazily
#      evaluated candidates (panel_lu_leaf.hip: lu3_*).  (a) Forced onto panels the resident kernel also takes
#      (GSI_LU_TALL=1): the factors must be BIT-identical to the resident kernel's and the pivots dgetrf's -- ties, several
#      64-column blocks, a ragged last leaf, tiny panels; (b) a panel that only this path takes, against dgetrf. ------------
@pytest.mark.gpu
def test_lu_streamed_leaves_bit_identical_to_resident(gsi, tmp_path):
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gsi_amd as gsi
from oracle import oracle as orc
out = sys.argv[1]
ctx = gsi.Context(0)
res = {}
for idx, (m, l, ties) in enumerate([(100, 25, False), (9, 2, False), (64, 64, False), (5000, 160, True), (3000, 33, True),
                                    (300000, 136, True), (70001, 72, False), (600001, 24, True),
                                    # above 4096 rows per CU the default path is the resident kernel with lazily evaluated
                                    # OVERFLOW rows (panel_lu_leaf.hip, OV): also bit-identical to the streamed leaves
                                    (1100000, 72, True), (1048585, 16, True)]):
    rng = np.random.default_rng(m + l)
    Y = rng.standard_normal((m, l))
    if ties and m >= 3000:
        Y[1500:1600] = Y[100:200]
        for j in range(0, l, 5):
            r = rng.choice(m, size=3, replace=False)
            Y[r, j] = [9.5, -9.5, 9.5]
    L, piv = gsi.lu_L(Y, return_pivots=True, ctx=ctx)
    assert np.array_equal(piv, orc.lu_pivots(Y)), (m, l)
    if m <= 70001:
        Lref = orc.lu_L(Y)
        assert np.abs(L - Lref).max() < 1e-10 * max(1.0, np.abs(Lref).max()), (m, l)
    res["L%d" % idx] = L
    res["p%d" % idx] = piv
np.savez(out, **res)
print("lu-ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for tag, extra in (("resident", {}), ("streamed", {"GSI_LU_TALL": "1"})):
        env = dict(os.environ)
        env.update(extra)
        f = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0 and "lu-ok" in r.stdout, tag + "\n" + r.stdout[-2000:] + r.stderr[-4000:]
        outs.append(np.load(f))
    for k in outs[0].files:
        assert np.array_equal(outs[0][k], outs[1][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("m,l", [(1300000, 24), (1048577, 72)])
def test_lu_taller_than_the_register_file_matches_lapack(gsi, ctx, m, l):
    rng = np.random.default_rng(m + l)
    Y = rng.standard_normal((m, l))
    for j in range(0, l, 3):                       # equal-magnitude maxima far apart: the lowest row must win
        r = rng.choice(m, size=3, replace=False)
        Y[r, j] = [7.5, -7.5, 7.5]
    L, piv = gsi.lu_L(Y, return_pivots=True, ctx=ctx)
    assert np.array_equal(piv, orc.lu_pivots(Y))
    Lref = orc.lu_L(Y)
    assert np.abs(L - Lref).max() < 1e-11 * max(1.0, np.abs(Lref).max())
    assert np.all(np.triu(L[:l], 1) == 0.0) and np.all(np.diag(L[:l]) == 1.0)


# ---- the row-sharded LU with REAL row offsets on one GPU: G virtual ranks, every lus_* kernel launched with
#      its shard's row0 / mloc, the exchanges as device copies; bit-identical to the single-rank factorization.  Cases: a
#      pivot search that crosses shards, ties across the shard boundary (the lowest global row must win), a ragged last
#      shard, an empty last shard, several 64-column blocks (rank-64 updates with row0 != 0) -----------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("m,l,G", [(100, 25, 2), (625, 50, 3), (5000, 160, 3), (3000, 33, 4), (70001, 72, 8),
                                   (300000, 136, 2), (9, 2, 4), (1000, 100, 7)])
def test_lu_sharded_virtual_ranks_bit_identical(gsi, ctx, m, l, G):
    rng = np.random.default_rng(m + l + G)
    Y = rng.standard_normal((m, l))
    pad = -(-m // G)
    if m == 3000:                       # exact ties between rows of different shards
        Y[pad + 10:pad + 110] = Y[100:200]
        Y[2 * pad + 5:2 * pad + 55] = -Y[100:150]
    Lv, pv = gsi.lu_L_sharded_virtual(Y, G, return_pivots=True, ctx=ctx)
    L1, p1 = gsi.lu_L(Y, return_pivots=True, ctx=ctx)
    assert np.array_equal(pv, p1)
    assert np.array_equal(Lv, L1)
    if m <= 5000:
        assert np.array_equal(pv, orc.lu_pivots(Y))


# ---- the persistent leaf kernel's time-out path (info = -1): one workgroup stays silent at a pivot step (test knob), every
#      other workgroup runs out of polls, the launch drains, the call fails with GSI_ERR_INTERNAL (GSI_NO_RETRY) -- and
#      without that switch the entry point re-runs on the streamed leaves (no spin-waits) and returns LAPACK's factorization.  Then: two
#      contexts on ONE GPU factoring concurrently (plain launches may interleave their workgroups; whatever happens both
#      results must be dgetrf's). ---------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_lu_lost_coresidency_paths(gsi):
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, threading, numpy as np
sys.path.insert(0, os.getcwd())
import gsi_amd as gsi
from oracle import oracle as orc
rng = np.random.default_rng(11)
Y = rng.standard_normal((40000, 40))
pref, Lref = orc.lu_pivots(Y), orc.lu_L(Y)
os.environ["GSI_LU_POLL_LIMIT"] = "20000"
os.environ["GSI_LU_TEST_MUTE_EPOCH"] = "13"
os.environ["GSI_NO_RETRY"] = "1"
ctx = gsi.Context(0)
try:
    gsi.lu_L(Y, ctx=ctx)
    raise SystemExit("the muted exchange did not time out")
except gsi.GsiError as e:
    assert e.code == 8 and "timed out" in str(e), str(e)
L2, p2 = gsi.lu_L(Y, return_pivots=True, ctx=ctx)          # the context has switched to the streamed leaves
assert np.array_equal(p2, pref) and np.abs(L2 - Lref).max() < 1e-11
del os.environ["GSI_NO_RETRY"]
ctx3 = gsi.Context(0)                                       # fresh context: time-out, then the automatic re-run
L3, p3 = gsi.lu_L(Y, return_pivots=True, ctx=ctx3)
assert np.array_equal(p3, pref) and np.abs(L3 - Lref).max() < 1e-11
A = rng.standard_normal((3000, 40)) @ rng.standard_normal((40, 3000))
Om = rng.standard_normal((3000, 24))
ctx4 = gsi.Context(0)
Z4, S4 = gsi.randsvd(A, 16, 8, 2, Omega=Om, return_S=True, ctx=ctx4)   # time-out inside randsvd -> re-run on its inputs
del os.environ["GSI_LU_TEST_MUTE_EPOCH"]
ctx5 = gsi.Context(0)
Z5, S5 = gsi.randsvd(A, 16, 8, 2, Omega=Om, return_S=True, ctx=ctx5)
assert np.abs(S4 - S5).max() < 1e-11 * S5[0]
print("abort-path-ok")
# two contexts, one GPU, concurrent factorizations
os.environ["GSI_LU_POLL_LIMIT"] = "400000"
Yb = rng.standard_normal((1000000, 16))      # <512, 8> leaves: one workgroup per CU, 245 of 256 CUs each -- two do not fit
pb = orc.lu_pivots(Yb)
out = {}
def work(tag):
    c = gsi.Context(0)
    for _ in range(3):
        out[tag] = gsi.lu_L(Yb, return_pivots=True, ctx=c)[1]
ts = [threading.Thread(target=work, args=(t,)) for t in ("a", "b")]
[t.start() for t in ts]; [t.join() for t in ts]
assert np.array_equal(out["a"], pb) and np.array_equal(out["b"], pb)
print("two-contexts-ok")
'''
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "abort-path-ok" in r.stdout and "two-contexts-ok" in r.stdout, \
        r.stdout[-2000:] + r.stderr[-4000:]



# ---- scattered-point covariance as an implicit, row-streamed operator (SURVEY 8b "coords + kernel id + params"): products
#      and randsvd against the dense kernel matrix built by numpy from the same coordinates; every kernel kind, 1-3
#      dimensions, a nugget -----------------------------------------------------------------------------------------------
def _dense_pointcov(P, kind, ell, sigma2, nugget):
    d2 = ((P[:, :, None] - P[:, None, :]) ** 2).sum(axis=0)
    r = np.sqrt(d2) / ell
    if kind == "gaussian":
        K = np.exp(-0.5 * r * r)
    elif kind == "exponential":
        K = np.exp(-r)
    elif kind == "matern32":
        K = (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)
    else:
        K = (1.0 + np.sqrt(5.0) * r + 5.0 * r * r / 3.0) * np.exp(-np.sqrt(5.0) * r)
    return sigma2 * K + nugget * np.eye(P.shape[1])


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,kind,l", [(300, 2, "exponential", 7), (1000, 3, "gaussian", 48), (777, 1, "matern32", 33),
                                         (2500, 2, "matern52", 160), (4097, 2, "exponential", 320),
                                         # the 64-row x 320-column kernel (pointcov_gemm.hip: 160 < l): ragged widths, three
                                         # dimensions, fewer rows than one tile, two column chunks, an even leading dimension
                                         (1500, 3, "matern52", 250), (3000, 2, "gaussian", 400), (50, 2, "exponential", 170),
                                         (5000, 1, "matern32", 192), (2048, 3, "exponential", 320),
                                         # the 192-row x 160-column arrangement (97 <= l <= 160): ragged widths on both of its
                                         # column counts (128, 160), fewer rows than one tile, three dimensions, a nugget
                                         (1000, 2, "exponential", 97), (3001, 3, "gaussian", 128), (190, 2, "matern52", 144),
                                         (4099, 1, "exponential", 150)])
def test_pointcov_implicit_products(gsi, ctx, n, d, kind, l):
    rng = np.random.default_rng(n + l)
    P = rng.uniform(0.0, 30.0, size=(d, n))
    ell, sigma2, nugget = 4.0, 2.5, (0.1 if kind == "exponential" else 0.0)
    A = _dense_pointcov(P, kind, ell, sigma2, nugget)
    op = gsi.pointcov_implicit_operator(ctx, P, kind, ell=ell, sigma2=sigma2, nugget=nugget)
    X = rng.standard_normal((n, l))
    Y = op.matmul(X)
    assert np.abs(Y - A @ X).max() < 1e-11 * np.abs(A @ X).max()
    assert np.abs(op.rmatmul_t(X) - A @ X).max() < 1e-11 * np.abs(A @ X).max()
    op.close()


@pytest.mark.gpu
@pytest.mark.parametrize("l", [32, 200])           # the 128 x 160 kernel (GEN 2) and the 96 x 320 one
@pytest.mark.parametrize("kind", ["exponential", "gaussian", "matern52"])
def test_pointcov_entry_extremes(gsi, ctx, kind, l):
    """The in-loader entry at the ends of its range (pointcov_gen.hpp: no clamp in front of the exponential, r^2 + 1e-280 in
    front of the square root): coincident points off the diagonal, arguments far beyond exp's underflow (the exponent must
    saturate to 0, not wrap), arguments ~ 1e-9 (everything ~ sigma^2), sigma^2 at both ends of the double range."""
    rng = np.random.default_rng(7)
    n = 700
    P = rng.uniform(0.0, 1.0, size=(2, n))
    P[:, 100:120] = P[:, 300:320]                       # twenty coincident pairs
    X = rng.standard_normal((n, l))
    for scale, ell, sigma2 in ((1.0e6, 1.0e-3, 3.0), (1.0, 1.0e9, 1.0e-30), (50.0, 1.0, 1.0e30), (1.0e12, 1.0, 1.0)):
        Ps = P * scale
        A = _dense_pointcov(Ps, kind, ell, sigma2, 0.0)
        op = gsi.pointcov_implicit_operator(ctx, Ps, kind, ell=ell, sigma2=sigma2)
        Y = op.matmul(X)
        op.close()
        ref = A @ X
        assert np.isfinite(Y).all()
        assert np.abs(Y - ref).max() <= 1e-11 * np.abs(ref).max(), (kind, l, scale, ell, sigma2)


@pytest.mark.gpu
def test_pointcov_implicit_randsvd(gsi, ctx):
    rng = np.random.default_rng(5)
    n, K, p, q = 3000, 40, 10, 2
    P = rng.uniform(0.0, 50.0, size=(2, n))
    A = _dense_pointcov(P, "exponential", 12.0, 1.0, 0.0)
    op = gsi.pointcov_implicit_operator(ctx, P, "exponential", ell=12.0)
    X = rng.standard_normal((n, 24))
    assert np.abs(op.matmul(X) - A @ X).max() < 1e-11 * np.abs(A @ X).max()
    Om = rng.standard_normal((n, K + p))
    Z, S = gsi.randsvd(op, K, p, q, Omega=Om, return_S=True)
    op.close()
    Zr, Sr, _ = orc.randsvd_full(A, K, p, q, Om)
    assert rel_sv_err(S, Sr, K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, Zr, K) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind,l", [("exponential", 48), ("gaussian", 320), ("matern52", 200)])
def test_pointcov_translation_invariance(gsi, ctx, kind, l):
    """Entries depend on coordinate differences only: UTM-like coordinates -- a common offset of 5e5 / 4.6e6 on points
    that are tens of correlation lengths apart, and an ell whose reciprocal is not a power of two -- must cost no accuracy:
    the same 1e-12 against the dense matrix as the points centred at the origin, and the two operators agree to rounding."""
    rng = np.random.default_rng(17)
    n, ell = 1800, 20.0
    P0 = rng.uniform(0.0, 600.0, size=(2, n))
    P1 = P0 + np.array([[5.0e5], [4.6e6]])
    X = rng.standard_normal((n, l))
    A = _dense_pointcov(P0, kind, ell, 1.0, 0.0)
    ref = A @ X
    Y = []
    for P in (P0, P1):
        op = gsi.pointcov_implicit_operator(ctx, P, kind, ell=ell)
        Y.append(op.matmul(X))
        op.close()
    # (P1 - offset is P0 only to rounding of the offset addition itself: ~1e-10 absolute on a coordinate = 5e-12 of ell)
    A1 = _dense_pointcov(P1, kind, ell, 1.0, 0.0)
    assert np.abs(Y[0] - ref).max() < 1e-12 * np.abs(ref).max()
    assert np.abs(Y[1] - A1 @ X).max() < 1e-12 * np.abs(ref).max()


@pytest.mark.gpu
def test_pointcov_rejects_unbounded_coordinates(gsi, ctx):
    """The generator has no clamp in front of its exponential; the accepted range is checked on the host (gsi_hip.h)."""
    P = np.random.default_rng(1).uniform(0, 1, size=(2, 50))
    for bad in (np.inf, np.nan, 1e300):
        Pb = P.copy()
        Pb[1, 17] = bad
        with pytest.raises(gsi.GsiError, match="finite and within"):
            gsi.pointcov_implicit_operator(ctx, Pb, "exponential", ell=1.0)
    # far, but inside the range: the entries underflow to exactly zero, nothing is poisoned
    Pf = P.copy()
    Pf[:, 3] = 1e20
    op = gsi.pointcov_implicit_operator(ctx, Pf, "gaussian", ell=1.0)
    X = np.random.default_rng(2).standard_normal((50, 8))
    Y = op.matmul(X)
    op.close()
    A = _dense_pointcov(Pf, "gaussian", 1.0, 1.0, 0.0)
    assert np.isfinite(Y).all() and np.abs(Y - A @ X).max() < 1e-11 * np.abs(A @ X).max()


# ---- the WHOLE multi-rank pipeline on the HIP backend, on one GPU.  RCCL refuses two ranks on one device, so the ranks are
#      joined by the library's two RCCL-free communicators:
#        GSI_LOCAL_COMM=1 -- ranks as THREADS of one process, one context each;
#        GSI_SHM_COMM=1   -- ranks as PROCESSES (one per rank, as on a multi-GPU node): host barriers in POSIX shared memory,
#                            every device buffer a peer touches mapped with hipIpcOpenMemHandle -- the staging buffers of
#                            the collectives and, what this is for, the pivot-exchange buffer the persistent LU leaves of
#                            DIFFERENT processes write into and poll.
#      Every kernel runs with real row offsets and real exchanges between the ranks' buffers: row-sharded products, the
#      sharded LU, TSQR, the FFT operator's all-to-alls, gsi_randsvd_rows, the implicit operators' transposed products, a
#      row-sharded xi-basis.  (tests/test_distributed_gloo.py runs the same host code over the CPU reference backend.) ------
_MULTIRANK_BODY = r'''
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import hashlib
import numpy as np
import gsi_amd as gsi
from oracle import oracle as orc
from helpers import gaussian_cov, powerlaw_fields, rel_sv_err

def run_rank(rank, world, ctx, exchange, out):
    # exchange(key, array) -> the ranks' arrays in rank order (a host-side all-gather supplied by the harness)
    gather_rows = lambda key, loc: np.concatenate(exchange(key, np.ascontiguousarray(loc)), axis=0)
    assert ctx.rank() == (rank, world)
    rng = np.random.default_rng(7)
    def my_rows(full):
        r0, nl = ctx.shard(full.shape[0])
        return np.asfortranarray(full[r0:r0 + nl])
    A = gaussian_cov(23, 17, 3.0)                              # n = 391, not divisible by the world size
    for qq in (0, 2):
        K, p = 20, 12
        Om = rng.standard_normal((391, K + p))
        Z, S = gsi.randsvd(A, K, p, qq, Omega=Om, return_S=True, ctx=ctx)
        Zr, Sr, _ = orc.randsvd_full(A, K, p, qq, Om)
        out[f"dense_q{qq}_sv"] = rel_sv_err(S, Sr, K)
        out[f"dense_q{qq}_xis"] = orc.xis_error_up_to_sign(Z, Zr, K)
    fields = powerlaw_fields(rng, (21, 19), 40)                # n = 399
    Om = rng.standard_normal((399, 24))
    lr = gsi.LowRankCovMatrix(fields, ctx=ctx)
    Z = gsi.randsvd(lr, 16, 8, 3, Omega=Om)                    # sharded LU + TSQR on the HIP kernels, row0 != 0
    xr, _ = orc.getxis_fields(fields, 16, 8, 3, Om)
    out["lowrank_xis"] = orc.xis_error_up_to_sign(Z, np.array(xr).T, 16)
    Zrows = gsi.randsvd_rows(lr._device_operator(), 16, 8, 3, my_rows(Om))
    out["lowrank_rows_xis"] = orc.xis_error_up_to_sign(gather_rows("lr", Zrows.to_host()), np.array(xr).T, 16)
    Zrows.close(); lr.close()
    Yp = rng.standard_normal((5000, 72)); Yp[3000:3100] = Yp[100:200]
    Ls, ps = gsi.lu_L_sharded(Yp, return_pivots=True, ctx=ctx)     # persistent leaves, pivot exchange between the ranks' kernels
    out["lu_pivots"] = 0.0 if np.array_equal(ps, orc.lu_pivots(Yp)) else 1.0
    out["lu_L"] = float(np.abs(Ls - orc.lu_L(Yp)).max())
    Yq = rng.standard_normal((300000, 136))                        # <512, 4> leaves, three 64-column blocks, shards of 1e5+ rows
    for jj in range(0, 136, 5):
        rr = rng.choice(300000, size=3, replace=False)
        Yq[rr, jj] = [9.5, -9.5, 9.5]
    Lq, pq = gsi.lu_L_sharded(Yq, return_pivots=True, ctx=ctx)
    out["lu_big_pivots"] = 0.0 if np.array_equal(pq, orc.lu_pivots(Yq)) else 1.0
    out["lu_big_L"] = float(np.abs(Lq - orc.lu_L(Yq)).max())
    digest = np.frombuffer(hashlib.sha256(np.ascontiguousarray(Lq).tobytes()).digest(), dtype=np.uint8)
    out["lu_big_same_on_all_ranks"] = 0.0 if all(np.array_equal(d, digest) for d in exchange("Lq", digest)) else 1.0
    Ns, beta = [25, 18], -3.5                                  # FFTRF convention on a grid that is not a power of two
    nf = 450
    Af = orc.fft_powerlaw_apply(np.eye(nf), Ns, beta, fftrf=True)
    fop = gsi.fft_powerlaw_operator(ctx, Ns, beta, fftrf=True)
    X = rng.standard_normal((nf, 9))
    out["fft_mul"] = float(np.abs(fop.matmul(X) - Af @ X).max())
    out["fft_mul_t"] = float(np.abs(fop.rmatmul_t(X) - Af @ X).max())
    K, p, qq = 20, 10, 2
    Om = rng.standard_normal((nf, K + p))
    Zr, Sr, _ = orc.randsvd_full(Af, K, p, qq, Om)
    Zrows, S2 = gsi.randsvd_rows(fop, K, p, qq, my_rows(Om), return_S=True)
    out["fft_rows_sv"] = rel_sv_err(S2, Sr, K)
    out["fft_rows_xis"] = orc.xis_error_up_to_sign(gather_rows("fft", Zrows.to_host()), Zr, K)
    Zrows.close(); fop.close()
    G = gaussian_cov(19, 13, 2.5)                              # n = 247: implicit operators, transposed products sharded
    gop = gsi.gridcov_implicit_operator(ctx, 19, 13, 2.5)
    X = rng.standard_normal((247, 5))
    out["implicit_mul"] = float(np.abs(gop.matmul(X) - G @ X).max())
    out["implicit_mul_t"] = float(np.abs(gop.rmatmul_t(X) - G @ X).max())
    gop.close()
    Pp = rng.uniform(0.0, 20.0, size=(2, 333))
    dd = np.sqrt(((Pp[:, :, None] - Pp[:, None, :]) ** 2).sum(axis=0)) / 5.0
    Ap = np.exp(-dd)
    pop = gsi.pointcov_implicit_operator(ctx, Pp, "exponential", ell=5.0)
    X = rng.standard_normal((333, 4))
    out["pointcov_mul"] = float(np.abs(pop.matmul(X) - Ap @ X).max())
    out["pointcov_mul_t"] = float(np.abs(pop.rmatmul_t(X) - Ap @ X).max())
    Xw = rng.standard_normal((333, 200))                       # 160 < l: the 64 x 320-tile kernel with row / reduction offsets
    out["pointcov_wide_mul"] = float(np.abs(pop.matmul(Xw) - Ap @ Xw).max())
    out["pointcov_wide_mul_t"] = float(np.abs(pop.rmatmul_t(Xw) - Ap @ Xw).max())
    Om = rng.standard_normal((333, 24))
    Z, S = gsi.randsvd(pop, 16, 8, 2, Omega=Om, return_S=True)
    Zr, Sr, _ = orc.randsvd_full(Ap, 16, 8, 2, Om)
    out["pointcov_sv"] = rel_sv_err(S, Sr, 16)
    pop.close()
    # BASELINE configs[4] on HIP kernels: pcgalsqr over a row-sharded xi-basis against the oracle
    Np, Mp = 192, 8
    xs = rng.standard_normal(Np); Q0 = rng.standard_normal((Mp, Np)); Qc = Q0.T @ Q0
    truep = np.linalg.cholesky(Qc + 1e-9 * np.eye(Np)) @ rng.standard_normal(Np) + 1.0
    forward = lambda pv: pv * xs
    yobs = forward(truep) + 1e-4 * rng.standard_normal(Np)
    import scipy.sparse as sp
    Rn = 1e-8 * sp.identity(Np, format="csc")
    Omp = rng.standard_normal((Np, Mp + 2))
    qop = gsi.dense_operator(ctx, Qc)
    Zp = gsi.randsvd_rows(qop, Mp, 2, 3, my_rows(Omp))
    basis = gsi.ShardedDeviceBasis(Zp, Mp, lambda loc: gather_rows("pcga", loc))
    X0 = np.full(Np, 1.0)
    r0p, nlp = ctx.shard(Np)
    s_loc = gsi.pcgalsqr(forward, X0[r0p:r0p + nlp], X0[r0p:r0p + nlp], basis, Rn, yobs, ctx=ctx)
    s_full = gather_rows("s", s_loc)
    s_ref = orc.pcgalsqr(forward, X0, X0, orc.getxis_dense(Qc, Mp, 2, 3, Omp), Rn, yobs)
    out["pcgalsqr_sharded_basis"] = float(np.linalg.norm(s_full - s_ref) / np.linalg.norm(s_ref))
    basis.close(); Zp.close(); qop.close()

def check_rank(r, out):
    for k, v in out.items():
        tol = 1e-6 if (k.endswith("xis") or k == "pcgalsqr_sharded_basis") else 1e-9
        if k.endswith("mul") or k.endswith("mul_t") or k in ("lu_L", "lu_big_L"):
            tol = 1e-10
        assert v < tol, (r, k, v)
'''

_MULTIRANK_THREADS = r'''
import threading, traceback
world = int(sys.argv[1])
ctx0 = gsi.Context(0)
uid = ctx0.unique_id()
res, errs = [dict() for _ in range(world)], []
bar = threading.Barrier(world)
box = {}

def make_exchange(rank):
    def exchange(key, arr):
        box[(key, rank)] = arr
        bar.wait()
        parts = [box[(key, r)] for r in range(world)]
        bar.wait()
        return parts
    return exchange

def run(rank):
    try:
        ctx = ctx0 if rank == 0 else gsi.Context(0)
        ctx.comm_init(world, rank, uid)
        run_rank(rank, world, ctx, make_exchange(rank), res[rank])
        if rank != 0:
            ctx.close()
    except Exception:
        errs.append((rank, traceback.format_exc()))
        try:
            bar.abort()
        except Exception:
            pass

ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in ts]; [t.join() for t in ts]
if errs:
    print(errs[0][1]); raise SystemExit(1)
for r in range(world):
    check_rank(r, res[r])
    assert res[r].keys() == res[0].keys()
print("multirank-one-gpu-ok", len(res[0]))
'''

_MULTIRANK_PROCESS = r'''
import time, traceback
world, rank, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seq = {}
def wait_for(path, what):
    t0 = time.time()
    while not os.path.exists(path):
        if os.path.exists(os.path.join(tmp, "failed")):
            raise RuntimeError("another rank failed")
        if time.time() - t0 > 600:
            raise RuntimeError("timed out waiting for " + what)
        time.sleep(0.005)
def exchange(key, arr):                    # host-side all-gather between the rank processes: files in the test's directory
    seq[key] = seq.get(key, 0) + 1
    stem = os.path.join(tmp, "%s_%d" % (key, seq[key]))
    np.save(stem + "_%d.tmp.npy" % rank, arr)
    os.rename(stem + "_%d.tmp.npy" % rank, stem + "_%d.npy" % rank)
    parts = []
    for r in range(world):
        wait_for(stem + "_%d.npy" % r, "rank %d's %s" % (r, key))
        parts.append(np.load(stem + "_%d.npy" % r))
    return parts
try:
    ctx = gsi.Context(0)
    idfile = os.path.join(tmp, "uid")
    if rank == 0:                          # rank 0 makes the id and ships it to the others (here: a file)
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ctx.unique_id()))
        os.rename(idfile + ".tmp", idfile)
    wait_for(idfile, "the communicator id")
    with open(idfile, "rb") as f:
        uid = f.read()
    ctx.comm_init(world, rank, uid)
    out = {}
    run_rank(rank, world, ctx, exchange, out)
    check_rank(rank, out)
    assert all(int(k[0]) == len(out) for k in exchange("keys", np.array([len(out)])))
    ctx.close()
    print("multirank-processes-ok", rank, len(out))
except Exception:
    with open(os.path.join(tmp, "failed"), "w") as f:
        f.write(traceback.format_exc())
    traceback.print_exc()
    raise SystemExit(1)
'''


def _spawn_multirank(code, args, env_extra):
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(env_extra)
    env["GSI_LU_MR_REQUIRE"] = "1"       # the sharded LU must run its persistent leaves with the in-kernel exchange between ranks
    return subprocess.Popen([sys.executable, "-c", _MULTIRANK_BODY + code] + [str(a) for a in args], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, env=env,
                            cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _join_multirank(procs, marker, timeout=900):
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=timeout))
    except Exception:
        for pr in procs:
            pr.kill()
        raise
    for r, (pr, (so, se)) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and marker in so, "rank %d\n" % r + so[-3000:] + se[-4000:]


# hier: the sharded LU's TWO-hop pivot exchange (ranks reduce among their own workgroups, then exchange one record per rank:
# what shards of more than 256 / nranks workgroups take -- 10^6 rows per rank), forced onto these small shards
# "ov": at most 8 workgroups per rank, so that the 300 000 x 136 panel's shards are TALLER than the resident window and the
# persistent leaves evaluate their overflow rows lazily (what shards of more than ~10^6 rows per rank take)
@pytest.mark.gpu
@pytest.mark.parametrize("world,hier", [(2, False), (3, False), (2, True), (2, "ov")])
def test_multirank_pipeline_on_one_gpu(gsi, world, hier):
    env = {"GSI_LOCAL_COMM": "1"}
    if hier == "ov":
        env["GSI_LU_MR_OV_GRID"] = "8"
    elif hier:
        env["GSI_LU_MR_HIER"] = "1"
    _join_multirank([_spawn_multirank(_MULTIRANK_THREADS, [world], env)], "multirank-one-gpu-ok")


@pytest.mark.gpu
@pytest.mark.parametrize("world,hier", [(2, False), (3, False), (3, True), (3, "ov")])
def test_multirank_processes_on_one_gpu(gsi, world, hier, tmp_path):
    """One PROCESS per rank, all on GPU 0: what the rank threads cannot cover -- hipIpc mappings of another process's buffers,
    and persistent kernels of different processes exchanging pivots through them."""
    env = {"GSI_SHM_COMM": "1", "GSI_SHM_TIMEOUT_S": "120"}
    if hier == "ov":
        env["GSI_LU_MR_OV_GRID"] = "8"
    elif hier:
        env["GSI_LU_MR_HIER"] = "1"
    _join_multirank([_spawn_multirank(_MULTIRANK_PROCESS, [world, r, str(tmp_path)], env) for r in range(world)],
                    "multirank-processes-ok")


@pytest.mark.gpu
def test_shared_memory_communicator_single_rank_and_foreign_id(gsi):
    """GSI_SHM_COMM=1 with ONE rank (every collective degenerates to a copy through the staging buffer) and an id that
    did not come from gsi_comm_unique_id under that switch (must be refused, not dereferenced as a name)."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gsi_amd as gsi
from oracle import oracle as orc
from helpers import gaussian_cov, rel_sv_err
ctx = gsi.Context(0)
try:
    ctx.comm_init(1, 0, bytes(128))
    raise SystemExit("a foreign id was accepted")
except gsi.GsiError as e:
    assert "gsi_comm_unique_id" in str(e), str(e)
ctx.comm_init(1, 0, ctx.unique_id())
assert ctx.rank() == (0, 1)
A = gaussian_cov(20, 15, 3.0)
Om = np.random.default_rng(0).standard_normal((300, 24))
Z, S = gsi.randsvd(A, 16, 8, 2, Omega=Om, return_S=True, ctx=ctx)
Zr, Sr, _ = orc.randsvd_full(A, 16, 8, 2, Om)
assert rel_sv_err(S, Sr, 16) < 1e-9
Y = np.random.default_rng(1).standard_normal((4000, 40))
L, p = gsi.lu_L_sharded(Y, return_pivots=True, ctx=ctx)
assert np.array_equal(p, orc.lu_pivots(Y))
print("shm-single-ok")
"""
    env = dict(os.environ)
    env["GSI_SHM_COMM"] = "1"
    env["GSI_FORCE_COMM"] = "1"          # a 1-rank job gets no communicator otherwise
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "shm-single-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- BASELINE configs[1] as configured: dense fp64 65536 x 65536 Gaussian covariance (256 x 256 grid, ell = 16), K = 128,
#      p = 32, q = 2 -- HIP vs the oracle (dgemm / dgetrf / dgeqp3 / dgesdd) on the same host matrix and Omega: a 34 GB host
#      matrix, its upload and ~20 s of host LAPACK (32 s in all on the GPU box; GSI_SKIP_C2_PARITY=1 skips it) -------------
@pytest.mark.gpu
@pytest.mark.skipif(bool(__import__("os").environ.get("GSI_SKIP_C2_PARITY")), reason="GSI_SKIP_C2_PARITY set")
def test_c2_parity_vs_oracle(gsi):
    ctx = gsi.default_context()
    g, ell, K, p, q = 256, 16.0, 128, 32, 2
    n = g * g
    ex = np.exp(-(np.arange(g, dtype=np.float64) ** 2) / (2.0 * ell * ell))
    T = ex[np.abs(np.arange(g)[:, None] - np.arange(g)[None, :])]              # g x g Toeplitz factor of one axis
    A = np.empty((n, n), order="F")
    for bx in range(g):                                                        # A = T (x) T, block row by block row (point = x * g + y)
        A[bx * g:(bx + 1) * g, :] = np.kron(T[bx:bx + 1, :], T)
    rng = np.random.default_rng(2)
    Om = np.asfortranarray(rng.standard_normal((n, K + p)))
    Z, S = gsi.randsvd(A, K, p, q, Omega=Om, return_S=True, ctx=ctx)
    Zref, Sref, _ = orc.randsvd_full(A, K, p, q, Om)
    assert rel_sv_err(S, Sref, K) < 1e-9
    # the 256 x 256 grid is exactly symmetric: singular values come in (near-)equal pairs, whose vectors are determined only
    # up to a rotation within the pair -- compare the projectors of the leading subspace instead of single vectors
    k2 = 64
    P1 = Z[:, :k2] / np.sqrt(S[:k2])
    P2 = Zref[:, :k2] / np.sqrt(Sref[:k2])
    gap_ok = Sref[k2 - 1] - Sref[k2] > 1e-6 * Sref[0]
    if gap_ok:
        assert np.linalg.norm(P1 - P2 @ (P2.T @ P1)) < 1e-6
    assert np.all(Z[:, K:] == 0.0)
