"""Pin the CPU oracle against every known-answer / property test the reference's own
test-suite holds for the RandMatFact path.  CPU only."""
import numpy as np
import pytest
import scipy.linalg as sl
import scipy.sparse as sp

from oracle import oracle as orc
from helpers import exact_rank_matrix, powerlaw_fields


# ---- test/testrmf.jl:11-19, 31-36 -------------------------------------------------
@pytest.mark.parametrize("n,m", [(10, 2), (10, 5), (100, 5), (100, 10), (100, 25)])
def test_rangefinder_exact_rank(n, m):
    rng = np.random.default_rng(100 * n + m)
    A = exact_rank_matrix(rng, n, m)
    Q = orc.rangefinder_adaptive(A, rng.standard_normal)
    assert abs(Q.shape[1] - m) <= 1
    assert np.linalg.norm(A - Q @ Q.T @ A) < 1e-8
    Q = orc.rangefinder(A, m, 2, rng.standard_normal((n, m)))
    assert abs(Q.shape[1] - m) <= 1
    assert np.linalg.norm(A - Q @ Q.T @ A) < 1e-8


# ---- test/testrmf.jl:21-29 ---------------------------------------------------------
def test_eig_nystrom_kat():
    A = np.array([[2.0, -1, 0], [-1, 2, -1], [0, -1, 2]])
    rng = np.random.default_rng(7)
    Q = orc.rangefinder_adaptive(A, rng.standard_normal)
    U, Sigmavec = orc.eig_nystrom(A, Q)
    lam = Sigmavec ** 2
    kat = np.array([2 + np.sqrt(2), 2.0, 2 - np.sqrt(2)])
    assert np.linalg.norm(np.sort(np.linalg.eigvalsh(A))[::-1] - lam) < 1e-8
    assert np.linalg.norm(kat - lam) < 1e-8


# ---- lu(Y).L semantics (Julia LinearAlgebra.LU; RandMatFact.jl:60-61) -----------------
def test_lu_L_is_pivoted_row_order():
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((40, 7))
    L = orc.lu_L(Y)
    P, L2, U = sl.lu(Y)                       # Y = P @ L2 @ U  ->  L2 @ U = P.T @ Y
    assert np.allclose(L, L2, atol=1e-14)
    piv = orc.lu_pivots(Y)
    perm = np.arange(40)
    for i, pi in enumerate(piv):
        perm[[i, pi]] = perm[[pi, i]]
    assert np.allclose(L @ U, Y[perm, :], atol=1e-12)
    assert np.allclose(np.diag(L[:7]), 1.0) and np.allclose(np.triu(L[:7], 1), 0.0)
    assert np.abs(L).max() <= 1.0 + 1e-15


# ---- test/testrpcga.jl:46-58 -------------------------------------------------------
def test_lowrankcov_kat():
    samples = [[-.5, 0., .5], [1., -1., 0.], [-.5, 1., -.5]]
    lrcm = orc.LowRankCovMatrix(samples)
    fullcm = lrcm.todense()
    assert np.allclose(fullcm, lrcm.matmul(np.eye(3)))
    S = np.asarray(samples)
    assert np.allclose(sum(np.outer(x, x) for x in S) / (len(S) - 1), fullcm)
    assert np.allclose(fullcm, [[.75, -.75, 0], [-.75, 1, -.25], [0, -.25, .25]])
    rng = np.random.default_rng(0)
    for _ in range(100):
        x = rng.standard_normal((3, 3))
        assert np.allclose(fullcm @ x, lrcm.matmul(x))
        assert np.allclose(fullcm.T @ x, lrcm.matmul(x))


# ---- test/testrpcga.jl:60-81 -------------------------------------------------------
def test_lowrankcov_consistency():
    rng = np.random.default_rng(2017)
    N, M = 10000, 100
    sqrtcov = rng.standard_normal((M, M))
    cov = sqrtcov @ sqrtcov.T
    samples = (sqrtcov @ rng.standard_normal((M, N))).T
    lrcm = orc.LowRankCovMatrix(samples)
    # the N-rank-1 loop of lowrank.jl:115-121 is O(N) python iterations; use the same algebra densely
    Sc = lrcm.samples
    full = Sc.T @ Sc / (N - 1)
    assert np.linalg.norm(full - cov, 2) < M ** 2 / np.sqrt(N) + 10
    x = rng.standard_normal(M)
    assert np.allclose(lrcm.matmul(x), full @ x)


# ---- test/testrpcga.jl:83-102 ------------------------------------------------------
def test_getxis_lrcm_vs_dense_same_omega():
    rng = np.random.default_rng(0)
    numfields, numxis, p, q = 100, 30, 20, 3
    fields = powerlaw_fields(rng, (25, 25), numfields)
    Omega = rng.standard_normal((625, numxis + p))      # "seed = 0 for both calls"
    lrcmxis, _ = orc.getxis_fields(fields, numxis, p, q, Omega)
    fullcm = orc.LowRankCovMatrix(fields).todense()
    fullxis = orc.getxis_dense(fullcm, numxis, p, q, Omega)
    for a, b in zip(fullxis, lrcmxis):
        assert min(np.linalg.norm(a - b), np.linalg.norm(a + b)) < 1e-6


# ---- test/testrpcga.jl:10-44 -------------------------------------------------------
def test_pcgalowrank_operator():
    rng = np.random.default_rng(5)
    numetas, numobs = 10, 20
    for noise in (1e16, 0.0):
        for etagen in (np.zeros, rng.standard_normal):
            for hxgen in (np.zeros, rng.standard_normal):
                etas = [etagen(numobs) for _ in range(numetas)]
                HQH = sum(np.outer(e, e) for e in etas)
                HX = hxgen(numobs)
                R = noise * sp.identity(numobs, format="csc")
                bigA = np.block([[HQH + R.toarray(), HX[:, None]], [HX[None, :], np.zeros((1, 1))]])
                lr = orc.PCGALowRankMatrix(etas, HX, R)
                assert lr.shape == (numobs + 1, numobs + 1)
                for i in range(numobs + 1):
                    x = np.zeros(numobs + 1)
                    x[i] = 1.0
                    assert np.allclose(bigA @ x, lr.matvec(x))


# ---- test/testrpcga.jl:104-131 (a reduced sweep of the end-to-end PCGA bound) -----------
def _setup_simple(rng, M, N, mu):
    x = rng.standard_normal(N)
    Q0 = rng.standard_normal((M, N))
    Q = Q0.T @ Q0
    w, V = np.linalg.eigh(Q)
    sqrtQ = (V * np.sqrt(np.clip(w, 0, None))) @ V.T
    truep = sqrtQ @ rng.standard_normal(N) + mu
    forward = lambda p: p * x
    K, p_ = M, int(round(0.1 * M))
    Omega = rng.standard_normal((N, K + p_))
    xis = orc.getxis_dense(Q, K, p_, 3, Omega)
    X = np.full(N, float(mu))
    noise = 1e-4
    R = noise ** 2 * sp.identity(N, format="csc")
    yobs = forward(truep) + noise * rng.standard_normal(N)
    return forward, np.full(N, float(mu)), X, xis, R, yobs, truep


@pytest.mark.parametrize("M,N,mu", [(1, 4, 0.0), (2, 16, 10.0), (8, 64, 0.0), (16, 128, 10.0), (32, 256, 0.0)])
def test_pcga_end_to_end(M, N, mu):
    rng = np.random.default_rng(2017 + M + N)
    forward, p0, X, xis, R, yobs, truep = _setup_simple(rng, M, N, mu)
    popt = orc.pcgadirect(forward, p0, X, xis, R, yobs)
    assert np.linalg.norm(popt - truep) / np.linalg.norm(truep) < 2e-2
    if M < N / 6:
        popt = orc.pcgalsqr(forward, p0, X, xis, R, yobs)
        assert np.linalg.norm(popt - truep) / np.linalg.norm(truep) < 2e-2
