"""CPU oracle for the RandMatFact hot path of GeostatInversion.jl.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product path
(``geostatinversion.jl_amd``) never imports this module and fails loudly when
its HIP library is missing.

What this is: a numpy/scipy restatement of the reference algorithm, operation
for operation, calling the *same LAPACK routines* the Julia reference reaches
through ``LinearAlgebra`` (Julia stdlib -> bundled OpenBLAS/LAPACK; here scipy's
bundled OpenBLAS 0.3.29):

    A*Omega, A'*Q, Q'*A      dgemm           RandMatFact.jl:55,67,70,85
    lu(Y).L                  dgetrf          RandMatFact.jl:60-61,68-69,72-73
    qr(Y, Val(true)) -> Q    dgeqp3+dorgqr   RandMatFact.jl:57-58,75-76
    svd(B)                   dgesdd (thin)   RandMatFact.jl:86
    cholesky(Hermitian(B2))  dpotrf          RandMatFact.jl:95

Pinning: the reference is 100 % Julia and ``julia`` is not
installed in this image, so the reference itself cannot be run here.  It ships
no golden vectors.  The oracle is therefore pinned by every known-answer and
property test the reference's own test-suite holds for this path
(tests/test_oracle_reference_kats.py): the Nystrom 3x3 tridiagonal KAT
(test/testrmf.jl:21-29), rank recovery and ||A-QQ'A||<1e-8 on exact-rank inputs
for both range finders (test/testrmf.jl:5-19), the LowRankCovMatrix 3-sample KAT
and operator identities (test/testrpcga.jl:46-58), the same-Omega
LowRankCovMatrix-vs-dense xis parity (test/testrpcga.jl:83-102) and the PCGA
operator test (test/testrpcga.jl:10-44).  Element-wise parity of Q/Z/S with a
Julia run is NOT pinned (no Julia, no reference fixtures) -- "parity pinned by
reference KATs/properties only".

RNG: the reference draws Omega with Julia's ``randn`` (RandMatFact.jl:54) after
an optional ``Random.seed!`` (GeostatInversion.jl:24-27).  That stream cannot be
reproduced outside Julia, so Omega is an explicit *input* everywhere here -- the
same array is handed to the oracle and to the HIP path.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import blas as _blas
import scipy.linalg as sl

__all__ = [
    "lu_L", "qr_thinQ", "rangefinder", "randsvd", "randsvd_full", "eig_nystrom",
    "rangefinder_adaptive", "colnorms", "LowRankCovMatrix", "getxis_dense",
    "getxis_fields", "PCGALowRankMatrix", "pcgadirect", "pcgalsqr", "lsqr",
    "subspace_sin", "xis_error_up_to_sign",
]


# --------------------------------------------------------------------------
# panel factorizations (Julia stdlib LinearAlgebra semantics)
# --------------------------------------------------------------------------
def lu_L(Y: np.ndarray) -> np.ndarray:
    """``F = LinearAlgebra.lu(Y); Q = F.L``  (RandMatFact.jl:60-61, 68-69, 72-73).

    dgetrf with partial pivoting.  ``F.L`` is the m x min(m,l) unit-lower-
    trapezoidal factor in *pivoted row order*: ``L @ U == Y[p, :]``.  The
    reference never undoes the permutation, so neither do we.  Julia's
    ``lu(...; check=true)`` throws only on an exactly zero pivot (info > 0).
    """
    Y = np.asarray(Y, dtype=np.float64)
    m, l = Y.shape
    lu, piv, info = sl.lapack.dgetrf(np.asfortranarray(Y))
    if info < 0:
        raise ValueError(f"dgetrf: illegal argument {-info}")
    if info > 0:
        raise np.linalg.LinAlgError(f"SingularException({info})")
    k = min(m, l)
    L = np.tril(lu[:, :k], -1)
    L[np.arange(k), np.arange(k)] = 1.0
    return L


def lu_pivots(Y: np.ndarray) -> np.ndarray:
    """0-based LAPACK ipiv of ``lu(Y)`` (row i was swapped with row ipiv[i])."""
    _, piv, _ = sl.lapack.dgetrf(np.asfortranarray(np.asarray(Y, dtype=np.float64)))
    return piv


def qr_thinQ(Y: np.ndarray) -> np.ndarray:
    """``F = qr(Y, Val(true)); Matrix(F.Q)``  (RandMatFact.jl:57-58, 75-76).

    dgeqp3 column-pivoted Householder QR, explicit thin Q (m x min(m,l)).  The
    permutation and R are discarded by the reference.
    """
    Q, _, _ = sl.qr(np.asarray(Y, dtype=np.float64), mode="economic", pivoting=True)
    return Q


# --------------------------------------------------------------------------
# RandMatFact.jl
# --------------------------------------------------------------------------
def _mul(A, X):
    return A.matmul(X) if hasattr(A, "matmul") else A @ X


def _tmul(A, X):
    """``A' * X``: dense -> dgemm('T'); LowRankCovMatrix -> adjoint(A) is A (lowrank.jl:38-40)."""
    return A.matmul(X) if hasattr(A, "matmul") else A.T @ X


def rangefinder(A, l: int, numiterations: int, Omega: np.ndarray) -> np.ndarray:
    """``rangefinder(A, l::Int64, numiterations::Int64)``  (RandMatFact.jl:50-80).

    ``Omega`` (n x l) replaces ``randn(n, l)`` at RandMatFact.jl:54.
    """
    n = A.shape[1]
    Omega = np.asarray(Omega, dtype=np.float64)
    if Omega.shape != (n, l):
        raise ValueError(f"Omega must be {(n, l)}, got {Omega.shape}")
    Y = _mul(A, Omega)                                   # :55
    if numiterations == 0:
        return qr_thinQ(Y)                               # :57-58
    elif numiterations > 0:
        Q = lu_L(Y)                                      # :60-61
    else:
        raise ValueError("parameter numiterations should be positive, but "
                         f"numiterations={numiterations}")   # :63
    for i in range(1, numiterations + 1):                # :66
        Q = _tmul(A, Q)                                  # :67
        Q = lu_L(Q)                                      # :68-69
        Q = _mul(A, Q)                                   # :70
        if i < numiterations:
            Q = lu_L(Q)                                  # :72-73
        else:
            Q = qr_thinQ(Q)                              # :75-76
    return Q


def randsvd_full(A, K: int, p: int, q: int, Omega: np.ndarray):
    """randsvd plus the intermediate (Q, S) -- S is ``svd(B).S`` (RandMatFact.jl:86)."""
    Q = rangefinder(A, K + p, q, Omega)                  # :84
    if hasattr(A, "matmul"):
        B = A.matmul(Q).T                                # lowrank.jl:131-133: (A * B.parent)'
    else:
        B = Q.T @ A                                      # :85
    _, S, Vt = np.linalg.svd(B, full_matrices=False)     # :86  dgesdd, thin
    V = Vt.T
    Sh = np.sqrt(np.concatenate([S[:K], np.zeros(p)]))   # :87
    Z = V * Sh[None, :]                                  # :88  (n x (K+p); last p columns zero)
    return Z, S, Q


def randsvd(A, K: int, p: int, q: int, Omega: np.ndarray) -> np.ndarray:
    """``randsvd(A, K::Int, p::Int, q::Int)``  (RandMatFact.jl:83-90) -> Z, n x (K+p)."""
    return randsvd_full(A, K, p, q, Omega)[0]


def eig_nystrom(A, Q: np.ndarray):
    """``eig_nystrom(A, Q)``  (RandMatFact.jl:92-102) -> (U, Sigmavec); eigenvalues = Sigmavec**2."""
    B1 = _mul(A, Q)                                      # :93
    B2 = Q.T @ B1                                        # :94
    # cholesky(Hermitian(B2)).U: Hermitian() reads the upper triangle (uplo=:U default)
    C = sl.cholesky(np.triu(B2) + np.triu(B2, 1).T, lower=False)   # :95
    F = B1 @ np.linalg.inv(C)                            # :96
    U, Sigmavec, _ = np.linalg.svd(F, full_matrices=False)         # :97
    return U, Sigmavec


def colnorms(Y: np.ndarray) -> np.ndarray:
    """``colnorms(Y)``  (RandMatFact.jl:7-13)."""
    return np.sqrt((np.asarray(Y) ** 2).sum(axis=0))


def rangefinder_adaptive(A: np.ndarray, randn, epsilon: float = 1e-8, r: int = 10) -> np.ndarray:
    """``rangefinder(A; epsilon=1e-8, r=10)`` -- Halko Alg 4.2  (RandMatFact.jl:15-48).

    ``randn(shape)`` is the Gaussian source standing in for Julia's ``randn`` /
    ``Random.randn!`` (called once with ``(n, r)`` at :20, then with ``(n,)`` per
    iteration at :36), in the same order the reference consumes its stream.
    """
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    kmax = min(n, m)
    Yfull = np.zeros((n, r + kmax))                      # :18 (n rows: the reference assumes m == n)
    Yfull[:m, :r] = A @ randn((n, r))                    # :20
    Qfull = np.zeros((m, kmax))                          # :23
    j = 0
    thresh = epsilon / np.sqrt(200.0 / np.pi)
    while colnorms(Yfull[:, j:j + r]).max() > thresh:    # :26
        if j >= kmax:
            break                                        # the reference would throw a BoundsError here
        j += 1
        Yj = Yfull[:m, j - 1].copy()                     # :28 (view; ``Yj -= ...`` at :31 rebinds, not in place)
        Q = Qfull[:, :j - 1]
        Yj = Yj - Q @ (Q.T @ Yj)                         # :30-31
        Qfull[:, j - 1] += Yj / np.linalg.norm(Yj)       # :32-34 (axpy! into a zero column)
        Q = Qfull[:, :j]                                 # :35
        omega = randn((n,))                              # :36
        Aomega = A @ omega                               # :37
        ynew = Aomega - Q @ (Q.T @ Aomega)               # :38-39
        Yfull[:m, r + j - 1] = ynew                      # :40
        Qj = Qfull[:, j - 1]
        for i in range(j + 1, j + r):                    # :42-45  i = j+1 : j+r-1 (1-based)
            Yi = Yfull[:m, i - 1]
            Yi -= np.dot(Qj, Yi) * Qj
    return Qfull[:, :j].copy()                           # :47


# --------------------------------------------------------------------------
# lowrank.jl
# --------------------------------------------------------------------------
class LowRankCovMatrix:
    """``LowRankCovMatrix(samples)``  (lowrank.jl:14-30): implicit A = sum_i s_i s_i' / (N-1)
    over mean-removed samples; symmetric, ``adjoint(A) === A`` (lowrank.jl:38-44)."""

    def __init__(self, samples, gemm_form=False):
        S = np.asarray(samples, dtype=np.float64)         # N x n (one sample per row)
        if S.ndim != 2:
            raise ValueError("samples must be a sequence of equal-length vectors")
        means = S.sum(axis=0) / S.shape[0]                # :18-24
        self.samples = S - means[None, :]                 # :25-27
        self.N, self.n = S.shape
        # gemm_form: the matrix products as S' (S B) / (N - 1) -- the sum of lowrank.jl:115-121's N rank-1 terms
        # associated as two dgemms (differs from the ger! loop by rounding order only).  For sizes where N
        # read-modify-write sweeps of the n x l result on the host would take hours (n = 1e6: bench.py's full-size
        # parity leg); the default stays the reference's loop.
        self.gemm_form = bool(gemm_form)

    @property
    def shape(self):                                      # size(A)  lowrank.jl:50-60
        return (self.n, self.n)

    def size(self, i: int) -> int:
        if i in (1, 2):
            return self.n
        raise IndexError(f"there is no {i}-th dimension in a LowRankCovMatrix")   # :58

    def matmul(self, B: np.ndarray) -> np.ndarray:
        """``*(A::LowRankCovMatrix, B::Matrix)`` (lowrank.jl:115-121) and the vector
        ``mul!`` (lowrank.jl:75-81): N rank-1 ger!/axpy! updates, in sample order."""
        B = np.asarray(B, dtype=np.float64)
        vec = B.ndim == 1
        B2 = B[:, None] if vec else B
        alpha = 1.0 / (self.N - 1)
        if vec:
            out = np.zeros(self.n)
            for s in self.samples:
                out += alpha * (s * np.dot(B, s))             # BLAS.axpy!(1/(N-1), s * dot(x, s), v)   :75-81
            return out
        if self.gemm_form:
            T = self.samples @ B2                         # T[i, :] = (B' s_i)'      the gemv of :117
            T *= alpha
            return np.asfortranarray((T.T @ self.samples).T)   # sum_i s_i T[i, :]      the ger! of :118, summed by dgemm
        # BLAS.ger!(1/(N-1), s, B's, result), in place like the reference (:115-121)
        out = np.zeros((self.n, B2.shape[1]), order="F")
        Bt = np.ascontiguousarray(B2.T)
        for s in self.samples:
            out = _blas.dger(alpha, s, Bt @ s, a=out, overwrite_a=1)
        return out

    def rmatmul(self, B: np.ndarray) -> np.ndarray:
        """``*(B::Matrix, A::LowRankCovMatrix)``  (lowrank.jl:123-129)."""
        B = np.asarray(B, dtype=np.float64)
        out = np.zeros((B.shape[0], self.n))
        alpha = 1.0 / (self.N - 1)
        for s in self.samples:
            out += alpha * np.outer(B @ s, s)
        return out

    def todense(self) -> np.ndarray:
        """``Matrix(I, n, n) * lrcm`` as the tests build it (test/testrpcga.jl:49, 90)."""
        return self.rmatmul(np.eye(self.n))

    def solve(self, b: np.ndarray) -> np.ndarray:
        """``\\(A::LowRankCovMatrix, b)``  (lowrank.jl:141-144): lsqr with maxiter = N."""
        return lsqr(self.matmul, self.matmul, np.asarray(b, float), self.n, maxiter=self.N)[0]


class PCGALowRankMatrix:
    """``PCGALowRankMatrix(etas, HX, R)``  (lowrank.jl:32-36): [(HQH+R) HX; HX' 0] with
    HQH = sum_i eta_i eta_i' kept implicit."""

    def __init__(self, etas, HX, R):
        self.etas = [np.asarray(e, dtype=np.float64) for e in etas]
        self.HX = np.asarray(HX, dtype=np.float64)
        self.R = R

    @property
    def shape(self):                                      # lowrank.jl:62-73
        s = len(self.etas[0]) + 1
        return (s, s)

    def matvec(self, x: np.ndarray) -> np.ndarray:
        """``mul!(v, A::PCGALowRankMatrix, x)``  (lowrank.jl:83-97)."""
        x = np.asarray(x, dtype=np.float64)
        xshort = x[:-1]
        v = np.empty(len(x))
        v[:-1] = self.R @ xshort
        v[-1] = np.dot(self.HX, xshort)
        for eta in self.etas:
            v[:-1] += eta * np.dot(eta, xshort)
        v[:-1] += self.HX * x[-1]
        return v


# --------------------------------------------------------------------------
# GeostatInversion.jl: getxis
# --------------------------------------------------------------------------
def getxis_dense(Qcov: np.ndarray, numxis: int, p: int, q: int, Omega: np.ndarray):
    """``getxis(Q::Matrix, numxis, p, q=3, seed)``  (GeostatInversion.jl:63-70)."""
    Z = randsvd(np.asarray(Qcov, dtype=np.float64), numxis, p, q, Omega)
    return [Z[:, i].copy() for i in range(numxis)]       # :66-68


def getxis_fields(fields, numxis: int, p: int, q: int, Omega: np.ndarray):
    """``getxis(Val{:iwantfields}, samplefield, numfields, ...)``  (GeostatInversion.jl:29-38),
    with the sampled ``fields`` passed in (sampling is user code run through rpmap at :30)."""
    lrcm = LowRankCovMatrix(fields)                      # :31
    Z = randsvd(lrcm, numxis, p, q, Omega)               # :32
    return [Z[:, i].copy() for i in range(numxis)], fields


# --------------------------------------------------------------------------
# consumers: direct.jl / lsqr.jl  (section 8 row a17 / f1)
# --------------------------------------------------------------------------
def lsqr(matvec, rmatvec, b, ncols, maxiter=None, atol=None, btol=None, conlim=1e8, damp=0.0):
    """Paige & Saunders LSQR (ACM TOMS 8(1), 1982) as IterativeSolvers.jl 0.9 ``lsqr`` runs it.

    IterativeSolvers is a third-party dependency (Project.toml:21, ``IterativeSolvers = "0.9"``)
    that is NOT under /root/reference; this restates the published algorithm with that
    package's defaults (atol = btol = sqrt(eps), conlim = 1e8, maxiter = max(m, n)).  The
    reference pins it only through the 2e-2 end-to-end PCGA bound (test/testrpcga.jl:129)
    -- "parity unpinned" at solver level.  Call sites: lsqr.jl:54, lowrank.jl:142.
    """
    b = np.asarray(b, dtype=np.float64)
    m = b.shape[0]
    n = ncols
    tol = np.sqrt(np.finfo(np.float64).eps)
    atol = tol if atol is None else atol
    btol = tol if btol is None else btol
    maxiter = max(m, n) if maxiter is None else maxiter
    ctol = 1.0 / conlim if conlim > 0 else 0.0
    x = np.zeros(n)
    u = b.copy()
    beta = np.linalg.norm(u)
    if beta == 0:
        return x, 0
    u /= beta
    v = rmatvec(u)
    alpha = np.linalg.norm(v)
    if alpha == 0:
        return x, 0
    v = v / alpha
    w = v.copy()
    rhobar, phibar = alpha, beta
    bnorm = beta
    Anorm = ddnorm = xxnorm = 0.0
    res2 = 0.0
    z = 0.0
    cs2, sn2 = -1.0, 0.0
    dampsq = damp * damp
    it = 0
    while it < maxiter:
        it += 1
        u = matvec(v) - alpha * u
        beta = np.linalg.norm(u)
        if beta > 0:
            u /= beta
            Anorm = np.sqrt(Anorm ** 2 + alpha ** 2 + beta ** 2 + dampsq)
            v = rmatvec(u) - beta * v
            alpha = np.linalg.norm(v)
            if alpha > 0:
                v /= alpha
        rhobar1 = np.sqrt(rhobar ** 2 + dampsq)
        cs1 = rhobar / rhobar1
        sn1 = damp / rhobar1
        psi = sn1 * phibar
        phibar = cs1 * phibar
        rho = np.sqrt(rhobar1 ** 2 + beta ** 2)
        cs = rhobar1 / rho
        sn = beta / rho
        theta = sn * alpha
        rhobar = -cs * alpha
        phi = cs * phibar
        phibar = sn * phibar
        tau = sn * phi
        t1 = phi / rho
        t2 = -theta / rho
        ddnorm += (np.linalg.norm(w) / rho) ** 2
        x = x + t1 * w
        w = v + t2 * w
        delta = sn2 * rho
        gambar = -cs2 * rho
        rhs = phi - delta * z
        zbar = rhs / gambar
        xnorm = np.sqrt(xxnorm + zbar ** 2)
        gamma = np.sqrt(gambar ** 2 + theta ** 2)
        cs2 = gambar / gamma
        sn2 = theta / gamma
        z = rhs / gamma
        xxnorm += z ** 2
        Acond = Anorm * np.sqrt(ddnorm)
        res1 = phibar ** 2
        res2 += psi ** 2
        rnorm = np.sqrt(res1 + res2)
        Arnorm = alpha * abs(tau)
        test1 = rnorm / bnorm
        test2 = Arnorm / (Anorm * rnorm) if Anorm * rnorm > 0 else 0.0
        test3 = 1.0 / Acond if Acond > 0 else 0.0
        t1c = test1 / (1.0 + Anorm * xnorm / bnorm)
        rtol = btol + atol * Anorm * xnorm / bnorm
        if (1 + test3 <= 1) or (1 + test2 <= 1) or (1 + t1c <= 1):
            break
        if test3 <= ctol or test2 <= atol or test1 <= rtol:
            break
    return x, it


def _pcga_setup(forwardmodel, s, X, xis, delta):
    """Shared head of pcgadirectiteration! (direct.jl:38-46) / pcgalsqriteration (lsqr.jl:36-44)."""
    K = len(xis)
    paramstorun = [s + delta * xis[i] for i in range(K)]
    paramstorun.append(s + delta * X)
    paramstorun.append(s + delta * s)
    paramstorun.append(s)
    results = [np.asarray(forwardmodel(pv), dtype=np.float64) for pv in paramstorun]   # pmap
    hs = results[K + 2]
    etas = [(results[i] - hs) / delta for i in range(K)]
    HX = (results[K] - hs) / delta
    Hs = (results[K + 1] - hs) / delta
    return etas, HX, Hs, hs


def pcgadirect(forwardmodel, s0, X, xis, R, y, maxiters=5, delta=np.sqrt(np.finfo(float).eps),
               xtol=1e-6, callback=None):
    """``pcgadirect``  (direct.jl:21-67)."""
    s = np.asarray(s0, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    Rd = R.toarray() if hasattr(R, "toarray") else np.asarray(R, dtype=np.float64)
    it = 0
    converged = False
    while not converged and it < maxiters:
        olds = s
        etas, HX, Hs, hs = _pcga_setup(forwardmodel, s, X, xis, delta)
        if callback is not None:
            callback(s, hs)                               # :47
        HQH = np.zeros((len(y), len(y)))
        for eta in etas:
            HQH += np.outer(eta, eta)                     # :49-53
        b = np.concatenate([y - hs + Hs, np.zeros(1)])    # :56
        bigA = np.block([[HQH + Rd, HX[:, None]], [HX[None, :], np.zeros((1, 1))]])   # :57
        x = np.linalg.pinv(bigA) @ b                      # :58
        beta_bar, xi_bar = x[-1], x[:-1]
        s = X * beta_bar                                  # :61
        for i, eta in enumerate(etas):
            s = s + xis[i] * np.dot(eta, xi_bar)          # :62-65
        if np.linalg.norm(s - olds) < xtol:
            converged = True
        it += 1
    return s


def pcgalsqr(forwardmodel, s0, X, xis, R, y, maxiters=5, delta=np.sqrt(np.finfo(float).eps),
             xtol=1e-6):
    """``pcgalsqr``  (lsqr.jl:20-63)."""
    s = np.asarray(s0, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    it = 0
    converged = False
    while not converged and it < maxiters:
        olds = s
        etas, HX, Hs, hs = _pcga_setup(forwardmodel, s, X, xis, delta)
        b = np.concatenate([y - hs + Hs, np.zeros(1)])    # :52
        bigA = PCGALowRankMatrix(etas, HX, R)             # :53
        x, _ = lsqr(bigA.matvec, bigA.matvec, b, len(b))  # :54 (operator is symmetric: adjoint == self)
        beta_bar, xi_bar = x[-1], x[:-1]
        s = X * beta_bar
        for i, eta in enumerate(etas):
            s = s + xis[i] * np.dot(eta, xi_bar)
        if np.linalg.norm(s - olds) < xtol:
            converged = True
        it += 1
    return s


# --------------------------------------------------------------------------
# parity metrics shared by the tests and bench.py
# --------------------------------------------------------------------------
def subspace_sin(Q1: np.ndarray, Q2: np.ndarray) -> float:
    """sin of the largest principal angle between range(Q1) and range(Q2) (both orthonormal)."""
    M = Q2 - Q1 @ (Q1.T @ Q2)
    return float(np.linalg.norm(M, 2))


def xis_error_up_to_sign(Z1: np.ndarray, Z2: np.ndarray, K: int) -> float:
    """max_i min(||z1_i - z2_i||, ||z1_i + z2_i||): the reference's own parity measure
    (test/testrpcga.jl:100)."""
    err = 0.0
    for i in range(K):
        a, b = Z1[:, i], Z2[:, i]
        err = max(err, min(np.linalg.norm(a - b), np.linalg.norm(a + b)))
    return float(err)


# ---- f2: matrix-free circulant-embedding covariance (not in the reference as an operator) --------------------
def fft_powerlaw_spectrum(Ns, beta, fftrf=False):
    """lambda on the embedding grid (shape M_1 x ... x M_d, Fortran order of the flattened operator):
    (sum_a nu_a^2)^(beta/2), f_a = min(k_a, M_a - k_a), lambda(0) = 0; normalised to mean 1 so that the circulant
    F^-1 diag(lambda) F has a unit diagonal.
    fftrf=False: M_a = next power of two >= 2 N_a, nu_a = f_a / M_a (cycles per grid spacing).
    fftrf=True : FFTRF.jl's convention -- M_a = 2 N_a exactly, nu_a = f_a the integer wavenumbers
    0..N_a, -(N_a-1)..-1 (FFTRF.jl:86-89); `S_f ^ (.25*beta)` at FFTRF.jl:62 is the square root of this lambda."""
    if fftrf:
        Ms = [1 if N == 1 else 2 * N for N in Ns]
    else:
        Ms = [1 if N == 1 else 1 << int(np.ceil(np.log2(2 * N))) for N in Ns]
    k2 = np.zeros(Ms)
    for a, M in enumerate(Ms):
        k = np.arange(M)
        f = np.minimum(k, M - k) / (1.0 if fftrf else M)
        shape = [1] * len(Ms)
        shape[a] = M
        k2 = k2 + (f ** 2).reshape(shape)
    lam = np.zeros(Ms)
    nz = k2 > 0
    lam[nz] = k2[nz] ** (0.5 * beta)
    return lam / lam.mean(), Ms


def fft_powerlaw_apply(X, Ns, beta, fftrf=False):
    """A X for A = R F^-1 diag(lambda) F R' (zero padding R' of the N-grid into the embedding grid), columns of X =
    vec(field) in column-major (Julia) order."""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    lam, Ms = fft_powerlaw_spectrum(Ns, beta, fftrf)
    out = np.empty_like(X)
    box = tuple(slice(0, N) for N in Ns)
    for c in range(X.shape[1]):
        w = np.zeros(Ms)
        w[box] = X[:, c].reshape(Ns, order="F")
        y = np.fft.ifftn(lam * np.fft.fftn(w)).real
        out[:, c] = y[box].reshape(-1, order="F")
    return out


# ---- FFTRF.jl:8-100 restated (TEST INFRASTRUCTURE: FFTRF stays on the Julia/CPU side, BASELINE.json north_star; this
#      exists only so that a test can check which covariance its fields have) ------------------------------------------
def fftrf_powerlaw_structuredgrid(Ns, k0, dk, beta, rng, raw=False):
    """``powerlaw_structuredgrid(Ns, k0, dk, beta)``  (FFTRF.jl:83-100) for 2-D and 3-D grids.

    Returns the N_1 x N_2 (x N_3) array ``finalk``; ``raw=True`` returns it before the per-sample normalisation of
    FFTRF.jl:94-98 (the quantity whose covariance is the circulant-embedding operator).  ``rng.standard_normal``
    stands for Julia's ``randn`` in ``mulbyphi`` (FFTRF.jl:74-81)."""
    Ns = [int(v) for v in Ns]
    d = len(Ns)
    if d not in (2, 3):
        raise ValueError(f"unsupported dimension: {d}")                     # FFTRF.jl:58
    M = [2 * N for N in Ns]                                                   # :85
    coords = [np.concatenate([np.arange(0, N + 1), -np.arange(N - 1, 0, -1)]).astype(float) for N in Ns]   # :86-89
    # computesqrtS_f (:40-72): the array is (M_2, M_1[, M_3]); its FIRST axis runs over coordinate 2
    if d == 2:
        S = coords[1][:, None] ** 2 + coords[0][None, :] ** 2
    else:
        S = coords[1][:, None, None] ** 2 + coords[0][None, :, None] ** 2 + coords[2][None, None, :] ** 2
    with np.errstate(divide="ignore"):
        sqrtS = S ** (0.25 * beta)                                            # :62
    sqrtS[np.isinf(sqrtS)] = 0.0                                              # :63-65
    phi = rng.standard_normal(sqrtS.shape)                                    # mulbyphi :75
    result = sqrtS * (np.cos(2 * np.pi * phi) + 1j * np.sin(2 * np.pi * phi))   # :78 (cospi, sinpi)
    kcomplex = np.fft.ifftn(result)                                           # :92
    # reducek (:8-38): keep the first half along every axis, real part, axes 1 and 2 swapped back
    if d == 2:
        finalk = kcomplex[:Ns[1], :Ns[0]].real.T.copy()                       # finalk[j, i] = real(k[i, j])
    else:
        finalk = np.transpose(kcomplex[:Ns[1], :Ns[0], :Ns[2]].real, (1, 0, 2)).copy()
    if raw:
        return finalk
    std = finalk.std(ddof=1)                                                  # Statistics.std: corrected  :94
    mean = finalk.mean()                                                      # :95
    return dk * (finalk - mean) / std + k0                                    # :96-98
