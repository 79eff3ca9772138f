"""The LAPACK-free C restatement (oracle/gsi_oracle.c) against the scipy oracle, the golden
vectors and the reference's known-answer tests.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc
from helpers import gaussian_cov, exact_rank_matrix, rel_sv_err
import cpuref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def oc():
    return cpuref.load_oracle_c()


def F(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


@pytest.mark.parametrize("m,l", [(10, 2), (40, 7), (300, 25)])
def test_c_lu_matches_dgetrf(oc, m, l):
    rng = np.random.default_rng(m + l)
    Y = rng.standard_normal((m, l))
    A = F(Y)
    piv = np.empty(l, dtype=np.int32)
    assert oc.gsio_lu_L(A.ctypes.data_as(dp), m, l, m, piv.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert np.array_equal(piv, orc.lu_pivots(Y))
    assert np.abs(A - orc.lu_L(Y)).max() < 1e-12


def test_c_lu_zero_pivot_info(oc):
    A = F(np.zeros((6, 2)))
    assert oc.gsio_lu_L(A.ctypes.data_as(dp), 6, 2, 6, None) == 1


@pytest.mark.parametrize("m,l", [(12, 3), (200, 20)])
def test_c_pivoted_qr_matches_dgeqp3(oc, m, l):
    rng = np.random.default_rng(m * l)
    Y = rng.standard_normal((m, l)) @ np.diag(np.logspace(0, -5, l))
    A = F(Y)
    R = F(np.zeros((l, l)))
    jp = np.empty(l, dtype=np.int32)
    oc.gsio_qr_thinQ(A.ctypes.data_as(dp), m, l, m, 1, R.ctypes.data_as(dp), jp.ctypes.data_as(C.POINTER(C.c_int32)))
    assert np.abs(A.T @ A - np.eye(l)).max() < 1e-13
    assert np.abs(A @ R - Y[:, jp]).max() < 1e-13
    import scipy.linalg as sl
    Qs, Rs, Ps = sl.qr(Y, mode="economic", pivoting=True)
    assert np.array_equal(jp, Ps)                                  # same pivot order as dgeqp3
    assert orc.subspace_sin(Qs, A) < 1e-9


def test_c_svd_tall(oc):
    rng = np.random.default_rng(4)
    W = rng.standard_normal((150, 12)) @ np.diag(np.logspace(0, -7, 12)) @ rng.standard_normal((12, 12))
    A = F(W)
    S = np.empty(12)
    oc.gsio_svd_tall(A.ctypes.data_as(dp), 150, 12, 150, S.ctypes.data_as(dp))
    Sref = np.linalg.svd(W, compute_uv=False)
    assert np.abs(S - Sref).max() < 1e-13 * Sref[0]
    assert np.abs(A.T @ A - np.eye(12)).max() < 1e-10


@pytest.mark.parametrize("q", [0, 1, 3])
def test_c_randsvd_vs_scipy_oracle(oc, q):
    A = gaussian_cov(14, 10, 3.0)
    rng = np.random.default_rng(q)
    K, p = 10, 5
    Om = rng.standard_normal((140, K + p))
    Z = F(np.zeros((140, K + p)))
    S = np.empty(K + p)
    Af, Of = F(A), F(Om)
    assert oc.gsio_randsvd(Af.ctypes.data_as(dp), 140, 140, 140, Of.ctypes.data_as(dp), K, p, q, Z.ctypes.data_as(dp),
                           S.ctypes.data_as(dp)) == 0
    Zr, Sr, _ = orc.randsvd_full(A, K, p, q, Om)
    assert rel_sv_err(S, Sr, K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, Zr, K) < 1e-6
    assert np.all(Z[:, K:] == 0)


def test_c_randsvd_vs_golden(oc):
    g = np.load(os.path.join(GOLD, "dense_gauss_n192.npz"))
    A = gaussian_cov(int(g["grid"][0]), int(g["grid"][1]), float(g["ell"]))
    K, p, q = int(g["K"]), int(g["p"]), int(g["q"])
    Z = F(np.zeros((192, K + p)))
    S = np.empty(K + p)
    Af, Of = F(A), F(g["Omega"])
    assert oc.gsio_randsvd(Af.ctypes.data_as(dp), 192, 192, 192, Of.ctypes.data_as(dp), K, p, q, Z.ctypes.data_as(dp),
                           S.ctypes.data_as(dp)) == 0
    assert rel_sv_err(S, g["S"], K) < 1e-9
    assert orc.xis_error_up_to_sign(Z, g["Z"], K) < 1e-6


def test_c_negative_iterations(oc):
    A = F(np.eye(4))
    Om = F(np.ones((4, 2)))
    Q = F(np.zeros((4, 2)))
    assert oc.gsio_rangefinder(A.ctypes.data_as(dp), 4, 4, 4, Om.ctypes.data_as(dp), 2, -1, Q.ctypes.data_as(dp)) == -1


def test_c_nystrom_kat(oc):
    """test/testrmf.jl:21-29."""
    g = np.load(os.path.join(GOLD, "nystrom_kat.npz"))
    A = F(g["A"])
    Q = F(np.linalg.qr(np.random.default_rng(1).standard_normal((3, 3)))[0])
    U = F(np.zeros((3, 3)))
    Sig = np.empty(3)
    assert oc.gsio_eig_nystrom(A.ctypes.data_as(dp), 3, 3, Q.ctypes.data_as(dp), 3, U.ctypes.data_as(dp),
                               Sig.ctypes.data_as(dp)) == 0
    assert np.linalg.norm(Sig ** 2 - g["eigenvalues"]) < 1e-8


def test_scipy_oracle_vs_golden():
    """oracle.py itself still reproduces the committed vectors."""
    g = np.load(os.path.join(GOLD, "dense_gauss_n192.npz"))
    A = gaussian_cov(int(g["grid"][0]), int(g["grid"][1]), float(g["ell"]))
    Z, S, _ = orc.randsvd_full(A, int(g["K"]), int(g["p"]), int(g["q"]), g["Omega"])
    assert rel_sv_err(S, g["S"], int(g["K"])) < 1e-10
    assert orc.xis_error_up_to_sign(Z, g["Z"], int(g["K"])) < 1e-7
    Y = A @ g["Omega"]
    assert np.array_equal(orc.lu_pivots(Y), g["lu_pivots"])
    g3 = np.load(os.path.join(GOLD, "lowrank_n100_N24.npz"))
    xis, _ = orc.getxis_fields(list(g3["fields"]), int(g3["K"]), int(g3["p"]), int(g3["q"]), g3["Omega"])
    assert orc.xis_error_up_to_sign(np.array(xis).T, g3["xis"].T, int(g3["K"])) < 1e-7


@pytest.mark.parametrize("h,l", [(20, 7), (150, 24)])
def test_c_oracle_lu_ties_duplicate_rows(h, l):
    """Tie-breaking of the pivot search (lowest index, idamax) in the C restatement: Y = [R; R]."""
    import cpuref
    import ctypes as C
    lib = cpuref.load_oracle_c()
    rng = np.random.default_rng(h * l)
    R = rng.standard_normal((h, l))
    Y = np.asfortranarray(np.vstack([R, R]))
    piv = np.zeros(l, dtype=np.int32)
    Yc = Y.copy(order="F")
    lib.gsio_lu_L.restype = C.c_int
    info = lib.gsio_lu_L(Yc.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(2 * h), C.c_int64(l), C.c_int64(2 * h),
                         piv.ctypes.data_as(C.POINTER(C.c_int32)))
    assert info == 0
    assert np.array_equal(piv, orc.lu_pivots(Y))
