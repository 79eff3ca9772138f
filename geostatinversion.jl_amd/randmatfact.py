"""Host-side mirror of the reference module `RandMatFact` (src/RandMatFact.jl): same function
names, argument meaning and error behaviour, every operation executed by libgsi_hip.so.

The one deliberate difference: the Gaussian test matrix.  The reference draws `randn(n, l)` from
Julia's task-local RNG (RandMatFact.jl:54).  Here `Omega` may be passed explicitly (parity runs:
the same array goes to the oracle and to the GPU); otherwise it is drawn from this module's numpy
Generator in Julia's column-major fill order.  The Julia wrapper in julia/ passes Julia's own
randn and so reproduces the reference stream exactly.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .context import Operator, dense_operator, default_context

_rng = np.random.default_rng()


def seed(s):
    """`Random.seed!(s)` for this module's default stream."""
    global _rng
    _rng = np.random.default_rng(s)


def randn(*shape):
    """Column-major-order Gaussian fill, like Julia's `randn(n, l)`."""
    if len(shape) == 1:
        return _rng.standard_normal(shape[0])
    n, l = shape
    return np.asfortranarray(_rng.standard_normal((l, n)).T)


def _as_operator(A, ctx=None):
    if isinstance(A, Operator):
        return A, False
    if hasattr(A, "_device_operator"):          # LowRankCovMatrix
        return A._device_operator(ctx), False
    ctx = ctx or default_context()
    return dense_operator(ctx, A), True


def _is_host_matrix(A, ctx=None):
    """A plain host matrix on a single-rank context: the host-matrix entry points take it (with a communicator every rank
    uploads its own block of rows through `dense_operator`)."""
    if isinstance(A, Operator) or hasattr(A, "_device_operator"):
        return False
    return (ctx or default_context()).rank()[1] == 1


def colnorms(Y):
    """`colnorms(Y)`  (RandMatFact.jl:7-13)."""
    Y = np.asarray(Y, dtype=np.float64)
    return np.sqrt((Y * Y).sum(axis=0))


def rangefinder(A, l=None, numiterations=None, *, Omega=None, epsilon=1e-8, r=10, ctx=None):
    """Both reference methods:

    `rangefinder(A, l::Int64, numiterations::Int64)`  (RandMatFact.jl:50-80), and
    `rangefinder(A; epsilon=1e-8, r=10)`              (RandMatFact.jl:15-48) when `l` is omitted.
    Returns Q (m x l, orthonormal columns).
    """
    if l is not None and _is_host_matrix(A, ctx):
        # rangefinder(A::Matrix, l, numiterations): the matrix goes up in row blocks and the sketch runs under the upload
        cx = ctx or default_context()
        Af = L.fmat(A, "A")
        m, n = Af.shape
        l, q = int(l), int(numiterations)
        Om = L.fmat(randn(n, l) if Omega is None else Omega, "Omega")           # RandMatFact.jl:54
        if Om.shape != (n, l):
            raise ValueError(f"Omega must be {(n, l)}, got {Om.shape}")
        Q = np.empty((m, l), order="F")
        L.check(cx.lib.gsi_rangefinder_dense_host(cx.h, L.dptr(Af), m, n, Af.shape[0], L.dptr(Om), l, q, L.dptr(Q), None),
                cx.lib)
        return Q
    op, owned = _as_operator(A, ctx)
    try:
        m, n = op.shape
        lib, cx = op.ctx.lib, op.ctx
        if l is None:
            return _rangefinder_adaptive(op, epsilon, r)
        l = int(l)
        q = int(numiterations)
        if Omega is None:
            Omega = randn(n, l)                                  # RandMatFact.jl:54
        Om = L.fmat(Omega, "Omega")
        if Om.shape != (n, l):
            raise ValueError(f"Omega must be {(n, l)}, got {Om.shape}")
        Q = np.empty((m, l), order="F")
        L.check(lib.gsi_rangefinder(cx.h, op.h, L.dptr(Om), l, q, L.dptr(Q)), lib)
        return Q
    finally:
        if owned:
            op.close()


def _rangefinder_adaptive(op, epsilon, r):
    m, n = op.shape
    lib, cx = op.ctx.lib, op.ctx
    Q = np.empty((m, min(m, n)), order="F")

    def _fill(_user, buf, count):
        np.ctypeslib.as_array(buf, shape=(count,))[:] = _rng.standard_normal(count)

    cb = L.RANDN_FN(_fill)
    ncols = C.c_int64()
    L.check(lib.gsi_rangefinder_adaptive(cx.h, op.h, cb, None, float(epsilon), int(r), L.dptr(Q),
                                         C.byref(ncols)), lib)
    return np.ascontiguousarray(Q[:, :ncols.value])


def randsvd(A, K, p, q, *, Omega=None, return_S=False, ctx=None):
    """`randsvd(A, K::Int, p::Int, q::Int)`  (RandMatFact.jl:83-90): Z (n x (K+p)) with the last p
    columns zero and Z Z' ~ A.  `return_S` also returns svd(Q'A).S (RandMatFact.jl:86).

    A host matrix (what `getxis(Q::Matrix, ...)` hands over, GeostatInversion.jl:63-70) goes through
    `gsi_randsvd_dense_host`: uploaded in row blocks, the first pass under the upload."""
    if _is_host_matrix(A, ctx):
        cx = ctx or default_context()
        Af = L.fmat(A, "A")
        m, n = Af.shape
        K, p, q = int(K), int(p), int(q)
        l = K + p
        Om = L.fmat(randn(n, l) if Omega is None else Omega, "Omega")
        if Om.shape != (n, l):
            raise ValueError(f"Omega must be {(n, l)}, got {Om.shape}")
        Z = np.empty((n, l), order="F")
        S = np.empty(l)
        L.check(cx.lib.gsi_randsvd_dense_host(cx.h, L.dptr(Af), m, n, Af.shape[0], L.dptr(Om), K, p, q, L.dptr(Z),
                                              S.ctypes.data_as(L.c_dp), None), cx.lib)
        return (Z, S) if return_S else Z
    op, owned = _as_operator(A, ctx)
    try:
        m, n = op.shape
        lib, cx = op.ctx.lib, op.ctx
        K, p, q = int(K), int(p), int(q)
        l = K + p
        if Omega is None:
            Omega = randn(n, l)
        Om = L.fmat(Omega, "Omega")
        if Om.shape != (n, l):
            raise ValueError(f"Omega must be {(n, l)}, got {Om.shape}")
        Z = np.empty((n, l), order="F")
        S = np.empty(l)
        L.check(lib.gsi_randsvd(cx.h, op.h, L.dptr(Om), K, p, q, L.dptr(Z), S.ctypes.data_as(L.c_dp)), lib)
        return (Z, S) if return_S else Z
    finally:
        if owned:
            op.close()


def randsvd_rows(op, K, p, q, Omega_rows, *, return_S=False):
    """`randsvd` with row-sharded panels over the ranks of `op`'s communicator (`gsi_randsvd_rows`): `Omega_rows` = this
    rank's rows of Omega (host array or DeviceMatrix, nloc x (K+p)); returns this rank's rows of Z as a DeviceMatrix
    (nothing n x (K+p) is gathered), optionally S (host)."""
    from .context import DeviceMatrix
    cx = op.ctx
    l = int(K) + int(p)
    Om = Omega_rows if isinstance(Omega_rows, DeviceMatrix) else DeviceMatrix.from_host(cx, Omega_rows)
    Z = DeviceMatrix(cx, Om.shape[0], l)
    S = DeviceMatrix(cx, l, 1)
    L.check(cx.lib.gsi_randsvd_rows(cx.h, op.h, Om.h, int(K), int(p), int(q), Z.h, S.h), cx.lib)
    Sh = S.to_host()[:, 0].copy()
    S.close()
    if Om is not Omega_rows:
        Om.close()
    return (Z, Sh) if return_S else Z


def eig_nystrom(A, Q, *, ctx=None):
    """`eig_nystrom(A, Q)`  (RandMatFact.jl:92-102) -> (U, Sigmavec); eigenvalues are Sigmavec**2."""
    op, owned = _as_operator(A, ctx)
    try:
        m, n = op.shape
        lib, cx = op.ctx.lib, op.ctx
        Qf = L.fmat(Q, "Q")
        if Qf.shape[0] != n:
            raise ValueError("Q must have size(A,2) rows")
        j = Qf.shape[1]
        U = np.empty((m, j), order="F")
        S = np.empty(j)
        L.check(lib.gsi_eig_nystrom(cx.h, op.h, L.dptr(Qf), j, L.dptr(U), S.ctypes.data_as(L.c_dp)), lib)
        return U, S
    finally:
        if owned:
            op.close()


# ---- panel primitives (what the reference gets from LinearAlgebra) -----------------------------
def lu_L(Y, *, return_pivots=False, ctx=None):
    """`LinearAlgebra.lu(Y).L` in pivoted row order  (RandMatFact.jl:60-61)."""
    ctx = ctx or default_context()
    Yf = L.fmat(Y, "Y")
    m, l = Yf.shape
    out = np.empty((m, l), order="F")
    piv = np.empty(l, dtype=np.int32)
    L.check(ctx.lib.gsi_lu_L(ctx.h, L.dptr(Yf), m, l, L.dptr(out), piv.ctypes.data_as(C.POINTER(C.c_int32))),
            ctx.lib)
    return (out, piv) if return_pivots else out


def lu_L_dev(Y, *, return_pivots=False):
    """`lu(Y).L` of a device-resident panel (`DeviceMatrix`), in place (`gsi_lu_L_dev`)."""
    cx = Y.ctx
    piv = np.empty(Y.shape[1], dtype=np.int32)
    L.check(cx.lib.gsi_lu_L_dev(cx.h, Y.h, piv.ctypes.data_as(C.POINTER(C.c_int32))), cx.lib)
    return (Y, piv) if return_pivots else Y


def lu_L_sharded(Y, *, return_pivots=False, ctx=None):
    """`lu(Y).L` with the panel row-sharded over the ranks of `ctx`'s communicator (collective; every rank passes the
    whole panel and receives the whole L): bit-identical to `lu_L`."""
    ctx = ctx or default_context()
    Yf = L.fmat(Y, "Y")
    m, l = Yf.shape
    out = np.empty((m, l), order="F")
    piv = np.empty(l, dtype=np.int32)
    L.check(ctx.lib.gsi_lu_L_sharded(ctx.h, L.dptr(Yf), m, l, L.dptr(out), piv.ctypes.data_as(C.POINTER(C.c_int32))),
            ctx.lib)
    return (out, piv) if return_pivots else out


def lu_L_sharded_virtual(Y, nshards, *, return_pivots=False, ctx=None):
    """`lu(Y).L` through the row-sharded kernels with `nshards` virtual ranks on one GPU (`gsi_lu_L_sharded_virtual`):
    bit-identical to `lu_L`."""
    ctx = ctx or default_context()
    Yf = L.fmat(Y, "Y")
    m, l = Yf.shape
    out = np.empty((m, l), order="F")
    piv = np.empty(l, dtype=np.int32)
    L.check(ctx.lib.gsi_lu_L_sharded_virtual(ctx.h, L.dptr(Yf), m, l, int(nshards), L.dptr(out),
                                             piv.ctypes.data_as(C.POINTER(C.c_int32))), ctx.lib)
    return (out, piv) if return_pivots else out


def qr_thinQ(Y, *, return_R=False, ctx=None):
    """`Matrix(qr(Y, Val(true)).Q)` up to an orthogonal change of basis  (RandMatFact.jl:57-58)."""
    ctx = ctx or default_context()
    Yf = L.fmat(Y, "Y")
    m, l = Yf.shape
    Q = np.empty((m, l), order="F")
    R = np.empty((l, l), order="F")
    L.check(ctx.lib.gsi_qr_thinQ(ctx.h, L.dptr(Yf), m, l, L.dptr(Q), L.dptr(R)), ctx.lib)
    return (Q, R) if return_R else Q


def svd_tall(W, *, ctx=None):
    """(S, V) of `svd(W')` for tall W: what RandMatFact.jl:86 takes from `svd(B)`."""
    ctx = ctx or default_context()
    Wf = L.fmat(W, "W")
    n, l = Wf.shape
    V = np.empty((n, l), order="F")
    S = np.empty(l)
    L.check(ctx.lib.gsi_svd_tall(ctx.h, L.dptr(Wf), n, l, L.dptr(V), S.ctypes.data_as(L.c_dp)), ctx.lib)
    return S, V


def gemm(A, B, *, trans=False, alpha=1.0, ctx=None):
    """alpha * A * B (trans=False) or alpha * A' * B through the MFMA kernels."""
    ctx = ctx or default_context()
    Af, Bf = L.fmat(A, "A"), L.fmat(B, "B")
    if trans:
        k, m = Af.shape
    else:
        m, k = Af.shape
    if Bf.shape[0] != k:
        raise ValueError("dimension mismatch")
    l = Bf.shape[1]
    Cm = np.empty((m, l), order="F")
    L.check(ctx.lib.gsi_gemm(ctx.h, int(trans), m, l, k, float(alpha), L.dptr(Af), Af.shape[0], L.dptr(Bf), k,
                             L.dptr(Cm), m), ctx.lib)
    return Cm
