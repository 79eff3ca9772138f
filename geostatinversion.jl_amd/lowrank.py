"""Host-side mirror of src/lowrank.jl: `LowRankCovMatrix` (device-resident sample matrix) and
`PCGALowRankMatrix`."""
import ctypes as C

import numpy as np

from . import _lib as L
from .context import Operator, default_context


class LowRankCovMatrix:
    """`LowRankCovMatrix(samples)`  (lowrank.jl:14-30): A = sum_i s_i s_i'/(N-1) over the
    mean-removed samples, never formed.  The samples are uploaded once as an n x N column-major
    matrix and centred on the device; products run as two tall-skinny MFMA GEMMs,
    S (S'B)/(N-1), instead of the reference's N rank-1 `ger!` sweeps (lowrank.jl:115-121)."""

    def __init__(self, samples, ctx=None):
        S = np.asarray(samples, dtype=np.float64)
        if S.ndim != 2 or S.shape[0] < 2:
            raise ValueError("samples must be a sequence of >= 2 equal-length vectors")
        self.N, self.n = S.shape
        self._host_samples = S
        self._ctx = ctx
        self._op = None

    def _device_operator(self, ctx=None):
        if self._op is None or self._op.h is None:
            ctx = ctx or self._ctx or default_context()
            Sf = np.asfortranarray(self._host_samples.T)           # n x N, sample i = column i
            row0, nloc = ctx.shard(self.n)
            h = C.c_void_p()
            base = Sf[row0:, :] if nloc > 0 else Sf
            L.check(ctx.lib.gsi_op_lowrank(ctx.h, C.byref(h), base.ctypes.data_as(L.c_dp), self.n, self.N,
                                           Sf.shape[0], 1, row0, nloc), ctx.lib)
            self._op = Operator(ctx, h)
        return self._op

    # ---- the reference's method table ----
    @property
    def shape(self):                       # size(A)            lowrank.jl:50-52
        return (self.n, self.n)

    def size(self, i):                     # size(A, i)         lowrank.jl:54-60
        if i in (1, 2):
            return self.n
        raise IndexError(f"there is no {i}-th dimension in a LowRankCovMatrix")

    @property
    def T(self):                           # adjoint / transpose return the operator itself  :38-44
        return self

    eltype = np.float64                    # lowrank.jl:46-48
    __array_ufunc__ = None                 # let `ndarray @ lrcm` reach __rmatmul__

    def matmul(self, B):
        """`*(A::LowRankCovMatrix, B::Matrix)` (lowrank.jl:115-121) / vector `*` (:135-139)."""
        return self._device_operator().matmul(B)

    __matmul__ = matmul

    def __rmatmul__(self, B):
        """`*(B::Matrix, A::LowRankCovMatrix)`  (lowrank.jl:123-129): B*A = (A*B')' by symmetry."""
        return self.matmul(np.asarray(B, dtype=np.float64).T).T

    def todense(self):
        return self.matmul(np.eye(self.n))

    @property
    def samples(self):
        """`A.samples` (lowrank.jl:14-16): the mean-removed samples as the device holds them, N x n (one per row)."""
        return device_samples(self._device_operator(), self.N)

    def solve(self, b, *, return_iterations=False):
        """`\\(A::LowRankCovMatrix, b::Vector)`  (lowrank.jl:141-144): `lsqr(A, b; maxiter=length(A.samples))` with
        IterativeSolvers' defaults, every vector resident in HBM (`gsi_op_lowrank_solve`)."""
        op = self._device_operator()
        bv = np.ascontiguousarray(b, dtype=np.float64)
        if bv.shape != (self.n,):
            raise ValueError("dimension mismatch")
        x = np.empty(self.n)
        it = C.c_int64()
        L.check(op.ctx.lib.gsi_op_lowrank_solve(op.ctx.h, op.h, bv.ctypes.data_as(L.c_dp), x.ctypes.data_as(L.c_dp),
                                               C.byref(it)), op.ctx.lib)
        return (x, it.value) if return_iterations else x

    def close(self):
        if self._op is not None:
            self._op.close()
            self._op = None


def device_samples(op, N):
    """Centred samples of a device LowRankCovMatrix operator (`gsi_op_lowrank_samples`), N x n, one sample per row
    (this rank's columns when the operator is row-sharded)."""
    m = C.c_int64()
    r0 = C.c_int64()
    ml = C.c_int64()
    L.check(op.ctx.lib.gsi_op_size(op.h, C.byref(m), None, C.byref(r0), C.byref(ml)), op.ctx.lib)
    out = np.empty((int(N), max(ml.value, 1)))                # C order N x n_local == column-major n_local x N
    L.check(op.ctx.lib.gsi_op_lowrank_samples(op.ctx.h, op.h, out.ctypes.data_as(L.c_dp), max(ml.value, 1)), op.ctx.lib)
    return out[:, :ml.value]


class PCGALowRankMatrix:
    """`PCGALowRankMatrix(etas, HX, R)`  (lowrank.jl:32-36, 62-73, 83-97): the saddle-point matrix
    [(HQH+R) HX; HX' 0] with HQH = sum eta_i eta_i' kept implicit -- the LSQR operator of pcgalsqr
    (lsqr.jl:53-54).  etas, HX and R are uploaded once (`gsi_pcgamat_create`); the product and the LSQR solve run on
    the device (`gsi_pcgamat_mul`, `gsi_pcgamat_lsqr`)."""

    def __init__(self, etas, HX, R, ctx=None):
        self.ctx = ctx or default_context()
        E = np.asfortranarray(np.stack([np.asarray(e, dtype=np.float64) for e in etas], axis=1))
        self.nobs, self.K = E.shape
        hx = np.ascontiguousarray(HX, dtype=np.float64)
        if hasattr(R, "diagonal") and hasattr(R, "nnz") and R.nnz == np.count_nonzero(R.diagonal()):
            Rv, diag = np.ascontiguousarray(R.diagonal(), dtype=np.float64), 1     # sparse diagonal (the tests' R)
        else:
            Rd = R.toarray() if hasattr(R, "toarray") else np.asarray(R, dtype=np.float64)
            Rv, diag = np.asfortranarray(Rd), 0
        h = C.c_void_p()
        lib = self.ctx.lib
        L.check(lib.gsi_pcgamat_create(self.ctx.h, C.byref(h), L.dptr(E), self.nobs, self.K, hx.ctypes.data_as(L.c_dp),
                                       Rv.ctypes.data_as(L.c_dp), diag), lib)
        self.h = h
        self.ctx._children.append(self)       # destroyed before the context is (its buffers belong to the context)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.gsi_pcgamat_destroy(self.h)
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def shape(self):
        return (self.nobs + 1, self.nobs + 1)

    def size(self, i):
        if i in (1, 2):
            return self.nobs + 1
        raise IndexError(f"there is no {i}-th dimension in a PCGALowRankMatrix")      # lowrank.jl:71

    def matvec(self, x):
        """`*(A::PCGALowRankMatrix, x::Vector)` / `mul!`  (lowrank.jl:83-97, 109-113)."""
        xv = np.ascontiguousarray(x, dtype=np.float64)
        if xv.shape != (self.nobs + 1,):
            raise ValueError("dimension mismatch")
        y = np.empty(self.nobs + 1)
        L.check(self.ctx.lib.gsi_pcgamat_mul(self.ctx.h, self.h, xv.ctypes.data_as(L.c_dp), y.ctypes.data_as(L.c_dp)),
                self.ctx.lib)
        return y

    __matmul__ = matvec

    def lsqr(self, b, *, return_iterations=False):
        """`IterativeSolvers.lsqr(bigA, b)` with that package's defaults  (lsqr.jl:54)."""
        bv = np.ascontiguousarray(b, dtype=np.float64)
        x = np.empty(self.nobs + 1)
        it = C.c_int64()
        L.check(self.ctx.lib.gsi_pcgamat_lsqr(self.ctx.h, self.h, bv.ctypes.data_as(L.c_dp), x.ctypes.data_as(L.c_dp),
                                              C.byref(it)), self.ctx.lib)
        return (x, it.value) if return_iterations else x
