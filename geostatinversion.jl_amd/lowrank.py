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

    def close(self):
        if self._op is not None:
            self._op.close()
            self._op = None


class PCGALowRankMatrix:
    """`PCGALowRankMatrix(etas, HX, R)`  (lowrank.jl:32-36, 62-73, 83-97): the saddle-point matrix
    [(HQH+R) HX; HX' 0] with HQH = sum eta_i eta_i' kept implicit.  nobs-sized host algebra, as in
    the reference; it is the LSQR operator of pcgalsqr (lsqr.jl:53-54)."""

    def __init__(self, etas, HX, R):
        self.E = np.asfortranarray(np.stack([np.asarray(e, dtype=np.float64) for e in etas], axis=1))
        self.HX = np.asarray(HX, dtype=np.float64)
        self.R = R

    @property
    def shape(self):
        s = self.E.shape[0] + 1
        return (s, s)

    def size(self, i):
        if i in (1, 2):
            return self.E.shape[0] + 1
        raise IndexError(f"there is no {i}-th dimension in a PCGALowRankMatrix")

    def matvec(self, x):
        x = np.asarray(x, dtype=np.float64)
        xs = x[:-1]
        v = np.empty(len(x))
        v[:-1] = self.R @ xs + self.E @ (self.E.T @ xs) + self.HX * x[-1]
        v[-1] = np.dot(self.HX, xs)
        return v

    __matmul__ = matvec
