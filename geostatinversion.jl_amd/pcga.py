"""Host-side mirror of the consumers of the xi-basis: `pcgadirect` (src/direct.jl), `pcgalsqr`
(src/lsqr.jl) and `rga` (src/GeostatInversion.jl:101-103).  The forward model is user code and runs
on the host exactly as the reference's `pmap` does; the package's own n-sized algebra (the
perturbation batch and the update s = X*beta + sum xis[i]*dot(eta_i, xi_bar)) and pcgalsqr's
saddle-point LSQR (PCGALowRankMatrix products, IterativeSolvers defaults) run on the GPU."""
import numpy as np

from . import _lib as L
from .context import default_context
from .lowrank import PCGALowRankMatrix

SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))


class DeviceBasis:
    """The xi-basis resident in HBM: the n x (K+p) matrix Z a device-side randsvd left
    there; `xis[i]` is its column i.  Pass it to `pcgadirect` / `pcgalsqr` / `rga` in place of the
    reference's `xis::Array{Array{Float64,1},1}`: the basis then crosses PCIe never (only the n x (K+3)
    perturbation batch and the updated s do, because the forward model is host code).
    `precision=32` keeps an fp32 copy of the K columns instead (BASELINE configs[4], "fp32 mixed precision": half
    the HBM bytes of every iteration's own algebra; all sums in fp64)."""

    def __init__(self, Zmat, K, precision=64):
        import ctypes as C
        self.Zmat = Zmat
        self.ctx = Zmat.ctx
        self.n = Zmat.shape[0]
        self.K = int(K)
        self.precision = int(precision)
        if not 1 <= self.K <= Zmat.shape[1]:
            raise ValueError("K out of range for this basis")
        h = C.c_void_p()
        L.check(self.ctx.lib.gsi_basis_create(self.ctx.h, C.byref(h), Zmat.h, self.K, self.precision), self.ctx.lib)
        self.h = h
        self.ctx._children.append(self)   # destroyed before the context is
        if self.precision == 32:
            self.Zmat = None            # the fp64 matrix is no longer needed by this basis

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.gsi_basis_destroy(self.h)
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.K

    def __getitem__(self, i):
        """xis[i] on the host (one column)."""
        if not 0 <= i < self.K:
            raise IndexError(i)
        out = np.empty(self.n)
        L.check(self.ctx.lib.gsi_basis_download_col(self.ctx.h, self.h, i, out.ctypes.data_as(L.c_dp)), self.ctx.lib)
        return out

    def params(self, s, X, delta):
        out = np.empty((self.n, self.K + 3), order="F")
        s = np.ascontiguousarray(s, dtype=np.float64)
        X = np.ascontiguousarray(X, dtype=np.float64)
        L.check(self.ctx.lib.gsi_pcga_params_basis(self.ctx.h, self.h, s.ctypes.data_as(L.c_dp),
                                                   X.ctypes.data_as(L.c_dp), float(delta), L.dptr(out)), self.ctx.lib)
        return out

    def update(self, X, beta_bar, etas, xi_bar):
        E = np.asfortranarray(np.stack(etas, axis=1))
        X = np.ascontiguousarray(X, dtype=np.float64)
        xb = np.ascontiguousarray(xi_bar, dtype=np.float64)
        out = np.empty(self.n)
        L.check(self.ctx.lib.gsi_pcga_update_basis(self.ctx.h, self.h, X.ctypes.data_as(L.c_dp), float(beta_bar),
                                                   L.dptr(E), E.shape[0], xb.ctypes.data_as(L.c_dp),
                                                   out.ctypes.data_as(L.c_dp)), self.ctx.lib)
        return out


class ShardedDeviceBasis(DeviceBasis):
    """A ROW-SHARDED xi-basis (BASELINE configs[4]): this rank's rows of Z as `gsi_randsvd_rows` left them,
    one process per GPU.  `params` / `update` act on this rank's rows of s, X and paramstorun (nothing of size n x K is
    replicated); `gather(rows) -> whole` is the host-side all-gather of row blocks the caller's launcher provides
    (torch.distributed, MPI, Julia's Distributed ...), used only to hand whole parameter vectors to the forward model
    (user code on the host, as the reference's pmap does) and for the convergence norm."""

    def __init__(self, Zmat_rows, K, gather, precision=64):
        super().__init__(Zmat_rows, K, precision=precision)
        self.gather = gather


class _Basis:
    """xis given on the host (the reference's Vector{Vector{Float64}}) as one n x K column-major block."""

    def __init__(self, xis, ctx):
        self.ctx = ctx or default_context()
        self.Z = np.asfortranarray(np.stack([np.asarray(x, dtype=np.float64) for x in xis], axis=1))
        self.n, self.K = self.Z.shape

    def params(self, s, X, delta):
        out = np.empty((self.n, self.K + 3), order="F")
        lib, cx = self.ctx.lib, self.ctx
        s = np.ascontiguousarray(s, dtype=np.float64)
        X = np.ascontiguousarray(X, dtype=np.float64)
        L.check(lib.gsi_pcga_params(cx.h, L.dptr(self.Z), self.n, self.K, s.ctypes.data_as(L.c_dp),
                                    X.ctypes.data_as(L.c_dp), float(delta), L.dptr(out)), lib)
        return out

    def update(self, X, beta_bar, etas, xi_bar):
        E = np.asfortranarray(np.stack(etas, axis=1))
        X = np.ascontiguousarray(X, dtype=np.float64)
        xb = np.ascontiguousarray(xi_bar, dtype=np.float64)
        out = np.empty(self.n)
        lib, cx = self.ctx.lib, self.ctx
        L.check(lib.gsi_pcga_update(cx.h, L.dptr(self.Z), self.n, self.K, X.ctypes.data_as(L.c_dp), float(beta_bar),
                                    L.dptr(E), E.shape[0], xb.ctypes.data_as(L.c_dp), out.ctypes.data_as(L.c_dp)),
                lib)
        return out


def _as_basis(xis, ctx):
    return xis if isinstance(xis, DeviceBasis) else _Basis(xis, ctx)


def _iteration_head(forwardmodel, basis, s, X, delta):
    """direct.jl:38-46 / lsqr.jl:36-51."""
    K = basis.K
    P = basis.params(s, X, delta)                                     # paramstorun
    if getattr(basis, "gather", None) is not None:                    # row-sharded basis: whole vectors for the forward model
        P = basis.gather(P)
    results = [np.asarray(forwardmodel(np.ascontiguousarray(P[:, i])), dtype=np.float64) for i in range(K + 3)]
    hs = results[K + 2]
    etas = [(results[i] - hs) / delta for i in range(K)]
    HX = (results[K] - hs) / delta
    Hs = (results[K + 1] - hs) / delta
    return etas, HX, Hs, hs


def _global_norm(basis, d):
    """norm(s - olds) (direct.jl:29, lsqr.jl:27); over all ranks' rows when the basis is row-sharded"""
    if getattr(basis, "gather", None) is not None:
        d = basis.gather(d)
    return float(np.linalg.norm(d))


def pcgadirect(forwardmodel, s0, X, xis, R, y, *, maxiters=5, delta=SQRT_EPS, xtol=1e-6,
               callback=lambda s, obs_cal: None, ctx=None):
    """`pcgadirect(forwardmodel, s0, X, xis, R, y; maxiters=5, delta=sqrt(eps), xtol=1e-6, callback)`
    (direct.jl:21-67)."""
    basis = _as_basis(xis, ctx)
    s = np.asarray(s0, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    Rd = R.toarray() if hasattr(R, "toarray") else np.asarray(R, dtype=np.float64)
    converged, it = False, 0
    while not converged and it < maxiters:
        olds = s
        etas, HX, Hs, hs = _iteration_head(forwardmodel, basis, s, X, delta)
        callback(s, hs)                                               # :47
        E = np.stack(etas, axis=1)
        HQH = E @ E.T                                                 # :49-53
        b = np.concatenate([y - hs + Hs, np.zeros(1)])                # :56
        bigA = np.block([[HQH + Rd, HX[:, None]], [HX[None, :], np.zeros((1, 1))]])   # :57
        x = np.linalg.pinv(bigA) @ b                                  # :58
        s = basis.update(X, x[-1], etas, x[:-1])                      # :59-65
        if _global_norm(basis, s - olds) < xtol:
            converged = True
        it += 1
    return s


def pcgalsqr(forwardmodel, s0, X, xis, R, y, *, maxiters=5, delta=SQRT_EPS, xtol=1e-6, ctx=None):
    """`pcgalsqr(forwardmodel, s0, X, xis, R, y; maxiters=5, delta=sqrt(eps), xtol=1e-6)`
    (lsqr.jl:20-63).  No `callback` keyword, as in the reference."""
    basis = _as_basis(xis, ctx)
    s = np.asarray(s0, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    converged, it = False, 0
    while not converged and it < maxiters:
        olds = s
        etas, HX, Hs, hs = _iteration_head(forwardmodel, basis, s, X, delta)
        b = np.concatenate([y - hs + Hs, np.zeros(1)])                # :52
        bigA = PCGALowRankMatrix(etas, HX, R, ctx=basis.ctx)          # :53
        x = bigA.lsqr(b)                                              # :54  (LSQR on the device)
        bigA.close()
        s = basis.update(X, x[-1], etas, x[:-1])                      # :55-61
        if _global_norm(basis, s - olds) < xtol:
            converged = True
        it += 1
    return s


def rga(forwardmodel, s0, X, xis, R, y, S, *, maxiters=5, delta=SQRT_EPS, xtol=1e-6, pcgafunc=pcgadirect,
        callback=lambda s, obs_cal: None):
    """`rga(...; pcgafunc=pcgadirect, callback)`  (GeostatInversion.jl:101-103)."""
    S = np.asarray(S, dtype=np.float64)
    Rd = R.toarray() if hasattr(R, "toarray") else np.asarray(R, dtype=np.float64)
    return pcgafunc(lambda x: S @ forwardmodel(x), s0, X, xis, S @ Rd @ S.T, S @ np.asarray(y, dtype=np.float64),
                    maxiters=maxiters, delta=delta, xtol=xtol, callback=callback)


pcga = pcgadirect     # GeostatInversion.jl:105
