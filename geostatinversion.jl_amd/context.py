"""Context / operator / device-matrix handles over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib as L


class Context:
    """One MI355X: wraps `gsi_ctx`.  `Context(device)`; multi-GPU: one per process, then
    `comm_init(nranks, rank, unique_id)`."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or L.load()
        h = C.c_void_p()
        L.check(self.lib.gsi_ctx_create(C.byref(h), int(device)), self.lib)
        self.h = h
        self.device = int(device)
        self._children = []

    def close(self):
        if getattr(self, "h", None):
            for ch in list(self._children):
                ch.close()
            self.lib.gsi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        L.check(self.lib.gsi_ctx_sync(self.h), self.lib)

    # ---- multi-GPU ----
    def unique_id(self):
        buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
        L.check(self.lib.gsi_comm_unique_id(buf), self.lib)
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        buf = C.create_string_buffer(bytes(unique_id), L.UNIQUE_ID_BYTES)
        L.check(self.lib.gsi_ctx_comm_init(self.h, int(nranks), int(rank), buf), self.lib)

    def rank(self):
        r, n = C.c_int(), C.c_int()
        L.check(self.lib.gsi_ctx_rank(self.h, C.byref(r), C.byref(n)), self.lib)
        return r.value, n.value

    def host_allgather(self, values):
        """Every rank's `values` (a few host doubles) on every rank, rank-major (`gsi_ctx_host_allgather`): the job's barrier
        and its way to the slowest rank's time -- no second communication layer beside the library's communicator."""
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64)
        out = np.empty((self.rank()[1], v.size))
        L.check(self.lib.gsi_ctx_host_allgather(self.h, L.dptr(v), v.size, L.dptr(out)), self.lib)
        return out

    def barrier(self):
        self.host_allgather([0.0])

    def shard(self, m):
        """Block-row layout the library expects: pad = ceil(m/nranks), row0 = rank*pad."""
        r, n = self.rank()
        pad = -(-m // n)
        row0 = min(r * pad, m)
        return row0, min(pad, m - row0)

    # ---- measurement ----
    def profile(self, on=True):
        """0 / False: off; 1 / True: phase events; 2: additionally skew barriers in front of collectives and panel LUs
        (`comm_wait` phase; several ranks, diagnostic steps only)."""
        L.check(self.lib.gsi_ctx_profile(self.h, int(on)), self.lib)

    def phase_reset(self):
        L.check(self.lib.gsi_ctx_phase_reset(self.h), self.lib)

    def phase_times(self):
        ms = (C.c_double * len(L.PHASES))()
        cnt = (C.c_int64 * len(L.PHASES))()
        L.check(self.lib.gsi_ctx_phase_times(self.h, ms, cnt), self.lib)
        return {name: (ms[i], cnt[i]) for i, name in enumerate(L.PHASES)}

    def counters(self):
        out = (C.c_int64 * 4)()
        L.check(self.lib.gsi_ctx_counters(self.h, out), self.lib)
        return {"cholqr2": out[0], "householder": out[1], "jacobi_sweeps": out[2], "scholqr3": out[3]}

    LU_FORMS = ["none", "replicated", "per-step", "persistent-1hop", "persistent-2hop", "persistent-ov"]

    def path_info(self):
        """Which path ran under the communicator (`gsi_ctx_path_info`): the form of the panel LUs (and how many ran in each
        form since the last phase_reset), the self-test mask of the in-kernel pivot exchange, collectives entered since the
        last phase_reset, the ranks the communicator joined, LU time-outs seen / hidden by a transparent re-run."""
        n = 13
        out = (C.c_int64 * n)()
        L.check(self.lib.gsi_ctx_path_info(self.h, out, n), self.lib)
        forms = {self.LU_FORMS[f]: int(out[6 + f]) for f in range(1, 6) if out[6 + f]}
        return {"lu_form": self.LU_FORMS[out[0]] if 0 <= out[0] < 6 else int(out[0]), "lu_forms_run": forms,
                "lu_selftest_mask": int(out[1]), "collectives": int(out[2]), "n_ranks_seen": int(out[3]),
                "lu_timeouts": int(out[4]), "lu_timeouts_recovered": int(out[5]), "svd_sweep_cap_hits": int(out[12])}

    def release_cache(self):
        """Return cached device memory (released panels, idle workspaces) to the driver (`gsi_ctx_release_cache`)."""
        L.check(self.lib.gsi_ctx_release_cache(self.h), self.lib)

    def device_bytes(self):
        b = C.c_int64()
        L.check(self.lib.gsi_ctx_device_bytes(self.h, C.byref(b)), self.lib)
        return b.value


class _Handle:
    _destroy = None

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        ctx._children.append(self)

    def close(self):
        if getattr(self, "h", None):
            getattr(self.ctx.lib, self._destroy)(self.h)
            self.h = None
            if self in self.ctx._children:
                self.ctx._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Operator(_Handle):
    """Device-resident linear operator (`gsi_op`): the duck type RandMatFact needs
    (size, A*X, A'*X -- RandMatFact.jl:52-55,67,70,85)."""
    _destroy = "gsi_op_destroy"

    @property
    def shape(self):
        m, n = C.c_int64(), C.c_int64()
        L.check(self.ctx.lib.gsi_op_size(self.h, C.byref(m), C.byref(n), None, None), self.ctx.lib)
        return (m.value, n.value)

    def size(self, i):
        if i in (1, 2):
            return self.shape[i - 1]
        raise IndexError(f"there is no {i}-th dimension in a {type(self).__name__}")   # lowrank.jl:58

    def _mul(self, X, trans):
        Xf = L.fmat(X, "X")
        m, n = self.shape
        rows_in, rows_out = (m, n) if trans else (n, m)
        if Xf.shape[0] != rows_in:
            raise ValueError(f"dimension mismatch: operator is {m}x{n}, X has {Xf.shape[0]} rows")
        Y = np.empty((rows_out, Xf.shape[1]), order="F")
        L.check(self.ctx.lib.gsi_op_mul(self.ctx.h, self.h, int(trans), L.dptr(Xf), Xf.shape[0], Xf.shape[1],
                                        L.dptr(Y), rows_out), self.ctx.lib)
        return Y[:, 0].copy() if np.ndim(X) == 1 else Y

    def matmul(self, X):
        """`A * X`"""
        return self._mul(X, 0)

    def rmatmul_t(self, X):
        """`A' * X`"""
        return self._mul(X, 1)

    __matmul__ = matmul


def dense_operator(ctx, A):
    """`Matrix{Float64}` operator.  With a communicator, `A` is the full matrix and only this rank's
    block of rows is uploaded."""
    Af = L.fmat(A, "A")
    m, n = Af.shape
    row0, mloc = ctx.shard(m)
    h = C.c_void_p()
    base = Af[row0:, :] if mloc > 0 else Af
    L.check(ctx.lib.gsi_op_dense(ctx.h, C.byref(h), base.ctypes.data_as(L.c_dp), m, n, Af.shape[0], row0, mloc),
            ctx.lib)
    return Operator(ctx, h)


def gridcov_operator(ctx, nx, ny, ell, kind=0):
    """Synthetic covariance of an nx x ny unit grid generated in HBM.
    kind 0: exp(-d^2/(2 ell^2)); kind 1: exp(-d/ell)."""
    row0, mloc = ctx.shard(nx * ny)
    h = C.c_void_p()
    L.check(ctx.lib.gsi_op_dense_gridcov(ctx.h, C.byref(h), nx, ny, float(ell), int(kind), row0, mloc), ctx.lib)
    return Operator(ctx, h)


def lowrank_synthetic_operator(ctx, n, N, seed=0, decay=0.75):
    """`LowRankCovMatrix` over N synthetic sample fields of n points generated and centred in HBM (SURVEY.md 8d,
    C4-ii): sample j = (j+1)^-decay * iid N(0,1).  Row-sharded like every operator."""
    row0, nloc = ctx.shard(n)
    h = C.c_void_p()
    L.check(ctx.lib.gsi_op_lowrank_synthetic(ctx.h, C.byref(h), int(n), int(N), int(seed), float(decay), row0, nloc),
            ctx.lib)
    return Operator(ctx, h)


def gridcov_implicit_operator(ctx, nx, ny, ell, kind=0, table=None):
    """A stationary grid covariance as an implicit operator: entries are regenerated inside the product kernel, nothing of
    size n^2 is stored.  kind 0: Gaussian exp(-d^2/(2 ell^2)) (two 1-D tables); kind 1:
    exponential exp(-d/ell) (a 2-D table over grid offsets); `table` (nx x ny array, table[dx, dy] = k(dx, dy)): any
    stationary kernel on the grid."""
    row0, mloc = ctx.shard(nx * ny)
    h = C.c_void_p()
    if table is not None:
        t = np.ascontiguousarray(table, dtype=np.float64)
        if t.shape != (nx, ny):
            raise ValueError("table must be nx x ny")
        L.check(ctx.lib.gsi_op_gridcov_implicit_table(ctx.h, C.byref(h), nx, ny, t.ctypes.data_as(L.c_dp), row0, mloc),
                ctx.lib)
    else:
        L.check(ctx.lib.gsi_op_gridcov_implicit_kind(ctx.h, C.byref(h), nx, ny, float(ell), int(kind), row0, mloc),
                ctx.lib)
    return Operator(ctx, h)


POINTCOV_KINDS = {"gaussian": 0, "exponential": 1, "matern32": 2, "matern52": 3}


def pointcov_implicit_operator(ctx, points, kind="exponential", ell=1.0, sigma2=1.0, nugget=0.0):
    """The covariance of scattered points, A[i, j] = sigma2 * k(|x_i - x_j| / ell) (+ nugget on the diagonal), as an
    implicit operator (`gsi_op_pointcov_implicit`): `points` is d x n (point i = column i, d = 1..3 -- the layout of the
    reference's `points::Matrix`, FFTRF.jl:102); kind in "gaussian", "exponential", "matern32", "matern52".  Row panels of
    A are generated on a second stream while the MFMA contraction consumes the previous one; nothing n x n is stored."""
    P = np.asfortranarray(np.asarray(points, dtype=np.float64))
    if P.ndim != 2 or not 1 <= P.shape[0] <= 3:
        raise ValueError("points must be d x n with d = 1, 2 or 3")
    d, n = P.shape
    k = POINTCOV_KINDS[kind] if isinstance(kind, str) else int(kind)
    row0, mloc = ctx.shard(n)
    h = C.c_void_p()
    L.check(ctx.lib.gsi_op_pointcov_implicit(ctx.h, C.byref(h), P.ctypes.data_as(L.c_dp), n, d, k, float(ell), float(sigma2),
                                             float(nugget), row0, mloc), ctx.lib)
    return Operator(ctx, h)


def fft_powerlaw_operator(ctx, Ns, beta, fftrf=False):
    """Matrix-free power-law covariance on a structured grid (circulant embedding, spectrum |k|^beta, unit diagonal).
    `Ns` = grid points per axis (1 to 3 axes); the operator acts on vec(field) in Julia's (column-major) order.
    `fftrf=False` (`gsi_op_fft_powerlaw`): embedding on the next power of two >= 2N, |k| in cycles per grid spacing --
    the covariance family FFTRF samples from, for any grid.  `fftrf=True` (`gsi_op_fft_powerlaw_fftrf`): FFTRF.jl's own
    convention (exactly 2N points per axis, integer wavenumbers, FFTRF.jl:83-90) = the covariance of
    `powerlaw_structuredgrid(Ns, k0, dk, beta)` fields up to dk^2 and the per-sample normalisation; any grid (those that
    are not powers of two through a re-embedded spectrum, same matrix).  With a communicator on `ctx` every rank creates it
    and the products run on row shards."""
    Ns = [int(v) for v in Ns]
    arr = (C.c_int64 * len(Ns))(*Ns)
    h = C.c_void_p()
    fn = ctx.lib.gsi_op_fft_powerlaw_fftrf if fftrf else ctx.lib.gsi_op_fft_powerlaw
    L.check(fn(ctx.h, C.byref(h), len(Ns), arr, float(beta)), ctx.lib)
    return Operator(ctx, h)


class DeviceMatrix(_Handle):
    """Column-major Float64 matrix resident in HBM (`gsi_mat`)."""
    _destroy = "gsi_mat_destroy"

    def __init__(self, ctx, rows, cols):
        h = C.c_void_p()
        L.check(ctx.lib.gsi_mat_create(ctx.h, C.byref(h), int(rows), int(cols)), ctx.lib)
        super().__init__(ctx, h)
        self.shape = (int(rows), int(cols))

    @classmethod
    def from_host(cls, ctx, a):
        af = L.fmat(a)
        m = cls(ctx, *af.shape)
        L.check(ctx.lib.gsi_mat_upload(ctx.h, m.h, L.dptr(af), af.shape[0]), ctx.lib)
        return m

    def randn(self, seed):
        L.check(self.ctx.lib.gsi_mat_randn(self.ctx.h, self.h, int(seed)), self.ctx.lib)
        return self

    def to_host(self):
        out = np.empty(self.shape, order="F")
        L.check(self.ctx.lib.gsi_mat_download(self.ctx.h, self.h, L.dptr(out), self.shape[0]), self.ctx.lib)
        return out


_default_ctx = None


def default_context():
    """Process-wide context on device 0 (LOCAL_RANK if set), created on first use."""
    global _default_ctx
    if _default_ctx is None or _default_ctx.h is None:
        import os
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx
