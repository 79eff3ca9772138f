"""Host-side mirror of `getxis` / `randsvdwithseed`  (src/GeostatInversion.jl:20-70)."""
import numpy as np

from . import randmatfact as RMF
from .lowrank import LowRankCovMatrix


def randsvdwithseed(Q, numxis, p, q, seed=None, *, Omega=None):
    """`randsvdwithseed(Q, numxis, p, q, seed::Nothing|Int)`  (GeostatInversion.jl:20-27)."""
    if seed is not None:
        if not isinstance(seed, (int, np.integer)):
            raise TypeError("seed must be an Int or None")      # the reference has no other method
        RMF.seed(int(seed))                                      # Random.seed!(seed)   :25
    return RMF.randsvd(Q, numxis, p, q, Omega=Omega)


def getxis_iwantfields(samplefield, numfields, numxis, p, q=3, seed=None, *, Omega=None, ctx=None):
    """`getxis(Val{:iwantfields}, samplefield, numfields, numxis, p, q=3, seed)`  (:29-38) -> (xis, fields)."""
    fields = [np.asarray(samplefield(), dtype=np.float64) for _ in range(numfields)]   # rpmap   :30
    lrcm = LowRankCovMatrix(fields, ctx=ctx)                     # :31
    try:
        Z = randsvdwithseed(lrcm, numxis, p, q, seed, Omega=Omega)   # :32
    finally:
        lrcm.close()
    xis = [np.ascontiguousarray(Z[:, i]) for i in range(numxis)]     # :33-36
    return xis, fields


def getxis(first, *args, **kwargs):
    """Both reference methods (GeostatInversion.jl:58-70):

    `getxis(samplefield::Function, numfields, numxis, p, q=3, seed=nothing)`
    `getxis(Q::Matrix, numxis, p, q=3, seed=nothing)`
    """
    if callable(first):
        xis, _ = getxis_iwantfields(first, *args, **kwargs)      # :58-61
        return xis
    return _getxis_matrix(first, *args, **kwargs)


def _getxis_matrix(Q, numxis, p, q=3, seed=None, *, Omega=None):
    Z = randsvdwithseed(Q, numxis, p, q, seed, Omega=Omega)     # :65
    return [np.ascontiguousarray(Z[:, i]) for i in range(numxis)]   # :66-68


def getxis_device(Q, numxis, p, q=3, seed=None, *, Omega=None, ctx=None, precision=64):
    """`getxis` whose result stays in HBM: returns a `DeviceBasis` usable wherever the
    reference takes `xis` (`pcgadirect`, `pcgalsqr`, `rga`).  Q: a matrix, a `LowRankCovMatrix` or a device
    `Operator`.  `precision=32`: the basis is kept in fp32 (mixed precision, BASELINE configs[4])."""
    import ctypes as C
    from . import _lib as L
    from .context import DeviceMatrix, Operator, dense_operator, default_context
    from .pcga import DeviceBasis
    if seed is not None:
        RMF.seed(int(seed))
    owned = False
    if isinstance(Q, Operator):
        op = Q
    elif hasattr(Q, "_device_operator"):
        op = Q._device_operator(ctx)
    else:
        op, owned = dense_operator(ctx or default_context(), Q), True
    try:
        cx = op.ctx
        m, n = op.shape
        l = int(numxis) + int(p)
        Om = RMF.randn(n, l) if Omega is None else Omega
        Omd = DeviceMatrix.from_host(cx, Om)
        Z = DeviceMatrix(cx, n, l)
        L.check(cx.lib.gsi_randsvd_dev(cx.h, op.h, Omd.h, int(numxis), int(p), int(q), Z.h, None), cx.lib)
        Omd.close()
        return DeviceBasis(Z, numxis, precision=precision)
    finally:
        if owned:
            op.close()
