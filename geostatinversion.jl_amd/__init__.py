"""geostatinversion.jl_amd -- MI355X (gfx950) implementation of the randomized low-rank
factorization hot path of GeostatInversion.jl (RandMatFact.jl reached through getxis()).

Product = libgsi_hip.so (hand-written HIP kernels behind the C ABI of include/gsi_hip.h).
This package is the host-side mirror of the reference's interface for that path -- the same
function names and argument meaning as the Julia module -- and is nothing but ctypes calls
into the library.  Importing it never silently degrades: no library, or no gfx950 GPU, raises.

The directory name contains a dot, so import it through the `gsi_amd` alias at the repo root:
    import gsi_amd as gsi
"""
from . import _lib
from ._lib import GsiError, load, LIB_PATH
from .context import (Context, Operator, DeviceMatrix, dense_operator, gridcov_operator,
                      gridcov_implicit_operator, pointcov_implicit_operator, fft_powerlaw_operator, lowrank_synthetic_operator,
                      default_context)
from . import randmatfact as RandMatFact
from .randmatfact import rangefinder, randsvd, randsvd_rows, eig_nystrom, colnorms, lu_L, lu_L_dev, lu_L_sharded, lu_L_sharded_virtual, qr_thinQ, svd_tall, gemm
from .lowrank import LowRankCovMatrix, PCGALowRankMatrix, device_samples
from .getxis import getxis, getxis_iwantfields, getxis_device, randsvdwithseed
from .pcga import pcgadirect, pcgalsqr, rga, pcga, DeviceBasis, ShardedDeviceBasis

__all__ = [
    "GsiError", "load", "LIB_PATH", "Context", "Operator", "DeviceMatrix", "dense_operator",
    "gridcov_operator", "gridcov_implicit_operator", "pointcov_implicit_operator", "fft_powerlaw_operator", "lowrank_synthetic_operator", "default_context", "RandMatFact", "rangefinder", "randsvd", "randsvd_rows", "eig_nystrom",
    "colnorms", "lu_L", "lu_L_dev", "lu_L_sharded", "lu_L_sharded_virtual", "qr_thinQ", "svd_tall", "gemm", "LowRankCovMatrix", "PCGALowRankMatrix", "device_samples",
    "getxis", "getxis_iwantfields", "getxis_device", "randsvdwithseed", "pcgadirect", "pcgalsqr", "rga", "pcga",
    "DeviceBasis", "ShardedDeviceBasis",
]
