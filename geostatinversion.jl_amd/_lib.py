"""ctypes binding of libgsi_hip.so (the C ABI in include/gsi_hip.h).

There is NO fallback: if the library is missing, or no gfx950 GPU is visible when a context is
created, this raises.  Nothing here computes anything -- it is the same thin shim a Julia
`ccall` wrapper is, written in Python because Julia is not in this image.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgsi_hip.so")

GSI_OK = 0
ERR_NAMES = {1: "GSI_ERR_ARG", 2: "GSI_ERR_NEG_ITERS", 3: "GSI_ERR_SINGULAR", 4: "GSI_ERR_HIP",
             5: "GSI_ERR_RCCL", 6: "GSI_ERR_OOM", 7: "GSI_ERR_NOT_POSDEF", 8: "GSI_ERR_INTERNAL"}
PHASES = ["gemm_n", "gemm_t", "lu", "qr", "svd", "small_gemm", "comm", "other", "comm_wait"]
UNIQUE_ID_BYTES = 128

c_i64 = C.c_int64
c_dp = C.POINTER(C.c_double)
c_vp = C.c_void_p
RANDN_FN = C.CFUNCTYPE(None, c_vp, c_dp, c_i64)

# name -> (restype, argtypes): every symbol include/gsi_hip.h declares
SIGNATURES = {
    "gsi_version": (C.c_int, []),
    "gsi_last_error": (C.c_char_p, []),
    "gsi_backend_name": (C.c_char_p, []),
    "gsi_ctx_create": (C.c_int, [C.POINTER(c_vp), C.c_int]),
    "gsi_ctx_destroy": (C.c_int, [c_vp]),
    "gsi_ctx_sync": (C.c_int, [c_vp]),
    "gsi_comm_unique_id": (C.c_int, [c_vp]),
    "gsi_ctx_comm_init": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp]),
    "gsi_ctx_rank": (C.c_int, [c_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gsi_ctx_host_allgather": (C.c_int, [c_vp, c_dp, c_i64, c_dp]),
    "gsi_op_dense": (C.c_int, [c_vp, C.POINTER(c_vp), c_dp, c_i64, c_i64, c_i64, c_i64, c_i64]),
    "gsi_op_lowrank": (C.c_int, [c_vp, C.POINTER(c_vp), c_dp, c_i64, c_i64, c_i64, C.c_int, c_i64, c_i64]),
    "gsi_op_lowrank_synthetic": (C.c_int, [c_vp, C.POINTER(c_vp), c_i64, c_i64, C.c_uint64, C.c_double, c_i64, c_i64]),
    "gsi_op_lowrank_samples": (C.c_int, [c_vp, c_vp, c_dp, c_i64]),
    "gsi_op_dense_gridcov": (C.c_int, [c_vp, C.POINTER(c_vp), c_i64, c_i64, C.c_double, C.c_int, c_i64, c_i64]),
    "gsi_op_gridcov_implicit": (C.c_int, [c_vp, C.POINTER(c_vp), c_i64, c_i64, C.c_double, c_i64, c_i64]),
    "gsi_op_gridcov_implicit_kind": (C.c_int, [c_vp, C.POINTER(c_vp), c_i64, c_i64, C.c_double, C.c_int, c_i64, c_i64]),
    "gsi_op_gridcov_implicit_table": (C.c_int, [c_vp, C.POINTER(c_vp), c_i64, c_i64, c_dp, c_i64, c_i64]),
    "gsi_op_pointcov_implicit": (C.c_int, [c_vp, C.POINTER(c_vp), c_dp, c_i64, C.c_int, C.c_int, C.c_double, C.c_double,
                                           C.c_double, c_i64, c_i64]),
    "gsi_op_fft_powerlaw": (C.c_int, [c_vp, C.POINTER(c_vp), C.c_int, C.POINTER(c_i64), C.c_double]),
    "gsi_op_fft_powerlaw_fftrf": (C.c_int, [c_vp, C.POINTER(c_vp), C.c_int, C.POINTER(c_i64), C.c_double]),
    "gsi_op_destroy": (C.c_int, [c_vp]),
    "gsi_op_size": (C.c_int, [c_vp, C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(c_i64)]),
    "gsi_op_mul": (C.c_int, [c_vp, c_vp, C.c_int, c_dp, c_i64, c_i64, c_dp, c_i64]),
    "gsi_rangefinder": (C.c_int, [c_vp, c_vp, c_dp, c_i64, c_i64, c_dp]),
    "gsi_randsvd": (C.c_int, [c_vp, c_vp, c_dp, c_i64, c_i64, c_i64, c_dp, c_dp]),
    "gsi_randsvd_dense_host": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_i64, c_dp, c_i64, c_i64, c_i64, c_dp, c_dp, C.POINTER(c_vp)]),
    "gsi_rangefinder_dense_host": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_i64, c_dp, c_i64, c_i64, c_dp, C.POINTER(c_vp)]),
    "gsi_eig_nystrom": (C.c_int, [c_vp, c_vp, c_dp, c_i64, c_dp, c_dp]),
    "gsi_rangefinder_adaptive": (C.c_int, [c_vp, c_vp, RANDN_FN, c_vp, C.c_double, c_i64, c_dp, C.POINTER(c_i64)]),
    "gsi_mat_create": (C.c_int, [c_vp, C.POINTER(c_vp), c_i64, c_i64]),
    "gsi_mat_destroy": (C.c_int, [c_vp]),
    "gsi_mat_upload": (C.c_int, [c_vp, c_vp, c_dp, c_i64]),
    "gsi_mat_download": (C.c_int, [c_vp, c_vp, c_dp, c_i64]),
    "gsi_mat_randn": (C.c_int, [c_vp, c_vp, C.c_uint64]),
    "gsi_op_mul_dev": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, c_vp]),
    "gsi_rangefinder_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "gsi_randsvd_dev": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "gsi_randsvd_rows": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "gsi_lu_L": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_dp, C.POINTER(C.c_int32)]),
    "gsi_lu_L_dev": (C.c_int, [c_vp, c_vp, C.POINTER(C.c_int32)]),
    "gsi_lu_L_sharded": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_dp, C.POINTER(C.c_int32)]),
    "gsi_lu_L_sharded_virtual": (C.c_int, [c_vp, c_dp, c_i64, c_i64, C.c_int, c_dp, C.POINTER(C.c_int32)]),
    "gsi_qr_thinQ": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_dp, c_dp]),
    "gsi_svd_tall": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_dp, c_dp]),
    "gsi_gemm": (C.c_int, [c_vp, C.c_int, c_i64, c_i64, c_i64, C.c_double, c_dp, c_i64, c_dp, c_i64, c_dp, c_i64]),
    "gsi_pcga_params": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_dp, c_dp, C.c_double, c_dp]),
    "gsi_pcga_update": (C.c_int, [c_vp, c_dp, c_i64, c_i64, c_dp, C.c_double, c_dp, c_i64, c_dp, c_dp]),
    "gsi_pcga_params_dev": (C.c_int, [c_vp, c_vp, c_i64, c_dp, c_dp, C.c_double, c_dp]),
    "gsi_pcga_update_dev": (C.c_int, [c_vp, c_vp, c_i64, c_dp, C.c_double, c_dp, c_i64, c_dp, c_dp]),
    "gsi_mat_download_col": (C.c_int, [c_vp, c_vp, c_i64, c_dp]),
    "gsi_op_lowrank_solve": (C.c_int, [c_vp, c_vp, c_dp, c_dp, C.POINTER(c_i64)]),
    "gsi_pcgamat_create": (C.c_int, [c_vp, C.POINTER(c_vp), c_dp, c_i64, c_i64, c_dp, c_dp, C.c_int]),
    "gsi_pcgamat_destroy": (C.c_int, [c_vp]),
    "gsi_pcgamat_mul": (C.c_int, [c_vp, c_vp, c_dp, c_dp]),
    "gsi_pcgamat_lsqr": (C.c_int, [c_vp, c_vp, c_dp, c_dp, C.POINTER(c_i64)]),
    "gsi_basis_create": (C.c_int, [c_vp, C.POINTER(c_vp), c_vp, c_i64, C.c_int]),
    "gsi_basis_destroy": (C.c_int, [c_vp]),
    "gsi_pcga_params_basis": (C.c_int, [c_vp, c_vp, c_dp, c_dp, C.c_double, c_dp]),
    "gsi_pcga_update_basis": (C.c_int, [c_vp, c_vp, c_dp, C.c_double, c_dp, c_i64, c_dp, c_dp]),
    "gsi_basis_download_col": (C.c_int, [c_vp, c_vp, c_i64, c_dp]),
    "gsi_ctx_profile": (C.c_int, [c_vp, C.c_int]),
    "gsi_ctx_phase_reset": (C.c_int, [c_vp]),
    "gsi_ctx_phase_times": (C.c_int, [c_vp, c_dp, C.POINTER(c_i64)]),
    "gsi_ctx_counters": (C.c_int, [c_vp, C.POINTER(c_i64)]),
    "gsi_ctx_path_info": (C.c_int, [c_vp, C.POINTER(c_i64), c_i64]),
    "gsi_ctx_pinned_copy_rate": (C.c_int, [c_vp, c_i64, c_dp, c_dp]),
    "gsi_ctx_release_cache": (C.c_int, [c_vp]),
    "gsi_ctx_device_bytes": (C.c_int, [c_vp, C.POINTER(c_i64)]),
}


class GsiError(RuntimeError):
    """Raised for every non-zero status: the analogue of Julia's `error(...)` on the reference path
    (RandMatFact.jl:63, lowrank.jl:58)."""

    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


_lib = None


def load(path=None):
    """dlopen the product library and bind every symbol of the header; raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GSI_HIP_LIB") or LIB_PATH     # GSI_HIP_LIB: another build of the same library (A/B runs)
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build it with `python {os.path.join(_HERE, 'build.py')}` (needs hipcc). "
            "This package has no CPU fallback.")
    lib = C.CDLL(p, mode=C.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status, lib=None):
    if status != GSI_OK:
        lib = lib or load()
        raise GsiError(status, (lib.gsi_last_error() or b"").decode("utf-8", "replace"))


def fmat(a, name="matrix"):
    """A column-major float64 view/copy of `a` (what a Julia Matrix{Float64} is in memory)."""
    arr = np.asarray(a, dtype=np.float64)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    if arr.ndim != 2:
        raise ValueError(f"{name} must be 1-D or 2-D")
    return np.asfortranarray(arr)


def dptr(arr):
    return arr.ctypes.data_as(c_dp)
