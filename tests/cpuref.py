"""Test helper: load the CPU reference build of the SAME C ABI (oracle/_build/libgsi_cpuref.so =
pipeline.cpp + api.cpp + the C restatement) so the host-side logic can run without a GPU.
Only tests use this; the product loader never opens it."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# GSI_CPUREF_LIB / GSI_ORACLE_C_LIB: point the CPU suite at the sanitizer builds (`make -C oracle asan`)
CPUREF = os.environ.get("GSI_CPUREF_LIB") or os.path.join(ROOT, "oracle", "_build", "libgsi_cpuref.so")
ORACLE_C = os.environ.get("GSI_ORACLE_C_LIB") or os.path.join(ROOT, "oracle", "_build", "libgsi_oracle.so")

ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int64)
ALLGATHER_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def load_cpuref():
    import gsi_amd
    if not os.path.exists(CPUREF):
        build()
    lib = gsi_amd._lib.load(CPUREF)        # binds every header symbol on this library too
    lib.gsi_cpuref_set_collectives.restype = None
    lib.gsi_cpuref_set_collectives.argtypes = [ALLREDUCE_FN, ALLGATHER_FN]
    return lib


def load_oracle_c():
    if not os.path.exists(ORACLE_C):
        build()
    lib = C.CDLL(ORACLE_C)
    dp, i64 = C.POINTER(C.c_double), C.c_int64
    lib.gsio_lu_L.restype = C.c_int
    lib.gsio_lu_L.argtypes = [dp, i64, i64, i64, C.POINTER(C.c_int32)]
    lib.gsio_qr_thinQ.restype = None
    lib.gsio_qr_thinQ.argtypes = [dp, i64, i64, i64, C.c_int, dp, C.POINTER(C.c_int32)]
    lib.gsio_svd_tall.restype = C.c_int
    lib.gsio_svd_tall.argtypes = [dp, i64, i64, i64, dp]
    lib.gsio_rangefinder.restype = C.c_int
    lib.gsio_rangefinder.argtypes = [dp, i64, i64, i64, dp, i64, i64, dp]
    lib.gsio_randsvd.restype = C.c_int
    lib.gsio_randsvd.argtypes = [dp, i64, i64, i64, dp, i64, i64, i64, dp, dp]
    lib.gsio_eig_nystrom.restype = C.c_int
    lib.gsio_eig_nystrom.argtypes = [dp, i64, i64, dp, i64, dp, dp]
    return lib
