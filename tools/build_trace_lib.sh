#!/bin/bash
# debug build of the library with phase stamps in the LU leaf kernel (-DGSI_LU_TRACE) -> tools/libgsi_hip_trace.so
set -e
cd "$(dirname "$0")/../geostatinversion.jl_amd"
python build.py > /dev/null
mkdir -p /tmp/tracebuild
OBJS=""
for f in gemm_f64.hip panel_qr.hip cholqr.hip jacobi_svd.hip misc.hip fft_cov.hip hip_backend.hip pipeline.cpp api.cpp; do OBJS="$OBJS build/${f/./_}.o"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGSI_LU_TRACE -x hip -c csrc/panel_lu_leaf.hip -o /tmp/tracebuild/panel_lu_leaf_hip.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-Bsymbolic -o ../tools/libgsi_hip_trace.so $OBJS /tmp/tracebuild/panel_lu_leaf_hip.o -ldl
