// pointcov.hip -- row panels of a scattered-point covariance matrix, generated where they are consumed (SURVEY.md 8b:
// "kernel-function covariance (coords + kernel id + params)"; the reference takes such covariances as dense matrices,
// getxis(Q::Matrix, ...) GeostatInversion.jl:63, and places unstructured points as a d x n matrix, FFTRF.jl:102).
// A(i, j) = sigma2 k(|x_i - x_j| / ell) (+ nugget on the diagonal) has no table to look up -- every entry costs a distance,
// a sqrt and an exp -- so the product is ROW-STREAMED: this kernel fills an R x k panel of A in HBM on a second stream
// while the MFMA contraction (gemm_f64.hip, stored operand) consumes the previous panel; two panels ping-pong
// (hip_backend.hip:gemm_nn_pointcov).  VALU + HBM-write bound; the contraction it feeds is MFMA-bound, so the two overlap.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "hip_common.hpp"
#include "pointcov.hpp"

namespace gsi { namespace hipk {

// P[r + j ldp] = A(roff + r, koff + j), r < rows, j < cols.  Thread = one row point (coordinates in registers) walking a
// chunk of 64 columns: a wavefront writes 512 contiguous bytes per column; the column point is a wave-uniform (scalar) load.
template <int D>
__global__ __launch_bounds__(256) void pointcov_panel_kernel(double* __restrict__ P, int64_t ldp, int64_t rows, int64_t cols,
                                                             const double* __restrict__ pts, pointcov::Params prm,
                                                             int64_t roff, int64_t koff) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.y * 64;
  const bool live = r < rows;
  const int64_t gi = roff + (live ? r : 0);
  double xi[D];
#pragma unroll
  for (int a = 0; a < D; ++a) xi[a] = pts[gi * D + a];
  const int64_t jend = (j0 + 64 < cols) ? j0 + 64 : cols;
  for (int64_t j = j0; j < jend; ++j) {
    const int64_t gj = koff + j;
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) { const double t = xi[a] - pts[gj * D + a]; d2 += t * t; }
    if (live) P[r + j * ldp] = pointcov::kernel(prm, d2, gi == gj);
  }
}

void pointcov_panel(hipStream_t st, double* P, int64_t ldp, int64_t rows, int64_t cols, const double* pts,
                    const pointcov::Params& prm, int64_t roff, int64_t koff) {
  if (rows <= 0 || cols <= 0) return;
  dim3 grid((unsigned)((rows + 255) / 256), (unsigned)((cols + 63) / 64));
  if (prm.d == 1) hipLaunchKernelGGL(pointcov_panel_kernel<1>, grid, dim3(256), 0, st, P, ldp, rows, cols, pts, prm, roff, koff);
  else if (prm.d == 2) hipLaunchKernelGGL(pointcov_panel_kernel<2>, grid, dim3(256), 0, st, P, ldp, rows, cols, pts, prm, roff, koff);
  else hipLaunchKernelGGL(pointcov_panel_kernel<3>, grid, dim3(256), 0, st, P, ldp, rows, cols, pts, prm, roff, koff);
}

}}  // namespace gsi::hipk
