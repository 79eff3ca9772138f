// lsqr_dev.hip -- the vector side of IterativeSolvers.lsqr (lsqr.jl:54, lowrank.jl:142 of the reference) with every scalar
// of the recurrence resident in HBM (lsqr_state.hpp): three fused launch groups per iteration around the two operator
// products, no host synchronisation inside an iteration.  Norms are two-stage reductions with a fixed grid and a fixed
// summation order (deterministic).  HBM-bound BLAS-1 work; at nobs ~ 4096 it is launch latency that counts, which is why
// nothing here waits for the host.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "hip_common.hpp"
#include "lsqr_state.hpp"

namespace gsi { namespace hipk {

namespace {
using namespace gsi::lsqrst;
constexpr int LSQR_PARTS = 128;      // partial sums per norm

__device__ inline double block_sum_256(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  return sh[0];
}
__device__ inline double sum_parts(const double* part) {     // fixed order
  double s = 0.0;
  for (int i = 0; i < LSQR_PARTS; ++i) s += part[i];
  return s;
}

// u = t - alpha u, partial sums of |u|^2
__global__ __launch_bounds__(256) void lsqr_u_kernel(int64_t m, const double* __restrict__ t, double* __restrict__ u,
                                                     const double* __restrict__ s, double* __restrict__ part) {
  __shared__ double sh[256];
  double acc = 0.0;
  if (s[STOPPED] == 0.0 && s[ITERS] < s[MAXITER]) {
    const double alpha = s[ALPHA];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
      const double v = t[i] - alpha * u[i];
      u[i] = v;
      acc += v * v;
    }
  }
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ void lsqr_after_u_kernel(double* __restrict__ s, const double* __restrict__ part) {
  if (threadIdx.x == 0 && blockIdx.x == 0) after_u(s, sum_parts(part));
}
// u *= 1 / beta
__global__ __launch_bounds__(256) void lsqr_scale_u_kernel(int64_t m, double* __restrict__ u, const double* __restrict__ s) {
  if (s[APPLY] == 0.0 || !(s[BETA] > 0.0)) return;
  const double inv = 1.0 / s[BETA];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) u[i] *= inv;
}
// v = t - beta v (beta > 0 only), partial sums of |v|^2
__global__ __launch_bounds__(256) void lsqr_v_kernel(int64_t n, const double* __restrict__ t, double* __restrict__ v,
                                                     const double* __restrict__ s, double* __restrict__ part) {
  __shared__ double sh[256];
  double acc = 0.0;
  if (s[APPLY] != 0.0 && s[BETA] > 0.0) {
    const double beta = s[BETA];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const double x = t[i] - beta * v[i];
      v[i] = x;
      acc += x * x;
    }
  }
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ void lsqr_after_v_kernel(double* __restrict__ s, const double* __restrict__ partv, const double* __restrict__ partw) {
  if (threadIdx.x == 0 && blockIdx.x == 0) after_v(s, sum_parts(partv), sum_parts(partw));
}
// v *= 1 / alpha (when it was recomputed and is positive); x += t1 w; w = v + t2 w; partial sums of the new |w|^2
__global__ __launch_bounds__(256) void lsqr_update_kernel(int64_t n, double* __restrict__ v, double* __restrict__ w,
                                                          double* __restrict__ x, const double* __restrict__ s,
                                                          double* __restrict__ partw) {
  __shared__ double sh[256];
  if (s[APPLY] == 0.0) return;                 // partw keeps |w|^2 of the last applied iteration
  const bool rescale = (s[BETA] > 0.0 && s[ALPHA] > 0.0);
  const double inv = rescale ? 1.0 / s[ALPHA] : 1.0;
  const double t1 = s[T1], t2 = s[T2];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double vi = v[i];
    if (rescale) { vi *= inv; v[i] = vi; }
    const double wi = w[i];
    x[i] += t1 * wi;
    const double wn = vi + t2 * wi;
    w[i] = wn;
    acc += wn * wn;
  }
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partw[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void lsqr_sumsq_kernel(int64_t n, const double* __restrict__ w, double* __restrict__ part) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += w[i] * w[i];
  const double tot = block_sum_256(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
}  // namespace

size_t lsqr_work_doubles() { return (size_t)gsi::lsqrst::COUNT + 3 * LSQR_PARTS; }

// work = [state COUNT | part_u | part_v | part_w]
void lsqr_begin(hipStream_t st, int64_t n, const double* w, double* work) {
  hipLaunchKernelGGL(lsqr_sumsq_kernel, dim3(LSQR_PARTS), dim3(256), 0, st, n, w, work + COUNT + 2 * LSQR_PARTS);
}
void lsqr_step_u(hipStream_t st, int64_t m, const double* t, double* u, double* work) {
  double* s = work;
  double* pu = work + COUNT;
  hipLaunchKernelGGL(lsqr_u_kernel, dim3(LSQR_PARTS), dim3(256), 0, st, m, t, u, s, pu);
  hipLaunchKernelGGL(lsqr_after_u_kernel, dim3(1), dim3(64), 0, st, s, pu);
  hipLaunchKernelGGL(lsqr_scale_u_kernel, dim3(LSQR_PARTS), dim3(256), 0, st, m, u, s);
}
void lsqr_step_v(hipStream_t st, int64_t n, const double* t, double* v, double* w, double* x, double* work) {
  double* s = work;
  double* pv = work + COUNT + LSQR_PARTS;
  double* pw = work + COUNT + 2 * LSQR_PARTS;
  hipLaunchKernelGGL(lsqr_v_kernel, dim3(LSQR_PARTS), dim3(256), 0, st, n, t, v, s, pv);
  hipLaunchKernelGGL(lsqr_after_v_kernel, dim3(1), dim3(64), 0, st, s, pv, pw);
  hipLaunchKernelGGL(lsqr_update_kernel, dim3(LSQR_PARTS), dim3(256), 0, st, n, v, w, x, s, pw);
}

}}  // namespace gsi::hipk
