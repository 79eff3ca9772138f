// misc.hip -- the HBM-bound odds and ends of the path: LowRankCovMatrix mean removal
// (lowrank.jl:17-27), Z = V*sqrt(S) column scaling (RandMatFact.jl:87-88, folded into the
// l x l factor), the Cholesky / triangular solve of eig_nystrom (RandMatFact.jl:95-96),
// vector helpers of the adaptive range finder (RandMatFact.jl:15-48), the perturbation
// batch of pcgadirect (direct.jl:39-45), plus benchmark input generators (a counter-based
// Gaussian fill and a synthetic grid covariance).
#include "hip_common.hpp"

namespace gsi { namespace hipk {

static inline int grid_for(int64_t total, int cap = 4096) {
  int64_t g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- LowRankCovMatrix constructor: subtract the per-row mean over the N samples ------------
// thread = one row; both sweeps are coalesced column segments of the n x N sample matrix
__global__ void center_rows_kernel(double* __restrict__ S, int64_t n, int64_t N, int64_t ld) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x) {
    double mean = 0.0;
    for (int64_t i = 0; i < N; ++i) mean += S[r + i * ld];   // lowrank.jl:19-23 (sample order)
    mean /= (double)N;                                       // :24
    for (int64_t i = 0; i < N; ++i) S[r + i * ld] -= mean;    // :25-27
  }
}
void center_rows(hipStream_t st, double* S, int64_t n, int64_t N, int64_t ld) {
  hipLaunchKernelGGL(center_rows_kernel, dim3(grid_for(n)), dim3(256), 0, st, S, n, N, ld);
}

// ---- U <- U * diag(sqrt(S_i) for i < K, 0 otherwise)   (RandMatFact.jl:87-88) ---------------
__global__ void scale_cols_sqrt_kernel(double* __restrict__ U, int64_t l, const double* __restrict__ S,
                                       int64_t K) {
  const int64_t total = l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = e / l;
    U[e] = (c < K) ? U[e] * sqrt(S[c]) : 0.0;
  }
}
void scale_cols_sqrt(hipStream_t st, double* U, int64_t l, const double* S, int64_t K) {
  hipLaunchKernelGGL(scale_cols_sqrt_kernel, dim3(grid_for(l * l)), dim3(256), 0, st, U, l, S, K);
}

// ---- counter-based N(0,1): Philox4x32-10 + Box-Muller (benchmark inputs; not Julia's stream) ---
__device__ inline void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                    uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__global__ void randn_kernel(double* __restrict__ p, uint64_t count, uint64_t seed) {
  const uint64_t npairs = (count + 1) / 2;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs;
       i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t c0 = (uint32_t)i, c1 = (uint32_t)(i >> 32), c2 = 0x5eed5eedu, c3 = 0;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    const uint64_t a = ((uint64_t)c0 << 32) | c1;
    const uint64_t b = ((uint64_t)c2 << 32) | c3;
    const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0, 1]
    const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);          // [0, 1)
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    p[2 * i] = rad * cs;
    if (2 * i + 1 < count) p[2 * i + 1] = rad * sn;
  }
}
void randn_fill(hipStream_t st, double* p, size_t count, uint64_t seed) {
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for((int64_t)((count + 1) / 2), 8192)), dim3(256), 0, st, p,
                     (uint64_t)count, seed);
}

// ---- synthetic sample fields for a LowRankCovMatrix: S(i, j) = g(i, j) (j+1)^-decay with
// g iid N(0,1) addressed by the GLOBAL (row, sample) index, so every row shard of every rank layout draws the
// same matrix.  Benchmark / test input only (the reference's fields come from FFTRF on the host).
__global__ void lowrank_samples_kernel(double* __restrict__ S, int64_t ld, int64_t nloc, int64_t N, int64_t row0,
                                       uint64_t seed, double decay) {
  const int64_t j = blockIdx.y;
  for (int64_t jj = j; jj < N; jj += gridDim.y) {
    const double scale = pow((double)(jj + 1), -decay);
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nloc; r += (int64_t)gridDim.x * blockDim.x) {
      const uint64_t g = (uint64_t)(row0 + r);
      const uint64_t pair = g >> 1;
      uint32_t c0 = (uint32_t)pair, c1 = (uint32_t)(pair >> 32), c2 = (uint32_t)jj, c3 = 0x10c0ffeeu;
      uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
      for (int rd = 0; rd < 10; ++rd) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
      }
      const uint64_t a = ((uint64_t)c0 << 32) | c1;
      const uint64_t b = ((uint64_t)c2 << 32) | c3;
      const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740992.0);
      const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
      const double rad = sqrt(-2.0 * log(u1));
      double sn, cs;
      sincospi(2.0 * u2, &sn, &cs);
      S[r + jj * ld] = scale * rad * ((g & 1) ? sn : cs);
    }
  }
}
void fill_lowrank_samples(hipStream_t st, double* S, int64_t ld, int64_t nloc, int64_t N, int64_t row0, uint64_t seed,
                          double decay) {
  if (nloc <= 0 || N <= 0) return;
  int gx = (int)((nloc + 255) / 256);
  if (gx > 256) gx = 256;
  const int gy = (int)((N < 4096) ? N : 4096);
  hipLaunchKernelGGL(lowrank_samples_kernel, dim3(gx, gy), dim3(256), 0, st, S, ld, nloc, N, row0, seed, decay);
}

// ---- synthetic covariance of an nx x ny unit grid ------------------------------
// point i = (i / ny, i % ny).  kind 0: exp(-d^2/(2 ell^2)); kind 1: exp(-d/ell).
__global__ void gridcov_kernel(double* __restrict__ A, int64_t lda, int64_t nx, int64_t ny, double ell,
                               int kind, int64_t row0, int64_t mloc) {
  const int64_t n = nx * ny;
  const int64_t col = blockIdx.y;
  const double cx = (double)(col / ny), cy = (double)(col % ny);
  const double inv2 = 1.0 / (2.0 * ell * ell), inv1 = 1.0 / ell;
  for (int64_t col_it = col; col_it < n; col_it += gridDim.y) {
    const double qx = (double)(col_it / ny), qy = (double)(col_it % ny);
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < mloc;
         r += (int64_t)gridDim.x * blockDim.x) {
      const int64_t gi = row0 + r;
      const double dx = (double)(gi / ny) - qx, dy = (double)(gi % ny) - qy;
      const double d2 = dx * dx + dy * dy;
      A[r + col_it * lda] = (kind == 0) ? exp(-d2 * inv2) : exp(-sqrt(d2) * inv1);
    }
  }
  (void)cx; (void)cy;
}
void fill_gridcov(hipStream_t st, double* A, int64_t lda, int64_t nx, int64_t ny, double ell, int kind,
                  int64_t row0, int64_t mloc) {
  const int64_t n = nx * ny;
  int gx = (int)((mloc + 255) / 256);
  if (gx > 64) gx = 64;
  int gy = (int)((n < 16384) ? n : 16384);
  hipLaunchKernelGGL(gridcov_kernel, dim3(gx, gy), dim3(256), 0, st, A, lda, nx, ny, ell, kind, row0, mloc);
}

// ---- squared column norms: one workgroup per column ----------------------------------------
__global__ __launch_bounds__(256) void colnorms_sq_kernel(const double* __restrict__ Y, int64_t m,
                                                          int64_t ld, double* __restrict__ out) {
  __shared__ double s[4];
  const double* col = Y + (int64_t)blockIdx.x * ld;
  double v = 0.0;
  for (int64_t i = threadIdx.x; i < m; i += 256) { const double x = col[i]; v += x * x; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
void colnorms_sq(hipStream_t st, const double* Y, int64_t m, int64_t c, int64_t ld, double* out_dev) {
  if (c <= 0) return;
  hipLaunchKernelGGL(colnorms_sq_kernel, dim3((unsigned)c), dim3(256), 0, st, Y, m, ld, out_dev);
}

// ---- dot product (single workgroup, fixed order) ---------------------------------------------
__global__ __launch_bounds__(1024) void dot_kernel(int64_t n, const double* __restrict__ x,
                                                   const double* __restrict__ y, double* __restrict__ out) {
  __shared__ double s[16];
  double v = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) v += x[i] * y[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += s[w];
    *out = t;
  }
}
// long vectors: 256 workgroups leave partial sums, one more launch adds them in index order (deterministic)
__global__ __launch_bounds__(256) void dot_partial_kernel(int64_t n, const double* __restrict__ x,
                                                          const double* __restrict__ y, double* __restrict__ part) {
  __shared__ double s[4];
  double v = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v += x[i] * y[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ __launch_bounds__(64) void dot_final_kernel(int nparts, const double* __restrict__ part, double* __restrict__ out) {
  double v = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) v += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (threadIdx.x == 0) *out = v;
}
void dot_dev(hipStream_t st, int64_t n, const double* x, const double* y, double* out_dev) {
  if (n < 65536) {
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, st, n, x, y, out_dev);
    return;
  }
  double* part = out_dev + 8;     // the backend's scalar scratch holds 64 + 256 doubles
  hipLaunchKernelGGL(dot_partial_kernel, dim3(256), dim3(256), 0, st, n, x, y, part);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(64), 0, st, 256, part, out_dev);
}

__global__ void axpy_kernel(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}
void axpy(hipStream_t st, int64_t n, double a, const double* x, double* y) {
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, a, x, y);
}
__global__ void scal_copy_kernel(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a * x[i];
}
void scal_copy(hipStream_t st, int64_t n, double a, const double* x, double* y) {
  hipLaunchKernelGGL(scal_copy_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, a, x, y);
}

// ---- BLAS-2 for the adaptive range finder (RandMatFact.jl:30-45: gemv!, axpy!, dot) and the nobs-sized saddle-point
//      products of pcgalsqr: HBM-bound, no matrix core involved, no host round trip per scalar --------------------
// y (m) = alpha A x + beta y, A m x k column-major: thread = one row (coalesced column segments), x in LDS chunks
__global__ __launch_bounds__(256) void gemv_n_kernel(int64_t m, int64_t k, double alpha, const double* __restrict__ A,
                                                     int64_t lda, const double* __restrict__ x, double beta,
                                                     double* __restrict__ y) {
  __shared__ double xs[1024];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int64_t k0 = 0; k0 < k; k0 += 1024) {
    const int kc = (int)((k - k0 < 1024) ? (k - k0) : 1024);
    __syncthreads();
    for (int j = threadIdx.x; j < kc; j += 256) xs[j] = x[k0 + j];
    __syncthreads();
    if (i < m) {
      const double* row = A + i + k0 * lda;
      int j = 0;
      for (; j + 4 <= kc; j += 4) {
        a0 += row[(int64_t)(j + 0) * lda] * xs[j + 0];
        a1 += row[(int64_t)(j + 1) * lda] * xs[j + 1];
        a2 += row[(int64_t)(j + 2) * lda] * xs[j + 2];
        a3 += row[(int64_t)(j + 3) * lda] * xs[j + 3];
      }
      for (; j < kc; ++j) a0 += row[(int64_t)j * lda] * xs[j];
    }
  }
  if (i < m) {
    const double v = alpha * ((a0 + a1) + (a2 + a3));
    y[i] = (beta != 0.0) ? v + beta * y[i] : v;
  }
}
void gemv_n(hipStream_t st, int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double beta,
            double* y) {
  if (m <= 0) return;
  hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, k, alpha, A, lda, x, beta, y);
}
// y (k) = alpha A' x, A m x k: workgroup (column j, slab s) leaves a partial sum, a second launch adds the slabs in
// index order (deterministic).  part: k * GEMV_T_SLABS doubles.
constexpr int GEMV_T_SLABS = 32;
__global__ __launch_bounds__(256) void gemv_t_partial_kernel(int64_t m, const double* __restrict__ A, int64_t lda,
                                                             const double* __restrict__ x, double* __restrict__ part) {
  __shared__ double s[4];
  const int64_t j = blockIdx.x;
  const int64_t per = (m + GEMV_T_SLABS - 1) / GEMV_T_SLABS;
  const int64_t i0 = (int64_t)blockIdx.y * per, i1 = (i0 + per < m) ? i0 + per : m;
  const double* col = A + j * lda;
  double v = 0.0;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) v += col[i] * x[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[j * GEMV_T_SLABS + blockIdx.y] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ void gemv_t_final_kernel(int64_t k, double alpha, const double* __restrict__ part, double* __restrict__ y) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  double v = 0.0;
  for (int s = 0; s < GEMV_T_SLABS; ++s) v += part[j * GEMV_T_SLABS + s];
  y[j] = alpha * v;
}
size_t gemv_t_workspace_doubles(int64_t k) { return (size_t)k * GEMV_T_SLABS; }
void gemv_t(hipStream_t st, int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double* y,
            double* part) {
  if (k <= 0) return;
  hipLaunchKernelGGL(gemv_t_partial_kernel, dim3((unsigned)k, GEMV_T_SLABS), dim3(256), 0, st, m, A, lda, x, part);
  hipLaunchKernelGGL(gemv_t_final_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, k, alpha, part, y);
}
// Y[:, c] -= dot(q, Y[:, c]) q for c < ncols (RandMatFact.jl:42-45): the dots stay on the device.  part: ncols * GEMV_T_SLABS.
__global__ __launch_bounds__(256) void project_apply_kernel(int64_t m, int64_t ncols, const double* __restrict__ q,
                                                            const double* __restrict__ part, double* __restrict__ Y,
                                                            int64_t ld) {
  __shared__ double d[64];
  if (threadIdx.x < ncols) {
    double v = 0.0;
    for (int s = 0; s < GEMV_T_SLABS; ++s) v += part[threadIdx.x * GEMV_T_SLABS + s];
    d[threadIdx.x] = v;
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    const double qi = q[i];
    for (int64_t c = 0; c < ncols; ++c) Y[i + c * ld] -= d[c] * qi;
  }
}
void project_out(hipStream_t st, int64_t m, int64_t ncols, const double* q, double* Y, int64_t ld, double* part) {
  if (ncols <= 0) return;
  hipLaunchKernelGGL(gemv_t_partial_kernel, dim3((unsigned)ncols, GEMV_T_SLABS), dim3(256), 0, st, m, Y, ld, q, part);
  hipLaunchKernelGGL(project_apply_kernel, dim3(grid_for(m, 1024)), dim3(256), 0, st, m, ncols, q, part, Y, ld);
}

__global__ void scal_kernel(int64_t n, double a, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= a;
}
void scal(hipStream_t st, int64_t n, double a, double* x) {
  hipLaunchKernelGGL(scal_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, a, x);
}
__global__ void diag_mul_add_kernel(int64_t n, const double* __restrict__ d, const double* __restrict__ x,
                                    double* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] += d[i] * x[i];
}
void diag_mul_add(hipStream_t st, int64_t n, const double* d, const double* x, double* y) {
  hipLaunchKernelGGL(diag_mul_add_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, d, x, y);
}

// ---- fp32-stored xi-basis (mixed precision: fp32 storage, fp64 arithmetic) ------------------------------------
__global__ void f64_to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (float)src[i];
}
void f64_to_f32(hipStream_t st, const double* src, float* dst, size_t count) {
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3(grid_for((int64_t)count, 8192)), dim3(256), 0, st, src, dst, count);
}
__global__ void pcga_params_f32_kernel(const float* __restrict__ Z, int64_t n, int64_t K, const double* __restrict__ s,
                                       const double* __restrict__ X, double delta, double* __restrict__ out) {
  const int64_t col = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double si = s[i];
    double v;
    if (col < K) v = si + delta * (double)Z[i + col * n];
    else if (col == K) v = si + delta * X[i];
    else if (col == K + 1) v = si + delta * si;
    else v = si;
    out[i + col * n] = v;
  }
}
void pcga_params_f32(hipStream_t st, const float* Z, int64_t n, int64_t K, const double* s, const double* X,
                     double delta, double* out) {
  hipLaunchKernelGGL(pcga_params_f32_kernel, dim3(grid_for(n, 256), (unsigned)(K + 3)), dim3(256), 0, st, Z, n, K, s, X,
                     delta, out);
}
// y = beta X + Z w: thread = one row, the K weights in LDS; every global access is a coalesced column segment of the
// fp32 basis (HBM-bound: 4 n K bytes)
__global__ __launch_bounds__(256) void basis_gemv_f32_kernel(const float* __restrict__ Z, int64_t n, int64_t K,
                                                             const double* __restrict__ w, double beta,
                                                             const double* __restrict__ X, double* __restrict__ y) {
  extern __shared__ double ws[];
  for (int64_t k = threadIdx.x; k < K; k += 256) ws[k] = w[k];
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double acc0 = beta * X[i], acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int64_t k = 0;
    for (; k + 4 <= K; k += 4) {
      acc0 += (double)Z[i + (k + 0) * n] * ws[k + 0];
      acc1 += (double)Z[i + (k + 1) * n] * ws[k + 1];
      acc2 += (double)Z[i + (k + 2) * n] * ws[k + 2];
      acc3 += (double)Z[i + (k + 3) * n] * ws[k + 3];
    }
    for (; k < K; ++k) acc0 += (double)Z[i + k * n] * ws[k];
    y[i] = (acc0 + acc1) + (acc2 + acc3);
  }
}
void basis_gemv_f32(hipStream_t st, const float* Z, int64_t n, int64_t K, const double* w, double beta, const double* X,
                    double* y) {
  hipLaunchKernelGGL(basis_gemv_f32_kernel, dim3(grid_for(n, 4096)), dim3(256), (size_t)K * sizeof(double), st, Z, n, K, w,
                     beta, X, y);
}

// ---- R (l x l, ld l) <- upper triangle of the top of Y --------------------------------------
__global__ void extract_upper_kernel(const double* __restrict__ Y, int64_t ld, int64_t l,
                                     double* __restrict__ R) {
  const int64_t total = l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % l, c = e / l;
    R[e] = (r <= c) ? Y[r + c * ld] : 0.0;
  }
}
void extract_upper(hipStream_t st, const double* Y, int64_t ld, int64_t l, double* R) {
  hipLaunchKernelGGL(extract_upper_kernel, dim3(grid_for(l * l, 1024)), dim3(256), 0, st, Y, ld, l, R);
}

// ---- cholesky(Hermitian(B)).U for the small j x j Nystrom core (RandMatFact.jl:95) -----------
// one workgroup, upper triangle read, left-looking column Cholesky; *info = k+1 if not PD
__global__ __launch_bounds__(256) void chol_upper_kernel(double* __restrict__ B, int j, int32_t* info) {
  const int tid = threadIdx.x;
  __shared__ double s_d;
  for (int k = 0; k < j; ++k) {
    // U[k][k] = sqrt(B[k][k] - sum_{p<k} U[p][k]^2)
    if (tid == 0) {
      double d = B[k + (int64_t)k * j];
      for (int p = 0; p < k; ++p) { const double u = B[p + (int64_t)k * j]; d -= u * u; }
      if (!(d > 0.0)) { if (*info == 0) *info = k + 1; d = 1.0; }
      d = sqrt(d);
      B[k + (int64_t)k * j] = d;
      s_d = d;
    }
    __syncthreads();
    const double dk = s_d;
    // U[k][c] = (B[k][c] - sum_{p<k} U[p][k] U[p][c]) / U[k][k],  c > k
    for (int c = k + 1 + tid; c < j; c += 256) {
      double v = B[k + (int64_t)c * j];
      for (int p = 0; p < k; ++p) v -= B[p + (int64_t)k * j] * B[p + (int64_t)c * j];
      B[k + (int64_t)c * j] = v / dk;
    }
    __syncthreads();
  }
  // zero the strict lower triangle so B is exactly U
  for (int e = tid; e < j * j; e += 256) {
    const int r = e % j, c = e / j;
    if (r > c) B[e] = 0.0;
  }
}
void chol_upper(hipStream_t st, double* B, int64_t j, int32_t* info) {
  hipLaunchKernelGGL(chol_upper_kernel, dim3(1), dim3(256), 0, st, B, (int)j, info);
}

// ---- F <- F * inv(C), C upper triangular j x j (RandMatFact.jl:96); thread = one row of F ------
__global__ void trsm_right_upper_kernel(double* __restrict__ F, int64_t m, int j, int64_t ldf,
                                        const double* __restrict__ C) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < m;
       r += (int64_t)gridDim.x * blockDim.x) {
    // x C = f  ->  x[c] = (f[c] - sum_{p<c} x[p] C[p][c]) / C[c][c]
    for (int c = 0; c < j; ++c) {
      double v = F[r + c * ldf];
      for (int p = 0; p < c; ++p) v -= F[r + p * ldf] * C[p + (int64_t)c * j];
      F[r + c * ldf] = v / C[c + (int64_t)c * j];
    }
  }
}
void trsm_right_upper(hipStream_t st, double* F, int64_t m, int64_t j, int64_t ldf, const double* C) {
  hipLaunchKernelGGL(trsm_right_upper_kernel, dim3(grid_for(m)), dim3(256), 0, st, F, m, (int)j, ldf, C);
}

// ---- paramstorun of pcgadirect / pcgalsqr (direct.jl:39-45, lsqr.jl:37-43) --------------------
// out[:, i] = s + delta*Z[:, i] (i < K); out[:, K] = s + delta*X; out[:, K+1] = s + delta*s; out[:, K+2] = s
__global__ void pcga_params_kernel(const double* __restrict__ Z, int64_t n, int64_t K,
                                   const double* __restrict__ s, const double* __restrict__ X, double delta,
                                   double* __restrict__ out) {
  const int64_t col = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double si = s[i];
    double v;
    if (col < K) v = si + delta * Z[i + col * n];
    else if (col == K) v = si + delta * X[i];
    else if (col == K + 1) v = si + delta * si;
    else v = si;
    out[i + col * n] = v;
  }
}
void pcga_params(hipStream_t st, const double* Z, int64_t n, int64_t K, const double* s, const double* X,
                 double delta, double* out) {
  int gx = grid_for(n, 256);
  hipLaunchKernelGGL(pcga_params_kernel, dim3(gx, (unsigned)(K + 3)), dim3(256), 0, st, Z, n, K, s, X, delta,
                     out);
}

}}  // namespace gsi::hipk
