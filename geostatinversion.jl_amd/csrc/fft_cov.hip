// fft_cov.hip -- matrix-free stationary covariance on a structured grid by circulant embedding
// (BASELINE.json configs[2]: "FFTRF power-law covariance ... matrix-free A*Omega via FFT").
//
// The reference only SAMPLES such fields (FFTRF.jl:83-100: ifft of sqrt(S_f) .* exp(2 pi i phi) on the doubled
// periodic grid, first octant kept); the covariance of those samples is, up to the per-sample normalisation,
//     A = R F^-1 diag(lambda) F R',   lambda(k) = |k|^beta  (S_f = |k|^(beta/2) squared, lambda(0) = 0),
// R' = zero padding of the N_1 x .. x N_d grid into the periodic embedding grid.  This file applies A to the
// columns of a panel without ever forming it -- the role getxis(::Function) + LowRankCovMatrix play in the
// reference, but exact instead of a sample estimate.
//
// gfx950 mapping.  lambda is real and even, so F^-1 diag(lambda) F is a REAL operator: two real columns ride in
// one complex transform (x_a + i x_b -> A x_a + i A x_b) with no untangling pass.  The embedding is the next
// power of two >= 2 N per axis.  A d-dimensional transform is d passes of batched 1-D radix-2 transforms done
// entirely in LDS (a line of <= 4096 complex doubles is 64 KB): axis 0 lines are contiguous; for the other axes
// a workgroup takes a tile of T neighbouring lines so that every global access is T*16 contiguous bytes.  Each
// pass reads and writes the work array once: HBM-bound (2 d + 1 passes forward/scale/back per column pair,
// 16 B per embedded point per pass).  Twiddles sit in LDS; the spectrum is a precomputed real table with the
// 1/M of the inverse transform and the unit-diagonal normalisation folded in.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <vector>
#include "hip_common.hpp"

namespace gsi { namespace hipk {

constexpr int FFT_THREADS = 256;


__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// W[b][embedded(i)] = X[i, 2b] + i X[i, 2b+1] inside the N-box, 0 outside.  One thread per embedded point.
__global__ __launch_bounds__(256) void fft_pack_kernel(double2* __restrict__ W, int64_t Mtot, int nd, int64_t N0, int64_t N1,
                                                       int64_t N2, int64_t M0, int64_t M1, const double* __restrict__ X,
                                                       int64_t ldx, int64_t col0, int64_t l) {
  const int64_t b = blockIdx.y;
  const int64_t ca = col0 + 2 * b, cb = ca + 1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256) {
    const int64_t i0 = e % M0, r = e / M0, i1 = r % M1, i2 = r / M1;
    double2 v = make_double2(0.0, 0.0);
    if (i0 < N0 && i1 < N1 && i2 < N2) {
      const int64_t i = i0 + N0 * (i1 + N1 * i2);
      v.x = X[i + ca * ldx];
      if (cb < l) v.y = X[i + cb * ldx];
    }
    W[b * Mtot + e] = v;
  }
  (void)nd;
}

__global__ __launch_bounds__(256) void fft_unpack_kernel(const double2* __restrict__ W, int64_t Mtot, int64_t N0, int64_t N1,
                                                         int64_t N2, int64_t M0, int64_t M1, double* __restrict__ Y,
                                                         int64_t ldy, int64_t col0, int64_t l) {
  const int64_t b = blockIdx.y;
  const int64_t ca = col0 + 2 * b, cb = ca + 1;
  const int64_t n = N0 * N1 * N2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t i0 = i % N0, r = i / N0, i1 = r % N1, i2 = r / N1;
    const double2 v = W[b * Mtot + i0 + M0 * (i1 + M1 * i2)];
    Y[i + ca * ldy] = v.x;
    if (cb < l) Y[i + cb * ldy] = v.y;
  }
}

// W[b][e] *= lam[e]   (lam real: spectrum / (M * mean(spectrum)))
__global__ __launch_bounds__(256) void fft_scale_kernel(double2* __restrict__ W, int64_t Mtot, const double* __restrict__ lam) {
  double2* Wb = W + (int64_t)blockIdx.y * Mtot;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256) {
    const double s = lam[e];
    double2 v = Wb[e];
    v.x *= s; v.y *= s;
    Wb[e] = v;
  }
}

// Batched in-place 1-D transforms of length Ma (power of two) along one axis.
//   element k of line (inner, outer): W[inner + estride * k + ostride * outer],  inner < estride, outer < nouter
// A workgroup owns T lines: AXIS0 (estride == 1): T consecutive `outer`; otherwise T consecutive `inner`.
// LDS: T lines of Ma complex + Ma/2 twiddles.  sign = -1 forward, +1 inverse (unnormalised).
template <bool AXIS0>
__global__ __launch_bounds__(FFT_THREADS) void fft_axis_kernel(double2* __restrict__ W, int64_t Mtot, int Ma, int log2Ma,
                                                               int64_t estride, int64_t ostride, int64_t nouter, int T,
                                                               double sign) {
  extern __shared__ double2 fsm[];
  double2* tw = fsm;                 // [Ma/2]
  double2* buf = fsm + Ma / 2;       // [T][Ma + 1]   (+1: lines start in different banks)
  const int tid = threadIdx.x;
  const int lstride = Ma + 1;
  double2* Wb = W + (int64_t)blockIdx.y * Mtot;
  for (int k = tid; k < Ma / 2; k += FFT_THREADS) {
    double s, c;
    sincospi(2.0 * (double)k / (double)Ma, &s, &c);
    tw[k] = make_double2(c, sign * s);
  }
  // tile -> (inner0, outer0)
  int64_t inner0, outer0;
  int nlines;
  if (AXIS0) {
    outer0 = (int64_t)blockIdx.x * T; inner0 = 0;
    nlines = (int)((nouter - outer0 < T) ? (nouter - outer0) : T);
  } else {
    const int64_t tiles_per_outer = (estride + T - 1) / T;
    outer0 = blockIdx.x / tiles_per_outer;
    inner0 = (blockIdx.x % tiles_per_outer) * T;
    nlines = (int)((estride - inner0 < T) ? (estride - inner0) : T);
  }
  const unsigned shift = 32u - (unsigned)log2Ma;
  // load with bit reversal of k
  for (int e = tid; e < nlines * Ma; e += FFT_THREADS) {
    int j, k;
    if (AXIS0) { k = e % Ma; j = e / Ma; } else { j = e % nlines; k = e / nlines; }
    const int64_t g = AXIS0 ? ((int64_t)k + ostride * (outer0 + j)) : (inner0 + j + estride * (int64_t)k + ostride * outer0);
    const unsigned kr = (log2Ma == 0) ? 0u : (__brev((unsigned)k) >> shift);
    buf[j * lstride + kr] = Wb[g];
  }
  __syncthreads();
  const int nbf = nlines * (Ma / 2);
  for (int st = 0; st < log2Ma; ++st) {
    const int half = 1 << st;
    const int tws = Ma >> (st + 1);
    for (int e = tid; e < nbf; e += FFT_THREADS) {
      const int j = e / (Ma / 2), p = e % (Ma / 2);
      const int q = p & (half - 1);
      const int i0 = ((p >> st) << (st + 1)) + q;
      double2* x = buf + j * lstride;
      const double2 a = x[i0], b = cmul(tw[q * tws], x[i0 + half]);
      x[i0] = make_double2(a.x + b.x, a.y + b.y);
      x[i0 + half] = make_double2(a.x - b.x, a.y - b.y);
    }
    __syncthreads();
  }
  for (int e = tid; e < nlines * Ma; e += FFT_THREADS) {
    int j, k;
    if (AXIS0) { k = e % Ma; j = e / Ma; } else { j = e % nlines; k = e / nlines; }
    const int64_t g = AXIS0 ? ((int64_t)k + ostride * (outer0 + j)) : (inner0 + j + estride * (int64_t)k + ostride * outer0);
    Wb[g] = buf[j * lstride + k];
  }
}

// lam[e] = (sum_i (f_i / M_i)^2)^(beta/2), f_i = min(k_i, M_i - k_i); lam[0] = 0
__global__ __launch_bounds__(256) void fft_spectrum_kernel(double* __restrict__ lam, int64_t Mtot, int64_t M0, int64_t M1,
                                                           int64_t M2, double beta) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256) {
    const int64_t k0 = e % M0, r = e / M0, k1 = r % M1, k2 = r / M1;
    const double f0 = (double)((k0 <= M0 - k0) ? k0 : M0 - k0) / (double)M0;
    const double f1 = (double)((k1 <= M1 - k1) ? k1 : M1 - k1) / (double)M1;
    const double f2 = (double)((k2 <= M2 - k2) ? k2 : M2 - k2) / (double)M2;
    const double k2sum = f0 * f0 + f1 * f1 + f2 * f2;
    lam[e] = (k2sum > 0.0) ? pow(k2sum, 0.5 * beta) : 0.0;
  }
}

// partial sums of lam -> part[blockIdx.x]; then lam *= 1 / sum (one more tiny kernel on the host side order)
__global__ __launch_bounds__(256) void fft_sum_kernel(const double* __restrict__ lam, int64_t Mtot, double* __restrict__ part) {
  __shared__ double s[256];
  double acc = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256) acc += lam[e];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(256) void fft_normalise_kernel(double* __restrict__ lam, int64_t Mtot, const double* __restrict__ part,
                                                            int nparts) {
  double tot = 0.0;
  for (int i = 0; i < nparts; ++i) tot += part[i];       // fixed order: deterministic
  const double inv = (tot > 0.0) ? 1.0 / tot : 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256) lam[e] *= inv;
}

static inline int grid_for(int64_t total, int cap) {
  int64_t g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

static int ilog2(int64_t v) { int r = 0; while (((int64_t)1 << r) < v) ++r; return r; }

int64_t fft_embed_size(int64_t N) { int64_t m = 1; while (m < 2 * N) m <<= 1; return (N == 1) ? 1 : m; }

void fft_spectrum(hipStream_t st, double* lam, double* part64, const int64_t M[3], double beta) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  hipLaunchKernelGGL(fft_spectrum_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, M[0], M[1], M[2], beta);
  hipLaunchKernelGGL(fft_sum_kernel, dim3(64), dim3(256), 0, st, lam, Mtot, part64);
  hipLaunchKernelGGL(fft_normalise_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, part64, 64);
}

static void fft_axis(hipStream_t st, double2* W, int64_t Mtot, int nb, const int64_t M[3], int axis, double sign) {
  const int Ma = (int)M[axis];
  if (Ma == 1) return;
  const int lg = ilog2(Ma);
  int64_t estride = 1;
  for (int a = 0; a < axis; ++a) estride *= M[a];
  const int64_t ostride = estride * Ma;
  const int64_t nouter = Mtot / ostride;
  // lines per workgroup: as many as fit ~96 KB of LDS, at most 16 (256-byte segments for the strided axes)
  int T = (int)((96 * 1024) / ((size_t)(Ma + 1) * sizeof(double2)));
  if (T > 16) T = 16;
  if (T < 1) T = 1;
  const size_t shmem = ((size_t)Ma / 2 + (size_t)T * (Ma + 1)) * sizeof(double2);
  if (axis == 0) {
    static bool attr0 = false;
    if (!attr0) { (void)hipFuncSetAttribute((const void*)fft_axis_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr0 = true; }
    const int64_t ntiles = (nouter + T - 1) / T;
    hipLaunchKernelGGL(fft_axis_kernel<true>, dim3((unsigned)ntiles, (unsigned)nb), dim3(FFT_THREADS), shmem, st, W, Mtot, Ma, lg,
                       estride, ostride, nouter, T, sign);
  } else {
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute((const void*)fft_axis_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); attr1 = true; }
    const int64_t tiles_per_outer = (estride + T - 1) / T;
    hipLaunchKernelGGL(fft_axis_kernel<false>, dim3((unsigned)(tiles_per_outer * nouter), (unsigned)nb), dim3(FFT_THREADS), shmem,
                       st, W, Mtot, Ma, lg, estride, ostride, nouter, T, sign);
  }
}

// Y (n x l, ld ldy) = A X for the embedded-circulant covariance; W holds nb_max * Mtot complex doubles.
void fft_cov_apply(hipStream_t st, const int64_t N[3], const int64_t M[3], const double* lam, double2* W, int nb_max,
                   int64_t l, const double* X, int64_t ldx, double* Y, int64_t ldy) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  const int64_t n = N[0] * N[1] * N[2];
  const int64_t npairs = (l + 1) / 2;
  for (int64_t p0 = 0; p0 < npairs; p0 += nb_max) {
    const int nb = (int)((npairs - p0 < nb_max) ? (npairs - p0) : nb_max);
    const int64_t col0 = 2 * p0;
    hipLaunchKernelGGL(fft_pack_kernel, dim3(grid_for(Mtot, 2048), nb), dim3(256), 0, st, W, Mtot, 3, N[0], N[1], N[2], M[0],
                       M[1], X, ldx, col0, l);
    for (int a = 0; a < 3; ++a) fft_axis(st, W, Mtot, nb, M, a, -1.0);
    hipLaunchKernelGGL(fft_scale_kernel, dim3(grid_for(Mtot, 2048), nb), dim3(256), 0, st, W, Mtot, lam);
    for (int a = 2; a >= 0; --a) fft_axis(st, W, Mtot, nb, M, a, +1.0);
    hipLaunchKernelGGL(fft_unpack_kernel, dim3(grid_for(n, 2048), nb), dim3(256), 0, st, W, Mtot, N[0], N[1], N[2], M[0], M[1], Y,
                       ldy, col0, l);
  }
}

}}  // namespace gsi::hipk
