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
// power of two >= 2 N per axis.  A d-dimensional transform is d passes of batched 1-D transforms done entirely in
// LDS (a line of <= 4096 complex doubles is 64 KB): axis 0 lines are contiguous; for the other axes a workgroup
// takes a tile of T neighbouring lines so that every global access is T*16 contiguous bytes.
// HBM-bound, so the passes move as little as possible: the zero padding is never stored or read (forward passes
// read only the N_a valid entries of a line and skip lines that lie in the padding of a later axis; inverse passes
// write only the N_a entries that survive the restriction), the first pass reads X and the last writes Y directly,
// and the spectrum (with the 1/M of the inverse transform and the unit-diagonal normalisation folded in) is
// applied as the last forward pass stores.  2-D, M = 2N: 20 n complex moved per column pair instead of 48 n.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include "hip_common.hpp"

namespace gsi { namespace hipk {

__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// One pass: batched in-place 1-D transforms of length Ma (power of two) along one axis of the embedded grid.
//   element k of line (inner, o):  W[inner + estride * k + off(o)],  inner < estride (all earlier axes, full length),
//   o < R1 * R2 enumerates the LATER axes restricted to the original grid (i_b < N_b): off(o) = (o % R1) S1 + (o / R1) S2.
// MODE bits: AXIS0 lines are contiguous (else a workgroup takes T neighbouring `inner`); LOADX the first forward
// pass reads the two real columns of X; SCALE the last forward pass multiplies by the spectrum as it stores; STOREY
// the last inverse pass writes the two real columns of Y.  sign = -1 forward, +1 inverse (unnormalised).
enum { FFT_AXIS0 = 1, FFT_LOADX = 2, FFT_SCALE = 4, FFT_STOREY = 8 };
constexpr int FFT_TW_LEN = 4096;     // longest supported line; the plan stores exp(-2 pi i k / 4096), k < 2048
struct FftPass {
  int Ma, log2Ma, nin, nout, T;
  int64_t estride, R1, S1, R2, S2;
  double sign;
};

template <int MODE>
__global__ __launch_bounds__(1024) void fft_pass_kernel(double2* __restrict__ W, int64_t Mtot, FftPass ps,
                                                       const double* __restrict__ lam, const double* __restrict__ X,
                                                       int64_t ldx, double* __restrict__ Y, int64_t ldy, int64_t N0,
                                                       int64_t col0, int64_t l) {
  constexpr bool AXIS0 = (MODE & FFT_AXIS0) != 0;
  extern __shared__ double2 fsm[];
  const int Ma = ps.Ma, log2Ma = ps.log2Ma, T = ps.T;
  double2* tw = fsm;                 // [Ma/2]
  double2* buf = fsm + Ma / 2;       // [T][Ma + 1]   (+1: neighbouring lines start in different banks)
  const int tid = threadIdx.x, nth = blockDim.x;
  const int lstride = Ma + 1;
  double2* Wb = W + (int64_t)blockIdx.y * Mtot;
  const int64_t ca = col0 + 2 * (int64_t)blockIdx.y, cb = ca + 1;
  // twiddles exp(sign 2 pi i k / Ma) from the plan's table for the longest line (FFT_TW_LEN points): an L2 read
  // instead of Ma/2 sincospi evaluations per workgroup
  {
    const double2* twg = reinterpret_cast<const double2*>(lam + Mtot + 64);
    const int tstep = FFT_TW_LEN / Ma;
    for (int k = tid; k < Ma / 2; k += nth) {
      const double2 w = twg[k * tstep];
      tw[k] = make_double2(w.x, -ps.sign * w.y);      // table holds the forward sign
    }
  }
  // tile -> first line; line j of the tile: AXIS0: outer o0 + j, else inner i0 + j of outer o0
  const int64_t nouter = ps.R1 * ps.R2;
  int64_t o0, i0 = 0;
  int nlines;
  if (AXIS0) {
    o0 = (int64_t)blockIdx.x * T;
    nlines = (int)((nouter - o0 < T) ? (nouter - o0) : T);
  } else {
    const int64_t tiles_per_outer = (ps.estride + T - 1) / T;
    o0 = blockIdx.x / tiles_per_outer;
    i0 = (blockIdx.x % tiles_per_outer) * T;
    nlines = (int)((ps.estride - i0 < T) ? (ps.estride - i0) : T);
  }
  auto off = [&](int64_t o) -> int64_t { return (o % ps.R1) * ps.S1 + (o / ps.R1) * ps.S2; };
  const int64_t off0 = AXIS0 ? 0 : off(o0);
  const unsigned shift = 32u - (unsigned)log2Ma;
  if (ps.nin < Ma) {      // the padded part of the lines
    for (int e = tid; e < nlines * lstride; e += nth) buf[e] = make_double2(0.0, 0.0);
    __syncthreads();
  }
  // load k < nin, bit-reversed
  for (int e = tid; e < nlines * ps.nin; e += nth) {
    int j, k;
    if (AXIS0) { k = e % ps.nin; j = e / ps.nin; } else { j = e % nlines; k = e / nlines; }
    double2 v;
    if (MODE & FFT_LOADX) {
      const int64_t o = o0 + j;                          // (i1, i2) of the original grid
      const int64_t i = k + N0 * ((o % ps.R1) + ps.R1 * (o / ps.R1));
      v.x = X[i + ca * ldx];
      v.y = (cb < l) ? X[i + cb * ldx] : 0.0;
    } else {
      const int64_t g = AXIS0 ? ((int64_t)k + off(o0 + j)) : (i0 + j + ps.estride * (int64_t)k + off0);
      v = Wb[g];
    }
    buf[j * lstride + (int)(__brev((unsigned)k) >> shift)] = v;
  }
  __syncthreads();
  // decimation in time on the bit-reversed lines.  Two radix-2 stages at a time (radix 4 in registers: 4 reads,
  // 3 twiddles, 4 writes per 4 points instead of 8 + 4 + 8) -- the pass is as much LDS- as HBM-bound; one radix-2
  // stage first when log2(Ma) is odd.
  int st = 0;
  if (log2Ma & 1) {
    const int nbf = nlines * (Ma / 2);
    for (int e = tid; e < nbf; e += nth) {
      const int j = e / (Ma / 2), p = e % (Ma / 2);
      double2* x = buf + j * lstride + 2 * p;
      const double2 a = x[0], b = x[1];
      x[0] = cadd(a, b);
      x[1] = csub(a, b);
    }
    __syncthreads();
    st = 1;
  }
  const int nq = nlines * (Ma / 4);
  for (; st < log2Ma; st += 2) {
    const int h = 1 << st;
    const int tw1 = Ma >> (st + 1);      // twiddle stride of stage st   (block 2h)
    const int tw2 = Ma >> (st + 2);      //                    stage st+1 (block 4h)
    for (int e = tid; e < nq; e += nth) {
      const int j = e / (Ma / 4), p = e % (Ma / 4);
      const int q = p & (h - 1);
      double2* x = buf + j * lstride + ((p >> st) << (st + 2)) + q;
      const double2 w1 = tw[q * tw1], w2 = tw[q * tw2], w3 = tw[(q + h) * tw2];
      const double2 a0 = x[0], a1 = cmul(w1, x[h]), a2 = x[2 * h], a3 = cmul(w1, x[3 * h]);
      const double2 b0 = cadd(a0, a1), b1 = csub(a0, a1);
      const double2 b2 = cmul(w2, cadd(a2, a3)), b3 = cmul(w3, csub(a2, a3));
      x[0] = cadd(b0, b2);
      x[2 * h] = csub(b0, b2);
      x[h] = cadd(b1, b3);
      x[3 * h] = csub(b1, b3);
    }
    __syncthreads();
  }
  for (int e = tid; e < nlines * ps.nout; e += nth) {
    int j, k;
    if (AXIS0) { k = e % ps.nout; j = e / ps.nout; } else { j = e % nlines; k = e / nlines; }
    double2 v = buf[j * lstride + k];
    if (MODE & FFT_STOREY) {
      const int64_t o = o0 + j;
      const int64_t i = k + N0 * ((o % ps.R1) + ps.R1 * (o / ps.R1));
      Y[i + ca * ldy] = v.x;
      if (cb < l) Y[i + cb * ldy] = v.y;
    } else {
      const int64_t g = AXIS0 ? ((int64_t)k + off(o0 + j)) : (i0 + j + ps.estride * (int64_t)k + off0);
      if (MODE & FFT_SCALE) { const double sc = lam[g]; v.x *= sc; v.y *= sc; }
      Wb[g] = v;
    }
  }
}

// lam[e] = (sum_i nu_i^2)^(beta/2), f_i = min(k_i, M_i - k_i), lam[0] = 0.
// fftrf == 0: nu_i = f_i / M_i (cycles per grid spacing: the correlation length does not depend on the embedding);
// fftrf != 0: nu_i = f_i, the INTEGER wavenumbers FFTRF.jl:86-89 + computesqrtS_f (:40-72) use on its 2N embedding.
__global__ __launch_bounds__(256) void fft_spectrum_kernel(double* __restrict__ lam, int64_t Mtot, int64_t M0, int64_t M1,
                                                           int64_t M2, double beta, int fftrf) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256) {
    const int64_t k0 = e % M0, r = e / M0, k1 = r % M1, k2 = r / M1;
    const double f0 = (double)((k0 <= M0 - k0) ? k0 : M0 - k0) / (fftrf ? 1.0 : (double)M0);
    const double f1 = (double)((k1 <= M1 - k1) ? k1 : M1 - k1) / (fftrf ? 1.0 : (double)M1);
    const double f2 = (double)((k2 <= M2 - k2) ? k2 : M2 - k2) / (fftrf ? 1.0 : (double)M2);
    const double k2sum = f0 * f0 + f1 * f1 + f2 * f2;
    lam[e] = (k2sum > 0.0) ? pow(k2sum, 0.5 * beta) : 0.0;
  }
}

// partial sums of lam -> part[blockIdx.x]; then lam *= 1 / sum
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

__global__ __launch_bounds__(256) void fft_twiddle_kernel(double2* __restrict__ twg) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < FFT_TW_LEN / 2) {
    double s, c;
    sincospi(2.0 * (double)k / (double)FFT_TW_LEN, &s, &c);
    twg[k] = make_double2(c, -s);
  }
}

// lam layout: [Mtot spectrum | 64 scratch | FFT_TW_LEN doubles of twiddles]
size_t fft_plan_doubles(const int64_t M[3]) { return (size_t)(M[0] * M[1] * M[2]) + 64 + FFT_TW_LEN; }

void fft_spectrum(hipStream_t st, double* lam, double* part64, const int64_t M[3], double beta, int fftrf) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  hipLaunchKernelGGL(fft_twiddle_kernel, dim3(FFT_TW_LEN / 2 / 256), dim3(256), 0, st, reinterpret_cast<double2*>(lam + Mtot + 64));
  hipLaunchKernelGGL(fft_spectrum_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, M[0], M[1], M[2], beta, fftrf);
  hipLaunchKernelGGL(fft_sum_kernel, dim3(64), dim3(256), 0, st, lam, Mtot, part64);
  hipLaunchKernelGGL(fft_normalise_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, part64, 64);
}

template <int MODE>
static void launch_pass(hipStream_t st, dim3 grid, int threads, size_t shmem, double2* W, int64_t Mtot, const FftPass& ps,
                        const double* lam, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t N0, int64_t col0,
                        int64_t l) {
  static std::atomic<uint64_t> attr_mask{0};
  if (first_use_on_this_device(attr_mask))
    (void)hipFuncSetAttribute((const void*)fft_pass_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  hipLaunchKernelGGL((fft_pass_kernel<MODE>), grid, dim3(threads), shmem, st, W, Mtot, ps, lam, X, ldx, Y, ldy, N0, col0, l);
}

// one pass along `axis` (N, M: squeezed grid, axis 0 is never a singleton)
static void fft_pass(hipStream_t st, double2* W, int nb, const int64_t N[3], const int64_t M[3], int axis, bool inverse,
                     bool first_fwd, bool last_fwd, bool last_inv, const double* lam, const double* X, int64_t ldx,
                     double* Y, int64_t ldy, int64_t col0, int64_t l) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  FftPass ps;
  ps.Ma = (int)M[axis];
  ps.log2Ma = ilog2(ps.Ma);
  ps.nin = inverse ? ps.Ma : (int)N[axis];
  ps.nout = inverse ? (int)N[axis] : ps.Ma;
  ps.sign = inverse ? 1.0 : -1.0;
  const int64_t stride[3] = {1, M[0], M[0] * M[1]};
  ps.estride = stride[axis];
  ps.R1 = 1; ps.S1 = 0; ps.R2 = 1; ps.S2 = 0;
  if (axis == 0) { ps.R1 = N[1]; ps.S1 = stride[1]; ps.R2 = N[2]; ps.S2 = stride[2]; }
  else if (axis == 1) { ps.R1 = N[2]; ps.S1 = stride[2]; }
  // lines per workgroup (measured, tools/fft_budget_sweep.sh): contiguous lines need no neighbours -- <= 32 KB of
  // LDS so that several workgroups per CU overlap their load / butterfly / store phases; strided lines want long
  // segments (T up to 16 = 256 bytes) but still two workgroups per CU (<= 76 KB) -- except that fewer than 4 lines
  // (64-byte segments) is worse than one workgroup per CU, so 2048-point lines (32 KB each) take 152 KB for T = 4.
  const size_t line_bytes = (size_t)(ps.Ma + 1) * sizeof(double2);
  static const int b0 = getenv("GSI_FFT_B0") ? atoi(getenv("GSI_FFT_B0")) : 32;
  static const int b1 = getenv("GSI_FFT_B1") ? atoi(getenv("GSI_FFT_B1")) : 76;
  const size_t twb = (size_t)ps.Ma / 2 * sizeof(double2);
  auto lines_in = [&](size_t kb) -> int { return (kb * 1024 > twb + line_bytes) ? (int)((kb * 1024 - twb) / line_bytes) : 1; };
  int T = lines_in((size_t)(axis == 0 ? b0 : b1));
  if (axis != 0 && T < 4) { const int t2 = lines_in(152); T = t2 < 4 ? t2 : 4; }
  if (T > 16) T = 16;
  if (T < 1) T = 1;
  ps.T = T;
  const size_t shmem = twb + (size_t)T * line_bytes;
  const int threads = ((int64_t)T * ps.Ma >= 4096) ? 1024 : (((int64_t)T * ps.Ma >= 2048) ? 512 : 256);
  const int64_t nouter = ps.R1 * ps.R2;
  const int64_t tiles = (axis == 0) ? (nouter + T - 1) / T : ((ps.estride + T - 1) / T) * nouter;
  dim3 grid((unsigned)tiles, (unsigned)nb);
  const int64_t N0 = N[0];
#define GSI_FFT_LAUNCH(MODE) launch_pass<MODE>(st, grid, threads, shmem, W, Mtot, ps, lam, X, ldx, Y, ldy, N0, col0, l)
  if (axis == 0) {
    if (!inverse) {
      if (first_fwd && last_fwd) GSI_FFT_LAUNCH(FFT_AXIS0 | FFT_LOADX | FFT_SCALE);
      else if (first_fwd) GSI_FFT_LAUNCH(FFT_AXIS0 | FFT_LOADX);
      else GSI_FFT_LAUNCH(FFT_AXIS0);
    } else {
      if (last_inv) GSI_FFT_LAUNCH(FFT_AXIS0 | FFT_STOREY);
      else GSI_FFT_LAUNCH(FFT_AXIS0);
    }
  } else {
    if (!inverse && last_fwd) GSI_FFT_LAUNCH(FFT_SCALE);
    else GSI_FFT_LAUNCH(0);
  }
#undef GSI_FFT_LAUNCH
}

// Y (n x l, ld ldy) = A X for the embedded-circulant covariance; W holds nb_max * Mtot complex doubles.
// N, M are the SQUEEZED dimensions (singleton axes removed, trailing ones = 1): axis 0 is a real axis.
void fft_cov_apply(hipStream_t st, const int64_t N[3], const int64_t M[3], const double* lam, double2* W, int nb_max,
                   int64_t l, const double* X, int64_t ldx, double* Y, int64_t ldy) {
  int d = 1;
  if (M[1] > 1) d = 2;
  if (M[2] > 1) d = 3;
  const int64_t npairs = (l + 1) / 2;
  for (int64_t p0 = 0; p0 < npairs; p0 += nb_max) {
    const int nb = (int)((npairs - p0 < nb_max) ? (npairs - p0) : nb_max);
    const int64_t col0 = 2 * p0;
    for (int a = 0; a < d; ++a)
      fft_pass(st, W, nb, N, M, a, false, a == 0, a == d - 1, false, lam, X, ldx, Y, ldy, col0, l);
    for (int a = d - 1; a >= 0; --a)
      fft_pass(st, W, nb, N, M, a, true, false, false, a == 0, lam, X, ldx, Y, ldy, col0, l);
  }
}

}}  // namespace gsi::hipk
