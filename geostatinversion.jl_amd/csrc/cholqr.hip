// cholqr.hip -- CholeskyQR2 for the tall-skinny panels of the range finder (the thin-Q step of
// `qr(Y, Val(true))`, RandMatFact.jl:57-58,75-76, and the QR half of `svd(B)`, :86).
//
// Why: Householder on an m x l panel costs ~l/NB sweeps of the panel plus 2 launches per column; at
// the metric's nominal size (m = 10^6, l = 320) that is ~100 GB of HBM traffic per factorization.
// CholeskyQR2 is two rounds of { G = Y'Y (MFMA), R = chol(G) (l x l), X = R^-1 (l x l, explicit),
// Y_next = Y X (ONE out-of-place MFMA product) }: ~8 panel passes, all through the contraction kernel of
// gemm_f64.hip, and the input panel stays intact until the method is known to have worked (no save copy).  The computed Q has the same range as Y (all the reference keeps of its pivoted QR),
// is orthonormal to machine precision after the second round, and W = Q (R2 R1) holds to O(eps)|W|,
// so singular values keep the absolute O(eps sigma_1) accuracy of a Householder QR.
//
// Safety: the method needs cond(Y) < ~1e7.  The first Cholesky can break down (or silently lose
// everything) beyond that, and the reference's own tests produce exactly rank-deficient sketches
// (SURVEY.md H7).  Both rounds therefore raise a device flag on a non-positive / negligible pivot and
// the second round checks |Q1'Q1 - I|_max; on any doubt the caller runs the Householder path (panel_qr.hip)
// on the untouched panel.  Decision = one 4-byte read per factorization, taken before the last product.
#include "hip_common.hpp"
#include <cfloat>

namespace gsi { namespace hipk {

constexpr int CQ_TB = 32;   // column block of the triangular solve

// ---- blocked upper Cholesky of the small l x l Gram matrix ----------------------------------------------
// tiny[0] = 16 l eps max_i G_ii : pivots below it mean "numerically rank deficient" (flag |= 1)
__global__ __launch_bounds__(256) void cq_diagmax_kernel(const double* __restrict__ G, int l, double* __restrict__ tiny) {
  __shared__ double s[256];
  double mx = 0.0;
  for (int i = threadIdx.x; i < l; i += 256) mx = fmax(mx, G[i + (int64_t)i * l]);
  s[threadIdx.x] = mx;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) s[threadIdx.x] = fmax(s[threadIdx.x], s[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) tiny[0] = s[0] * (double)l * DBL_EPSILON * 16.0;
}

// broadcast of lane `src` (compile-time after unrolling) of a double: two v_readlane_b32, the result is wave-uniform
__device__ __forceinline__ double cq_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// One block column jb: U11 = chol(G11) and X = U11^-1, then the block row U12 = U11^-T G12.
// The trailing update G22 -= U12' U12 is an MFMA GEMM issued by the host.  One workgroup.
// The 32 x 32 factorization and inverse run in ONE WAVE, lane c = column c of the block held in registers, the pivot
// row reaching the other lanes through v_readlane (as 32 column steps x 3 __syncthreads over LDS, plus a
// one-thread-per-column back substitution, the same arithmetic took 63 us per block: 10 % of a CholeskyQR2 at l = 320).
// The operations and their order are those of the LDS form: results are bit-identical to it.
__global__ __launch_bounds__(256) void cq_chol_block_kernel(double* __restrict__ G, int l, int j0, int b,
                                                            const double* __restrict__ tiny_p,
                                                            double* __restrict__ Xout /* TB x TB */,
                                                            int32_t* __restrict__ flag) {
  __shared__ double X[CQ_TB][CQ_TB + 1];
  const int tid = threadIdx.x;
  if (tid < 64) {
    const double tiny = tiny_p[0];
    const int c = tid & (CQ_TB - 1);                  // lanes 32..63 mirror lanes 0..31 (no stores)
    double col[CQ_TB], x[CQ_TB];
#pragma unroll
    for (int r = 0; r < CQ_TB; ++r)
      col[r] = (r < b && c < b) ? G[(j0 + r) + (int64_t)(j0 + c) * l] : (r == c ? 1.0 : 0.0);
    bool bad = false;
#pragma unroll
    for (int k = 0; k < CQ_TB; ++k) {
      double d = cq_bcast(col[k], k);
      if (k < b && !(d > tiny)) { bad = true; d = 1.0; }
      const double ukk = sqrt(d);
      col[k] = (c == k) ? ukk : col[k] / ukk;          // row k of U (lanes c < k hold nothing that is read)
#pragma unroll
      for (int r = k + 1; r < CQ_TB; ++r) col[r] -= cq_bcast(col[k], r) * col[k];
      __builtin_amdgcn_sched_barrier(0);               // or the scheduler hoists hundreds of broadcasts (SGPR pairs) at once and spills
    }
    if (bad && tid == 0) atomicOr(flag, 1);
    // X = U^-1 (upper): column c by back substitution, the terms with p > c multiply exact zeros
#pragma unroll
    for (int r = CQ_TB - 1; r >= 0; --r) {
      double sacc = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int p = r + 1; p < CQ_TB; ++p) sacc -= cq_bcast(col[r], p) * x[p];
      x[r] = (r <= c) ? sacc / cq_bcast(col[r], r) : 0.0;
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tid < CQ_TB) {
#pragma unroll
      for (int r = 0; r < CQ_TB; ++r) {
        X[r][c] = x[r];
        Xout[r + c * CQ_TB] = x[r];
        if (r < b && c < b) G[(j0 + r) + (int64_t)(j0 + c) * l] = (r <= c) ? col[r] : 0.0;
      }
    }
  }
  __syncthreads();
  // block row: U12[:, c] = X' * G12[:, c]  (thread = one trailing column), and zero the block below U11
  // (r is a run-time loop on purpose: fully unrolled, the 528 entries of X were hoisted out of the column loop into
  // registers and from there to scratch; the entries below the diagonal of X are exact zeros, so every row sums all 32)
  for (int c = j0 + b + tid; c < l; c += 256) {
    double g[CQ_TB];
#pragma unroll
    for (int r = 0; r < CQ_TB; ++r) g[r] = (r < b) ? G[(j0 + r) + (int64_t)c * l] : 0.0;
#pragma unroll 1
    for (int r = 0; r < b; ++r) {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < CQ_TB; ++p) s += X[p][r] * g[p];
      G[(j0 + r) + (int64_t)c * l] = s;
    }
  }
  for (int e = tid; e < b * (l - j0 - b); e += 256) {   // strictly-lower part of this block column -> 0
    const int c = j0 + e % b, r = j0 + b + e / b;
    G[r + (int64_t)c * l] = 0.0;
  }
}

// ---- Y[:, j0:j0+b] <- Y[:, j0:j0+b] * X  (X upper triangular b x b, ld CQ_TB, zeros below the diagonal) ----
// A workgroup takes 8 rows: the 8 x 32 inputs go through LDS, thread (row, c) sums its 32 products in the order
// p = 0..31 (the terms past the diagonal are exact zeros).  With one thread per row the l x l inverse ran on 320
// threads, each a chain of 32 strided loads and 528 FMAs: 26-37 us per block, ten blocks per Cholesky round.
__global__ __launch_bounds__(256) void cq_right_mult_kernel(double* __restrict__ Y, int64_t m, int64_t ld,
                                                            int64_t j0, int b, const double* __restrict__ X) {
  __shared__ double Xs[CQ_TB][CQ_TB + 1];
  __shared__ double xs[8][CQ_TB + 1];
  const int tid = threadIdx.x, c = tid & (CQ_TB - 1), rl = tid >> 5;
  for (int e = tid; e < CQ_TB * CQ_TB; e += 256) Xs[e & (CQ_TB - 1)][e >> 5] = X[e];       // Xs[p][c] = X[p + c * TB]
  for (int64_t r0 = (int64_t)blockIdx.x * 8; r0 < m; r0 += (int64_t)gridDim.x * 8) {
    const int64_t r = r0 + rl;
    __syncthreads();                                  // Xs written / the previous pass's reads of xs done
    xs[rl][c] = (r < m && c < b) ? Y[r + (j0 + c) * ld] : 0.0;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < CQ_TB; ++p) s += xs[rl][p] * Xs[p][c];
    if (r < m && c < b) Y[r + (j0 + c) * ld] = s;
  }
}

// ---- flag |= 2 when max |G - I| > thresh (second-round Gram matrix of a would-be orthonormal Q1) ----
__global__ __launch_bounds__(256) void cq_orth_check_kernel(const double* __restrict__ G, int l, double thresh,
                                                            int32_t* __restrict__ flag) {
  const int64_t total = (int64_t)l * l;
  bool bad = false;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e % l), c = (int)(e / l);
    const double v = G[e] - (r == c ? 1.0 : 0.0);
    if (!(fabs(v) <= thresh)) bad = true;   // also catches NaN
  }
  if (bad) atomicOr(flag, 2);
}

// ---- G <- symmetric: the strictly lower triangle mirrors the upper one (the symmetric product computes only the
// tiles that touch the upper triangle) ----
__global__ void cq_mirror_upper_kernel(double* __restrict__ G, int l) {
  const int64_t total = (int64_t)l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % l), c = (int)(e / l);
    if (r > c) G[e] = G[c + (int64_t)r * l];
  }
}

// ---- R = R2 * R1 (both upper triangular l x l) ----
__global__ void cq_triprod_kernel(const double* __restrict__ R2, const double* __restrict__ R1, int l,
                                  double* __restrict__ R) {
  const int64_t total = (int64_t)l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % l), c = (int)(e / l);
    double s = 0.0;
    if (r <= c)
      for (int p = r; p <= c; ++p) s += R2[r + (int64_t)p * l] * R1[p + (int64_t)c * l];
    R[e] = s;
  }
}

static inline int grid_for(int64_t total, int cap = 2048) {
  int64_t g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// X (l x l) <- identity
__global__ void cq_identity_kernel(double* __restrict__ X, int l) {
  const int64_t total = (int64_t)l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    X[e] = (e % l == e / l) ? 1.0 : 0.0;
}

// G_ii += s: the shift of shifted CholeskyQR3 (Fukaya, Kannan, Nakatsukasa, Yamamoto, Yanagisawa 2020).  With
// chol(G + sI) the first round cannot break down and Y R^-1 has a condition number of order sqrt(s)/sigma_min,
// which two further plain rounds finish.  The paper's s = 11 (m l + l (l + 1)) u |Y|^2 is a worst-case bound on
// the rounding error of the Gram matrix; here s = 4 l sqrt(m) u trace(G) (the probabilistic size of that error,
// u = 2^-53, trace(G) = |Y|_F^2), which extends the reach from cond ~1e11 to ~1e13: if it is ever too small the
// Cholesky pivots flag it and the caller falls back -- a wrong guess costs time, never accuracy.
__global__ __launch_bounds__(256) void cq_shift_kernel(double* __restrict__ G, int l, double m) {
  __shared__ double s[256];
  double tr = 0.0;
  for (int i = threadIdx.x; i < l; i += 256) tr += G[i + (int64_t)i * l];
  s[threadIdx.x] = tr;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  const double shift = 4.0 * (double)l * sqrt(m) * (0.5 * DBL_EPSILON) * s[0];
  for (int i = threadIdx.x; i < l; i += 256) G[i + (int64_t)i * l] += shift;
}

size_t cholqr_small_doubles(int64_t l) {
  const int64_t nblk = (l + CQ_TB - 1) / CQ_TB;
  return (size_t)(7 * l * l + nblk * CQ_TB * CQ_TB + 64);
}

namespace {
struct CqBufs { double *R1, *R2, *R3, *Gt, *X1, *X2, *X3, *Rinv; };
inline CqBufs cq_bufs(double* small, int l) {
  const size_t ll = (size_t)l * l;
  return {small, small + ll, small + 2 * ll, small + 3 * ll, small + 4 * ll, small + 5 * ll, small + 6 * ll,
          small + 7 * ll};
}

// One round: Rp <- chol(src' src) (upper), X <- Rp^-1 (explicit, l x l), dst <- src X.
// The inverse costs 2 * l/32 launches on l x l data (the blocked solve applied to the identity); what it buys
// is ONE pass over the panel for Y R^-1 (read src, write dst) instead of a blocked in-place solve that re-reads
// the already solved block columns for every later one (l/64 panel reads), and an untouched src.
void cq_round(hipStream_t st, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t m, int l, double* Rp,
              double* X, const CqBufs& b, bool check, int32_t* flag, double* gemm_ws, bool apply, bool shift = false) {
  const int nblk = (l + CQ_TB - 1) / CQ_TB;
  if (!syrk_full_from_upper(st, l, m, src, lds, Rp, l, gemm_ws)) {                    // G = Y'Y: panel read once (syrk_f64.hip)
    gemm_f64_syrk_upper(st, l, m, src, lds, Rp, l, gemm_ws);                          //          or upper tiles of the general kernel
    hipLaunchKernelGGL(cq_mirror_upper_kernel, dim3(64), dim3(256), 0, st, Rp, l);
  }
  if (check)
    hipLaunchKernelGGL(cq_orth_check_kernel, dim3(grid_for((int64_t)l * l, 64)), dim3(256), 0, st, Rp, l, 0.1, flag);
  if (shift) hipLaunchKernelGGL(cq_shift_kernel, dim3(1), dim3(256), 0, st, Rp, l, (double)m);
  // R = chol(G), blocked: per 32-column block one small kernel (diagonal block + its inverse + block
  // row) and one MFMA GEMM for the trailing update; the block inverses are what the solve below needs
  hipLaunchKernelGGL(cq_diagmax_kernel, dim3(1), dim3(256), 0, st, Rp, l, b.Gt);
  for (int jb = 0; jb < nblk; ++jb) {
    const int j0 = jb * CQ_TB;
    const int bb = (l - j0 < CQ_TB) ? (l - j0) : CQ_TB;
    hipLaunchKernelGGL(cq_chol_block_kernel, dim3(1), dim3(256), 0, st, Rp, l, j0, bb, b.Gt,
                       b.Rinv + (size_t)jb * CQ_TB * CQ_TB, flag);
    const int t = l - j0 - bb;
    if (t > 0)
      gemm_f64(st, true, t, t, bb, -1.0, Rp + j0 + (int64_t)(j0 + bb) * l, l, Rp + j0 + (int64_t)(j0 + bb) * l, l, 1.0,
               Rp + (j0 + bb) + (int64_t)(j0 + bb) * l, l, gemm_ws);
  }
  hipLaunchKernelGGL(cq_identity_kernel, dim3(grid_for((int64_t)l * l, 256)), dim3(256), 0, st, X, l);
  for (int jb = 0; jb < nblk; ++jb) {                                                 // X <- I R^-1, blocked
    const int64_t j0 = (int64_t)jb * CQ_TB;
    const int bb = (int)((l - j0 < CQ_TB) ? (l - j0) : CQ_TB);
    if (j0 > 0)
      gemm_f64(st, false, j0 + bb, bb, j0, -1.0, X, l, Rp + j0 * (int64_t)l, l, 1.0, X + j0 * l, l, gemm_ws);
    hipLaunchKernelGGL(cq_right_mult_kernel, dim3((unsigned)((j0 + bb + 7) / 8)), dim3(256), 0, st, X, j0 + bb, (int64_t)l,
                       j0, bb, b.Rinv + (size_t)jb * CQ_TB * CQ_TB);
  }
  if (apply && !trmm_upper_tall(st, m, l, src, lds, X, l, dst, ldd))                      // dst = src R^-1 (R^-1 upper)
    gemm_f64_trmm_upper(st, m, l, l, src, lds, X, l, dst, ldd, gemm_ws);
}
}  // namespace

// CholeskyQR2 on Y (m x l, ld) in two host-visible steps.
// cholqr2_factor: T (m x l, ld ldt) <- Y R1^-1, then R2 = chol(T'T) and R2^-1; Y is NOT modified.  flag (zeroed
// device int) != 0 afterwards means "do not trust it" -- the caller runs Householder on the untouched Y.
// cholqr2_apply (after the host has read flag == 0): Y <- T R2^-1, R (may be null) <- R2 R1.
void cholqr2_factor(hipStream_t st, const double* Y, int64_t m, int64_t l64, int64_t ld, double* T, int64_t ldt,
                    double* small, int32_t* flag, double* gemm_ws) {
  const int l = (int)l64;
  const CqBufs b = cq_bufs(small, l);
  cq_round(st, Y, ld, T, ldt, m, l, b.R1, b.X1, b, false, flag, gemm_ws, true);
  cq_round(st, T, ldt, nullptr, 0, m, l, b.R2, b.X2, b, true, flag, gemm_ws, false);
}

void cholqr2_apply(hipStream_t st, double* Y, int64_t m, int64_t l64, int64_t ld, const double* T, int64_t ldt,
                   double* R, double* small, double* gemm_ws) {
  const int l = (int)l64;
  const CqBufs b = cq_bufs(small, l);
  if (!trmm_upper_tall(st, m, l, T, ldt, b.X2, l, Y, ld)) gemm_f64_trmm_upper(st, m, l, l, T, ldt, b.X2, l, Y, ld, gemm_ws);
  if (R != nullptr)
    hipLaunchKernelGGL(cq_triprod_kernel, dim3(grid_for((int64_t)l * l, 256)), dim3(256), 0, st, b.R2, b.R1, l, R);
}

// after cholqr2_factor: R (l x l) <- R2 R1, and where R2^-1 lives in the small workspace (svd(B) without the thin Q:
// hip_backend.hip:svd_tall_fused)
void cholqr2_R(hipStream_t st, int64_t l64, double* small, double* R) {
  const int l = (int)l64;
  const CqBufs b = cq_bufs(small, l);
  hipLaunchKernelGGL(cq_triprod_kernel, dim3(grid_for((int64_t)l * l, 256)), dim3(256), 0, st, b.R2, b.R1, l, R);
}
const double* cholqr2_X2(const double* small, int64_t l) { return small + 5 * (size_t)l * (size_t)l; }

// Shifted CholeskyQR3, the tier between CholeskyQR2 and Householder: panels with cond up to ~1e15 (sketches of
// fast-decaying covariance spectra after the power iterations).  Y -> T (shifted round) -> S (plain round); third
// Gram matrix, its orthogonality check, R3 and R3^-1.  Y is NOT modified.  Afterwards flag != 0 means "not trusted".
void scholqr3_factor(hipStream_t st, const double* Y, int64_t m, int64_t l64, int64_t ld, double* T, int64_t ldt,
                     double* S, int64_t lds_, double* small, int32_t* flag, double* gemm_ws) {
  const int l = (int)l64;
  const CqBufs b = cq_bufs(small, l);
  cq_round(st, Y, ld, T, ldt, m, l, b.R1, b.X1, b, false, flag, gemm_ws, true, true);
  cq_round(st, T, ldt, S, lds_, m, l, b.R2, b.X2, b, false, flag, gemm_ws, true);
  cq_round(st, S, lds_, nullptr, 0, m, l, b.R3, b.X3, b, true, flag, gemm_ws, false);
}

// after the host has read flag == 0: Y <- S R3^-1, R (may be null) <- R3 R2 R1
void scholqr3_apply(hipStream_t st, double* Y, int64_t m, int64_t l64, int64_t ld, const double* S, int64_t lds_,
                    double* R, double* small, double* gemm_ws) {
  const int l = (int)l64;
  const CqBufs b = cq_bufs(small, l);
  if (!trmm_upper_tall(st, m, l, S, lds_, b.X3, l, Y, ld)) gemm_f64_trmm_upper(st, m, l, l, S, lds_, b.X3, l, Y, ld, gemm_ws);
  if (R != nullptr) {
    const int g = grid_for((int64_t)l * l, 256);
    hipLaunchKernelGGL(cq_triprod_kernel, dim3(g), dim3(256), 0, st, b.R2, b.R1, l, b.X1);   // X1 is free: R2 R1
    hipLaunchKernelGGL(cq_triprod_kernel, dim3(g), dim3(256), 0, st, b.R3, b.X1, l, R);
  }
}

}}  // namespace gsi::hipk
