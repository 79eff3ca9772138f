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
#include <atomic>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

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

// ---- R = chol(G) and X = R^-1 of the l x l Gram matrix in ONE launch (round 5) ---------------------------
// The blocked form above is 2 + 4 l / 32 dependent launches per Cholesky round on l x l data (41 at l = 320: one-wave
// factorizations of the diagonal blocks, trailing updates and the products of the blocked inverse through the big
// contraction kernel, which is all latency at this size): 0.6 ms per round, two rounds per thin QR.  A dependent launch costs
// ~4.5 us on this chip whatever it does, and the small products ran at 12 - 25 us each.  Here one workgroup of 8 waves does
// the whole job out of L2 and LDS:
//   per 32-column block   wave 0: chol of the diagonal block in registers (lane = column, pivot row through v_readlane: the
//                         code of cq_chol_block_kernel);  all: the block row U12 = U11^-T G12 by forward substitution, one
//                         thread per trailing column (no explicit inverse on the critical path), kept in LDS;  all: the
//                         trailing update G22 -= U12' U12 of the UPPER triangle as 16 x 16 MFMA tiles fed from LDS, the
//                         accumulator's lane index along the rows of G (contiguous loads and stores);
//   then                  the diagonal blocks' inverses, one wave per block, all at once;  block column j of X = R^-1 as
//                         -X[0:j0, 0:j0] (R[0:j0, j] X_jj): the small factor row-wise in LDS, the product as MFMA tiles with
//                         the X operand read where it lies (lane index along its rows).
// Measured at l = 320 (GSI_CQ_TRACE=1, 100 MHz stamps by thread 0): 0.39 ms per round -- set-up 12 us, the first diagonal block
// 13, block rows 105, trailing updates with the next diagonal block inside them 145, diagonal inverses 19, W 28, the inverse's
// tiles 68 -- against ~0.55 for the blocked form: the thin QR of a 10^6 x 320 panel 10.9 -> 10.5 ms, of a 125 000-row shard
// 2.87 -> 2.42, 65536 x 160 0.82 -> 0.66 (profiles/r05_qr_fused_ab.log).  First version 0.49 ms: the W rows through the matrix
// cores (72 -> 27 us), reciprocal square roots in the pivots, four trailing tiles' loads in flight, one block of look-ahead
// (diagonal blocks 98 + trailing 93 -> 13 + 145) and 16-byte accesses in the block row (118 -> 105) took it here.  What is left
// is a chain per block: a block row (10.5 us: one thread per column, LDS-broadcast reads of U11 the limit), the next block's
// tiles, max(diagonal block, rest of the update).  Results differ from the blocked form in rounding only (summation order of
// the updates; reciprocal pivots).
constexpr int CQF_THREADS = 512;     // 8 waves, 2 per SIMD: 256 VGPRs each (the in-register 32 x 32 factorizations hold 64 - 128 of them)
constexpr int CQF_MAXL = 384;                    // beyond: the blocked multi-launch form (its products use the whole chip)
constexpr int CQF_LDU = CQF_MAXL - CQ_TB + 16;   // row stride of the block row image Us[p][c]: = 16 mod 32 (bank spread of the MFMA operand reads)
constexpr int CQF_LDK = CQF_MAXL - CQ_TB + 4;    // row stride of Wt[c][k]: = 4 mod 32
constexpr int CQF_DS = CQ_TB + 4;               // row stride of the 32 x 32 LDS blocks: rows 16-byte aligned (broadcast reads of 8 entries)
static_assert(CQF_LDU % 32 == 16 && CQF_LDK % 32 == 4, "LDS strides");
typedef double cq_double4 __attribute__((ext_vector_type(4)));
constexpr size_t CQF_LDS_BYTES = sizeof(double) * ((size_t)CQ_TB * CQF_LDU + (size_t)CQ_TB * CQF_DS + 2 * CQ_TB + 64);

// wave 0: U11 = chol(G11) in registers; Ds <- U11 (zeros below the diagonal), Dinv <- reciprocal pivots, G11 <- U11
__device__ __forceinline__ void cqf_chol_diag(double* __restrict__ G, int l, int j0, int b, double tiny, double* __restrict__ Ds,
                                                        double* __restrict__ Dinv, int32_t* __restrict__ flag, int lane) {
  const int c = lane & (CQ_TB - 1);                     // lanes 32..63 mirror lanes 0..31 (no stores)
  double col[CQ_TB];
#pragma unroll
  for (int r = 0; r < CQ_TB; ++r)
    col[r] = (r < b && c < b && r <= c) ? G[(j0 + r) + (int64_t)(j0 + c) * l] : (r == c ? 1.0 : 0.0);
  bool bad = false;
  double rdiag = 1.0;
#pragma unroll
  for (int k = 0; k < CQ_TB; ++k) {
    double d = cq_bcast(col[k], k);
    if (k < b && !(d > tiny)) { bad = true; d = 1.0; }
    // 1 / sqrt(d) to fp64 (v_rsq_f64 + two Newton steps) and multiplications: the sqrt + two divisions of the blocked form are
    // ~100 dependent instructions per pivot on the one wave everything else waits for
    double rs = __builtin_amdgcn_rsq(d);
    rs = rs * (1.5 - 0.5 * d * rs * rs);
    rs = rs * (1.5 - 0.5 * d * rs * rs);
    const double ukk = d * rs;
    col[k] = (c == k) ? ukk : col[k] * rs;             // row k of U (lanes c < k hold nothing that is read)
    if (c == k) rdiag = rs;
#pragma unroll
    for (int r = k + 1; r < CQ_TB; ++r) col[r] -= cq_bcast(col[k], r) * col[k];
    __builtin_amdgcn_sched_barrier(0);                 // or the scheduler hoists hundreds of broadcasts (SGPR pairs) at once and spills
  }
  if (bad && lane == 0) atomicOr(flag, 1);
  if (lane < CQ_TB) {
    Dinv[c] = rdiag;
#pragma unroll
    for (int r = 0; r < CQ_TB; ++r) {
      const double v = (r <= c) ? col[r] : 0.0;
      Ds[r * CQF_DS + c] = v;
      if (r < b && c < b) G[(j0 + r) + (int64_t)(j0 + c) * l] = v;
    }
  }
}
// one wave: X_jj = U_jj^-1 by back substitution along the lanes (lane = column)
__device__ __forceinline__ void cqf_inv_diag(const double* __restrict__ G, int l, int j0, int b, double* __restrict__ X, int lane) {
  const int c = lane & (CQ_TB - 1);
  double col[CQ_TB], x[CQ_TB];
#pragma unroll
  for (int r = 0; r < CQ_TB; ++r)
    col[r] = (r < b && c < b && r <= c) ? G[(j0 + r) + (int64_t)(j0 + c) * l] : (r == c ? 1.0 : 0.0);
#pragma unroll
  for (int r = CQ_TB - 1; r >= 0; --r) {
    double sacc = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int p = r + 1; p < CQ_TB; ++p) sacc -= cq_bcast(col[r], p) * x[p];
    x[r] = (r <= c) ? sacc / cq_bcast(col[r], r) : 0.0;
    __builtin_amdgcn_sched_barrier(0);
  }
  if (lane < CQ_TB && c < b) {
#pragma unroll
    for (int r = 0; r < CQ_TB; ++r)
      if (r < b) X[(j0 + r) + (int64_t)(j0 + c) * l] = x[r];
  }
}
// one thread = one trailing column c: u = U11^-T g by forward substitution; Us[:, c] <- u, G12[:, c] <- u.
// Eight rows at a time: the finished rows are re-read from the thread's own column of Us (LDS), so the registers hold eight
// accumulators, not the whole column and the 496 entries of U11 the fully unrolled form loads (which the compiler kept live
// all at once: 4 KB of scratch per lane).
// VEC (l even and a full 32-row block): the thread's 256 contiguous bytes as sixteen 16-byte loads and stores -- across the lanes
// the accesses are a column apart (l x 8 bytes) whatever the width, so a wave instruction touches 64 lines; halving the
// instruction count halves the address traffic of the one texture path the CU's waves share.
template <bool VEC>
__device__ __forceinline__ void cqf_block_row_col(double* __restrict__ gcol /* G + j0 + (j0 + b + c) l */, int b,
                                                  const double* __restrict__ Ds, const double* __restrict__ Dinv,
                                                  double* __restrict__ us /* Us + c */) {
  double g[CQ_TB];                                       // the whole column first: ONE L2 round trip, not one per chunk
  if constexpr (VEC) {
    const double2* g2 = reinterpret_cast<const double2*>(gcol);
#pragma unroll
    for (int p = 0; p < CQ_TB; p += 2) { const double2 v = g2[p >> 1]; g[p] = v.x; g[p + 1] = v.y; }
  } else {
#pragma unroll
    for (int p = 0; p < CQ_TB; ++p) g[p] = (p < b) ? gcol[p] : 0.0;
  }
#pragma unroll
  for (int pb = 0; pb < CQ_TB; pb += 8) {
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = g[pb + j];
#pragma unroll 4
    for (int q = 0; q < pb; ++q) {
      const double uq = us[q * CQF_LDU];
      const double* dr = Ds + q * CQF_DS + pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] -= dr[j] * uq;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < j; ++i) acc[j] -= Ds[(pb + i) * CQF_DS + pb + j] * acc[i];
      acc[j] *= Dinv[pb + j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double v = (pb + j < b) ? acc[j] : 0.0;
      us[(pb + j) * CQF_LDU] = v;
      if constexpr (!VEC) { if (pb + j < b) gcol[pb + j] = v; }
    }
    if constexpr (VEC) {
      double2* o2 = reinterpret_cast<double2*>(gcol + pb);
#pragma unroll
      for (int j = 0; j < 8; j += 2) o2[j >> 1] = make_double2(acc[j], acc[j + 1]);
    }
    asm volatile("" ::: "memory");                       // chunk by chunk: the next chunk's LDS reads are not hoisted over this one's
  }
}
// trace (null in production): 100 MHz ticks per phase, accumulated by thread 0 -- [0] set-up, [1] the first diagonal block, [2]
// block rows, [3] trailing updates with the next diagonal block inside them, [4] diagonal inverses, [5] W = R X_jj, [6] the
// inverse's tiles (GSI_CQ_TRACE=1)
#define CQF_STAMP(ph) do { if (trace != nullptr && tid == 0) { const unsigned long long now_ = wall_clock64(); trace[ph] += now_ - t_last; t_last = now_; } } while (0)
__global__ __launch_bounds__(CQF_THREADS) void cq_chol_inv_kernel(double* __restrict__ G, int l, double* __restrict__ X,
                                                                  int32_t* __restrict__ flag, unsigned long long* __restrict__ trace) {
  unsigned long long t_last = (trace != nullptr) ? wall_clock64() : 0ull;
  extern __shared__ double cqf_lds[];
  double* Us = cqf_lds;                                   // [32][CQF_LDU] block row / (inverse) Wt[32][CQF_LDK]
  double* Ds = Us + CQ_TB * CQF_LDU;                      // [32][33] the diagonal block's factor
  double* Dinv = Ds + CQ_TB * CQF_DS;                // [32] reciprocal pivots
  double* red = Dinv + CQ_TB;                             // [32 + ...] reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = (l + CQ_TB - 1) / CQ_TB;
  // tiny = 16 l eps max_i G_ii : pivots below it mean "numerically rank deficient" (flag |= 1)
  {
    double mx = 0.0;
    for (int i = tid; i < l; i += CQF_THREADS) mx = fmax(mx, G[i + (int64_t)i * l]);
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (tid == 0) {
      double m2 = 0.0;
      for (int w2 = 0; w2 < CQF_THREADS / 64; ++w2) m2 = fmax(m2, red[w2]);
      red[32] = m2 * (double)l * DBL_EPSILON * 16.0;
    }
    // X <- 0 below the diagonal blocks and everywhere it is not written later (the consumers read whole column chunks)
    for (int64_t e = tid; e < (int64_t)l * l; e += CQF_THREADS) X[e] = 0.0;
    __syncthreads();
  }
  const double tiny = red[32];
  CQF_STAMP(0);

  // ================= R = chol(G), right-looking, 32-column blocks, one block of look-ahead =================
  // Per block: B the block row (all waves); C1 the trailing tiles that touch the NEXT block's 32 rows (all waves); C2 wave 0
  // factors the next diagonal block while waves 1..7 finish the rest of the trailing update -- the 9.8 us of a diagonal
  // block hide behind the ~8 us the other tiles take anyway.
  auto trailing_tiles = [&](int j0, int b, int nc, int ncp, bool first_rows, int w0, int nw) __attribute__((always_inline)) {
    // 16 x 16 tiles (tr <= tc) of the upper triangle of the trailing matrix; first_rows: the tiles with tr < 2 (the next block's
    // rows), else the others; dealt over waves w0, w0 + 1, .., w0 + nw - 1.  Four tiles per pass: their 16 loads of G are in
    // flight together (one L2 round trip per pass, not per tile).
    const int nt = ncp >> 4;
    const int nfirst = (nt >= 2) ? (2 * nt - 1) : nt;         // tiles with tr < 2: (0, 0..nt-1), (1, 1..nt-1)
    const int ntot = nt * (nt + 1) / 2;
    const int cnt = first_rows ? nfirst : ntot - nfirst;
    const int base = first_rows ? 0 : nfirst;                 // row-major over the upper triangle: tr = 0, then tr = 1, ...
    const int jl = lane & 15, kq = lane >> 4;
    double* Gt = G + (int64_t)(j0 + b) + (int64_t)(j0 + b) * l;     // trailing matrix (0, 0)
    constexpr int TPP = 4;
    if (wave < w0 || wave >= w0 + nw) return;
    for (int t0 = wave - w0; t0 < cnt; t0 += nw * TPP) {
      cq_double4 acc[TPP];
      int trs[TPP], tcs[TPP];
#pragma unroll
      for (int u = 0; u < TPP; ++u) {
        const int tl = t0 + u * nw;
        int tr = 0, rem = base + ((tl < cnt) ? tl : 0);        // linear index -> (tr <= tc)
        while (rem >= nt - tr) { rem -= nt - tr; ++tr; }
        trs[u] = tr; tcs[u] = tr + rem;
        const int r = tr * 16 + jl;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = tcs[u] * 16 + kq + 4 * v;
          acc[u][v] = (tl < cnt && r < nc && c < nc) ? Gt[r + (int64_t)c * l] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < TPP; ++u) {
#pragma unroll
        for (int k0 = 0; k0 < CQ_TB; k0 += 4) {
          const double fa = -Us[(k0 + kq) * CQF_LDU + tcs[u] * 16 + jl];      // a-operand: i <-> column of G
          const double fb = Us[(k0 + kq) * CQF_LDU + trs[u] * 16 + jl];       // b-operand: j (= lane & 15) <-> row of G
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[u], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < TPP; ++u) {
        const int tl = t0 + u * nw;
        const int r = trs[u] * 16 + jl;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = tcs[u] * 16 + kq + 4 * v;
          if (tl < cnt && r < nc && c < nc) Gt[r + (int64_t)c * l] = acc[u][v];
        }
      }
    }
  };
  if (wave == 0) cqf_chol_diag(G, l, 0, (l < CQ_TB) ? l : CQ_TB, tiny, Ds, Dinv, flag, lane);     // the first diagonal block
  __syncthreads();
  CQF_STAMP(1);
  for (int jb = 0; jb < nblk; ++jb) {
    const int j0 = jb * CQ_TB;
    const int b = (l - j0 < CQ_TB) ? (l - j0) : CQ_TB;
    const int nc = l - j0 - b;                              // trailing columns
    if (nc <= 0) break;
    // ---- B: block row U12 = U11^-T G12, thread = trailing column; and zeros below U11 ----
    const int ncp = (nc + 15) & ~15;                      // the MFMA tiles read whole 16-column groups: zero padding
    const bool vec = ((l & 1) == 0) && b == CQ_TB && ((reinterpret_cast<uintptr_t>(G) & 15) == 0);
    for (int c = tid; c < ncp; c += CQF_THREADS) {
      if (c < nc) {
        if (vec) cqf_block_row_col<true>(G + j0 + (int64_t)(j0 + b + c) * l, b, Ds, Dinv, Us + c);
        else cqf_block_row_col<false>(G + j0 + (int64_t)(j0 + b + c) * l, b, Ds, Dinv, Us + c);
      } else {
#pragma unroll
        for (int p = 0; p < CQ_TB; ++p) Us[p * CQF_LDU + c] = 0.0;
      }
    }
    for (int e = tid; e < b * nc; e += CQF_THREADS) {     // strictly-lower part of this block column -> 0 (rows along the lanes)
      const int r = j0 + b + e % nc, c = j0 + e / nc;
      G[r + (int64_t)c * l] = 0.0;
    }
    __syncthreads();
    CQF_STAMP(2);
    // ---- C1: the trailing tiles of the next block's rows (all waves) ----
    trailing_tiles(j0, b, nc, ncp, true, 0, CQF_THREADS / 64);
    __syncthreads();
    // ---- C2: wave 0 factors the next diagonal block; the other waves update the rest of the trailing matrix ----
    {
      const int j1 = j0 + b;
      const int b1 = (l - j1 < CQ_TB) ? (l - j1) : CQ_TB;
      if (wave == 0) cqf_chol_diag(G, l, j1, b1, tiny, Ds, Dinv, flag, lane);
      trailing_tiles(j0, b, nc, ncp, false, 1, CQF_THREADS / 64 - 1);
    }
    __syncthreads();
    CQF_STAMP(3);
  }

  // ================= X = R^-1 =================
  // diagonal blocks: X_jj = U_jj^-1, one wave per block, all at once
  for (int jb = wave; jb < nblk; jb += CQF_THREADS / 64) {
    const int j0 = jb * CQ_TB;
    cqf_inv_diag(G, l, j0, (l - j0 < CQ_TB) ? (l - j0) : CQ_TB, X, lane);
  }
  __syncthreads();
  CQF_STAMP(4);
  // block columns j = 1 .. nblk - 1:  X[0:j0, j] = -X[0:j0, 0:j0] * (R[0:j0, j] * X_jj)
  double* Wt = Us;                                           // Wt[c][k], row stride CQF_LDK
  double* Xjj = Ds;                                          // [32][33]
  for (int jb = 1; jb < nblk; ++jb) {
    const int j0 = jb * CQ_TB;
    const int b = (l - j0 < CQ_TB) ? (l - j0) : CQ_TB;
    for (int e = tid; e < CQ_TB * CQ_TB; e += CQF_THREADS) {
      const int r = e & (CQ_TB - 1), c = e >> 5;
      Xjj[r * CQF_DS + c] = (r < b && c < b) ? X[(j0 + r) + (int64_t)(j0 + c) * l] : 0.0;
    }
    __syncthreads();
    // W = R[0:j0, jcols] X_jj as MFMA tiles, R read where it lies (lane index along its rows), straight into Wt[c][k]:
    // D[i][j] = sum_p X_jj[p][c0 + i] R[k0 + j][j0 + p]
    {
      const int jl = lane & 15, kq = lane >> 4;
      const int nkt = j0 >> 4;
      for (int t = wave; t < 2 * nkt; t += CQF_THREADS / 64) {
        const int kt = t >> 1, ct = t & 1;
        const double* rp = G + (kt * 16 + jl) + (int64_t)(j0 + kq) * l;
        double fb[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) fb[s8] = (4 * s8 + kq < b) ? rp[(int64_t)(4 * s8) * l] : 0.0;
        cq_double4 acc = (cq_double4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const double fa = Xjj[(4 * s8 + kq) * CQF_DS + ct * 16 + jl];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb[s8], acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) Wt[(ct * 16 + kq + 4 * v) * CQF_LDK + kt * 16 + jl] = acc[v];
      }
    }
    __syncthreads();
    CQF_STAMP(5);
    // tiles: 16 rows (r0 < j0) x the 32 columns of the block column (two accumulators sharing the X operand); reduction
    // k = r0 .. j0 (X is upper triangular) in chunks of 16, the X operand of the next chunk requested before this chunk's MFMAs
    const int nrt = j0 >> 4;                                  // j0 is a multiple of 32
    const int jl = lane & 15, kq = lane >> 4;
    for (int rt = wave; rt < nrt; rt += CQF_THREADS / 64) {   // (the long reductions, small r0, are dealt first)
      const int r0 = rt * 16;
      cq_double4 acc0 = (cq_double4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
      const double* xp = X + (r0 + jl) + (int64_t)kq * l;     // b-operand: j <-> row of X (contiguous along the lanes)
      double fbn[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) fbn[s4] = xp[(int64_t)(r0 + 4 * s4) * l];
      for (int kb = r0; kb < j0; kb += 16) {
        double fb[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) fb[s4] = fbn[s4];
        if (kb + 16 < j0) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) fbn[s4] = xp[(int64_t)(kb + 16 + 4 * s4) * l];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int k0 = kb + 4 * s4;
          const double fa0 = -Wt[jl * CQF_LDK + k0 + kq];              // a-operand: i <-> column within the block column
          const double fa1 = -Wt[(16 + jl) * CQF_LDK + k0 + kq];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0, fb[s4], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1, fb[s4], acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int c = kq + 4 * v;
        if (c < b) X[(r0 + jl) + (int64_t)(j0 + c) * l] = acc0[v];
        if (16 + c < b) X[(r0 + jl) + (int64_t)(j0 + 16 + c) * l] = acc1[v];
      }
    }
    __syncthreads();
    CQF_STAMP(6);
  }
}
#undef CQF_STAMP

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
  // R = chol(G) and X = R^-1: one launch up to l = 384 (GSI_CQ_FUSED=0: the blocked form below, A/B)
  static const bool fused_off = (getenv("GSI_CQ_FUSED") != nullptr && getenv("GSI_CQ_FUSED")[0] == '0');
  if (!fused_off && l <= CQF_MAXL && l >= 1) {
    static std::atomic<uint64_t> attr_mask{0};
    if (first_use_on_this_device(attr_mask))
      (void)hipFuncSetAttribute((const void*)cq_chol_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CQF_LDS_BYTES);
    static const bool trace_on = (getenv("GSI_CQ_TRACE") != nullptr);            // measurement only: synchronises and prints
    unsigned long long* trace = nullptr;
    if (trace_on && hipMalloc((void**)&trace, 8 * sizeof(unsigned long long)) == hipSuccess) hipMemsetAsync(trace, 0, 8 * sizeof(unsigned long long), st);
    hipLaunchKernelGGL(cq_chol_inv_kernel, dim3(1), dim3(CQF_THREADS), CQF_LDS_BYTES, st, Rp, l, X, flag, trace);
    if (trace != nullptr) {
      unsigned long long h[8];
      hipMemcpyAsync(h, trace, sizeof(h), hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
      hipFree(trace);
      fprintf(stderr, "[gsi cq trace] l = %d: set-up %.1f us, diagonal blocks %.1f, block rows %.1f, trailing updates %.1f, diagonal inverses %.1f, "
              "W rows %.1f, inverse tiles %.1f\n", l, h[0] / 100.0, h[1] / 100.0, h[2] / 100.0, h[3] / 100.0, h[4] / 100.0, h[5] / 100.0, h[6] / 100.0);
    }
    if (apply && !trmm_upper_tall(st, m, l, src, lds, X, l, dst, ldd))                    // dst = src R^-1 (R^-1 upper)
      gemm_f64_trmm_upper(st, m, l, l, src, lds, X, l, dst, ldd, gemm_ws);
    return;
  }
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
