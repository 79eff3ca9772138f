// jacobi_svd.hip -- the small dense SVD inside `svd(B)` (RandMatFact.jl:86) on gfx950.
// After the tall QR of B' only the l x l triangular factor R is left; this file computes
// R = U S V' and returns U (l x l) and S, which is all randsvd needs (Z = Q_B U sqrt(S)).
//
// One-sided (Hestenes) Jacobi on the columns of R: right rotations orthogonalise the
// columns, which converge to U*S -- high relative accuracy for the small singular values,
// unlike an eigen-decomposition of R'R.  V is never accumulated.
//
// MI355X mapping: block Jacobi.  The columns are cut into blocks of SVD_W columns; a
// round-robin tournament pairs the blocks, one workgroup (16 waves) per block pair per
// round keeps its 2*SVD_W columns (<= 128 KB of the CU's 160 KB LDS) resident and runs a
// full inner round-robin sweep over them out of LDS -- one wave per column pair, the three
// inner products by wavefront shuffles -- then writes the block back.  l <= 2*SVD_W needs
// a single workgroup and no inter-workgroup traffic at all.
#include "../../include/gsi_hip.h"
#include "hip_common.hpp"
#include "backend.hpp"
#include <array>
#include <atomic>
#include <cstdlib>
#include <cfloat>
#include <vector>

namespace gsi { namespace hipk {

constexpr int SVD_THREADS = 256;   // 4 waves; a 16-lane quarter wave per column pair of an inner round

// round-robin ("circle") tournament on n (even) players: pair q of round r
__device__ __host__ inline void rr_pair(int n, int r, int q, int* a, int* b) {
  if (q == 0) { *a = n - 1; *b = r % (n - 1); }
  else {
    *a = (r + q) % (n - 1);
    *b = ((r - q) % (n - 1) + (n - 1)) % (n - 1);
  }
}

__device__ inline double wave_allsum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// sum over the 16 lanes of a quarter wave = one DPP row: four row rotations (row_ror 8, 4, 2, 1), VALU only.
// (__shfl_xor is ds_bpermute on gfx9: four dependent LDS-pipe round trips per sum, three sums per column pair and
// round.)  The partial sums are periodic in the lane index, so each rotation adds the same two numbers the xor
// butterfly added: bit-identical results.
template <int CTRL>
__device__ __forceinline__ double dpp_row_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quarter_allsum(double v) {
  v += dpp_row_mov<0x128>(v);      // row_ror:8
  v += dpp_row_mov<0x124>(v);      // row_ror:4
  v += dpp_row_mov<0x122>(v);      // row_ror:2
  v += dpp_row_mov<0x121>(v);      // row_ror:1
  return v;
}

// One column pair handled by a quarter wave (16 lanes): the three inner products, the rotation, its application.
// NI > 0: l == 16 NI exactly -- every lane holds its NI elements of both columns in registers from the inner products to
// the rotation (one burst of LDS reads per round, a third less LDS traffic; the common sketch widths l = 160, 256, 320).
// NI == 0: any l, two walks over the columns with eight reads in flight.  Same sums in the same order either way.
template <int NI>
__device__ __forceinline__ int jacobi_pair(double* __restrict__ gp, double* __restrict__ gq, int l, int l16, bool active,
                                           double tol2) {
  double a = 0.0, b = 0.0, c = 0.0;
  double xr[NI > 0 ? NI : 1], yr[NI > 0 ? NI : 1];
  if constexpr (NI > 0) {
#pragma unroll
    for (int k = 0; k < NI; ++k) { xr[k] = gp[l16 + 16 * k]; yr[k] = gq[l16 + 16 * k]; }
#pragma unroll
    for (int k = 0; k < NI; ++k) { a += xr[k] * xr[k]; b += yr[k] * yr[k]; c += xr[k] * yr[k]; }
  } else {
#pragma unroll 8
    for (int i = l16; i < l; i += 16) {
      const double x = gp[i], y = gq[i];
      a += x * x; b += y * y; c += x * y;
    }
  }
  a = quarter_allsum(a); b = quarter_allsum(b); c = quarter_allsum(c);
  if (!(active && a > 0.0 && b > 0.0 && c * c > tol2 * (a * b))) return 0;
  // Rutishauser rotation, t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with zeta = (b-a)/(2c),
  // rewritten as t = sign(d*e) |e| / (|d| + hypot(d, e)).  The angle only has to be approximately
  // optimal -- what must hold to fp64 rounding is cs^2 + sn^2 = 1 -- so t comes from the
  // one-instruction v_rsq_f64 / v_rcp_f64 approximations and only cs gets Newton steps.
  const double d = b - a, e = 2.0 * c;
  const double q2 = d * d + e * e;
  const double r = q2 * __builtin_amdgcn_rsq(q2);                      // ~ hypot(d, e)
  const double t = copysign(fabs(e), d * e) * __builtin_amdgcn_rcp(fabs(d) + r);
  const double w = 1.0 + t * t;
  double cs = __builtin_amdgcn_rsq(w);
  cs = cs * (1.5 - 0.5 * w * cs * cs);                                 // Newton: 1/sqrt(w) to fp64
  cs = cs * (1.5 - 0.5 * w * cs * cs);
  const double sn = cs * t;
  if constexpr (NI > 0) {
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      gp[l16 + 16 * k] = cs * xr[k] - sn * yr[k];
      gq[l16 + 16 * k] = sn * xr[k] + cs * yr[k];
    }
  } else {
#pragma unroll 8
    for (int i = l16; i < l; i += 16) {
      const double x = gp[i], y = gq[i];
      gp[i] = cs * x - sn * y;
      gq[i] = sn * x + cs * y;
    }
  }
  return l16 == 0 ? 1 : 0;
}

// The same pair step with column p RESIDENT in the quarter wave's registers (xr, loaded once per launch): in the bipartite
// (cross-block) tournament pair slot `pairidx` keeps column p = pairidx for all SVD_W rounds and only its partner q walks,
// so p's 2 NI LDS reads and NI writes per round go away -- the round is LDS-bandwidth and LDS-latency bound (a round moved
// 160 KB through LDS per workgroup; now 100 KB).  Same operations on the same values in the same order: bit-identical.
template <int NI>
__device__ __forceinline__ int jacobi_pair_resident(double (&xr)[NI], double* __restrict__ gq, int l16, bool active, double tol2) {
  double a = 0.0, b = 0.0, c = 0.0;
  double yr[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) yr[k] = gq[l16 + 16 * k];
#pragma unroll
  for (int k = 0; k < NI; ++k) { a += xr[k] * xr[k]; b += yr[k] * yr[k]; c += xr[k] * yr[k]; }
  a = quarter_allsum(a); b = quarter_allsum(b); c = quarter_allsum(c);
  if (!(active && a > 0.0 && b > 0.0 && c * c > tol2 * (a * b))) return 0;
  const double d = b - a, e = 2.0 * c;
  const double q2 = d * d + e * e;
  const double r = q2 * __builtin_amdgcn_rsq(q2);
  const double t = copysign(fabs(e), d * e) * __builtin_amdgcn_rcp(fabs(d) + r);
  const double w = 1.0 + t * t;
  double cs = __builtin_amdgcn_rsq(w);
  cs = cs * (1.5 - 0.5 * w * cs * cs);
  cs = cs * (1.5 - 0.5 * w * cs * cs);
  const double sn = cs * t;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const double x = xr[k], y = yr[k];
    xr[k] = cs * x - sn * y;
    gq[l16 + 16 * k] = sn * x + cs * y;
  }
  return l16 == 0 ? 1 : 0;
}
template <int NI, int SVD_W>
__device__ __forceinline__ int jacobi_cross_rounds(double* __restrict__ cols, int lp, int pairidx, bool active, int l16, double tol2) {
  const int p = active ? pairidx : 0;
  double* gp = cols + p * lp;
  double xr[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) xr[k] = gp[l16 + 16 * k];
  int rots = 0;
  for (int r = 0; r < SVD_W; ++r) {
    const int q = SVD_W + (p + r) % SVD_W;
    rots += jacobi_pair_resident<NI>(xr, cols + q * lp, l16, active, tol2);
    __syncthreads();
  }
  if (active) {
#pragma unroll
    for (int k = 0; k < NI; ++k) gp[l16 + 16 * k] = xr[k];
  }
  return rots;
}

// ---- one block-pair visit, in three acts: load the two blocks into LDS, sweep the resident columns, write them back ----
// local column c <- global column (c < W ? ba*W + c : bb*W + c - W), zero beyond l
template <int SVD_W>
__device__ __forceinline__ void jacobi_load_blocks(double* __restrict__ cols, const double* __restrict__ G, int l, int lp, int ba, int bb, int tid) {
  constexpr int SVD_C = 2 * SVD_W;
  if ((l & 1) == 0) {
    // even l (lp is even by construction): row pairs as 16-byte loads, column by column -- no division per element,
    // half the load instructions
    for (int r2 = tid; 2 * r2 < lp; r2 += SVD_THREADS) {
#pragma unroll 8
      for (int c = 0; c < SVD_C; ++c) {
        const int gc = (c < SVD_W) ? ba * SVD_W + c : bb * SVD_W + (c - SVD_W);
        double2 v = make_double2(0.0, 0.0);
        if (gc < l && 2 * r2 < l) v = *reinterpret_cast<const double2*>(G + 2 * r2 + (int64_t)gc * l);
        *reinterpret_cast<double2*>(cols + c * lp + 2 * r2) = v;
      }
    }
  } else {
    for (int e = tid; e < SVD_C * lp; e += SVD_THREADS) {
      const int c = e / lp, r = e % lp;
      const int gc = (c < SVD_W) ? ba * SVD_W + c : bb * SVD_W + (c - SVD_W);
      cols[e] = (gc < l && r < l) ? G[r + (int64_t)gc * l] : 0.0;
    }
  }
}
template <int SVD_W>
__device__ __forceinline__ void jacobi_store_blocks(const double* __restrict__ cols, double* __restrict__ G, int l, int lp, int ba, int bb, int tid) {
  constexpr int SVD_C = 2 * SVD_W;
  if ((l & 1) == 0) {
    for (int r2 = tid; 2 * r2 < l; r2 += SVD_THREADS) {
#pragma unroll 8
      for (int c = 0; c < SVD_C; ++c) {
        const int gc = (c < SVD_W) ? ba * SVD_W + c : bb * SVD_W + (c - SVD_W);
        if (gc < l) *reinterpret_cast<double2*>(G + 2 * r2 + (int64_t)gc * l) = *reinterpret_cast<const double2*>(cols + c * lp + 2 * r2);
      }
    }
  } else {
    for (int e = tid; e < SVD_C * lp; e += SVD_THREADS) {
      const int c = e / lp, r = e % lp;
      const int gc = (c < SVD_W) ? ba * SVD_W + c : bb * SVD_W + (c - SVD_W);
      if (gc < l && r < l) G[r + (int64_t)gc * l] = cols[e];
    }
  }
}
// the sweep over the resident columns (every thread of the workgroup calls it; the blocks are in LDS and a barrier has passed);
// returns the rotations this thread counted (lane 0 of a quarter wave counts its pair's)
template <int SVD_W>
__device__ __forceinline__ int jacobi_sweep_blocks(double* __restrict__ cols, int l, int lp, double tol2, int inner_sweeps, int cross_only, int tid) {
  constexpr int SVD_C = 2 * SVD_W;
  const int lane = tid & 63, wave = tid >> 6;
  const int quarter = lane >> 4, l16 = lane & 15;
  int rots = 0;
  const int ni = ((l & 15) == 0) ? (l >> 4) : 0;      // elements per lane when the columns divide evenly
  const bool resident_ok = cross_only && inner_sweeps == 1 && (ni == 20 || ni == 16 || ni == 10) && SVD_W <= SVD_THREADS / 16;
  if (resident_ok) {
    const int pairidx = 4 * wave + quarter;
    const bool active = pairidx < SVD_W;
    if (ni == 20) rots += jacobi_cross_rounds<20, SVD_W>(cols, lp, pairidx, active, l16, tol2);
    else if (ni == 16) rots += jacobi_cross_rounds<16, SVD_W>(cols, lp, pairidx, active, l16, tol2);
    else rots += jacobi_cross_rounds<10, SVD_W>(cols, lp, pairidx, active, l16, tol2);
  } else
  for (int sw = 0; sw < inner_sweeps; ++sw) {
    // cross_only: only pairs (column of block a, column of block b) -- SVD_W rounds of a bipartite
    // tournament; otherwise all pairs of the 2*SVD_W resident columns (2*SVD_W - 1 rounds).  The pairs
    // inside a block are swept once per sweep (round 0), not once per block pairing.
    const int nrounds = cross_only ? SVD_W : SVD_C - 1;
    for (int r = 0; r < nrounds; ++r) {
      int p, q;
      const int pairidx = 4 * wave + quarter;
      const bool active = pairidx < SVD_C / 2;
      if (cross_only) { p = active ? pairidx : 0; q = SVD_W + (p + r) % SVD_W; }
      else rr_pair(SVD_C, r, active ? pairidx : 0, &p, &q);
      double* gp = cols + p * lp;
      double* gq = cols + q * lp;
      if (ni == 20) rots += jacobi_pair<20>(gp, gq, l, l16, active, tol2);
      else if (ni == 16) rots += jacobi_pair<16>(gp, gq, l, l16, active, tol2);
      else if (ni == 10) rots += jacobi_pair<10>(gp, gq, l, l16, active, tol2);
      else rots += jacobi_pair<0>(gp, gq, l, l16, active, tol2);
      __syncthreads();
    }
  }
  return rots;
}

// grid.x = number of block pairs in this round; SVD_W = columns per block (16, or 8 for l > 600)
// sched != null (sparse sweeps): workgroup b handles the block pair sched[3 b], sched[3 b + 1] with cross_only = sched[3 b + 2]
template <int SVD_W>
__global__ __launch_bounds__(SVD_THREADS) void jacobi_block_kernel(double* __restrict__ G, int l, int lp,
                                                                   int nblk, int round, double tol2,
                                                                   int32_t* __restrict__ rotcount,
                                                                   int inner_sweeps, int cross_only,
                                                                   const int32_t* __restrict__ sched) {
  constexpr int SVD_C = 2 * SVD_W;   // columns resident per workgroup
  extern __shared__ double cols[];  // [SVD_C][lp] followed by one int slot (single LDS object)
  int& s_rot = *reinterpret_cast<int*>(cols + SVD_C * lp);
  const int tid = threadIdx.x;
  int ba, bb;
  if (sched != nullptr) { ba = sched[3 * blockIdx.x]; bb = sched[3 * blockIdx.x + 1]; cross_only = sched[3 * blockIdx.x + 2]; }
  else if (nblk <= 2) { ba = 0; bb = 1; }
  else rr_pair(nblk, round, blockIdx.x, &ba, &bb);
  if (tid == 0) s_rot = 0;
  jacobi_load_blocks<SVD_W>(cols, G, l, lp, ba, bb, tid);
  __syncthreads();
  const int rots = jacobi_sweep_blocks<SVD_W>(cols, l, lp, tol2, inner_sweeps, cross_only, tid);
  if ((tid & 15) == 0 && rots) atomicAdd(&s_rot, rots);
  __syncthreads();
  jacobi_store_blocks<SVD_W>(cols, G, l, lp, ba, bb, tid);
  if (tid == 0 && s_rot) atomicAdd(rotcount, s_rot);
}

// (Round 5 built the whole iteration as ONE persistent launch -- the workgroups of a round resident, a grid barrier between the
// rounds, exact skipping of pair slots by modification stamps, then a look-before-sweeping step through the matrix cores --
// and measured it against this form: SLOWER on the benchmark's factor, 4.2 - 4.4 against 3.8 ms (9 sweeps against 8), equal within
// 2 % on random factors.  A round is ~25 us of dependent rotations, not launch latency (the barrier costs what the launch did),
// and a barrier-synchronised round is as slow as its slowest workgroup, so nothing short of the host-built schedule of active
// pairs below makes the late sweeps cheap.  The kernel is kept as tools/rejected_kernels/jacobi_persistent.hip.txt;
// DESIGN.md 4.4, profiles/r05_svd_persistent_ab.log.)
// Which block pairs still hold a column pair that would be rotated (the rotation test itself: c^2 > tol^2 a b)?  One workgroup
// per pair ba <= bb of SVD_W-column blocks, one thread per column pair; flags[pair] = 1 / 0.  What it buys: the sweep that
// only FINDS that nothing is left to rotate (one in ten at l = 320) becomes one small launch, and the late sweeps -- a few
// percent of the pairs still active -- run on a schedule of the active block pairs only.
template <int SVD_W>
__global__ __launch_bounds__(SVD_W * SVD_W) void jacobi_activity_kernel(const double* __restrict__ G, int l, int lp, int nblk, double tol2,
                                                                        int32_t* __restrict__ flags) {
  extern __shared__ double cols[];            // [2 SVD_W][lp]
  __shared__ int s_any;
  // pair index -> (ba <= bb), row-major over the upper triangle
  int ba = 0, rem = (int)blockIdx.x;
  while (rem >= nblk - ba) { rem -= nblk - ba; ++ba; }
  const int bb = ba + rem;
  const int tid = threadIdx.x;
  if (tid == 0) s_any = 0;
  for (int e = tid; e < 2 * SVD_W * lp; e += SVD_W * SVD_W) {
    const int c = e / lp, r = e % lp;
    const int gc = (c < SVD_W) ? ba * SVD_W + c : bb * SVD_W + (c - SVD_W);
    cols[e] = (gc < l && r < l) ? G[r + (int64_t)gc * l] : 0.0;
  }
  __syncthreads();
  const int i = tid / SVD_W, j = tid % SVD_W;
  const double* x = cols + i * lp;
  const double* y = cols + (SVD_W + j) * lp;
  double a = 0.0, b = 0.0, c = 0.0;
  for (int r = 0; r < l; ++r) { const double xv = x[r], yv = y[r]; a += xv * xv; b += yv * yv; c += xv * yv; }
  const bool pairok = (ba != bb) || (i < j);
  if (pairok && a > 0.0 && b > 0.0 && c * c > tol2 * (a * b)) s_any = 1;
  __syncthreads();
  if (tid == 0) flags[blockIdx.x] = s_any;
}

__global__ void jacobi_norms_kernel(const double* __restrict__ G, int l, double* __restrict__ norms) {
  // one wave per column
  const int col = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (col >= l) return;
  double s = 0.0;
  for (int i = lane; i < l; i += 64) { const double x = G[i + (int64_t)col * l]; s += x * x; }
  s = wave_allsum(s);
  if (lane == 0) norms[col] = sqrt(s);
}

// U[:, rank(c)] = G[:, c] / norm_c ; S[rank(c)] = norm_c ; rank by descending norm (stable)
__global__ void jacobi_finish_kernel(const double* __restrict__ G, int l, const double* __restrict__ norms,
                                     double* __restrict__ U, double* __restrict__ S) {
  const int col = blockIdx.x;
  const double nc = norms[col];
  int rank = 0;
  for (int j = 0; j < l; ++j) {
    const double nj = norms[j];
    if (nj > nc || (nj == nc && j < col)) ++rank;
  }
  const double inv = (nc > 0.0) ? 1.0 / nc : 0.0;
  for (int i = threadIdx.x; i < l; i += blockDim.x) U[i + (int64_t)rank * l] = G[i + (int64_t)col * l] * inv;
  if (threadIdx.x == 0) S[rank] = nc;
}

template <int SVD_W>
static int svd_small_impl(hipStream_t st, double* G, int l, double* U, double* S, const SvdWork& w) {
  constexpr int SVD_C = 2 * SVD_W;
  int lp = l;   // column stride (doubles): lp % 32 == 16 spreads the quarter waves of a 32-lane group over both bank halves
  while ((lp & 31) != 16) ++lp;
  int nblk = (l + SVD_W - 1) / SVD_W;
  if (nblk < 2) nblk = 2;
  if (nblk & 1) ++nblk;
  const size_t shmem = (size_t)SVD_C * lp * sizeof(double) + 16;
  static std::atomic<uint64_t> attr_mask{0};
  if (first_use_on_this_device(attr_mask))
    (void)hipFuncSetAttribute((const void*)jacobi_block_kernel<SVD_W>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024 - 64);
  const double tol = sqrt((double)l) * DBL_EPSILON;
  const double tol2 = tol * tol;
  const int max_sweeps = 40;
  int sweeps = 0;
  // Activity-driven sweeps (16-column blocks, i.e. l <= 600; GSI_SVD_PLAIN=1: the plain loop below, A/B): after every sweep
  // one small launch flags the block pairs that still hold a rotatable column pair.  None: converged (no sweep is spent on
  // finding that out).  Few: the next sweep visits only those, packed greedily into rounds of disjoint block pairs.
  const int npairs = nblk * (nblk + 1) / 2;
  static const bool plain = (getenv("GSI_SVD_PLAIN") != nullptr);
  if constexpr (SVD_W == 16) {
    if (!plain && nblk > 2 && w.pairs != nullptr && npairs + 3 * npairs <= SVD_SCHED_INTS) {
      static std::atomic<uint64_t> attr_mask2{0};
      if (first_use_on_this_device(attr_mask2))
        (void)hipFuncSetAttribute((const void*)jacobi_activity_kernel<SVD_W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
      std::vector<int32_t> flags((size_t)npairs, 1), sched;
      int32_t* d_flags = w.pairs;
      int32_t* d_sched = w.pairs + npairs;
      bool sparse = false;
      bool converged = false;
      for (; sweeps < max_sweeps;) {
        if (!sparse) {
          for (int r = 0; r < nblk - 1; ++r)
            hipLaunchKernelGGL(jacobi_block_kernel<SVD_W>, dim3(nblk / 2), dim3(SVD_THREADS), shmem, st, G, l, lp, nblk,
                               r, tol2, w.rotcount, 1, r == 0 ? 0 : 1, (const int32_t*)nullptr);
        } else {
          // active cross pairs -> rounds of disjoint block pairs (greedy); a block whose own (diagonal) pair is active gets
          // its intra-block sweep with the first entry it appears in (cross_only = 0 sweeps both blocks and the cross pairs)
          std::vector<char> diag((size_t)nblk, 0);
          std::vector<std::array<int32_t, 3>> edges;
          for (int ba = 0, pi = 0; ba < nblk; ++ba)
            for (int bb = ba; bb < nblk; ++bb, ++pi) {
              if (!flags[(size_t)pi]) continue;
              if (ba == bb) diag[(size_t)ba] = 1;
              else edges.push_back({ba, bb, 1});
            }
          std::vector<char> touched((size_t)nblk, 0);
          for (auto& e : edges) { touched[(size_t)e[0]] = 1; touched[(size_t)e[1]] = 1; }
          for (int b = 0; b < nblk; ++b)
            if (diag[(size_t)b] && !touched[(size_t)b]) {        // a block that is only active within itself: any partner
              int partner = -1;
              for (int c = 0; c < nblk && partner < 0; ++c)
                if (c != b && diag[(size_t)c] && !touched[(size_t)c]) partner = c;
              if (partner < 0) partner = (b + 1) % nblk;
              edges.push_back({std::min(b, partner), std::max(b, partner), 1});
              touched[(size_t)b] = 1; touched[(size_t)partner] = 1;
            }
          std::vector<std::vector<std::array<int32_t, 3>>> rounds;
          std::vector<std::vector<char>> used;
          for (auto& e : edges) {
            size_t r = 0;
            for (; r < rounds.size(); ++r)
              if (!used[r][(size_t)e[0]] && !used[r][(size_t)e[1]]) break;
            if (r == rounds.size()) { rounds.emplace_back(); used.emplace_back((size_t)nblk, 0); }
            if (diag[(size_t)e[0]] || diag[(size_t)e[1]]) { e[2] = 0; diag[(size_t)e[0]] = 0; diag[(size_t)e[1]] = 0; }
            rounds[r].push_back(e);
            used[r][(size_t)e[0]] = 1; used[r][(size_t)e[1]] = 1;
          }
          sched.clear();
          for (auto& rd : rounds) for (auto& e : rd) { sched.push_back(e[0]); sched.push_back(e[1]); sched.push_back(e[2]); }
          if ((int)sched.size() > 3 * npairs) throw Error(GSI_ERR_INTERNAL, "svd_small: schedule overflow");
          hipMemcpyAsync(d_sched, sched.data(), sizeof(int32_t) * sched.size(), hipMemcpyHostToDevice, st);
          hipStreamSynchronize(st);          // (pageable source: the copy must be over before `sched` is rebuilt)
          size_t off = 0;
          for (auto& rd : rounds) {
            hipLaunchKernelGGL(jacobi_block_kernel<SVD_W>, dim3((unsigned)rd.size()), dim3(SVD_THREADS), shmem, st, G, l, lp, nblk,
                               0, tol2, w.rotcount, 1, 1, (const int32_t*)(d_sched + off));
            off += 3 * rd.size();
          }
        }
        ++sweeps;
        if (sweeps < 4 && l >= 128) continue;      // nobody converges in three sweeps at these widths: no look, no host round trip
        // The flag threshold is 4 tol, the rotation threshold tol: tol = sqrt(l) eps is the rounding level of the inner
        // products themselves, so which of the pairs near tol "still need a rotation" depends on the order of summation
        // (this kernel's differs from the quarter waves' of the sweep; with the same threshold a converged matrix kept a
        // handful of flagged pairs for ever).  A pair above 4 tol here is above tol there and gets rotated: progress; what
        // may be left unrotated is coupled below 4 sqrt(l) eps -- second order in the singular values.
        // (its own LDS stride: odd, so that the 16 columns a wave's lanes walk sit in 16 different banks -- with the sweep
        // kernel's stride, lp % 32 == 16, they collide eight ways: 53 us per launch)
        const int lpa = l | 1;
        hipLaunchKernelGGL(jacobi_activity_kernel<SVD_W>, dim3((unsigned)npairs), dim3(SVD_W * SVD_W),
                           (size_t)2 * SVD_W * lpa * sizeof(double), st, G, l, lpa, nblk, 16.0 * tol2, d_flags);
        hipMemcpyAsync(flags.data(), d_flags, sizeof(int32_t) * (size_t)npairs, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        int active = 0;
        for (int32_t f : flags) active += f ? 1 : 0;
        if (active == 0) { converged = true; break; }
        sparse = (2 * active <= npairs);     // at least half of the pairs clean: a schedule beats the full tournament
      }
      hipLaunchKernelGGL(jacobi_norms_kernel, dim3((l + 3) / 4), dim3(256), 0, st, G, l, w.norms);
      hipLaunchKernelGGL(jacobi_finish_kernel, dim3(l), dim3(64), 0, st, G, l, w.norms, U, S);
      return converged ? sweeps : -sweeps;   // negative: the sweep cap was reached with rotatable pairs left (the caller counts it)
    }
  }
  bool plain_converged = false;
  for (; sweeps < max_sweeps; ++sweeps) {
    hipMemsetAsync(w.rotcount, 0, sizeof(int32_t), st);
    if (nblk == 2) {
      hipLaunchKernelGGL(jacobi_block_kernel<SVD_W>, dim3(1), dim3(SVD_THREADS), shmem, st, G, l, lp, nblk, 0, tol2,
                         w.rotcount, 2, 0, (const int32_t*)nullptr);
    } else {
      for (int r = 0; r < nblk - 1; ++r)
        hipLaunchKernelGGL(jacobi_block_kernel<SVD_W>, dim3(nblk / 2), dim3(SVD_THREADS), shmem, st, G, l, lp, nblk,
                           r, tol2, w.rotcount, 1, r == 0 ? 0 : 1, (const int32_t*)nullptr);
    }
    int32_t rot = 0;
    hipMemcpyAsync(&rot, w.rotcount, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    if (rot == 0) { ++sweeps; plain_converged = true; break; }
  }
  hipLaunchKernelGGL(jacobi_norms_kernel, dim3((l + 3) / 4), dim3(256), 0, st, G, l, w.norms);
  hipLaunchKernelGGL(jacobi_finish_kernel, dim3(l), dim3(64), 0, st, G, l, w.norms, U, S);
  return plain_converged ? sweeps : -sweeps;
}

// 32 resident columns per workgroup up to l = 600, then 16 / 8 / 4 as the columns get longer
// (LDS: columns x (l padded) x 8 B <= 160 KB); l <= SVD_MAX_L = 5000.
int svd_small(hipStream_t st, double* G, int64_t l64, double* U, double* S, const SvdWork& w) {
  const int l = (int)l64;
  static const int forced = getenv("GSI_SVD_W") ? atoi(getenv("GSI_SVD_W")) : 0;      // A/B knob
  if (forced == 16 && l <= 600) return svd_small_impl<16>(st, G, l, U, S, w);
  if (forced == 8 && l <= 1200) return svd_small_impl<8>(st, G, l, U, S, w);
  if (forced == 4 && l <= 2500) return svd_small_impl<4>(st, G, l, U, S, w);
  if (l <= 600) return svd_small_impl<16>(st, G, l, U, S, w);
  if (l <= 1200) return svd_small_impl<8>(st, G, l, U, S, w);
  if (l <= 2500) return svd_small_impl<4>(st, G, l, U, S, w);
  return svd_small_impl<2>(st, G, l, U, S, w);
}

}}  // namespace gsi::hipk
