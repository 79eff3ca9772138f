// gemm_f64.hip -- the contraction kernels of the randomized range finder on gfx950:
//     NN:  C(M x L) = alpha * A(M x K)  * B(K x L) + beta*C      (Y = A*Omega, Q = A*L;
//                                                              RandMatFact.jl:55,70)
//     TN:  C(M x L) = alpha * A(K x M)' * B(K x L) + beta*C      (Q = A'*Q, B' = A'*Q;
//                                                              RandMatFact.jl:67,85)
// Column-major fp64 throughout (Julia Matrix{Float64}).  L = sketch width (<= a few
// hundred), M and K up to 10^5..10^6: "tall-skinny times huge".
//
// Design (MI355X-first, not a BLAS port):
//  * one workgroup owns 128 rows of C and ALL L columns (in chunks of NT*16 <= 160), so
//    every element of the big operand A is read from HBM exactly once per pass; the small
//    operand B is re-read per workgroup but lives in L2 / Infinity Cache.
//  * v_mfma_f64_16x16x4_f64 (64 cycles per instruction per SIMD, 77.5 TFLOP/s measured
//    ceiling), operands swapped (MFMA-A <- B', MFMA-B <- A') so that the accumulator's lane
//    index runs along C's ROWS: each 16-lane group stores 128 contiguous bytes of a column
//    of C, and the A fragment is a contiguous 16-double LDS read.
//  * arithmetic intensity per A byte is L/4 flop/B >> 9.8 (ridge) for L >= 40: MFMA-bound.
//    One 8-wave workgroup per CU (two waves per SIMD, so one wave's LDS/VMEM chores are covered
//    by its partner's MFMAs): wave tile 32 rows x (NT/2 * 16) columns = NT accumulator tiles
//    (80 VGPRs at NT = 10), 4 row groups x 2 column halves.  Per k4 step a wave reads
//    2 + NT/2 fragments for NT MFMAs (0.7 LDS reads per MFMA), and the B (X) tile is fetched
//    once per 128 rows (measured: the per-workgroup re-read of X through the vector memory
//    pipe is what limits a 64-row design).
//  * software pipeline, four stages deep: tiles t+3, t+2 in flight HBM -> registers (two register
//    sets; one tile of lead was measured to expose ~2 us of memory latency per tile), tile t+1
//    registers -> LDS (other buffer), tile t LDS -> fragments -> MFMA.  Fragments are
//    reloaded in place one k4 step ahead; the single barrier per tile sits BEFORE the last k4
//    step's MFMAs, so the barrier wait and the first fragment reads of the next tile hide behind
//    1280 cycles of matrix work.  Everything fits 256 VGPRs, so hipcc keeps the accumulators in
//    VGPR-form MFMAs (with more it switches to AGPR accumulators and copies all of them
//    AGPR<->VGPR around the loop back edge).
//  * LDS: two buffers of (A tile + B tile) = 2 x ~78 KB of the CU's 160 KB; images padded so
//    fragment reads are conflict-free for ds_read_b64 (bank = (addr/4) mod 64):
//        B tile  bs[c][k], row stride 34 doubles   A tile (NN) as[k][r], row stride 144
//        A tile (TN) at[r][k], row stride 34
//  * loads are `global_load_dwordx2 v, v_off, s[base]`: one uniform 64-bit base per operand
//    in SGPRs + per-thread 32-bit byte offsets (per-load 64-bit bases overflow the SGPR file
//    and end up as v_readlane/v_writelane chains between the MFMAs).
//  * split-K over grid.y with per-split slabs + a fixed-order reduction kernel when the
//    grid would not fill the chip (row shards on 8 GPUs, small M): deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <algorithm>
#include "hip_common.hpp"
#include "pointcov.hpp"
#include "pointcov_gen.hpp"

namespace gsi { namespace hipk {

#define GSI_GEMM_TILE_COUNTERS_32 1
#include "gemm_f64_kernel.inc.hpp"
// the table-generated operand's instantiations live in gemm_f64_gen1.hip (the same template, compiled with the 64-bit tile
// counters it was tuned with: see the include's header)
void gemm_dispatch_gen1(int nt, dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double* C, int64_t ldc, double alpha, double beta, double* slabs,
                        int64_t kchunk, int nchunks_x, int wide, int xmode, int tri, const GenA& gen, int64_t nitems);

// C = alpha * sum_s slab[s] + beta*C, fixed summation order (deterministic)
__global__ void splitk_reduce_kernel(int64_t M, int64_t L, int nsplit, const double* __restrict__ slabs,
                                     double* __restrict__ C, int64_t ldc, double alpha, double beta) {
  const int64_t total = M * L;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += slabs[(int64_t)sp * total + idx];
    const int64_t r = idx % M, c = idx / M;
    double v = alpha * s;
    if (beta != 0.0) v += beta * C[r + c * ldc];
    C[r + c * ldc] = v;
  }
}

// Number of K splits for a grid of `nwg` workgroups.  One workgroup per CU, 256 CUs: a grid of g equal
// workgroups takes ceil(g / 256) rounds, so 181 row blocks (or 392: 2 rounds for 1.53 rounds of work) leave a
// large part of the chip idle.  Splitting K by s makes the rounds s times shorter: pick the s that minimises
// ceil(nwg * s / 256) / s, charged 0.4 % per split for the slab write + reduce, and only when it buys > 3 %.
int gemm_choose_split(int64_t nwg, int64_t K) {
  constexpr int64_t CUS = 256;
  static const bool legacy = getenv("GSI_GEMM_SPLIT_LEGACY") != nullptr;   // A/B knob: fill-the-chip rule only
  static const int forced = getenv("GSI_GEMM_FORCE_SPLIT") ? atoi(getenv("GSI_GEMM_FORCE_SPLIT")) : 0;   // experiments
  if (forced > 0 && nwg >= 256 && K >= 64 * BK) return forced;
  if (legacy) {
    if (nwg >= 192 || K < 8 * BK) return 1;
    int64_t want = (CUS + nwg - 1) / nwg, cap = K / (4 * BK);
    if (want > cap) want = cap;
    return (int)(want < 1 ? 1 : (want > 64 ? 64 : want));
  }
  if (K < 8 * BK || nwg >= 16 * CUS) return 1;
  int64_t maxs = K / (4 * BK);
  // up to 256 ways: a Gram matrix of a narrow, very tall panel (l = 48 at n = 1.3e8: ONE output tile) is an HBM stream that
  // 64 workgroups cannot pull -- 23 ms per 6.4 GB at n = 1.7e7 where 256 workgroups take ~3 (the slabs stay tiny: s M L)
  if (maxs > 256) maxs = 256;
  int best = 1;
  double best_cost = (double)((nwg + CUS - 1) / CUS);
  for (int64_t s = 2; s <= maxs; ++s) {
    const double cost = (double)((nwg * s + CUS - 1) / CUS) / (double)s * (1.0 + 0.004 * (double)s);
    if (cost < 0.97 * best_cost) { best = (int)s; best_cost = cost; }
  }
  return best;
}

size_t gemm_workspace_doubles(int64_t M, int64_t L, int64_t K) {
  const int64_t nchunk_cols = (L + NTMAX * 16 - 1) / (NTMAX * 16);
  const int64_t nwg = ((M + BMT - 1) / BMT) * nchunk_cols;
  const int ns = gemm_choose_split(nwg, K);
  return ns > 1 ? (size_t)ns * (size_t)M * (size_t)L : 0;
}
// the symmetric product C = A'A (l x l, K = m): fewer active tiles, so possibly more splits; bound by the chooser's cap
size_t gemm_syrk_workspace_doubles(int64_t l, int64_t m) {
  if (m < 8 * BK) return 0;
  // the split the launcher will actually choose for the tiles that touch the upper triangle (256 l^2 regardless
  // of the split was 8 GB at l = 2000)
  const int64_t tiles = (l + 15) / 16, nchunks = (tiles + NTMAX - 1) / NTMAX, nt = (tiles + nchunks - 1) / nchunks;
  const int64_t rowblocks = (l + BMT - 1) / BMT;
  int64_t active = 0;
  for (int64_t rb = 0; rb < rowblocks; ++rb)
    for (int64_t cb = 0; cb < nchunks; ++cb)
      if (cb >= (rb * BMT) / (nt * 16)) ++active;
  const int ns = gemm_choose_split(active, m);
  return ns > 1 ? (size_t)ns * (size_t)l * (size_t)l : 0;
}

// number of 16-column tiles per workgroup pass.  (128-column chunks for the symmetric product, so that chunk and
// row-block boundaries coincide, were measured SLOWER at l = 320: 320 is not a multiple of 128, the product then runs
// through the irregular-X instantiation: 6.3 vs 5.3 ms.)
static int gemm_chunk_tiles(int) { return NTMAX; }

static void gemm_launch(hipStream_t st, bool transA, const GenA* gen, int64_t M, int64_t L, int64_t K, double alpha,
                        const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                        int64_t ldc, double* ws, int tri = 0, int gen_mode = 1, int force_nsplit = 0) {
  if (M <= 0 || L <= 0) return;
  // The kernel addresses a tile with one uniform 64-bit base per operand plus per-thread byte offsets spanning up to
  // 160 columns of B, 128 rows of a transposed A or 32 columns of a plain A: 32 bits reach panels of ~3.3 million
  // rows; beyond that the 64-bit-offset instantiation (XMODE 2) takes over.
  const uint64_t lim = (uint64_t)1 << 32;
  const uint64_t bspan = 8ull * ((uint64_t)(NTMAX * 16) * (uint64_t)ldb + BK);
  const uint64_t aspan = (gen != nullptr) ? 0ull : 8ull * ((uint64_t)(transA ? BMT : BK) * (uint64_t)lda + BMT);
  const bool big = bspan >= lim || aspan >= lim;
  // columns are processed in chunks of nt*16 <= 160; balance the chunks
  const int64_t tiles = (L + 15) / 16;
  const int ntmax = gemm_chunk_tiles(tri);
  const int64_t nchunks = (tiles + ntmax - 1) / ntmax;
  const int nt = (int)((tiles + nchunks - 1) / nchunks);
  const int64_t rowblocks = (M + BMT - 1) / BMT;
  int64_t active = rowblocks * nchunks;
  if (tri == 1) {                                 // tiles with a column at or right of the row block's first row
    active = 0;
    for (int64_t rb = 0; rb < rowblocks; ++rb)
      for (int64_t cb = 0; cb < nchunks; ++cb)
        if (cb >= (rb * BMT) / ((int64_t)nt * 16)) ++active;
  }
  const int nsplit = (force_nsplit > 0) ? force_nsplit : ((K > 0) ? gemm_choose_split(active, K) : 1);
  int64_t kchunk = (K + nsplit - 1) / nsplit;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  if (kchunk == 0) kchunk = BK;
  const int ns_eff = (K > 0) ? (int)((K + kchunk - 1) / kchunk) : 1;
  dim3 grid((unsigned)active, (unsigned)ns_eff, 1);
  double* slabs = (ns_eff > 1) ? ws : nullptr;
  // persistent mode (see the kernel): short reductions with several rounds of output tiles, stored operand, regular X
  static const int ncus = [] { int dev = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; return n; }();
  static const bool persist_on = !(getenv("GSI_GEMM_PERSIST") != nullptr && getenv("GSI_GEMM_PERSIST")[0] == '0');
  int64_t nitems = 0;
  // 16-byte loads need 16-B aligned bases and even leading dimensions (sub-panel views often are not)
  const bool a_ok = ((uintptr_t)A & 15) == 0 && (lda & 1) == 0;
  const bool b_ok = ((uintptr_t)B & 15) == 0 && (ldb & 1) == 0;
  // regular kernel: everything 16-byte loadable and full 16-column tiles; irregular-X kernel: the operator is,
  // X is ragged or only 8-byte aligned; otherwise (operator itself unaligned) the element-wise path of the regular one
  const bool irregular_x = a_ok && (!b_ok || L % ((int64_t)nt * 16) != 0);
  const int wide = a_ok ? 1 : 0;
  const int xmode = big ? 2 : (irregular_x ? 1 : 0);
  if (persist_on && gen == nullptr && xmode == 0 && tri == 0 && ns_eff == 1 && (ncus & 7) == 0 && active >= 2 * (int64_t)ncus &&
      K <= 128 * BK) {
    nitems = active;
    grid.x = (unsigned)ncus;
  }
  const GenA none = {nullptr, 1, 0, 0, 0, 0.0, 0.0, 0.0};
  if (gen != nullptr && gen_mode == 2)
    launch_dispatch<false, 2>(nt, grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, (int)nchunks, wide, xmode, tri, *gen, nitems);
  else if (gen != nullptr)
    gemm_dispatch_gen1(nt, grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, (int)nchunks, wide, xmode, tri, *gen, nitems);   // gemm_f64_gen1.hip
  else if (transA)
    launch_dispatch<true, 0>(nt, grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, (int)nchunks, wide, xmode, tri, none, nitems);
  else
    launch_dispatch<false, 0>(nt, grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk, (int)nchunks, wide, xmode, tri, none, nitems);
  if (ns_eff > 1) {
    const int64_t total = M * L;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, M, L, ns_eff, slabs, C, ldc,
                       alpha, beta);
  }
}

void gemm_splitk_reduce(hipStream_t st, int64_t M, int64_t L, int nsplit, const double* slabs, double* C, int64_t ldc) {
  const int64_t total = M * L;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, M, L, nsplit, slabs, C, ldc, 1.0, 0.0);
}

// Host launchers. `ws` must hold gemm_workspace_doubles(M, L, K) doubles (or be null if 0).
void gemm_f64(hipStream_t st, bool transA, int64_t M, int64_t L, int64_t K, double alpha, const double* A,
              int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, double* ws) {
  gemm_launch(st, transA, nullptr, M, L, K, alpha, A, lda, B, ldb, beta, C, ldc, ws);
}

// The K split gemm_f64 (NN or TN, plain operands) uses for an M x L x K product, and rows [r0, r0 + mb) of that product
// with the SAME split (per-element reduction order unchanged: the row blocks put together are bit-identical to the one
// launch).  What the streamed upload of a dense operator needs: Y = A * Omega block by block as the rows land.
int gemm_split_for(int64_t M, int64_t L, int64_t K) {
  if (M <= 0 || L <= 0 || K <= 0) return 1;
  const int64_t tiles = (L + 15) / 16, nchunks = (tiles + NTMAX - 1) / NTMAX;
  const int64_t active = ((M + BMT - 1) / BMT) * nchunks;
  return gemm_choose_split(active, K);          // the chooser's split, before the rounding of the K chunk to whole tiles
}
size_t gemm_rowblock_workspace_doubles(int64_t M_full, int64_t mb, int64_t L, int64_t K) {
  const int ns = gemm_split_for(M_full, L, K);    // (the launcher may end up with fewer, never more, slabs)
  return ns > 1 ? (size_t)ns * (size_t)mb * (size_t)L : 0;
}
void gemm_f64_nn_rowblock(hipStream_t st, int64_t M_full, int64_t r0, int64_t mb, int64_t L, int64_t K, const double* A,
                          int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double* ws) {
  gemm_launch(st, false, nullptr, mb, L, K, 1.0, A + r0, lda, B, ldb, 0.0, C + r0, ldc, ws, 0, 1, gemm_split_for(M_full, L, K));
}

// C (l x l, ld ldc) = A'A for A m x l: only tiles that touch the upper triangle are computed, the rest of C is
// unspecified (cholqr.hip mirrors the upper triangle afterwards).  ws: gemm_syrk_workspace_doubles(l, m).
void gemm_f64_syrk_upper(hipStream_t st, int64_t l, int64_t m, const double* A, int64_t lda, double* C, int64_t ldc,
                         double* ws) {
  gemm_launch(st, true, nullptr, l, l, m, 1.0, A, lda, A, lda, 0.0, C, ldc, ws, 1);
}
// C (M x L) = A (M x K) * B with B (K x L) upper triangular: entries below the diagonal must be ZERO (those beyond
// a column chunk's last column are not read, the ones inside the chunk are)
void gemm_f64_trmm_upper(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* A, int64_t lda, const double* B,
                         int64_t ldb, double* C, int64_t ldc, double* ws) {
  gemm_launch(st, false, nullptr, M, L, K, 1.0, A, lda, B, ldb, 0.0, C, ldc, ws, 2);
}

// C (M x L) = G * B with G(i, k) = tab[|x_i - x_k| * ny + |y_i - y_k|], i = roff + row, k = koff + reduction index;
// tab = the nx * ny kernel table in device memory.  The operand G is generated in registers, never stored.
void gemm_f64_gridcov(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* tab, int64_t nx, int64_t ny,
                      int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C, int64_t ldc, double* ws) {
  (void)nx;
  GenA g = {tab, (int32_t)ny, 0, roff, koff, 0.0, 0.0, 0.0};
  // A / lda only feed the 16-byte-load test for the stored operand: pass aligned dummies
  gemm_launch(st, false, &g, M, L, K, 1.0, nullptr, 2, B, ldb, 0.0, C, ldc, ws);
}

// C (M x L) = G * B with G(i, k) = sigma2 kfun(|p_i - p_k| / ell) (+ nugget if i == k), i = roff + row, k = koff + reduction
// index; pts4 = the points as 32-byte records (x, y, z, 0), npts of them, in device memory.  G is generated in the tile
// loader, never stored (GEN 2 above).
void gemm_f64_pointcov(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* pts4, int64_t npts, int d, int kind,
                       double sigma2, double nugget, int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C,
                       int64_t ldc, double* ws, double* xpack) {
  if (gemm_f64_pointcov_wide(st, M, L, K, pts4, npts, d, kind, sigma2, nugget, roff, koff, B, ldb, C, ldc, ws, xpack)) return;
  GenA g = {pts4, (int32_t)npts, kind, roff, koff, 0.0, sigma2, nugget, (int32_t)d};
  gemm_launch(st, false, &g, M, L, K, 1.0, pts4, 2, B, ldb, 0.0, C, ldc, ws, 0, 2);      // A = the points (16-byte aligned records)
}
// the factor the points are scaled by, so that the squared distance of two records is the squared ARGUMENT of the kernel's
// exponential: c1 / ell (c1 = 1, sqrt 3, sqrt 5 for exponential, Matern 3/2, 5/2), 1 / (ell sqrt 2) for the Gaussian
double pointcov_point_scale(int kind, double inv_ell) {
  switch (kind) {
    case pointcov::GAUSSIAN: return inv_ell * 0.70710678118654752440;
    case pointcov::MATERN32: return inv_ell * 1.7320508075688772;
    case pointcov::MATERN52: return inv_ell * 2.23606797749979;
    default: return inv_ell;
  }
}
// pts (d x n, point i = column i) -> 32-byte records scale * (x - x_0, y - y_0, z - z_0, 0).  The translation by the first
// point makes the scaling translation-invariant: the kernels subtract SCALED coordinates, and scaling raw
// coordinates rounds at the size of their common offset -- UTM-like points (x ~ 5e5, y ~ 4.6e6, ell = 20) lost 5 digits of
// every entry (1.5e-11 against 1e-16 for subtract-then-scale).  After the translation the rounding is relative to the EXTENT
// of the point set, as in pointcov::kernel's subtract-then-scale; it costs the loader nothing.
__global__ __launch_bounds__(256) void pointcov_pad_kernel(const double* __restrict__ pts, int d, int64_t n, double scale, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int a = 0; a < 4; ++a) out[4 * i + a] = (a < d) ? scale * (pts[i * d + a] - pts[a]) : 0.0;
}
void pointcov_pad_points(hipStream_t st, const double* pts, int d, int64_t n, double scale, double* out4) {
  hipLaunchKernelGGL(pointcov_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, pts, d, n, scale, out4);
}

}}  // namespace gsi::hipk
