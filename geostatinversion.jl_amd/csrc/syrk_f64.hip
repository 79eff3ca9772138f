// syrk_f64.hip -- G = Y'Y for a tall panel (m x l, l <= 320): the Gram matrix of CholeskyQR (cholqr.hip), upper
// triangle computed once, mirrored on the way out.
//
// Why not the general contraction kernel (gemm_f64.hip, tri = 1): its 128 x 160 output tiles cover a 320 x 320 upper
// triangle with 5 tiles = 102 400 MFMA outputs where 51 360 are needed, and every tile streams its two operand column
// blocks from HBM / L2 again (7.7 GB for a 2.56 GB panel): 4.2 ms per Gram matrix at n = 10^6, l = 320, a quarter of
// the step's QR time.  Here the panel is read ONCE per half: a workgroup owns a slab of rows and HALF of the upper
// 16 x 16 blocks (105 of 210 at l = 320 -- block rows 0..5 are exactly half), all of them accumulating in registers
// (13-14 blocks = 112 accumulator registers per wave, 8 waves), so the kernel is bound by the matrix pipe at the
// minimum flop count: 210 blocks x 256 x 2 x m.
//   LDS:   chunks of 16 panel rows, [k][column] with an odd row stride (conflict-free for the k-fastest stores and the
//          column-fastest fragment reads), double buffered; the next chunk travels HBM -> registers while this one feeds
//          the MFMAs.
//   MFMA:  v_mfma_f64_16x16x4_f64; the A and B fragments of a symmetric product are the SAME registers (fragment b =
//          rows k..k+3 of columns 16 b .. 16 b + 15), loaded once per k-step for the block rows and columns the wave needs.
//   Out:   per-slab partial blocks in the fragment layout, summed over slabs in a fixed order (deterministic, no atomics)
//          by sy_reduce_kernel, which also writes the mirrored lower triangle.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include "hip_common.hpp"

namespace gsi { namespace hipk {

typedef double sy_double4 __attribute__((ext_vector_type(4)));

constexpr int SY_KC = 16;          // panel rows per chunk
constexpr int SY_WAVES = 8;
constexpr int SY_THREADS = 64 * SY_WAVES;
constexpr int SY_SLABS = 128;      // x 2 halves = one workgroup per CU

// the upper 16 x 16 blocks in row-major order: t -> (bi, bj >= bi)
template <int NB>
struct SyGeom {
  static constexpr int NBLK = NB * (NB + 1) / 2;
  static constexpr int LDW = 16 * NB + 1;                   // odd row stride of the LDS chunk
  static constexpr int row_start(int bi) { return bi * NB - bi * (bi - 1) / 2; }
  static constexpr int split_row() { int r = 0; while (row_start(r) * 2 < NBLK) ++r; return r; }   // half 1 starts here
  static constexpr int bi_of(int t) { int bi = 0; while (row_start(bi + 1) <= t) ++bi; return bi; }
  static constexpr int bj_of(int t) { return bi_of(t) + (t - row_start(bi_of(t))); }
  static constexpr int half_begin(int h) { return h ? row_start(split_row()) : 0; }
  static constexpr int half_end(int h) { return h ? NBLK : row_start(split_row()); }
  static constexpr int per_wave(int h) { return (half_end(h) - half_begin(h) + SY_WAVES - 1) / SY_WAVES; }
  static constexpr int wave_begin(int h, int w) {
    const int b = half_begin(h) + w * per_wave(h);
    return b < half_end(h) ? b : half_end(h);
  }
  static constexpr int wave_end(int h, int w) {
    const int e = half_begin(h) + (w + 1) * per_wave(h);
    return e < half_end(h) ? e : half_end(h);
  }
  static constexpr bool needs(int h, int w, int b) {       // does wave w of half h touch fragment b?
    for (int t = wave_begin(h, w); t < wave_end(h, w); ++t)
      if (bi_of(t) == b || bj_of(t) == b) return true;
    return false;
  }
  static constexpr int col_lo(int h) { return h ? 16 * split_row() : 0; }   // half 1 never reads the first columns
};

// Geometry as template arguments: everything below folds at compile time (written as constexpr calls inside unrolled
// loops the front end left the searches in bi_of / needs to the optimiser: a 227 000-line kernel that ran 100x slow).
template <int NB, int H, int W, int B>
__device__ __forceinline__ void sy_ld(double (&F)[NB], const double* p) {
  if constexpr (SyGeom<NB>::needs(H, W, B)) F[B] = p[16 * B];
}
template <int NB, int H, int W, int... B>
__device__ __forceinline__ void sy_ld_all(double (&F)[NB], const double* p, std::integer_sequence<int, B...>) {
  (sy_ld<NB, H, W, B>(F, p), ...);
}
template <int NB, int H, int W, int B>
__device__ __forceinline__ void sy_cp(double (&F)[NB], const double (&Fn)[NB]) {
  if constexpr (SyGeom<NB>::needs(H, W, B)) F[B] = Fn[B];
}
template <int NB, int H, int W, int... B>
__device__ __forceinline__ void sy_cp_all(double (&F)[NB], const double (&Fn)[NB], std::integer_sequence<int, B...>) {
  (sy_cp<NB, H, W, B>(F, Fn), ...);
}
template <int NB, int H, int W, int CNT, int I>
__device__ __forceinline__ void sy_mma(sy_double4 (&acc)[CNT], const double (&F)[NB]) {
  constexpr int t = SyGeom<NB>::wave_begin(H, W) + I;
  constexpr int bi = SyGeom<NB>::bi_of(t), bj = SyGeom<NB>::bj_of(t);
  acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[bj], F[bi], acc[I], 0, 0, 0);
}
template <int NB, int H, int W, int CNT, int... I>
__device__ __forceinline__ void sy_mma_all(sy_double4 (&acc)[CNT], const double (&F)[NB], std::integer_sequence<int, I...>) {
  (sy_mma<NB, H, W, CNT, I>(acc, F), ...);
}

// FAST (round 4): l == 16 NB and every chunk of every slab has its 16 rows (m % 16 == 0) -- the staging loads are a uniform
// 64-bit base stepped by uniform strides plus a per-thread 32-bit offset computed once, no clamps, no selects, and the
// fragments alternate between two register sets instead of being copied (F <- Fn): round 3's body issued ~400 VALU
// instructions per 50 MFMAs, and on gfx950 an fp64 MFMA shares the vector ALU with them (tools/mfma_valu_f64_conflict.hip).
template <int NB, int HALF, int W, bool FAST>
__device__ __forceinline__ void sy_body(const double* __restrict__ Y, int64_t ld, int64_t row0, int64_t rend, int l,
                                        double* __restrict__ Pslab, double* smem) {
  using G = SyGeom<NB>;
  constexpr int T0 = G::wave_begin(HALF, W), T1 = G::wave_end(HALF, W), CNT = T1 - T0;
  constexpr int CLO = G::col_lo(HALF), NCOL = 16 * NB - CLO;
  constexpr int NLD = (NCOL * SY_KC + SY_THREADS - 1) / SY_THREADS;      // staged elements per thread and chunk
  constexpr int BUF = SY_KC * G::LDW;
  static_assert(SY_KC == 16, "the k-steps of a chunk are written out below");
  const int tid = threadIdx.x, lane = tid & 63, jl = lane & 15, kk = lane >> 4;

  sy_double4 acc[CNT > 0 ? CNT : 1];
#pragma unroll
  for (int i = 0; i < CNT; ++i) acc[i] = (sy_double4){0.0, 0.0, 0.0, 0.0};

  const int64_t nrows = rend - row0;
  const int nchunks = nrows > 0 ? (int)((nrows + SY_KC - 1) / SY_KC) : 0;
  double stage[NLD];
  // element e of a chunk: k = e % 16 (fastest: 16 lanes cover the 128 contiguous bytes of one column), column CLO + e / 16
  const int sk = tid & 15, sc = tid >> 4;
  const uint32_t y_voff = (uint32_t)(8 * ((int64_t)sk + (int64_t)sc * ld)), y_voff_h = (uint32_t)(8 * ((int64_t)sk + (int64_t)(sc & 15) * ld));
  const uint32_t s_off = (uint32_t)(sk * G::LDW + CLO + sc), s_off_h = (uint32_t)(sk * G::LDW + CLO + (sc & 15));
  const int64_t y_step = 8 * 32 * ld;
  const char* y_next = reinterpret_cast<const char*>(Y + row0 + (int64_t)CLO * ld);      // uniform; one chunk = 128 bytes further
  auto prefetch = [&](int c) {
    if constexpr (FAST) {
      (void)c;
      const char* yb = y_next;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        if (32 * i < NCOL) {                                // compile time; a last half slot re-loads its lower 16 columns
          stage[i] = *reinterpret_cast<const double*>(yb + ((32 * i + 32 > NCOL) ? y_voff_h : y_voff));
          yb += y_step;
        }
      }
      y_next += 8 * SY_KC;
    } else {
      const int64_t k0 = row0 + (int64_t)c * SY_KC;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * SY_THREADS;
        const int k = e & (SY_KC - 1);
        int col = CLO + (e >> 4);
        const bool ok = (col < l) && (k0 + k < rend) && (e < NCOL * SY_KC);
        if (col >= l) col = l - 1;
        int64_t r = k0 + k;
        if (r >= rend) r = rend - 1;
        const double v = Y[r + (int64_t)col * ld];            // unconditional load of a clamped address
        stage[i] = ok ? v : 0.0;
      }
    }
  };
  auto store = [&](int buf) {
    double* s = smem + buf * BUF;
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        if (32 * i < NCOL) s[((32 * i + 32 > NCOL) ? s_off_h : s_off) + 32 * i] = stage[i];
    } else {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = tid + i * SY_THREADS;
        if (e < NCOL * SY_KC) s[(e & (SY_KC - 1)) * G::LDW + CLO + (e >> 4)] = stage[i];
      }
    }
  };
  if (nchunks > 0) {
    prefetch(0);
    store(0);
    if (nchunks > 1) prefetch(1);
  }
  __syncthreads();
  double F0[NB], F1[NB];
  using Bs = std::make_integer_sequence<int, NB>;
  using Is = std::make_integer_sequence<int, CNT>;
  for (int c = 0; c < nchunks; ++c) {
    const double* s = smem + (c & 1) * BUF + kk * G::LDW + jl;       // fragment b of step st: s[4 st LDW + 16 b]
    sy_ld_all<NB, HALF, W>(F0, s, Bs{});
    sy_ld_all<NB, HALF, W>(F1, s + 4 * G::LDW, Bs{});
    if constexpr (CNT > 0) sy_mma_all<NB, HALF, W, (CNT > 0 ? CNT : 1)>(acc, F0, Is{});
    if (c + 1 < nchunks) {
      // the other buffer was last read in the previous iteration, behind its closing barrier
      store((c + 1) & 1);
      if (c + 2 < nchunks) prefetch(c + 2);
    }
    sy_ld_all<NB, HALF, W>(F0, s + 8 * G::LDW, Bs{});
    if constexpr (CNT > 0) sy_mma_all<NB, HALF, W, (CNT > 0 ? CNT : 1)>(acc, F1, Is{});
    sy_ld_all<NB, HALF, W>(F1, s + 12 * G::LDW, Bs{});
    if constexpr (CNT > 0) sy_mma_all<NB, HALF, W, (CNT > 0 ? CNT : 1)>(acc, F0, Is{});
    if constexpr (CNT > 0) sy_mma_all<NB, HALF, W, (CNT > 0 ? CNT : 1)>(acc, F1, Is{});
    __syncthreads();
  }
  // lane holds D[i = kk + 4 reg][j = jl] = sum_k Y[k][16 bj + i] Y[k][16 bi + j]: stored as is, 512 contiguous bytes per register
#pragma unroll
  for (int i = 0; i < CNT; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Pslab[((int64_t)(T0 + i) * 4 + reg) * 64 + lane] = acc[i][reg];
}

template <int NB, bool FAST>
__global__ __launch_bounds__(SY_THREADS) void sy_kernel(const double* __restrict__ Y, int64_t ld, int64_t m, int l,
                                                        int64_t rows_per_slab, double* __restrict__ P) {
  extern __shared__ double sy_smem[];
  const int half = blockIdx.x & 1, slab = blockIdx.x >> 1;
  const int64_t row0 = (int64_t)slab * rows_per_slab;
  int64_t rend = row0 + rows_per_slab;
  if (rend > m) rend = m;
  double* Pslab = P + (int64_t)slab * SyGeom<NB>::NBLK * 256;
  const int wave = threadIdx.x >> 6;
#define GSI_SY_CASE(H, Wv) case (H) * SY_WAVES + (Wv): sy_body<NB, H, Wv, FAST>(Y, ld, row0, rend, l, Pslab, sy_smem); break;
  switch (half * SY_WAVES + wave) {
    GSI_SY_CASE(0, 0) GSI_SY_CASE(0, 1) GSI_SY_CASE(0, 2) GSI_SY_CASE(0, 3)
    GSI_SY_CASE(0, 4) GSI_SY_CASE(0, 5) GSI_SY_CASE(0, 6) GSI_SY_CASE(0, 7)
    GSI_SY_CASE(1, 0) GSI_SY_CASE(1, 1) GSI_SY_CASE(1, 2) GSI_SY_CASE(1, 3)
    GSI_SY_CASE(1, 4) GSI_SY_CASE(1, 5) GSI_SY_CASE(1, 6) GSI_SY_CASE(1, 7)
    default: break;
  }
#undef GSI_SY_CASE
}

// G (l x l, ld ldg) <- sum over slabs, fixed order; block t of the upper triangle per workgroup, both triangles written
template <int NB>
__global__ __launch_bounds__(256) void sy_reduce_kernel(const double* __restrict__ P, int nslab, int l, double* __restrict__ Gm,
                                                        int64_t ldg) {
  using G = SyGeom<NB>;
  const int t = blockIdx.x, tid = threadIdx.x;
  double v = 0.0;
#pragma unroll 16
  for (int s = 0; s < nslab; ++s) v += P[((int64_t)s * G::NBLK + t) * 256 + tid];      // loads in flight, sums in slab order
  int bi = 0;
  while (G::row_start(bi + 1) <= t) ++bi;
  const int bj = bi + (t - G::row_start(bi));
  const int reg = tid >> 6, lane = tid & 63, jl = lane & 15, kk = lane >> 4;
  const int row = 16 * bi + jl, col = 16 * bj + kk + 4 * reg;
  if (row <= col && col < l) {
    Gm[row + (int64_t)col * ldg] = v;
    Gm[col + (int64_t)row * ldg] = v;
  }
}

size_t syrk_upper_workspace_doubles(int64_t l, int64_t m) {
  // l < 96: the Gram matrix is HBM-bound (l / 8 flop per panel byte against a ridge of ~10) and the general kernel reads
  // the panel once where this one reads it 1.7 times (once per half)
  if (l > 320 || l < 96 || m < 4096) return 0;
  const int nb = (int)((l + 15) / 16);
  const int NB = nb <= 8 ? 8 : (nb <= 10 ? 10 : (nb <= 16 ? 16 : 20));
  return (size_t)SY_SLABS * (size_t)(NB * (NB + 1) / 2) * 256;
}

template <int NB>
static void sy_launch(hipStream_t st, int64_t l, int64_t m, const double* Y, int64_t ld, double* Gm, int64_t ldg, double* ws) {
  static std::atomic<uint64_t> attr_mask{0};
  const size_t shmem = (size_t)2 * SY_KC * SyGeom<NB>::LDW * sizeof(double);
  if (first_use_on_this_device(attr_mask)) {
    (void)hipFuncSetAttribute((const void*)sy_kernel<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)sy_kernel<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  }
  int64_t rps = (m + SY_SLABS - 1) / SY_SLABS;
  rps = (rps + SY_KC - 1) / SY_KC * SY_KC;
  // FAST: exact fit of the instantiation, whole chunks everywhere, per-thread offsets within 32 bits
  static const bool no_fast = (getenv("GSI_SY_NO_FAST") != nullptr);
  if (!no_fast && l == 16 * NB && m % SY_KC == 0 && 31 * ld + 16 < ((int64_t)1 << 29))
    hipLaunchKernelGGL((sy_kernel<NB, true>), dim3(2 * SY_SLABS), dim3(SY_THREADS), shmem, st, Y, ld, m, (int)l, rps, ws);
  else
    hipLaunchKernelGGL((sy_kernel<NB, false>), dim3(2 * SY_SLABS), dim3(SY_THREADS), shmem, st, Y, ld, m, (int)l, rps, ws);
  hipLaunchKernelGGL((sy_reduce_kernel<NB>), dim3(SyGeom<NB>::NBLK), dim3(256), 0, st, ws, SY_SLABS, (int)l, Gm, ldg);
}

// G (l x l) = Y'Y, both triangles.  Returns false (nothing done) when the shape is not this kernel's: the caller
// falls back to the general contraction.  ws: syrk_upper_workspace_doubles(l, m).
bool syrk_full_from_upper(hipStream_t st, int64_t l, int64_t m, const double* Y, int64_t ld, double* Gm, int64_t ldg, double* ws) {
  static const bool off = (getenv("GSI_NO_SYRK_KERNEL") != nullptr);
  if (off || syrk_upper_workspace_doubles(l, m) == 0) return false;
  const int nb = (int)((l + 15) / 16);
  if (nb <= 8) sy_launch<8>(st, l, m, Y, ld, Gm, ldg, ws);
  else if (nb <= 10) sy_launch<10>(st, l, m, Y, ld, Gm, ldg, ws);
  else if (nb <= 16) sy_launch<16>(st, l, m, Y, ld, Gm, ldg, ws);
  else sy_launch<20>(st, l, m, Y, ld, Gm, ldg, ws);
  return true;
}

// ---- C (m x l) = A (m x l) * X, X (l x l) upper triangular: the Y R^-1 of a CholeskyQR round ----------------------
// The general kernel runs this shape (K = l = 320: ten K tiles per 128 x 160 output tile) mostly in its prologue and
// epilogue: 3.65 ms at m = 10^6 against 1.4 ms of MFMA time for the triangular flop count.  Here a workgroup owns 128
// rows and ALL l columns (wave r: rows 16 r .. 16 r + 15, 20 accumulator blocks = 160 registers), walks the 16-row
// chunks kb of X and skips the column blocks left of the diagonal (bj < kb): exactly the 210 x 4 MFMAs per row block the
// triangle needs, every wave the same schedule.  X (0.8 MB) streams from L2 once per tile, A and C move once.
template <class Fn, int... I>
__device__ __forceinline__ void sy_for_each(Fn& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
constexpr int TR_ROWS = 128;
constexpr int TR_LDA = TR_ROWS + 1;
constexpr int TR_G = 5;            // X fragments per group

// FAST: all 128 rows of the block exist and l == 16 NB -- no clamps, no predicates, and every address is a
// uniform 64-bit base (SGPRs) plus a per-thread 32-bit byte offset that is computed ONCE: round 3's staging code spent
// ~25 integer VALU instructions per load on 64-bit index arithmetic and clamps (5179 VALU for 840 MFMAs in the <20>
// instantiation), and on gfx950 nothing of that hides behind the fp64 MFMAs -- v_mfma_f64_16x16x4_f64 holds the SIMD's
// vector ALU for all of its 64 cycles, also against the partner wave (tools/mfma_valu_f64_conflict.hip: both = sum).
// It also loaded all 10 X slots of every chunk although chunk kb only has columns >= 16 kb: skipped now.
template <int NB, bool FAST>
__global__ __launch_bounds__(SY_THREADS) void tr_kernel(const double* __restrict__ A, int64_t lda, int64_t m,
                                                        const double* __restrict__ X, int64_t ldx, int l,
                                                        double* __restrict__ C, int64_t ldc, int64_t rb0) {
  constexpr int LDW = 16 * NB + 1;
  constexpr int ABUF = SY_KC * TR_LDA, XBUF = SY_KC * LDW, BUF = ABUF + XBUF;
  constexpr int NLA = TR_ROWS * SY_KC / SY_THREADS;                      // 4
  constexpr int NLX = (16 * NB * SY_KC + SY_THREADS - 1) / SY_THREADS;   // <= 10
  extern __shared__ double sy_smem[];
  const int tid = threadIdx.x, lane = tid & 63, jl = lane & 15, kk = lane >> 4, r = tid >> 6;
  const int64_t r0 = (rb0 + (int64_t)blockIdx.x) * TR_ROWS;
  const int nkb = (l + 15) >> 4;

  sy_double4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = (sy_double4){0.0, 0.0, 0.0, 0.0};
  double sa[NLA], sx[NLX];
  // per-thread pieces of the staging addresses, fixed for the kernel: A element (row = tid % 128, k = tid / 128 + 4 i + 16 kb),
  // X element (k = tid % 16 + 16 kb, column = tid / 16 + 32 i + 16 kb)
  const int arow = tid & (TR_ROWS - 1), akq = tid >> 7, xk = tid & 15, xc = tid >> 4;
  const uint32_t a_voff = (uint32_t)(8 * ((int64_t)arow + (int64_t)akq * lda));      // FAST: fits (launcher checks 3 lda + 128 < 2^29)
  const uint32_t x_voff = (uint32_t)(8 * ((int64_t)xk + (int64_t)xc * ldx));
  const char* const Ab = reinterpret_cast<const char*>(A + r0);
  const char* const Xb = reinterpret_cast<const char*>(X);
  const uint32_t as_off = (uint32_t)(akq * TR_LDA + arow), xs_off = (uint32_t)(xk * LDW + xc);
  // FAST also means l == 16 NB exactly (the launcher checks): with the chunk index a template argument, which X slots a
  // chunk has (32 columns each; the last one of an odd chunk count is half a slot) is known at compile time -- no branch, no
  // select.  The half slot's upper 16 x 16 lanes re-load and re-store the lower half's elements (same address, same value).
  const uint32_t x_voff_h = (uint32_t)(8 * ((int64_t)xk + (int64_t)(xc & 15) * ldx));
  const uint32_t xs_off_h = (uint32_t)(xk * LDW + (xc & 15));
  const int64_t a_step = 8 * 4 * lda, x_step = 8 * 32 * ldx;              // uniform byte strides between the slots of a chunk
  const char* a_next = Ab;                                                // chunk kb: A + r0 + 16 kb lda   (prefetch runs in order)
  const char* x_next = Xb;                                                //           X(16 kb, 16 kb)
  auto prefetch = [&](auto KBc) {
    constexpr int kb = decltype(KBc)::value;
    constexpr int k0 = 16 * kb;
    if constexpr (FAST) {
      constexpr int ncol = 16 * NB - k0;                                   // columns k0 .. l - 1
      const char* ab = a_next;
#pragma unroll
      for (int i = 0; i < NLA; ++i) { sa[i] = *reinterpret_cast<const double*>(ab + a_voff); ab += a_step; }
      const char* xb = x_next;
#pragma unroll
      for (int i = 0; i < NLX; ++i) {
        if constexpr (true) {
          if (32 * i < ncol) {                                            // compile time
            const bool half = (32 * i + 32 > ncol);
            sx[i] = *reinterpret_cast<const double*>(xb + (half ? x_voff_h : x_voff));
            xb += x_step;
          }
        }
      }
      a_next += 8 * 16 * lda;
      x_next += 8 * 16 * (1 + ldx);
    } else {
#pragma unroll
      for (int i = 0; i < NLA; ++i) {                  // A: rows fastest (512 contiguous bytes per wave)
        const int e = tid + i * SY_THREADS;
        const int row = e & (TR_ROWS - 1), k = k0 + (e >> 7);
        const bool ok = (r0 + row < m) && (k < l);
        const int64_t rr = (r0 + row < m) ? r0 + row : m - 1;
        const double v = A[rr + (int64_t)(k < l ? k : l - 1) * lda];
        sa[i] = ok ? v : 0.0;
      }
#pragma unroll
      for (int i = 0; i < NLX; ++i) {                  // X: rows k0..k0+15 of the columns right of the diagonal block
        const int e = tid + i * SY_THREADS;
        const int k = k0 + (e & 15), c = k0 + (e >> 4);
        const bool ok = (k < l) && (c < l);
        const double v = X[(k < l ? k : l - 1) + (int64_t)(c < l ? c : l - 1) * ldx];
        sx[i] = ok ? v : 0.0;
      }
    }
  };
  auto store = [&](auto KBc) {
    constexpr int kb = decltype(KBc)::value;
    constexpr int buf = kb & 1;
    double* as = sy_smem + buf * BUF;
    double* xs = as + ABUF;
    if constexpr (FAST) {
      constexpr int ncol = 16 * NB - 16 * kb;
#pragma unroll
      for (int i = 0; i < NLA; ++i) as[as_off + 4 * i * TR_LDA] = sa[i];
#pragma unroll
      for (int i = 0; i < NLX; ++i)
        if (32 * i < ncol) xs[((32 * i + 32 > ncol) ? xs_off_h : xs_off) + 16 * kb + 32 * i] = sx[i];
    } else {
#pragma unroll
      for (int i = 0; i < NLA; ++i) {
        const int e = tid + i * SY_THREADS;
        as[(e >> 7) * TR_LDA + (e & (TR_ROWS - 1))] = sa[i];
      }
#pragma unroll
      for (int i = 0; i < NLX; ++i) {
        const int e = tid + i * SY_THREADS;
        const int c = 16 * kb + (e >> 4);
        if (c < 16 * NB) xs[(e & 15) * LDW + c] = sx[i];
      }
    }
  };
  // FAST: column block bb of the tile is FINAL after chunk bb (chunk kb only touches the blocks bb >= kb), so its four stores per
  // lane go out during chunk bb + 1, behind that chunk's prefetch loads in program order (vmcnt retires in order: the loads never
  // queue behind stores of their own chunk) -- the 327 KB of result stores that a workgroup, alone on its CU, used to issue after
  // its last MFMA (16 of a block's 72 us) are spread over the block.  Same uniform-base + 32-bit-offset addressing as the loads.
  const char* const Cb = reinterpret_cast<const char*>(C + r0);
  const uint32_t c_voff = (uint32_t)(8 * ((int64_t)(16 * r + jl) + (int64_t)kk * ldc));   // FAST: fits (launcher: 3 ldc + 128 < 2^29)
  const int64_t c_step = 8 * 4 * ldc;                                     // uniform: 4 columns on
  auto store_block = [&](auto Bc) {
    constexpr int b = decltype(Bc)::value;
    const char* cb = Cb + (int64_t)(16 * b) * 8 * ldc;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) { *reinterpret_cast<double*>(const_cast<char*>(cb) + c_voff) = acc[b][reg]; cb += c_step; }
  };
  using KB0 = std::integral_constant<int, 0>;
  using KB1 = std::integral_constant<int, 1>;
  prefetch(KB0{});
  store(KB0{});
  if (nkb > 1) prefetch(KB1{});
  __syncthreads();
  // The chunk index is a template argument: which column blocks a chunk touches (bb >= kb) is then known at compile
  // time -- as uniform branches around every load and MFMA the same loop ran no faster than the general kernel.
  auto chunk = [&](auto KBc) {
    constexpr int kb = decltype(KBc)::value;
    if (kb >= nkb) return;
    const double* as = sy_smem + (kb & 1) * BUF + kk * TR_LDA + 16 * r + jl;
    const double* xs = sy_smem + (kb & 1) * BUF + ABUF + kk * LDW + jl;
#pragma unroll
    for (int st = 0; st < SY_KC / 4; ++st) {
      const double fa = as[4 * st * TR_LDA];
      // X fragments in groups of TR_G, one group ahead of the MFMAs that consume them (all NB at once would not
      // fit next to 160 accumulator registers)
      constexpr int G0 = kb / TR_G, NG = (NB + TR_G - 1) / TR_G;
      double F[2][TR_G];
#pragma unroll
      for (int b = 0; b < TR_G; ++b) {
        const int bb = G0 * TR_G + b;
        if (bb < NB && bb >= kb) F[G0 & 1][b] = xs[4 * st * LDW + 16 * bb];
      }
#pragma unroll
      for (int g = G0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
          for (int b = 0; b < TR_G; ++b) {
            const int bb = (g + 1) * TR_G + b;
            if (bb < NB && bb >= kb) F[(g + 1) & 1][b] = xs[4 * st * LDW + 16 * bb];
          }
        }
#pragma unroll
        for (int b = 0; b < TR_G; ++b) {
          const int bb = g * TR_G + b;
          if (bb < NB && bb >= kb) acc[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(F[g & 1][b], fa, acc[bb], 0, 0, 0);
        }
      }
      if (st == ((r & 4) ? 2 : 0) && kb + 1 < nkb) {      // waves r and r + 4 share a SIMD: staging chores half a chunk apart,
        __builtin_amdgcn_s_setprio(0);                     // at low priority (one of the two is always in an MFMA stretch)
        store(std::integral_constant<int, kb + 1>{});
        if (kb + 2 < nkb) prefetch(std::integral_constant<int, kb + 2>{});
        __builtin_amdgcn_s_setprio(1);
      }
      if constexpr (FAST && kb >= 1) {
        // half a chunk behind the staging chores (inside their low-priority section, right behind the prefetch loads: measured
        // slower, 10.38 against 10.05 ms per thin QR of a 10^6 x 320 panel)
        if (st == ((r & 4) ? 3 : 1)) store_block(std::integral_constant<int, kb - 1>{});   // final since the end of chunk kb - 1
      }
    }
    __syncthreads();
  };
  sy_for_each(chunk, std::make_integer_sequence<int, NB>{});
  // lane holds D[i = kk + 4 reg][j = jl] = C[r0 + 16 r + jl][16 b + kk + 4 reg]
  const int64_t row = r0 + 16 * r + jl;
  if constexpr (FAST) {
    store_block(std::integral_constant<int, NB - 1>{});      // the others went out as they became final
  } else if (row < m) {
    double* cp = C + row;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int col = 16 * b + kk + 4 * reg;
        if (col < l) cp[(int64_t)col * ldc] = acc[b][reg];
      }
  }
}

template <int NB>
static void tr_launch(hipStream_t st, int64_t m, int64_t l, const double* A, int64_t lda, const double* X, int64_t ldx,
                      double* C, int64_t ldc) {
  static std::atomic<uint64_t> attr_mask{0};
  const size_t shmem = (size_t)2 * SY_KC * (TR_LDA + 16 * NB + 1) * sizeof(double);
  if (first_use_on_this_device(attr_mask)) {
    (void)hipFuncSetAttribute((const void*)tr_kernel<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void*)tr_kernel<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  }
  // whole row blocks through the FAST instantiation (no clamps, 32-bit per-thread offsets); the last partial block, odd
  // sketch widths and leading dimensions beyond the 32-bit offsets through the general one
  static const bool no_fast = (getenv("GSI_TR_NO_FAST") != nullptr);
  const bool fast_ok = !no_fast && (l == 16 * NB) && (3 * lda + TR_ROWS < ((int64_t)1 << 29)) && (31 * ldx + 16 < ((int64_t)1 << 29)) &&
                       (3 * ldc + TR_ROWS < ((int64_t)1 << 29));
  const int64_t nfull = fast_ok ? m / TR_ROWS : 0, nblk = (m + TR_ROWS - 1) / TR_ROWS;
  if (nfull > 0)
    hipLaunchKernelGGL((tr_kernel<NB, true>), dim3((unsigned)nfull), dim3(SY_THREADS), shmem, st, A, lda, m, X, ldx, (int)l, C, ldc,
                       (int64_t)0);
  if (nblk > nfull)
    hipLaunchKernelGGL((tr_kernel<NB, false>), dim3((unsigned)(nblk - nfull)), dim3(SY_THREADS), shmem, st, A, lda, m, X, ldx, (int)l, C,
                       ldc, nfull);
}

// C = A X for square upper-triangular X (entries below the diagonal must be zero: the diagonal blocks are read whole).
// false = shape not covered (the caller runs the general kernel).  C must not alias A.
bool trmm_upper_tall(hipStream_t st, int64_t m, int64_t l, const double* A, int64_t lda, const double* X, int64_t ldx, double* C,
                     int64_t ldc) {
  static const bool off = (getenv("GSI_NO_TRMM_KERNEL") != nullptr);
  if (off || l > 320 || l < 1 || m < 4096 || (m + TR_ROWS - 1) / TR_ROWS > 0x7fffffff) return false;
  const int nb = (int)((l + 15) / 16);
  if (nb <= 8) tr_launch<8>(st, m, l, A, lda, X, ldx, C, ldc);
  else if (nb <= 10) tr_launch<10>(st, m, l, A, lda, X, ldx, C, ldc);
  else if (nb <= 16) tr_launch<16>(st, m, l, A, lda, X, ldx, C, ldc);
  else tr_launch<20>(st, m, l, A, lda, X, ldx, C, ldc);
  return true;
}

}}  // namespace gsi::hipk
