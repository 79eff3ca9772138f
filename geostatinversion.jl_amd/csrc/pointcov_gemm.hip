// pointcov_gemm.hip -- C (M x L) = G * X for the covariance of SCATTERED points, G(i, j) = sigma2 k(|p_i - p_j| / ell)
// (+ nugget on the diagonal), with 96-row (or 64-row) x 320-column output tiles: every entry of G is generated ONCE per product
// for sketches of up to 320 columns (SURVEY.md 8b "entries generated in the tile loader"; RandMatFact.jl:55,70 are the products).
//
// Why a second kernel beside gemm_f64.hip's GEN 2 path (128 rows x 160 columns per workgroup): on gfx950 an fp64 MFMA holds the
// vector ALU, so the generator's vector instructions ADD to the matrix time -- measured 2.0-2.6 ns per wave
// instruction whatever the instruction (tools/valu_f64_rates.hip; v_rsq_f64 7 ns).  At l = 320 the 160-column kernel walks the
// reduction twice and generates every entry twice.  Here a workgroup owns ALL 320 columns of its rows: eight waves, 2 row groups x
// 4 column quarters, wave tile 48 x 80 (MT = 3; 32 x 80 at MT = 2), half the generator work per flop.  The price is the X tile:
// 320 columns x 8 bytes x depth must sit in LDS twice (double buffering), so the reduction depth per tile is 16 instead of 32
// (2 x 60 KB) -- one barrier per 60 (40) MFMAs of a wave instead of 80 -- and X is re-read once per 96 (64) rows instead of 128.
//
// Per tile of 16 reduction indices: wave w generates the two columns k = 2w, 2w + 1 of the tile of G -- the row points stay in
// registers for the whole kernel, the two column points are loaded one tile ahead through the VECTOR pipe (every lane the same
// address; as scalar loads they shared lgkmcnt with the LDS reads), and a lane's two or three entries are evaluated side by side
// (pointcov_gen.hpp: pre-scaled points, v_rsq_f64 + one Goldschmidt step + residual, table-driven exponential; 27 vector
// instructions per entry).  X arrives PACKED in tile order (pointcov_pack_kernel), HBM/L2 -> registers -> LDS.  Rows beyond M are
// generated from a clamped index and never stored; reduction indices beyond the range meet zero rows of the X tile.  Split-K with
// per-split slabs and the fixed-order reduction of gemm_f64.hip, the split chosen so that the last round of workgroups is full.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include "hip_common.hpp"
#include "pointcov_gen.hpp"

namespace gsi { namespace hipk {
namespace {
typedef double double4_t __attribute__((ext_vector_type(4)));

// rows of C per workgroup: 32 MT (MT 16-row tiles per wave, two row groups): 96 at MT = 3, 64 at MT = 2
constexpr int WBK = 16;            // reduction depth per LDS tile
constexpr int WBKP = WBK + 2;      // padded k stride of the X image [column][k] (WBKP / 2 odd: conflict-free ds_read_b64)
constexpr int WTHREADS = 512;
constexpr int WCSTEP = 2 * WTHREADS / WBK;   // X tile: column advance per pair slot (64)

struct PointGen {
  int32_t npts, kind, dim, pad;
  int64_t roff, koff;      // global index of row 0 of the product / of reduction index 0
  double sigma2, nugget;
};

// MT = 3 (96 rows per workgroup, wave tile 48 x 80, 120 accumulator registers): per flop 2/3 of the X traffic and 2/3 of the
// barriers of the 64-row tile (60 MFMAs per wave between barriers instead of 40).  A wave generates 96 x 2 entries per tile, three
// per lane: (k0, row l), (k1, row 32 + l) and -- lanes 0..31: (k0, row 64 + l), lanes 32..63: (k1, row l - 32).
// RGN = 4 (round 5: sketches of 97 .. 160 columns): the SAME wave tile (48 x 80 at MT = 3, NTQ = 5) with the
// eight waves arranged as 4 row groups x 2 column halves -- a workgroup owns 192 rows x 160 columns, every entry is still
// generated once, and per flop the generator work, the X traffic and the barriers are those of the 96 x 320 form.  (The 128 x
// 160 form of gemm_f64.hip, GEN 2, generates 128 x 32 entries per 80 MFMAs of a wave and stayed at 50 TFLOP/s at l = 160.)  A wave
// then generates 192 x 2 entries per tile, six per lane: rows l, 64 + l, 128 + l of both of its columns.
template <int NTQ, int MT, int RGN>
__global__ __launch_bounds__(WTHREADS) void pointcov_wide_kernel(
    int64_t M, int64_t L, int64_t K, const double* __restrict__ P4, const double2* __restrict__ Xp, int64_t ktiles,
    double* __restrict__ C, int64_t ldc, double* __restrict__ slabs, int64_t kchunk, int nchunks_x, PointGen gen) {
  static_assert(RGN == 2 || (RGN == 4 && MT == 3), "arrangements: 2 row groups x 4 column quarters, or 4 x 2 with 48-row wave tiles");
  constexpr int NT = (8 / RGN) * NTQ;         // 16-column tiles per workgroup
  constexpr int BM = 16 * MT * RGN;           // rows of C per workgroup
  constexpr int NIT = (NT * 16 + WCSTEP - 1) / WCSTEP;   // 64-column groups of the X tile (the packed stream is zero beyond NT * 16)
  constexpr int WBMP = BM + 16;               // padded row stride of the G image [k][row]
  constexpr int A_ELEMS = WBK * WBMP;
  constexpr int B_ELEMS = NIT * WCSTEP * WBKP;
  constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS;
  extern __shared__ double smem_raw[];        // [64-entry table] [2][G tile | X tile]
  double* const gtab = smem_raw;
  double* const smem = smem_raw + 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave % RGN;        // row group: C rows 16 MT rg .. of the workgroup tile
  const int cq = wave / RGN;        // column quarter (half at RGN = 4): 16-column tiles cq * NTQ ..
  const int ch = wave >> 2;         // which of the two waves of a SIMD (w, w + 4): their chores are staggered
  const int jl = lane & 15;
  const int kk = lane >> 4;
  const int64_t tile_lin = blockIdx.x;
  const int split = (int)blockIdx.y;
  const int64_t r0 = (tile_lin / nchunks_x) * BM;
  const int64_t c0 = (tile_lin % nchunks_x) * (NT * 16);
  const int64_t kbeg = (int64_t)split * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  const int64_t ntiles = (kend > kbeg) ? (kend - kbeg + WBK - 1) / WBK : 0;

  double4_t acc[MT][NTQ];
#pragma unroll
  for (int h = 0; h < MT; ++h)
#pragma unroll
    for (int t = 0; t < NTQ; ++t) acc[h][t] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // X tile: pair (k = 2 (tid % 8), column = tid / 8 + 64 it), it < NTQ: one 16-byte load and one ds_write_b128 per pair.  X arrives
  // PACKED (pointcov_pack_kernel): tile kt of column chunk c is the contiguous block [it][tid] of pairs -- a wave's load
  // instruction reads 1 KB in a row and a tile touches one 40 KB run instead of 128 bytes in each of 320 columns 8 n bytes apart
  // (measured: the strided form cost 34 ms of a 480 ms product in misses alone), zero-filled beyond L and K: no predicates here.
  // register sets of the X prefetch: two (tiles t+3, t+2 in flight) at 64 rows; ONE at 96 rows -- 120 accumulator registers
  // leave no room for the second, and the packed stream comes out of L2 within the one tile (60 MFMAs per wave) it then has
  constexpr int NS = (MT == 3) ? 1 : 2;
  double2 b_reg[NS][NIT];
  const int b_c = tid >> 3;
  const int b_k = 2 * (tid & 7);
  const double2* const Xbase = Xp + ((tile_lin % nchunks_x) * ktiles * NIT) * WTHREADS + tid;

  // this thread's row point(s) (fixed for the whole kernel)
  const int64_t row_first = gen.roff + r0;
  double px, py, pz;
  double pbx = 0.0, pby = 0.0, pbz = 0.0, pcx = 0.0, pcy = 0.0, pcz = 0.0;       // MT = 3: rows 32 + lane and (64 + lane | lane - 32)
  const int rowC = (RGN == 4) ? 128 + lane : ((lane < 32) ? 64 + lane : lane - 32);
  {
    auto rowpt = [&](int r, double& x, double& y, double& z) {
      const int64_t gr = row_first + r;
      const int64_t i0 = (gr < gen.npts) ? gr : (int64_t)gen.npts - 1;
      x = P4[4 * i0]; y = P4[4 * i0 + 1]; z = P4[4 * i0 + 2];
    };
    rowpt(lane, px, py, pz);
    if constexpr (RGN == 4) { rowpt(64 + lane, pbx, pby, pbz); rowpt(128 + lane, pcx, pcy, pcz); }
    else if constexpr (MT == 3) { rowpt(32 + lane, pbx, pby, pbz); rowpt(rowC, pcx, pcy, pcz); }
  }
  const GenPointK gq = gen_point_setup(gen.dim, gen.kind);
  gen_table_init(gtab, tid, gen.sigma2);
  __syncthreads();

  auto prefetch = [&](int64_t k0, auto SET) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value;
    const double2* src = Xbase + (k0 / WBK) * (NIT * WTHREADS);   // (k0 is a multiple of the tile depth)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {                         // (element-wise: the struct copy kept the array in scratch memory)
      // the last 64-column group of a 160-column tile is half padding: waves 4..7 (b_c >= 32) have nothing to fetch there
      if ((it + 1) * WCSTEP <= NT * 16 || wave < (NT * 16 - it * WCSTEP) / 8) {
        const double2 v = src[it * WTHREADS];
        b_reg[set][it].x = v.x; b_reg[set][it].y = v.y;
      }
    }
  };

  // The two column points of this wave's slots, ONE tile ahead, through the VECTOR memory pipe (every lane the same address:
  // one line per load).  As scalar loads they shared the counter of the LDS reads -- every wait for a fragment waited for them
  // too (measured: 22 ms of a 517 ms product).
  double2 qxy0, qxy1;
  double qz0 = 0.0, qz1 = 0.0;
  int64_t qg0 = 0, qg1 = 0;                                   // their indices (uniform)
  uint32_t vzero = 0;
  asm volatile("" : "+v"(vzero));                             // an opaque vector zero: keeps the loads out of the scalar unit
  auto load_points = [&](int64_t k0) __attribute__((always_inline)) {
    const int kw = 2 * wave;                                                  // this wave's two columns of the G tile
    const int krem = (kend - k0 < WBK) ? (int)((kend - k0 > 0) ? kend - k0 : 0) : WBK;   // reduction indices of that tile that exist
    // beyond the range: any valid point (the X rows there are zero)
    const int64_t g0 = gen.koff + ((kw < krem) ? k0 + kw : kbeg), g1 = gen.koff + ((kw + 1 < krem) ? k0 + kw + 1 : kbeg);
    qg0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(g0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(g0 & 0xffffffff));
    qg1 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(g1 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(g1 & 0xffffffff));
    const char* const pb = reinterpret_cast<const char*>(P4);
    qxy0 = *reinterpret_cast<const double2*>(pb + 32 * qg0 + vzero);
    qxy1 = *reinterpret_cast<const double2*>(pb + 32 * qg1 + vzero);
    if (gq.flags & 1) {
      qz0 = *reinterpret_cast<const double*>(pb + 32 * qg0 + 16 + vzero);
      qz1 = *reinterpret_cast<const double*>(pb + 32 * qg1 + 16 + vzero);
    }
  };

  auto stage = [&](int buf, auto SET, int64_t k0) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value;
    double* a_s = smem + buf * BUF_ELEMS;
    double* b_s = a_s + A_ELEMS;
    {
      const int kw = 2 * wave;
      const int fl = gen_flags(gq.flags);
      const int rel0 = (int)(qg0 - row_first), rel1 = (int)(qg1 - row_first);  // (point indices are 31-bit)
      if constexpr (RGN == 4) {
        // six entries per lane: rows lane, 64 + lane, 128 + lane of column k0, then of column k1 (two triples side by side)
        auto triple = [&](const double2& qxy, double qz, int rel, int kcol) __attribute__((always_inline)) {
          double dx = px - qxy.x, dy = py - qxy.y;
          double s0 = fma(dx, dx, dy * dy);
          dx = pbx - qxy.x; dy = pby - qxy.y;
          double s1 = fma(dx, dx, dy * dy);
          dx = pcx - qxy.x; dy = pcy - qxy.y;
          double s2 = fma(dx, dx, dy * dy);
          if (fl & 1) {
            const double dz0 = pz - qz, dz1 = pbz - qz, dz2 = pcz - qz;
            s0 = fma(dz0, dz0, s0); s1 = fma(dz1, dz1, s1); s2 = fma(dz2, dz2, s2);
          }
          double v0, v1, v2;
          gen_point_triple(gq, fl, s0, s1, s2, gtab, v0, v1, v2);
          if ((unsigned)rel < (unsigned)BM) {                                    // uniform: the diagonal crosses this column
            v0 += (lane == rel) ? gen.nugget : 0.0;
            v1 += (64 + lane == rel) ? gen.nugget : 0.0;
            v2 += (128 + lane == rel) ? gen.nugget : 0.0;
          }
          a_s[kcol * WBMP + lane] = v0;
          a_s[kcol * WBMP + 64 + lane] = v1;
          a_s[kcol * WBMP + 128 + lane] = v2;
        };
        triple(qxy0, qz0, rel0, kw);
        triple(qxy1, qz1, rel1, kw + 1);
      } else if constexpr (MT == 2) {
        double dx = px - qxy0.x, dy = py - qxy0.y;
        double s0 = fma(dx, dx, dy * dy);
        dx = px - qxy1.x; dy = py - qxy1.y;
        double s1 = fma(dx, dx, dy * dy);
        if (fl & 1) {
          const double dz0 = pz - qz0, dz1 = pz - qz1;
          s0 = fma(dz0, dz0, s0); s1 = fma(dz1, dz1, s1);
        }
        double v0, v1;
        gen_point_pair(gq, fl, s0, s1, gtab, v0, v1);
        if ((unsigned)rel0 < (unsigned)BM || (unsigned)rel1 < (unsigned)BM) {  // uniform: the diagonal crosses this slot
          v0 += (lane == rel0) ? gen.nugget : 0.0;
          v1 += (lane == rel1) ? gen.nugget : 0.0;
        }
        a_s[kw * WBMP + lane] = v0;
        a_s[(kw + 1) * WBMP + lane] = v1;
      } else {
        const bool lo = lane < 32;                                             // the middle entry: column k0 (lanes 0..31) or k1
        const double qcx = lo ? qxy0.x : qxy1.x, qcy = lo ? qxy0.y : qxy1.y;
        double dx = px - qxy0.x, dy = py - qxy0.y;
        double s0 = fma(dx, dx, dy * dy);
        dx = pcx - qcx; dy = pcy - qcy;
        double s1 = fma(dx, dx, dy * dy);
        dx = pbx - qxy1.x; dy = pby - qxy1.y;
        double s2 = fma(dx, dx, dy * dy);
        if (fl & 1) {
          const double dz0 = pz - qz0, dz1 = pcz - (lo ? qz0 : qz1), dz2 = pbz - qz1;
          s0 = fma(dz0, dz0, s0); s1 = fma(dz1, dz1, s1); s2 = fma(dz2, dz2, s2);
        }
        double v0, v1, v2;
        gen_point_triple(gq, fl, s0, s1, s2, gtab, v0, v1, v2);
        if ((unsigned)rel0 < (unsigned)BM || (unsigned)rel1 < (unsigned)BM) {  // uniform: the diagonal crosses this slot
          v0 += (lane == rel0) ? gen.nugget : 0.0;
          v1 += (rowC == (lo ? rel0 : rel1)) ? gen.nugget : 0.0;
          v2 += (32 + lane == rel1) ? gen.nugget : 0.0;
        }
        a_s[kw * WBMP + lane] = v0;
        a_s[(lo ? kw : kw + 1) * WBMP + rowC] = v1;
        a_s[(kw + 1) * WBMP + 32 + lane] = v2;
      }
      load_points(k0 + WBK);                                                    // the next tile's (clamped when there is none)
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if ((it + 1) * WCSTEP <= NT * 16 || wave < (NT * 16 - it * WCSTEP) / 8)
        *reinterpret_cast<double2*>(b_s + (b_c + WCSTEP * it) * WBKP + b_k) = b_reg[set][it];
  };

  double fa[MT], fan[MT];
  double fb[NTQ];
  const int t0 = cq * NTQ;
  auto a_frag = [&](int buf, int s, int h) -> double {
    return (smem + buf * BUF_ELEMS)[(4 * s + kk) * WBMP + 16 * MT * rg + 16 * h + jl];
  };
  auto b_frag = [&](int buf, int s, int t) -> double {
    return (smem + buf * BUF_ELEMS + A_ELEMS)[(16 * (t0 + t) + jl) * WBKP + 4 * s + kk];
  };

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  auto do_tile = [&](int64_t t, auto PAR) __attribute__((always_inline)) {
    constexpr int cur = decltype(PAR)::value;
    using NextSet = std::integral_constant<int, (NS == 2) ? (cur ^ 1) : 0>;
    auto chores = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_s_setprio(0);
      if (t + 1 < ntiles) stage(cur ^ 1, NextSet{}, kbeg + (t + 1) * WBK);       // tile t+1: registers / generator -> other LDS buffer
      if (t + 1 + NS < ntiles) prefetch(kbeg + (t + 1 + NS) * WBK, NextSet{});    // HBM -> the set just drained
      __builtin_amdgcn_s_setprio(1);
    };
#pragma unroll
    for (int s = 0; s < WBK / 4; ++s) {
      if (s == 1 && ch == 0) chores();
      if (s == 2 && ch != 0) chores();
      const bool last = (s + 1 == WBK / 4);
      // the one barrier per tile: tile t+1 becomes visible, and after this step's MFMAs nobody reads buffer `cur` any more
      if (last) __syncthreads();
      const int nbuf = last ? (cur ^ 1) : cur;
      const int ns = last ? 0 : s + 1;
      const bool more = !last || (t + 1 < ntiles);
      if (more) {
#pragma unroll
        for (int h = 0; h < MT; ++h) fan[h] = a_frag(nbuf, ns, h);
      }
#pragma unroll
      for (int tt = 0; tt < NTQ; ++tt) {
#pragma unroll
        for (int h = 0; h < MT; ++h)
          acc[h][tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[tt], fa[h], acc[h][tt], 0, 0, 0);
        if (more) fb[tt] = b_frag(nbuf, ns, tt);
      }
#pragma unroll
      for (int h = 0; h < MT; ++h) fa[h] = fan[h];
    }
  };

  if (ntiles > 0) {
    load_points(kbeg);
    prefetch(kbeg, Set0{});
    if (NS == 2 && ntiles > 1) prefetch(kbeg + WBK, std::integral_constant<int, NS - 1>{});
    stage(0, Set0{}, kbeg);
    if (ntiles > NS) prefetch(kbeg + NS * WBK, Set0{});
    __syncthreads();
#pragma unroll
    for (int h = 0; h < MT; ++h) fa[h] = a_frag(0, 0, h);
#pragma unroll
    for (int t = 0; t < NTQ; ++t) fb[t] = b_frag(0, 0, t);
    int64_t t = 0;
    for (; t + 1 < ntiles; t += 2) {
      do_tile(t, Set0{});
      do_tile(t + 1, Set1{});
    }
    if (t < ntiles) do_tile(t, Set0{});
  }

  // epilogue: lane holds D[i = kk + 4 reg][j = jl]  ->  C[row = .. + jl][col = c0 + 16 (t0 + t) + kk + 4 reg]
  double* const W = (slabs != nullptr) ? slabs + (int64_t)split * M * L : nullptr;
#pragma unroll
  for (int h = 0; h < MT; ++h) {
    const int64_t row = r0 + 16 * MT * rg + 16 * h + jl;
    if (row < M) {
#pragma unroll
      for (int t = 0; t < NTQ; ++t) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t col = c0 + 16 * (t0 + t) + kk + 4 * reg;
          if (col < L) {
            if (W != nullptr) W[row + col * M] = acc[h][t][reg];
            else C[row + col * ldc] = acc[h][t][reg];
          }
        }
      }
    }
  }
}

// X (K x L, leading dimension ldb) -> the tile stream of the kernel above: [chunk][k tile][it][tid] pairs (k = 16 kt + 2 (tid % 8)
// + {0, 1}, column = 64 ntq chunk + tid / 8 + 64 it), zero beyond K and L.  One read and one write of the sketch panel.
__global__ __launch_bounds__(WTHREADS) void pointcov_pack_kernel(int64_t L, int64_t K, const double* __restrict__ B, int64_t ldb,
                                                                 int ntq, int64_t ktiles, double2* __restrict__ Xp) {
  const int tid = threadIdx.x;
  const int64_t kt = blockIdx.x, chunk = blockIdx.y;
  const int64_t k = kt * WBK + 2 * (tid & 7);
  for (int it = 0; it < ntq; ++it) {
    const int64_t c = chunk * 64 * ntq + (tid >> 3) + WCSTEP * it;
    double2 v;
    v.x = (c < L && k < K) ? B[k + c * ldb] : 0.0;
    v.y = (c < L && k + 1 < K) ? B[k + 1 + c * ldb] : 0.0;
    Xp[((chunk * ktiles + kt) * ntq + it) * WTHREADS + tid] = v;
  }
}

template <int NTQ, int MT, int RGN = 2>
void launch_wide2(dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K, const double* P4, const double2* Xp, int64_t ktiles,
                  double* C, int64_t ldc, double* slabs, int64_t kchunk, int nchunks, const PointGen& gen) {
  constexpr int NIT = ((8 / RGN) * NTQ * 16 + WCSTEP - 1) / WCSTEP;
  constexpr size_t shmem = (64 + 2 * (WBK * (16 * MT * RGN + 16) + NIT * WCSTEP * WBKP)) * sizeof(double);
  static std::atomic<uint64_t> attr_mask{0};
  if (first_use_on_this_device(attr_mask))
    (void)hipFuncSetAttribute((const void*)pointcov_wide_kernel<NTQ, MT, RGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL((pointcov_wide_kernel<NTQ, MT, RGN>), grid, dim3(WTHREADS), shmem, st, M, L, K, P4, Xp, ktiles, C, ldc, slabs,
                     kchunk, nchunks, gen);
}
template <int NTQ>
void launch_wide(int mt, dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K, const double* P4, const double2* Xp,
                 int64_t ktiles, double* C, int64_t ldc, double* slabs, int64_t kchunk, int nchunks, const PointGen& gen) {
  if (mt == 3) launch_wide2<NTQ, 3>(grid, st, M, L, K, P4, Xp, ktiles, C, ldc, slabs, kchunk, nchunks, gen);
  else launch_wide2<NTQ, 2>(grid, st, M, L, K, P4, Xp, ktiles, C, ldc, slabs, kchunk, nchunks, gen);
}

int forced_split() {
  static const int f = getenv("GSI_GEMM_FORCE_SPLIT") ? atoi(getenv("GSI_GEMM_FORCE_SPLIT")) : 0;   // experiments
  return f;
}
// the tiling of a product with L columns: chunks of 64 ntq columns, ntq in 3..5 (L > 160)
// rgn = 4: the 192-row x (128 | 160)-column arrangement for 97 <= L <= 160 (one chunk); pack_groups = 64-column groups per chunk of
// the packed X stream
struct WideTiling { int ntq, mt, rgn, pack_groups; int64_t nchunks, active; int nsplit; int64_t kchunk; int ns_eff; };
WideTiling wide_tiling(int64_t M, int64_t L, int64_t K) {
  WideTiling w;
  static const int rows = getenv("GSI_POINTCOV_ROWS") ? atoi(getenv("GSI_POINTCOV_ROWS")) : 96;   // 64 | 96 rows per workgroup (A/B)
  const int64_t tiles = (L + 15) / 16;
  if (L <= 160) {
    w.rgn = 4; w.mt = 3;
    w.ntq = (tiles <= 8) ? 4 : 5;                              // 128 or 160 columns per workgroup
    w.nchunks = 1;
    w.pack_groups = (2 * w.ntq * 16 + WCSTEP - 1) / WCSTEP;    // 2 or 3
    w.active = (M + 191) / 192;
  } else {
    w.rgn = 2;
    w.mt = (rows == 64) ? 2 : 3;
    const int64_t nch = (tiles + 19) / 20;
    const int64_t nt = (tiles + nch - 1) / nch;               // balanced: 11 .. 20 tiles per chunk
    w.ntq = (int)((nt + 3) / 4);
    if (w.ntq < 3) w.ntq = 3;
    w.nchunks = (L + 64 * w.ntq - 1) / (64 * w.ntq);
    w.pack_groups = w.ntq;
    w.active = ((M + 32 * w.mt - 1) / (32 * w.mt)) * w.nchunks;
  }
  // K splits: the grid takes ceil(g s / 256) / s rounds of workgroups, and with 2110 row blocks (n = 202 500, 96 rows) the
  // partial last round is the whole difference between 8.24 and 9 -- measured 478 (s = 1), 453 (2), 439 (4), 440 ms (8), exactly
  // the model.  So the best s is taken outright (the contraction kernel's chooser asks for a 3 % gain per step and stops at 2).
  w.nsplit = 1;
  if (K > 0 && forced_split() > 0) w.nsplit = forced_split();
  else if (K > 0) {
    static const int ncus = [] { int dev = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; return n; }();
    double best = 0.0;
    for (int sp = 1; sp <= 16 && (sp == 1 || K / sp >= 4096); ++sp) {
      const double cost = (double)((w.active * sp + ncus - 1) / ncus) / (double)sp * (1.0 + 0.002 * (double)sp);
      if (sp == 1 || cost < best) { best = cost; w.nsplit = sp; }
    }
  }
  w.kchunk = (K + w.nsplit - 1) / w.nsplit;
  w.kchunk = ((w.kchunk + 31) / 32) * 32;
  if (w.kchunk == 0) w.kchunk = 32;
  w.ns_eff = (K > 0) ? (int)((K + w.kchunk - 1) / w.kchunk) : 1;
  return w;
}
bool wide_applies(int64_t L) {
  static const bool on = !(getenv("GSI_POINTCOV_WIDE") != nullptr && getenv("GSI_POINTCOV_WIDE")[0] == '0');   // A/B
  static const bool tall_on = !(getenv("GSI_POINTCOV_TALL") != nullptr && getenv("GSI_POINTCOV_TALL")[0] == '0');   // A/B: l <= 160 on GEN 2
  return on && (L > 160 || (tall_on && L > 96));              // narrower sketches: gemm_f64.hip's 128 x 160 form (GEN 2)
}
}  // namespace

size_t gemm_pointcov_workspace_doubles(int64_t M, int64_t L, int64_t K) {
  size_t need = gemm_workspace_doubles(M, L, K);
  if (wide_applies(L)) {
    const WideTiling w = wide_tiling(M, L, K);
    if (w.ns_eff > 1) need = std::max(need, (size_t)w.ns_eff * (size_t)M * (size_t)L);
  }
  return need;
}
// doubles of the packed copy of X the wide kernel streams (0: the product is not that kernel's)
size_t gemm_pointcov_pack_doubles(int64_t M, int64_t L, int64_t K) {
  if (M <= 0 || L <= 0 || K <= 0 || !wide_applies(L)) return 0;
  const WideTiling w = wide_tiling(M, L, K);
  const int64_t ktiles = (K + WBK - 1) / WBK;
  return (size_t)w.nchunks * (size_t)ktiles * (size_t)(WCSTEP * w.pack_groups) * (size_t)WBK;
}

// returns false when the product is not this kernel's (narrow sketches: gemm_f64.hip's 160-column kernel generates once anyway).
// xpack: gemm_pointcov_pack_doubles(M, L, K) doubles, 16-byte aligned.
bool gemm_f64_pointcov_wide(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* pts4, int64_t npts, int d, int kind,
                            double sigma2, double nugget, int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C,
                            int64_t ldc, double* ws, double* xpack) {
  if (M <= 0 || L <= 0 || K <= 0 || xpack == nullptr || !wide_applies(L)) return false;
  const WideTiling w = wide_tiling(M, L, K);
  const int64_t ktiles = (K + WBK - 1) / WBK;
  double2* const Xp = reinterpret_cast<double2*>(xpack);
  hipLaunchKernelGGL(pointcov_pack_kernel, dim3((unsigned)ktiles, (unsigned)w.nchunks), dim3(WTHREADS), 0, st, L, K, B, ldb, w.pack_groups,
                     ktiles, Xp);
  const PointGen gen = {(int32_t)npts, kind, d, 0, roff, koff, sigma2, nugget};
  dim3 grid((unsigned)w.active, (unsigned)w.ns_eff, 1);
  double* slabs = (w.ns_eff > 1) ? ws : nullptr;
  if (w.rgn == 4) {
    if (w.ntq == 4) launch_wide2<4, 3, 4>(grid, st, M, L, K, pts4, Xp, ktiles, C, ldc, slabs, w.kchunk, (int)w.nchunks, gen);
    else launch_wide2<5, 3, 4>(grid, st, M, L, K, pts4, Xp, ktiles, C, ldc, slabs, w.kchunk, (int)w.nchunks, gen);
  } else
  switch (w.ntq) {
    case 3: launch_wide<3>(w.mt, grid, st, M, L, K, pts4, Xp, ktiles, C, ldc, slabs, w.kchunk, (int)w.nchunks, gen); break;
    case 4: launch_wide<4>(w.mt, grid, st, M, L, K, pts4, Xp, ktiles, C, ldc, slabs, w.kchunk, (int)w.nchunks, gen); break;
    default: launch_wide<5>(w.mt, grid, st, M, L, K, pts4, Xp, ktiles, C, ldc, slabs, w.kchunk, (int)w.nchunks, gen); break;
  }
  if (w.ns_eff > 1) gemm_splitk_reduce(st, M, L, w.ns_eff, slabs, C, ldc);
  return true;
}

}}  // namespace gsi::hipk
