// gemm_f64.hip -- the contraction kernels of the randomized range finder on gfx950:
//     NN:  C(M x L) = alpha * A(M x K)  * B(K x L) + beta*C      (Y = A*Omega, Q = A*L;
//                                                              RandMatFact.jl:55,70)
//     TN:  C(M x L) = alpha * A(K x M)' * B(K x L) + beta*C      (Q = A'*Q, B' = A'*Q;
//                                                              RandMatFact.jl:67,85)
// Column-major fp64 throughout (Julia Matrix{Float64}).  L = sketch width (<= a few
// hundred), M and K up to 10^5..10^6: "tall-skinny times huge".
//
// Design (MI355X-first, not a BLAS port):
//  * one workgroup owns 64 rows of C and ALL L columns (in chunks of NT*16 <= 160), so
//    every element of the big operand A is read from HBM exactly once per pass; the small
//    operand B is re-read per workgroup but lives in L2 / Infinity Cache.
//  * v_mfma_f64_16x16x4_f64, operands swapped (MFMA-A <- B', MFMA-B <- A') so that the
//    accumulator's lane index runs along C's ROWS: each 16-lane group stores 128
//    contiguous bytes of a column of C (column-major coalescing), and the A fragment is a
//    contiguous 16-double read from LDS.
//  * arithmetic intensity per A byte is L/4 flop/B >> 9.8 (ridge) for L >= 40: the kernel
//    is MFMA-bound, the stream of A needs only ~25% of HBM bandwidth at L = 160.  At
//    64 cycles per MFMA the LDS and VMEM pipes are nearly idle, so the loop is kept simple:
//    one LDS buffer, next tile prefetched into registers while the MFMAs of the current
//    tile run, two workgroups per CU (two waves per SIMD) to cover the barrier bubbles.
//  * LDS images are padded so the per-lane 8-byte fragment reads are bank-conflict free
//    (ds_read_b64: bank = (addr/4) mod 64 over 32-lane groups):
//        B tile  bs[c][k], row stride 34 doubles  -> lane (c = l&15, kk = l>>4) hits bank 4c+2kk
//        A tile (NN) as[k][r], row stride 80      -> kk adds 32 banks
//        A tile (TN) at[r][k], row stride 34      -> as the B tile
//  * split-K over grid.y with per-split slabs + a fixed-order reduction kernel when the
//    grid would not fill the chip (row shards on 8 GPUs, small M): deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "hip_common.hpp"

namespace gsi { namespace hipk {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BMT = 64;    // C rows per workgroup (4 waves x 16)
constexpr int BK = 32;     // reduction depth per LDS tile
constexpr int BKP = 34;    // padded k stride (doubles) of the [col][k] images
constexpr int BMP = 80;    // padded row stride (doubles) of the NN A image [k][r]
constexpr int NTMAX = 10;  // 16-column tiles per workgroup pass (160 columns)

template <int NT, bool TRANS_A>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(
    int64_t M, int64_t L, int64_t K, const double* __restrict__ A, int64_t lda,
    const double* __restrict__ B, int64_t ldb, double* __restrict__ C, int64_t ldc, double alpha,
    double beta, double* __restrict__ slabs, int64_t kchunk) {
  constexpr int A_ELEMS = TRANS_A ? BMT * BKP : BK * BMP;
  __shared__ double a_s[A_ELEMS];
  __shared__ double b_s[NT * 16 * BKP];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int jl = lane & 15;   // MFMA "column" index -> C row within the wave's 16 rows
  const int kk = lane >> 4;   // MFMA k index within a k4 step
  const int64_t r0 = (int64_t)blockIdx.x * BMT;
  const int64_t c0 = (int64_t)blockIdx.z * (NT * 16);
  const int64_t kbeg = (int64_t)blockIdx.y * kchunk;
  const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;

  double4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};

  double a_reg[8];
  double b_reg[2 * NT];

  // per-thread load coordinates
  // NN A tile: element (k = tid/64 + 4*it, r = tid%64)     -> 512 B contiguous per wave
  // TN A tile: element (r = tid/32 + 8*it, k = tid%32)     -> 256 B contiguous per half wave
  // B tile   : element (c = tid/32 + 8*it, k = tid%32)
  const int a_r = TRANS_A ? (tid >> 5) : (tid & 63);
  const int a_k = TRANS_A ? (tid & 31) : (tid >> 6);
  const int b_c = tid >> 5;
  const int b_k = tid & 31;

  // Addressing: ONE wave-uniform 64-bit base per operand (advanced per tile, lives in SGPRs) plus
  // per-thread 32-bit BYTE offsets, so every load is `global_load_dwordx2 v, v_off, s[base]`.
  // (Per-load 64-bit uniform bases overflow the SGPR file: hipcc then spills them to VGPR lanes
  // and threads v_readlane/v_writelane chains between the MFMAs -- measured 95 vs 64 cycles/MFMA.)
  const uint32_t a_off0 = 8u * (TRANS_A ? (uint32_t)(a_k + (int64_t)a_r * lda) : (uint32_t)(a_r + (int64_t)a_k * lda));
  const uint32_t a_step_c = 8u * (uint32_t)((TRANS_A ? 8 : 4) * lda);
  const uint32_t b_off0 = 8u * (uint32_t)(b_k + (int64_t)b_c * ldb);
  const uint32_t b_step_c = 8u * (uint32_t)(8 * ldb);
  const char* const Abase = reinterpret_cast<const char*>(TRANS_A ? A + r0 * lda : A + r0);
  const char* const Bbase = reinterpret_cast<const char*>(B + c0 * ldb);
  // interior workgroups (all 64 rows and all NT*16 columns in range) take an unpredicated
  // load path on full-depth tiles; the branch is workgroup-uniform
  const bool wg_full = (r0 + BMT <= M) && (c0 + NT * 16 <= L);
  auto prefetch = [&](int64_t k0) {
    const char* Ab = Abase + 8 * (TRANS_A ? k0 : k0 * lda);   // uniform
    const char* Bb = Bbase + 8 * k0;                          // uniform
    // launder the strides so the 28 per-load offsets are recomputed per tile (one VALU add each)
    // instead of staying live in 28 VGPRs across the MFMA loop
    uint32_t a_step = a_step_c, b_step = b_step_c;
    asm volatile("" : "+s"(a_step), "+s"(b_step));
    if (wg_full && k0 + BK <= kend) {
#pragma unroll
      for (int it = 0; it < 8; ++it)
        a_reg[it] = *reinterpret_cast<const double*>(Ab + (a_off0 + (uint32_t)it * a_step));
#pragma unroll
      for (int it = 0; it < 2 * NT; ++it)
        b_reg[it] = *reinterpret_cast<const double*>(Bb + (b_off0 + (uint32_t)it * b_step));
      return;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t r = TRANS_A ? r0 + a_r + 8 * it : r0 + a_r;
      const int64_t k = TRANS_A ? k0 + a_k : k0 + a_k + 4 * it;
      a_reg[it] = (r < M && k < kend) ? *reinterpret_cast<const double*>(Ab + (a_off0 + (uint32_t)it * a_step)) : 0.0;
    }
#pragma unroll
    for (int it = 0; it < 2 * NT; ++it) {
      const int64_t c = c0 + b_c + 8 * it;
      const int64_t k = k0 + b_k;
      b_reg[it] = (c < L && k < kend) ? *reinterpret_cast<const double*>(Bb + (b_off0 + (uint32_t)it * b_step)) : 0.0;
    }
  };

  auto stage = [&]() {
    if (TRANS_A) {
#pragma unroll
      for (int it = 0; it < 8; ++it) a_s[(a_r + 8 * it) * BKP + a_k] = a_reg[it];
    } else {
#pragma unroll
      for (int it = 0; it < 8; ++it) a_s[(a_k + 4 * it) * BMP + a_r] = a_reg[it];
    }
#pragma unroll
    for (int it = 0; it < 2 * NT; ++it) b_s[(b_c + 8 * it) * BKP + b_k] = b_reg[it];
  };

  if (kbeg < kend) prefetch(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < kend) prefetch(k0 + BK);
#pragma unroll
    for (int s = 0; s < BK / 4; ++s) {
      const double bop = TRANS_A ? a_s[(16 * wave + jl) * BKP + 4 * s + kk]
                                 : a_s[(4 * s + kk) * BMP + 16 * wave + jl];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double aop = b_s[(16 * t + jl) * BKP + 4 * s + kk];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: lane holds D[i = kk + 4*reg][j = jl]  ->  C[row = r0+16*wave+jl][col = c0+16t+kk+4*reg]
  const int64_t row = r0 + 16 * wave + jl;
  if (row < M) {
    if (slabs != nullptr) {
      double* W = slabs + (int64_t)blockIdx.y * M * L;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t col = c0 + 16 * t + kk + 4 * reg;
          if (col < L) W[row + col * M] = acc[t][reg];
        }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t col = c0 + 16 * t + kk + 4 * reg;
          if (col < L) {
            double v = alpha * acc[t][reg];
            if (beta != 0.0) v += beta * C[row + col * ldc];
            C[row + col * ldc] = v;
          }
        }
    }
  }
}

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

template <int NT, bool TRANS_A>
static void launch_nt(dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K, const double* A,
                      int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double alpha,
                      double beta, double* slabs, int64_t kchunk) {
  hipLaunchKernelGGL((gemm_f64_kernel<NT, TRANS_A>), grid, dim3(256), 0, st, M, L, K, A, lda, B, ldb, C,
                     ldc, alpha, beta, slabs, kchunk);
}

template <bool TRANS_A>
static void launch_dispatch(int nt, dim3 grid, hipStream_t st, int64_t M, int64_t L, int64_t K,
                            const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                            int64_t ldc, double alpha, double beta, double* slabs, int64_t kchunk) {
#define GSI_CASE(N)                                                                             \
  case N:                                                                                       \
    launch_nt<N, TRANS_A>(grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk); \
    break;
  switch (nt) {
    GSI_CASE(1) GSI_CASE(2) GSI_CASE(3) GSI_CASE(4) GSI_CASE(5)
    GSI_CASE(6) GSI_CASE(7) GSI_CASE(8) GSI_CASE(9) GSI_CASE(10)
    default: break;
  }
#undef GSI_CASE
}

// Number of K splits for a grid of `nwg` workgroups: fill ~2 workgroups per CU.
int gemm_choose_split(int64_t nwg, int64_t K) {
  if (nwg >= 384 || K < 8 * BK) return 1;
  int64_t want = (512 + nwg - 1) / nwg;
  int64_t maxs = K / (4 * BK);
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  return (int)want;
}

size_t gemm_workspace_doubles(int64_t M, int64_t L, int64_t K) {
  const int64_t nchunk_cols = (L + NTMAX * 16 - 1) / (NTMAX * 16);
  const int64_t nwg = ((M + BMT - 1) / BMT) * nchunk_cols;
  const int ns = gemm_choose_split(nwg, K);
  return ns > 1 ? (size_t)ns * (size_t)M * (size_t)L : 0;
}

// Host launcher. `ws` must hold gemm_workspace_doubles(M, L, K) doubles (or be null if 0).
void gemm_f64(hipStream_t st, bool transA, int64_t M, int64_t L, int64_t K, double alpha, const double* A,
              int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, double* ws) {
  if (M <= 0 || L <= 0) return;
  // columns are processed in chunks of nt*16 <= 160; balance the chunks
  const int64_t tiles = (L + 15) / 16;
  const int64_t nchunks = (tiles + NTMAX - 1) / NTMAX;
  const int nt = (int)((tiles + nchunks - 1) / nchunks);
  const int64_t rowblocks = (M + BMT - 1) / BMT;
  const int nsplit = (K > 0) ? gemm_choose_split(rowblocks * nchunks, K) : 1;
  int64_t kchunk = (K + nsplit - 1) / nsplit;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  if (kchunk == 0) kchunk = BK;
  const int ns_eff = (K > 0) ? (int)((K + kchunk - 1) / kchunk) : 1;
  dim3 grid((unsigned)rowblocks, (unsigned)ns_eff, (unsigned)nchunks);
  double* slabs = (ns_eff > 1) ? ws : nullptr;
  if (transA)
    launch_dispatch<true>(nt, grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk);
  else
    launch_dispatch<false>(nt, grid, st, M, L, K, A, lda, B, ldb, C, ldc, alpha, beta, slabs, kchunk);
  if (ns_eff > 1) {
    const int64_t total = M * L;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, M, L, ns_eff, slabs, C, ldc,
                       alpha, beta);
  }
}

}}  // namespace gsi::hipk
