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
#include <stdexcept>
#include "hip_common.hpp"

namespace gsi { namespace hipk {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 csqr(double2 a) { return make_double2(a.x * a.x - a.y * a.y, (a.x + a.x) * a.y); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// Loads and stores by component: an assignment of the double2 STRUCT between address spaces becomes an llvm.memcpy,
// and an array that is the source or target of one stays in scratch memory instead of registers.
__device__ __forceinline__ double2 ld2(const double2* p) { return make_double2(p->x, p->y); }
__device__ __forceinline__ void st2(double2* p, double2 v) { p->x = v.x; p->y = v.y; }

// One pass: batched 1-D transforms of length Ma (power of two) along one axis of the embedded grid.
//   element k of line (inner, o):  W[inner + estride * k + off(o)],  inner < estride (all earlier axes, full length),
//   o < R1 * R2 enumerates the LATER axes restricted to the original grid (i_b < N_b): off(o) = (o % R1) S1 + (o / R1) S2.
// MODE bits: AXIS0 lines are contiguous (else a workgroup takes T neighbouring `inner`); LOADX the first forward
// pass reads the two real columns of X; STOREY the last inverse pass writes the two real columns of Y; INVERSE the sign
// of a plain pass (+1, unnormalised); FUSED the LAST axis: forward transform, multiply by the spectrum, inverse
// transform, all on the lines while they sit in LDS / registers -- the fully transformed array never exists in HBM.
enum { FFT_AXIS0 = 1, FFT_LOADX = 2, FFT_FUSED = 4, FFT_STOREY = 8, FFT_INVERSE = 16 };
constexpr int FFT_TW_LEN = 8192;     // longest supported line; the plan stores exp(-2 pi i k / 8192), k < 4096
constexpr int FFT_MAX_TILE = 8192;   // points of a tile (T lines): 16 per thread, 512 threads, 128 KB of LDS
// Layout of the intermediate array W between the passes (round 4).  NATURAL: axis 0 fastest, element (k0, k1, k2) at
// k0 + M0 (k1 + M1 k2): a strided pass over T neighbouring lines moves T x 16-byte segments that sit a whole axis-0 line
// (a power of two of KB) apart.  BLOCKED (lb = log2 Tb > 0): axis 0 is cut into blocks of Tb points and the index ALONG
// THE LAST AXIS comes next,
//     element (k0, k1, k2)  at  ((k0 >> lb) D1 + k1) (N2 Tb) + k2 Tb + (k0 & (Tb - 1))        [2-D: (k0 >> lb) (N1 Tb) + k1 Tb + ...]
// so the fused last-axis pass reads and writes ONE contiguous run per Tb lines (tiles of T <= Tb lines interleave
// T x 16-byte pieces of it, completed in L2 by the neighbouring workgroups of the XCD), the axis-0 passes move Tb x 16-byte
// runs (256 B at Tb = 16) instead of whole lines, and a middle axis (3-D) keeps T x 16-byte segments at a stride of N2 Tb
// elements.  Every pass stays in place.  The padding of the last axis is still neither stored nor read.
struct FftPass {
  int Ma, log2Ma, nin, nout, T, log2T, lstride, log2es;
  // element offsets within one column pair's array are 32-bit (the launcher checks Mtot < 2^31): 64-bit index arithmetic --
  // and above all the 64-bit DIVISION of off() per thread and item -- cost the contiguous passes registers and VALU time
  uint32_t estride, R1, S1, R2, S2;
  int lb, log2M0;           // lb = 0: natural layout
  int kind;                 // strided passes: 0 natural, 1 blocked / lines along axis 1, 2 blocked / lines along axis 2
  uint32_t BK0, OS, ks;     // block stride of k0 >> lb; stride of the tile's other index; stride between consecutive k of a line
  int lsync;                // 1: stage boundaries order only the line's own waves (line_barrier); 0: workgroup barriers (A/B)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt -- every global load and store in
// flight -- which is exactly what the persistent pass must not do: the next item's loads and the previous item's
// stores are meant to stay in flight across the butterflies.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- the transform of one line: Stockham autosort, radix 16 in registers -------------------------------------------
// A thread owns 16 points of its line: slot s <-> position jt + s * Ma/16 (jt = the thread's index within the line).
// Every stage of a Stockham decimation-in-time transform reads exactly those positions whatever its radix R (the
// inputs of butterfly jb = jt + c Ma/16, c < 16/R, are jb + r Ma/R = jt + (c + r 16/R) Ma/16), so a stage is
//     load 16 slots | twiddle, 16/R R-point DFTs in registers | barrier | scatter to (jb - k) R + k + m Ns | barrier
// with k = jb mod Ns, Ns = the product of the earlier radices.  Ma = 16^a * {1, 2, 4, 8}: a radix-16 stages and at
// most one smaller one LAST, so a 2048-point line is three LDS round trips (the radix-2/4 butterflies this replaces
// took six, and 60 % of their LDS cycles were bank conflicts).  Natural order in and out: no bit reversal anywhere,
// and the last stage of a forward transform leaves the thread holding the very slots the first stage of the inverse
// wants -- the fused pass multiplies by the spectrum in registers in between.
// LDS banking: element i of a line lives at i ^ ((i >> 4) & 15) (16-byte elements, 16 to a 256-byte bank row).  The
// loads are aligned runs of 16 consecutive elements per 16 lanes (a permutation within the row: conflict-free); the
// scatter of the first stage (lane stride 16 elements) lands in 16 different rows at 16 different columns; later
// stages scatter aligned runs again.
__device__ __forceinline__ int swz(int i) { return i ^ ((i >> 4) & 15); }
// A value the optimiser may not treat as loop-invariant: the per-element offsets and masks of an item's fill and drain
// are cheap to recompute, and hoisted out of the persistent loop they occupied (and spilled) ~100 registers.
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// W_16^(sgn * i) for i < 8, folded at compile time once the caller's loops are unrolled
__device__ __forceinline__ double2 mul_w16(double2 v, int i, int sgn) {
  if (i == 0) return v;
  if (i == 4) return sgn > 0 ? make_double2(-v.y, v.x) : make_double2(v.y, -v.x);
  const double C1 = 0.92387953251128673848, S1 = 0.38268343236508977173, H = 0.70710678118654752440;
  const double c = (i == 1) ? C1 : (i == 2) ? H : (i == 3) ? S1 : (i == 5) ? -S1 : (i == 6) ? -H : -C1;
  const double sa = (i == 1 || i == 7) ? S1 : (i == 2 || i == 6) ? H : C1;
  const double s = sgn > 0 ? sa : -sa;
  return make_double2(v.x * c - v.y * s, v.x * s + v.y * c);
}

// R-point DFT of a[0..R), natural order in and out: decimation in frequency, then the even/odd halves interleaved
// (register renaming once everything is unrolled)
template <int R, int SGN>
__device__ __forceinline__ void dft_regs(double2* a) {
  if constexpr (R > 1) {
    constexpr int H = R / 2;
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const double2 t = csub(a[i], a[i + H]);
      a[i] = cadd(a[i], a[i + H]);
      a[i + H] = mul_w16(t, i * (16 / R), SGN);
    }
    dft_regs<H, SGN>(a);
    dft_regs<H, SGN>(a + H);
    double2 t[R];
#pragma unroll
    for (int i = 0; i < H; ++i) { t[2 * i] = a[i]; t[2 * i + 1] = a[i + H]; }
#pragma unroll
    for (int i = 0; i < R; ++i) a[i] = t[i];
  }
}

struct FftLine {            // what a thread knows about its line
  double2* x;               // the line in LDS
  const double2* tabA;      // W_Ma^a, a < 64 (forward sign)
  const double2* tabB;      // W_Ma^(64 b)
  int jt, tpl, L;           // index within the line, threads per line (Ma / points per thread), log2 Ma
  bool act;                 // writes anything at all (a thread past the tile's lines only keeps the barriers company)
  int wpl;                  // waves per line (1: a wave holds whole lines)
};

// Barrier between the stages of ONE line (round 4).  A line belongs to tpl = Ma / 16 threads -- one wave at 1024 points, two at
// 2048 -- and a stage boundary only orders the LDS traffic of that line's own waves.  With one wave per line (or several lines
// per wave) there is nothing to wait for: the LDS executes a wave's instructions in order, a later ds_read of any lane sees
// an earlier ds_write of any lane; only the compiler must not reorder them.  Measured at 512^3 (1024-point lines: 9 of the
// fused item's 12 workgroup barriers gone): 134.5 -> 131.1 ms per 16 columns -- 2.6 %, which says the barriers were never
// what the pass waits for.  With two waves per line an arrival counter in LDS (ds_add, poll) was built and measured SLOWER than
// s_barrier (1000^2: 6.68 -> 6.95 ms) and removed: lines that span waves keep the workgroup barrier.
// Fills and drains of a strided tile touch every line from every thread and keep the workgroup barrier too.
__device__ __forceinline__ void line_barrier(const FftLine& f) {
  if (f.wpl <= 1) { asm volatile("" ::: "memory"); return; }
  lds_barrier();
}

// twiddle and NB R-point DFTs over the slots c + r NB.  Ns = 1 << lNs.
template <int R, int NB, int SGN>
__device__ __forceinline__ void stage_compute(double2 (&v)[R * NB], const FftLine& f, int lNs) {
  constexpr int LR = (R == 16) ? 4 : (R == 8) ? 3 : (R == 4) ? 2 : 1;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double2 a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = v[c + r * NB];
    if (lNs > 0) {
      const int jb = f.jt + c * f.tpl;
      const int k = jb & ((1 << lNs) - 1);
      const int t = k << (f.L - lNs - LR);                // W_{Ns R}^k = W_Ma^t
      double2 w1 = cmul(ld2(&f.tabA[t & 63]), ld2(&f.tabB[t >> 6]));
      if (SGN > 0) w1.y = -w1.y;
      // W^r from the binary powers W, W^2, W^4, W^8 as it is needed: few live registers, short dependency chains
      double2 pw[4];
      pw[0] = w1;
#pragma unroll
      for (int b = 1; b < LR; ++b) pw[b] = csqr(pw[b - 1]);
#pragma unroll
      for (int r = 1; r < R; ++r) {
        double2 wr = make_double2(1.0, 0.0);
        bool have = false;
#pragma unroll
        for (int b = 0; b < LR; ++b)
          if (r & (1 << b)) { wr = have ? cmul(wr, pw[b]) : pw[b]; have = true; }
        a[r] = cmul(a[r], wr);
      }
    }
    dft_regs<R, SGN>(a);
#pragma unroll
    for (int m = 0; m < R; ++m) v[c + m * NB] = a[m];
  }
}
template <int R, int NB>
__device__ __forceinline__ void stage_scatter(const double2 (&v)[R * NB], const FftLine& f, int lNs) {
  constexpr int LR = (R == 16) ? 4 : (R == 8) ? 3 : (R == 4) ? 2 : 1;
  if (!f.act) return;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    const int jb = f.jt + c * f.tpl;
    const int k = jb & ((1 << lNs) - 1);
    const int base = ((jb - k) << LR) + k;
#pragma unroll
    for (int m = 0; m < R; ++m) st2(&f.x[swz(base + (m << lNs))], v[c + m * NB]);
  }
}
// positions >= nin are zero padding that was never written (zpad: the first stage of a forward transform)
template <int P>
__device__ __forceinline__ void stage_gather(double2 (&v)[P], const FftLine& f, bool zpad, int nin) {
#pragma unroll
  for (int s = 0; s < P; ++s) {
    const int pos = f.jt + s * f.tpl;
    const double2 t = ld2(&f.x[swz(pos)]);
    v[s] = (zpad && pos >= nin) ? make_double2(0.0, 0.0) : t;
  }
}
// The whole line transform: n16 radix-16 stages, then (LR > 0) one of radix 2^LR.  A thread holds P = 16 points, or the
// whole line when it is shorter (SHORT: n16 = 0, P = 2^LR).  IN_REGS: the slots are already in v; OUT_REGS: leave the
// result in the slots (else the line ends up in LDS, behind a barrier).
// `late` runs once, just before the final stage's butterflies (the pass issues the next item's loads there when it
// cannot afford to hold them through the whole transform).
// FULL_END: the barrier behind the final scatter is the workgroup's (a strided tile is drained by every thread).
template <int SGN, int LR, bool SHORT, bool IN_REGS, bool OUT_REGS, bool FULL_END, class Late>
__device__ __forceinline__ void fft_line(double2 (&v)[SHORT ? (1 << LR) : 16], const FftLine& f, int n16, bool zpad, int nin,
                                         Late late) {
  constexpr int P = SHORT ? (1 << LR) : 16;
  constexpr int RF = (LR == 0) ? 16 : (1 << LR);           // the final stage
  const int nfull = (LR == 0) ? n16 - 1 : n16;            // radix-16 stages that go back to LDS
  int lNs = 0;
  if constexpr (!SHORT) {
    for (int s = 0; s < nfull; ++s) {
      if (!(IN_REGS && s == 0)) stage_gather<16>(v, f, zpad && s == 0, nin);
      stage_compute<16, 1, SGN>(v, f, lNs);
      line_barrier(f);          // every thread of the line has gathered: it may be overwritten
      stage_scatter<16, 1>(v, f, lNs);
      line_barrier(f);
      lNs += 4;
    }
  }
  if (!(IN_REGS && nfull == 0)) stage_gather<P>(v, f, zpad && nfull == 0, nin);
  late();
  stage_compute<RF, P / RF, SGN>(v, f, lNs);
  if (!OUT_REGS) {
    line_barrier(f);
    stage_scatter<RF, P / RF>(v, f, lNs);
    if (FULL_END) lds_barrier(); else line_barrier(f);
  }
}

// The workgroup is PERSISTENT over (column pair, tile) work items: while item t's butterflies run, item t+1's input
// is on its way into registers and item t's results leave through stores nobody waits for (a one-shot kernel with
// one 130 KB workgroup per CU serialised HBM load / butterflies / HBM store).  Workgroups of one XCD (blockIdx % 8)
// take NEIGHBOURING tiles, so the 64-byte segments of a strided pass with 4 lines per tile complete each other's
// 128-byte lines in the same L2.
template <int MODE, int LR, bool SHORT>
__global__ __launch_bounds__(512) void fft_pass_kernel(double2* __restrict__ W, int64_t Mtot, FftPass ps,
                                                      const double* __restrict__ lam, const double2* __restrict__ twg,
                                                      const double* __restrict__ X, int64_t ldx, double* __restrict__ Y,
                                                      int64_t ldy, int64_t N0, int64_t col0, int64_t l, int tiles, int nitems) {
  constexpr bool AXIS0 = (MODE & FFT_AXIS0) != 0, LOADX = (MODE & FFT_LOADX) != 0, FUSED = (MODE & FFT_FUSED) != 0;
  constexpr bool STOREY = (MODE & FFT_STOREY) != 0, INV = (MODE & FFT_INVERSE) != 0;
  constexpr int SGN1 = INV ? 1 : -1;
  constexpr int P = SHORT ? (1 << LR) : 16;             // points (slots) per thread: slot s <-> position jt + s tpl
  constexpr int NPRE = INV ? P : P / 2;                 // a forward line is at most half full (Ma >= 2 N)
  constexpr int NOUT = (INV || FUSED) ? P / 2 : P;      // and only the first N <= Ma/2 entries of an inverse survive
  extern __shared__ double2 fsm[];
  const int Ma = ps.Ma, L = ps.log2Ma, T = ps.T;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int n16 = SHORT ? 0 : (L >> 2);
  const int ltpl = SHORT ? 0 : L - 4;
  FftLine f;
  f.L = L;
  f.tpl = 1 << ltpl;
  f.tabA = fsm;
  f.tabB = fsm + 64;
  const int ntabB = Ma >= 128 ? (Ma >> 7) : 1;
  double2* buf = fsm + 64 + ntabB;
  const int jl = tid >> ltpl;                      // line of the tile
  f.jt = tid & (f.tpl - 1);
  f.x = buf + (jl < T ? jl : 0) * ps.lstride;
  {
    const int tstep = FFT_TW_LEN / Ma;
    const int half = Ma >= 2 ? Ma / 2 : 1;
    if (tid < 64) st2(&fsm[tid], ld2(&twg[(tid < half ? tid : 0) * tstep]));
    for (int b = tid; b < ntabB; b += nth) st2(&fsm[64 + b], ld2(&twg[(b * 64 < half ? b * 64 : 0) * tstep]));
  }
  f.wpl = (f.tpl >= 64) ? (f.tpl >> 6) : 1;
  if (ps.lsync == 0) f.wpl = 8;                      // A/B: every stage boundary a workgroup barrier (GSI_FFT_LINE_SYNC=0)
  lds_barrier();
  const int nouter = (int)(ps.R1 * ps.R2);
  // contiguous-axis passes: W position of slot s of this thread = a0_tb + s a0_ss (natural: jt + s tpl; blocked, tpl a multiple of
  // Tb: (jt >> lb) BK0 + (jt & (Tb - 1)) + s (tpl >> lb) BK0)
  const uint32_t a0_tb = (uint32_t)(f.jt >> ps.lb) * ps.BK0 + (uint32_t)(f.jt & ((1 << ps.lb) - 1));
  const uint32_t a0_ss = (uint32_t)(f.tpl >> ps.lb) * ps.BK0;
  // strided passes: W offset of element i of this thread = s_tofs + i s_kstep (line tid % T, position tid / T + i nth / T)
  const uint32_t s_tofs = (uint32_t)(tid & (T - 1)) + (uint32_t)(tid >> ps.log2T) * ps.ks;
  const uint32_t s_kstep = (uint32_t)(nth >> ps.log2T) * ps.ks;
  const bool s_lin = (nth >> ps.log2T) * NPRE <= Ma;       // every slot of every thread lies inside its line (positions < Ma)
  const int tiles_per_outer = AXIS0 ? 1 : (int)(ps.estride >> ps.log2T);
  auto off = [&](uint32_t o) -> uint32_t {
    if (ps.R2 == 1) return o * ps.S1;                     // uniform: one outer axis (2-D grids), no division at all
    const uint32_t q = o / ps.R1;
    return (o - q * ps.R1) * ps.S1 + q * ps.S2;
  };

  struct Item { uint32_t off0, lamoff; int pair, o0, i0, nlines; };
  // item w of this workgroup's sequence; workgroups of one XCD walk neighbouring tiles
  const int G = gridDim.x;
  const int bperm = ((G & 7) == 0) ? ((int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3)) : (int)blockIdx.x;
  auto item_of = [&](int w) -> Item {
    Item it;
    it.pair = w / tiles;
    const int tile = w - it.pair * tiles;
    if (AXIS0) {
      it.o0 = tile * T; it.i0 = 0; it.off0 = 0;
      it.nlines = (nouter - it.o0 < T) ? (nouter - it.o0) : T;
      it.lamoff = 0;
    } else {
      it.o0 = tile / tiles_per_outer;
      it.i0 = (tile - it.o0 * tiles_per_outer) << ps.log2T;
      if (ps.kind == 0) {
        it.off0 = off(it.o0) + it.i0;
      } else {                                   // blocked: (k0 >> lb) BK0 + (other index) OS + (k0 & (Tb - 1))
        const int k0 = (ps.kind == 1) ? it.i0 : (it.i0 & ((1 << ps.log2M0) - 1));
        const uint32_t other = (ps.kind == 1) ? (uint32_t)it.o0 : (uint32_t)(it.i0 >> ps.log2M0);
        it.off0 = (uint32_t)(k0 >> ps.lb) * ps.BK0 + other * ps.OS + (uint32_t)(k0 & ((1 << ps.lb) - 1));
      }
      it.nlines = T;
      it.lamoff = (uint32_t)it.i0 * (uint32_t)Ma;      // the spectrum of the last axis is stored line by line (transposed)
    }
    return it;
  };
  // the input a thread brings in for an item.  No predicate on the loads (an element past the end re-reads the last
  // valid one and is dropped later): a conditional load is its own basic block, and the compiler then drains vmcnt
  // between them -- serial HBM round trips.
  auto fetch = [&](const Item& it, double2 (&pre)[NPRE]) {
    if (AXIS0) {
      const uint32_t o = (uint32_t)(it.o0 + (jl < it.nlines ? jl : it.nlines - 1));
      const int64_t lo = LOADX ? N0 * (int64_t)o : (int64_t)off(o);
      const int64_t ca = col0 + 2 * it.pair, cb = (ca + 1 < l) ? ca + 1 : ca;
      const double2* Wb = W + (int64_t)it.pair * Mtot;
#pragma unroll
      for (int r = 0; r < NPRE; ++r) {
        const int pos = f.jt + r * f.tpl;
        if (LOADX) {
          const int64_t i = lo + (pos < ps.nin ? pos : ps.nin - 1);
          pre[r] = make_double2(X[i + ca * ldx], X[i + cb * ldx]);
        } else {
          // an inverse pass (nin = Ma: every position exists).  Slot r of a thread sits r slot-strides behind its first
          // point in BOTH layouts (blocked: tpl is a multiple of Tb): thread base + r * uniform stride, as cheap as natural
          pre[r] = ld2(&Wb[(uint32_t)lo + a0_tb + (uint32_t)r * a0_ss]);
        }
      }
    } else {
      // element i of this thread: line j = tid % T, position k = tid / T + i (nth / T) -- a per-thread constant offset plus
      // a uniform stride, no multiply and no clamp per element (a position past nin reads allocated words of W that the
      // first gather replaces by zeros: forward lines are at most half full, see the layout bounds in fft_pass)
      const double2* Wb = W + (int64_t)it.pair * Mtot + it.off0;
      if (s_lin) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) pre[i] = ld2(&Wb[s_tofs + (uint32_t)i * s_kstep]);
      } else {                                   // short lines: more thread slots than the tile has elements -- clamp (uniform branch)
        const int t0 = opaque(tid);
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
          const int e = t0 + i * nth;
          const int j = e & (T - 1), k = e >> ps.log2T;
          pre[i] = ld2(&Wb[(uint32_t)j + (uint32_t)(k < ps.nin ? k : ps.nin - 1) * ps.ks]);
        }
      }
    }
  };

  double2 pre[NPRE];
  int w = bperm;
  if (w < nitems) { const Item it = item_of(w); fetch(it, pre); }
  for (; w < nitems; w += G) {
    const Item it = item_of(w);
    f.act = (jl < it.nlines);
    double2 v[P];
    // ---- this item's input: into the slots (contiguous lines) or into LDS (strided lines, line index fastest)
    if (AXIS0) {
      const int64_t ca = col0 + 2 * it.pair;
#pragma unroll
      for (int s = 0; s < P; ++s) {
        if (s < NPRE) {
          v[s] = (f.jt + s * f.tpl < ps.nin) ? pre[s] : make_double2(0.0, 0.0);
          if (LOADX && ca + 1 >= l) v[s].y = 0.0;
        } else {
          v[s] = make_double2(0.0, 0.0);
        }
      }
    } else {
      const int t0 = opaque(tid);
#pragma unroll
      for (int i = 0; i < NPRE; ++i) {
        const int e = t0 + i * nth;
        const int j = e & (T - 1), k = e >> ps.log2T;
        // unconditional: an element past nin lands in the padding of its line, which the first gather never reads
        // (predicated, the compiler kept pre[] in scratch memory and reloaded it element by element)
        st2(&buf[j * ps.lstride + swz(k < Ma ? k : Ma - 1)], pre[i]);
      }
      lds_barrier();
    }
    // ---- the spectrum of this item's lines (P per thread, coalesced), then the NEXT item's input: both land
    //      behind the butterflies
    double lamv[FUSED ? P : 1];
    if (FUSED) {
      const double* lp = lam + it.lamoff + (int64_t)(f.act ? jl : 0) * Ma;
#pragma unroll
      for (int s = 0; s < P; ++s) lamv[s] = lp[f.jt + s * f.tpl];
    }
    // Where the next item's loads are issued: as early as possible, unless the registers they occupy are what spills
    // the transform -- a fused pass holds the spectrum through its forward half and the prefetch through its inverse
    // half; a strided inverse pass (16 elements per thread) issues them before its final stage.
    constexpr bool PRE_LATE = !AXIS0 && INV && !FUSED;
    auto prefetch = [&]() { if (w + G < nitems) { const Item nx = item_of(w + G); fetch(nx, pre); } };
    auto nothing = []() {};
    if (!FUSED && !PRE_LATE) prefetch();
    // ---- transforms
    if (PRE_LATE) fft_line<SGN1, LR, SHORT, AXIS0, AXIS0 || FUSED, !AXIS0>(v, f, n16, !INV, ps.nin, prefetch);
    else fft_line<SGN1, LR, SHORT, AXIS0, AXIS0 || FUSED, !AXIS0>(v, f, n16, !INV, ps.nin, nothing);
    if (FUSED) {
#pragma unroll
      for (int s = 0; s < P; ++s) { v[s].x *= lamv[s]; v[s].y *= lamv[s]; }
      prefetch();
      fft_line<1, LR, SHORT, true, AXIS0, !AXIS0>(v, f, n16, false, 0, nothing);
    }
    // ---- results
    if (AXIS0) {
      if (f.act) {
        const uint32_t o = (uint32_t)(it.o0 + jl);
        const int64_t ca = col0 + 2 * it.pair, cb = ca + 1;
        double2* Wb = W + (int64_t)it.pair * Mtot + off(o);
#pragma unroll
        for (int s = 0; s < NOUT; ++s) {
          const int pos = f.jt + s * f.tpl;
          if (pos < ps.nout) {
            if (STOREY) {
              const int64_t i = pos + N0 * (int64_t)o;
              Y[i + ca * ldy] = v[s].x;
              if (cb < l) Y[i + cb * ldy] = v[s].y;
            } else {
              st2(&Wb[a0_tb + (uint32_t)s * a0_ss], v[s]);
            }
          }
        }
      }
    } else {
      double2* Wb = W + (int64_t)it.pair * Mtot + it.off0;
      const int t0 = opaque(tid);
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        const int e = t0 + i * nth;
        const int j = e & (T - 1), k = e >> ps.log2T;
        if (k < ps.nout) st2(&Wb[s_tofs + (uint32_t)i * s_kstep], ld2(&buf[j * ps.lstride + swz(k)]));
      }
      lds_barrier();      // the lines are free for the next item (the stores above are not waited for)
    }
  }
}

// lam = (sum_i nu_i^2)^(beta/2), f_i = min(k_i, M_i - k_i), 0 at k = 0.
// fftrf == 0: nu_i = f_i / M_i (cycles per grid spacing: the correlation length does not depend on the embedding);
// fftrf != 0: nu_i = f_i, the INTEGER wavenumbers FFTRF.jl:86-89 + computesqrtS_f (:40-72) use on its 2N embedding.
// Layout: the fused pass over the LAST axis reads the spectrum of a whole line at a time, so for d >= 2 it is stored
// line by line -- lam[inner * M_last + k_last], inner = the linear index over the earlier axes (a plain transpose of
// the natural order; lines = Mtot / M_last, mlast = M_last; lines = 1 keeps the natural order).
__global__ __launch_bounds__(256) void fft_spectrum_kernel(double* __restrict__ lam, int64_t Mtot, int64_t M0, int64_t M1,
                                                           int64_t M2, double beta, int fftrf, int64_t lines, int64_t mlast) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256) {
    const int64_t k0 = e % M0, r = e / M0, k1 = r % M1, k2 = r / M1;
    const double f0 = (double)((k0 <= M0 - k0) ? k0 : M0 - k0) / (fftrf ? 1.0 : (double)M0);
    const double f1 = (double)((k1 <= M1 - k1) ? k1 : M1 - k1) / (fftrf ? 1.0 : (double)M1);
    const double f2 = (double)((k2 <= M2 - k2) ? k2 : M2 - k2) / (fftrf ? 1.0 : (double)M2);
    const double k2sum = f0 * f0 + f1 * f1 + f2 * f2;
    lam[(e % lines) * mlast + e / lines] = (k2sum > 0.0) ? pow(k2sum, 0.5 * beta) : 0.0;
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
  const int64_t mlast = (M[2] > 1) ? M[2] : ((M[1] > 1) ? M[1] : Mtot);      // d = 1: one line, natural order
  hipLaunchKernelGGL(fft_spectrum_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, M[0], M[1], M[2], beta, fftrf,
                     Mtot / mlast, mlast);
  hipLaunchKernelGGL(fft_sum_kernel, dim3(64), dim3(256), 0, st, lam, Mtot, part64);
  hipLaunchKernelGGL(fft_normalise_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, part64, 64);
}

// ---- FFTRF's convention on a grid whose 2 N embedding is not a power of two ------------------------------------------
// A = R C R' touches the circulant kernel c = F^-1_{2N} lambda only at the lags |t_a| <= N_a - 1 (i, j both in the box), so
// ANY periodic embedding of M'_a >= 2 N_a - 1 points that carries those lags gives the same matrix.  The products keep their
// power-of-two Stockham passes (M' = next power of two >= 2 N); what changes is the spectrum they multiply by:
//     lambda' = F_{M'} c',   c'_t = c_{|t|} for |t_a| <= N_a - 1, 0 elsewhere,   c = F^-1_{2N} lambda (FFTRF.jl:83-90's lambda).
// lambda is real and even along every axis, so both transforms are cosine sums, and both are tensor products of small
// per-axis matrices: computed once per plan with the MFMA contraction kernel (hip_backend.hip:fftcov_create).
//   out[r + c rows] = (weighted && r > 0 ? 2 : 1) cos(2 pi r c / period)
__global__ __launch_bounds__(256) void fft_cos_matrix_kernel(double* __restrict__ out, int64_t rows, int64_t cols, int64_t period,
                                                             int weighted) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e % rows, c = e / rows;
    const int64_t a = (r * c) % period;                    // exact argument reduction
    const double v = cospi(2.0 * (double)a / (double)period);
    out[e] = (weighted && r > 0) ? 2.0 * v : v;
  }
}
void fft_cos_matrix(hipStream_t st, double* out, int64_t rows, int64_t cols, int64_t period, bool weighted) {
  hipLaunchKernelGGL(fft_cos_matrix_kernel, dim3(grid_for(rows * cols, 4096)), dim3(256), 0, st, out, rows, cols, period,
                     weighted ? 1 : 0);
}
// natural order (axis 0 fastest) -> the line-by-line layout of the fused last-axis pass: out[inner * mlast + k_last]
__global__ __launch_bounds__(256) void fft_lines_layout_kernel(const double* __restrict__ nat, double* __restrict__ out, int64_t Mtot,
                                                               int64_t lines, int64_t mlast) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < Mtot; e += (int64_t)gridDim.x * 256)
    out[(e % lines) * mlast + e / lines] = nat[e];
}
void fft_lines_layout(hipStream_t st, const double* nat, double* out, const int64_t M[3]) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  const int64_t mlast = (M[2] > 1) ? M[2] : ((M[1] > 1) ? M[1] : Mtot);
  hipLaunchKernelGGL(fft_lines_layout_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, nat, out, Mtot, Mtot / mlast, mlast);
}
// |k|^beta on the grid M (integer wavenumbers when fftrf), NATURAL order, not normalised
void fft_spectrum_natural(hipStream_t st, double* lam, const int64_t M[3], double beta, int fftrf) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  hipLaunchKernelGGL(fft_spectrum_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, M[0], M[1], M[2], beta, fftrf,
                     (int64_t)1, Mtot);
}
// lam[0, Mtot) *= 1 / sum (unit diagonal); twiddle table behind the 64 scratch doubles, as fft_spectrum leaves them
void fft_finish_plan(hipStream_t st, double* lam, double* part64, const int64_t M[3]) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  hipLaunchKernelGGL(fft_twiddle_kernel, dim3(FFT_TW_LEN / 2 / 256), dim3(256), 0, st, reinterpret_cast<double2*>(lam + Mtot + 64));
  hipLaunchKernelGGL(fft_sum_kernel, dim3(64), dim3(256), 0, st, lam, Mtot, part64);
  hipLaunchKernelGGL(fft_normalise_kernel, dim3(grid_for(Mtot, 4096)), dim3(256), 0, st, lam, Mtot, part64, 64);
}

template <int MODE, int LR, bool SHORT>
static void launch_pass_k(hipStream_t st, int ncus, int threads, size_t shmem, double2* W, int64_t Mtot, const FftPass& ps,
                        const double* lam, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t N0, int64_t col0,
                        int64_t l, int64_t tiles, int64_t nitems) {
  static std::atomic<uint64_t> attr_mask{0};
  if (first_use_on_this_device(attr_mask))
    (void)hipFuncSetAttribute((const void*)fft_pass_kernel<MODE, LR, SHORT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  // persistent workgroups over the (column pair, tile) items: as many as the chip holds at once
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)fft_pass_kernel<MODE, LR, SHORT>, threads, shmem) != hipSuccess || per_cu < 1)
    per_cu = 1;
  int64_t gx = (int64_t)ncus * per_cu;
  if (gx > nitems) gx = nitems;
  const double2* twg = reinterpret_cast<const double2*>(lam + Mtot + 64);
  hipLaunchKernelGGL((fft_pass_kernel<MODE, LR, SHORT>), dim3((unsigned)gx), dim3(threads), shmem, st, W, Mtot, ps, lam, twg, X, ldx,
                     Y, ldy, N0, col0, l, (int)tiles, (int)nitems);
}
// the kernel is specialised on the last radix (log2 Ma mod 4) and on lines shorter than 16 points
template <int MODE>
static void launch_pass(hipStream_t st, int ncus, int threads, size_t shmem, double2* W, int64_t Mtot, const FftPass& ps,
                        const double* lam, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t N0, int64_t col0,
                        int64_t l, int64_t tiles, int64_t nitems) {
#define GSI_FFT_K(LR, SH) launch_pass_k<MODE, LR, SH>(st, ncus, threads, shmem, W, Mtot, ps, lam, X, ldx, Y, ldy, N0, col0, l, tiles, nitems)
  const int lr = ps.log2Ma & 3;
  if (ps.Ma < 16) {
    if (lr == 1) GSI_FFT_K(1, true); else if (lr == 2) GSI_FFT_K(2, true); else GSI_FFT_K(3, true);
  } else {
    if (lr == 0) GSI_FFT_K(0, false); else if (lr == 1) GSI_FFT_K(1, false); else if (lr == 2) GSI_FFT_K(2, false); else GSI_FFT_K(3, false);
  }
#undef GSI_FFT_K
}

// one pass along `axis` (N, M: squeezed grid, axis 0 is never a singleton); fused = forward, spectrum, inverse
static void fft_pass(hipStream_t st, double2* W, int nb, const int64_t N[3], const int64_t M[3], int axis, bool inverse,
                     bool fused, bool loadx, bool storey, const double* lam, const double* X, int64_t ldx,
                     double* Y, int64_t ldy, int64_t col0, int64_t l) {
  const int64_t Mtot = M[0] * M[1] * M[2];
  FftPass ps;
  ps.Ma = (int)M[axis];
  ps.log2Ma = ilog2(ps.Ma);
  ps.nin = inverse ? ps.Ma : (int)N[axis];
  ps.nout = (inverse || fused) ? (int)N[axis] : ps.Ma;
  if (Mtot >= ((int64_t)1 << 31)) throw std::runtime_error("fft_pass: embedding grids of 2^31 points or more are not supported");
  const int64_t stride[3] = {1, M[0], M[0] * M[1]};
  ps.estride = (uint32_t)stride[axis];
  ps.log2es = ilog2(ps.estride);
  ps.R1 = 1; ps.S1 = 0; ps.R2 = 1; ps.S2 = 0;
  if (axis == 0) { ps.R1 = (uint32_t)N[1]; ps.S1 = (uint32_t)stride[1]; ps.R2 = (uint32_t)N[2]; ps.S2 = (uint32_t)stride[2]; }
  else if (axis == 1) { ps.R1 = (uint32_t)N[2]; ps.S1 = (uint32_t)stride[2]; }
  const int64_t nouter = (int64_t)ps.R1 * ps.R2;
  // ---- the layout of W (see FftPass): blocked unless switched off or the grid has no strided pass / a short axis 0
  const int d = (M[2] > 1) ? 3 : ((M[1] > 1) ? 2 : 1);
  static const int tb_env = getenv("GSI_FFT_TB") ? atoi(getenv("GSI_FFT_TB")) : 16;
  int lb = 0;
  if (d >= 2 && tb_env >= 4 && M[0] >= 256) {      // 16 points per thread on the contiguous axis: tpl = M0 / 16 must be a multiple of Tb
    while ((2 << lb) <= tb_env && (2 << lb) <= 64 && (2 << lb) <= M[0] / 16) ++lb;
  }
  const int Tb = 1 << lb;
  ps.lb = lb; ps.log2M0 = ilog2(M[0]); ps.kind = 0; ps.BK0 = 1; ps.OS = 0; ps.ks = ps.estride;
  if (lb > 0) {
    // element (k0, k1, k2) at ((k0 >> lb) D1 + k1) (N2 Tb) + k2 Tb + (k0 & (Tb - 1)); 2-D: D1 = 1, "k2" = k1, N2 -> N1
    const int64_t Nl = (d == 3) ? N[2] : N[1], D1 = (d == 3) ? M[1] : 1;
    ps.BK0 = (uint32_t)(D1 * Nl * Tb);
    if (axis == 0) {                       // lines o = (k1, k2) restricted to the grid: off(o) = k1 S1 + k2 S2
      if (d == 3) { ps.S1 = (uint32_t)(Nl * Tb); ps.S2 = (uint32_t)Tb; } else { ps.S1 = (uint32_t)Tb; ps.S2 = 0; }
    } else if (axis == 1 && d == 3) {      // tile = T neighbours in k0 at fixed k2 = o; k = k1
      ps.kind = 1; ps.OS = (uint32_t)Tb; ps.ks = (uint32_t)(Nl * Tb);
    } else if (axis == 1) {                // 2-D: the last axis; k = k1, no outer index
      ps.kind = 1; ps.OS = 0; ps.ks = (uint32_t)Tb;
    } else {                               // axis 2: tile = T neighbours in k0 at fixed k1 = inner / M0; k = k2
      ps.kind = 2; ps.OS = (uint32_t)(Nl * Tb); ps.ks = (uint32_t)Tb;
    }
  }
  // lines per workgroup, within 8192 points (16 per thread, 512 threads, 128 KB).  Contiguous lines need no neighbours:
  // GSI_FFT_B0 KB of LDS so that several workgroups share a CU.  Strided lines want long segments: a power of two of
  // neighbouring lines, up to 16 (256 bytes) within GSI_FFT_B1 KB, but at least 4 (64 bytes) whatever that costs.
  static const int b0 = getenv("GSI_FFT_B0") ? atoi(getenv("GSI_FFT_B0")) : 64;
  // 140 KB (round 3; was 76): 1024-point lines (512^3 grids) get 8 lines per tile = 128-byte segments, one workgroup per CU,
  // instead of 4 lines (64-byte segments) and two workgroups: 171.8 -> 144.1 ms per 16 columns at 512^3 (3.0 -> 3.6 TB/s),
  // 256^3 +3 %, 2048-point lines unchanged (4 lines are all that fit the 8192-point tile)
  static const int b1 = getenv("GSI_FFT_B1") ? atoi(getenv("GSI_FFT_B1")) : 140;
  const int64_t line_bytes = (int64_t)ps.Ma * (int64_t)sizeof(double2);
  const int tmax = FFT_MAX_TILE / ps.Ma;            // >= 1: Ma <= 8192
  int T;
  if (axis == 0) {
    T = (int)((int64_t)b0 * 1024 / line_bytes);
    if (T > tmax) T = tmax;
    if (T > 32) T = 32;
    if (T > nouter) T = (int)nouter;
    if (T < 1) T = 1;
    ps.log2T = 0;
    ps.lstride = ps.Ma;
  } else {
    int want = (int)((int64_t)b1 * 1024 / line_bytes);
    // at least 4 lines (64-byte segments) whatever that costs -- unless the layout is blocked, where the pieces of the
    // tiles of one block are neighbours in memory and GSI_FFT_MIN_T=2 lets two half-size workgroups share a CU (A/B)
    static const int min_t = getenv("GSI_FFT_MIN_T") ? atoi(getenv("GSI_FFT_MIN_T")) : 4;
    const int floor_t = (lb > 0 && min_t >= 1 && min_t < 4) ? min_t : 4;
    if (want < floor_t) want = floor_t;
    if (want > 16) want = 16;
    if (want > tmax) want = tmax;
    if (want > (int64_t)ps.estride) want = (int)ps.estride;   // estride is a power of two >= 4
    if (lb > 0 && want > Tb) want = Tb;              // a tile never leaves its block of Tb lines
    if (lb > 0 && want > M[0]) want = (int)M[0];
    T = 1; ps.log2T = 0;
    while (2 * T <= want) { T *= 2; ++ps.log2T; }
    ps.lstride = ps.Ma + (T >= 2 ? 16 / T : 0);      // line-fastest fills and drains: T lines x 16/T neighbours = 16 banks rows apart
  }
  ps.T = T;
  static const int lsync_env = getenv("GSI_FFT_LINE_SYNC") ? atoi(getenv("GSI_FFT_LINE_SYNC")) : 1;
  ps.lsync = lsync_env;
  const int tpl = ps.Ma >= 16 ? ps.Ma / 16 : 1;
  const int threads = (T * tpl + 63) / 64 * 64;
  if (threads > 512) throw std::runtime_error("fft_pass: tile exceeds 16 points per thread");
  const int ntabB = ps.Ma >= 128 ? ps.Ma / 128 : 1;
  const size_t shmem = ((size_t)64 + ntabB + (size_t)T * ps.lstride) * sizeof(double2);
  const int64_t tiles = (axis == 0) ? (nouter + T - 1) / T : (ps.estride / T) * nouter;
  const int64_t nitems = tiles * nb;
  if (nitems >= ((int64_t)1 << 31)) throw std::runtime_error("fft_pass: too many work items");
  static const int ncus = [] { int dev = 0, n = 256; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; return n; }();
  const int64_t N0 = N[0];
#define GSI_FFT_LAUNCH(MODE) launch_pass<MODE>(st, ncus, threads, shmem, W, Mtot, ps, lam, X, ldx, Y, ldy, N0, col0, l, tiles, nitems)
  if (axis == 0) {
    if (fused) GSI_FFT_LAUNCH(FFT_AXIS0 | FFT_LOADX | FFT_FUSED | FFT_STOREY);     // d = 1
    else if (!inverse && loadx) GSI_FFT_LAUNCH(FFT_AXIS0 | FFT_LOADX);
    else if (inverse && storey) GSI_FFT_LAUNCH(FFT_AXIS0 | FFT_STOREY | FFT_INVERSE);
    else throw std::runtime_error("fft_pass: axis 0 is always the first forward and the last inverse pass");
  } else {
    if (fused) GSI_FFT_LAUNCH(FFT_FUSED);
    else if (inverse) GSI_FFT_LAUNCH(FFT_INVERSE);
    else GSI_FFT_LAUNCH(0);
  }
#undef GSI_FFT_LAUNCH
}

// Y (n x l, ld ldy) = A X for the embedded-circulant covariance; W holds nb_max * Mtot complex doubles.
// N, M are the SQUEEZED dimensions (singleton axes removed, trailing ones = 1): axis 0 is a real axis.
// d passes forward, the last one fused with the spectrum and the first inverse pass, d - 1 passes back.
void fft_cov_apply(hipStream_t st, const int64_t N[3], const int64_t M[3], const double* lam, double2* W, int nb_max,
                   int64_t l, const double* X, int64_t ldx, double* Y, int64_t ldy) {
  int d = 1;
  if (M[1] > 1) d = 2;
  if (M[2] > 1) d = 3;
  const int64_t npairs = (l + 1) / 2;
  for (int64_t p0 = 0; p0 < npairs; p0 += nb_max) {
    const int nb = (int)((npairs - p0 < nb_max) ? (npairs - p0) : nb_max);
    const int64_t col0 = 2 * p0;
    for (int a = 0; a < d - 1; ++a)
      fft_pass(st, W, nb, N, M, a, false, false, a == 0, false, lam, X, ldx, Y, ldy, col0, l);
    fft_pass(st, W, nb, N, M, d - 1, false, true, d == 1, d == 1, lam, X, ldx, Y, ldy, col0, l);
    for (int a = d - 2; a >= 0; --a)
      fft_pass(st, W, nb, N, M, a, true, false, false, a == 0, lam, X, ldx, Y, ldy, col0, l);
  }
}

}}  // namespace gsi::hipk
