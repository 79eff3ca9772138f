// panel_lu.hip -- `F = lu(Y); Q = F.L` (RandMatFact.jl:60-61, 68-69, 72-73) for a tall-skinny
// m x l panel on gfx950.  Same pivot sequence as LAPACK dgetrf (partial pivoting, first
// maximal |entry| wins ties) and, like Julia's `F.L`, the result is the unit-lower-
// trapezoidal L in PIVOTED row order (L*U == Y[p,:]): the reference never undoes the row
// permutation, so the power iteration it runs is A' P A Omega and this kernel has to
// reproduce P exactly.
//
// Blocked right-looking factorization, block width LU_NB:
//   per column j of the active panel (thread = one row, loop over the <= NB live columns,
//   so every global access is a coalesced column segment):
//     lu_sweep  : ONE launch per column: finish column j-1 (redundant fixed-order reduction of the
//                 per-workgroup arg-maxes, row interchange by workgroup 0), apply its rank-1 update
//                 to the live leaf columns, leave the arg-max of |Y[j:m, j]|       (HBM/L2-bound)
//   per block: lu_trsm (U12 = L11^-1 A12) and the trailing update A22 -= L21*U12 through the
//   MFMA gemm kernel.  Blocks of LU_NB columns are split recursively down to LU_LEAF columns
//   (dgetrf2 style), so a sweep touches at most LU_LEAF live columns: the HBM traffic of the
//   sweeps is n*l*LU_LEAF*8 B instead of n*l*LU_NB*8 B.
// Everything is stream-ordered; the host never looks at a pivot.  An exactly zero pivot
// (Julia: SingularException) is recorded in the sticky *info flag, which the backend reads
// and clears at the end of the entry point.
#include "hip_common.hpp"

namespace gsi { namespace hipk {

int64_t lu_max_blocks(int64_t m) {
  int64_t rpt = (m + 256 * 1024 - 1) / (256 * 1024);
  if (rpt < 1) rpt = 1;
  return (m + 256 * rpt - 1) / (256 * rpt) + 1;
}

__device__ inline bool lu_better(double ov, int64_t oi, double v, int64_t i) {
  // idamax order: larger |value| wins, first (smallest) row index on ties
  return ov > v || (ov == v && oi >= 0 && (i < 0 || oi < i));
}

// ONE launch per column.  For column j of the leaf [jb, jb+b) this kernel
//   (1) finishes column j-1 (if do_update): every workgroup redundantly reduces the per-workgroup
//       arg-maxes the previous launch left (fixed order -> identical everywhere), reads the pivot
//       row's live values from the winner's published candidate row and the old row j-1 from
//       `rowsave` (both written by the PREVIOUS launch and read-only here, so no workgroup races
//       with the row interchange), workgroup 0 performs the interchange (non-live columns by a
//       swap, live columns by writing the pivot row into row j-1; the thread that owns the pivot's
//       old position computes with the old row j-1 and stores there), records ipiv/info;
//   (2) scales column j-1 and applies its rank-1 update to the live columns j..jb+b-1 of rows >= j;
//   (3) (if do_argmax) leaves the per-workgroup arg-max of |column j|, the candidate row's live
//       values and row j's live values for the next launch.
// Thread = one row (rows_per_thread rows for very tall panels); every global access is a coalesced
// column segment.
__global__ __launch_bounds__(256) void lu_sweep_kernel(
    double* __restrict__ Y, int64_t ld, int64_t m, int64_t l, int64_t jb, int b, int64_t j, int do_update,
    int do_argmax, int rows_per_thread, const double* __restrict__ pval_in, const int64_t* __restrict__ pidx_in,
    const double* __restrict__ cand_in, const double* __restrict__ rowsave_in, int nblocks_prev,
    double* __restrict__ pval_out, int64_t* __restrict__ pidx_out, double* __restrict__ cand_out,
    double* __restrict__ rowsave_out, int32_t* __restrict__ ipiv, int32_t* __restrict__ info) {
  __shared__ double s_val[256];
  __shared__ int64_t s_idx[256];
  __shared__ int s_blk[256];
  __shared__ double s_u[LU_LEAF + 1];    // pivot row: [0] = pivot, [1+k] = live column j+k
  __shared__ double s_old[LU_LEAF + 1];  // old row j-1, same layout
  const int tid = threadIdx.x;
  const int nlive = (int)(jb + b - j);   // live columns j .. jb+b-1
  const int64_t jp = j - 1;
  int64_t r = -1;
  double rpiv = 0.0;
  // Issue this thread's first row loads before the reduction prologue: they do not depend on the pivot
  // (the one row that does -- the pivot's old position -- is replaced from s_old below), and the
  // prologue is a chain of dependent global/LDS latencies that would otherwise sit in front of them.
  // Very tall panels give a thread up to LU_PRE rows (rows_per_thread); all of their loads are issued here, back
  // to back: inside the update loop every row's stores would otherwise sit between one row's loads and the next
  // (the compiler cannot move loads of Y above stores to Y), one exposed memory latency per row.
  const int64_t base = j + (int64_t)blockIdx.x * 256 * rows_per_thread;
  constexpr int LU_PRE = 4;
  double pre_x0[LU_PRE];
  double pre_a[LU_PRE][LU_LEAF];
#pragma unroll
  for (int rr = 0; rr < LU_PRE; ++rr) {
    pre_x0[rr] = 0.0;
#pragma unroll
    for (int k = 0; k < LU_LEAF; ++k) pre_a[rr][k] = 0.0;
    const int64_t i0 = base + tid + 256 * (int64_t)rr;
    if (rr < rows_per_thread && i0 < m) {
      const double* row0 = Y + i0;
      if (do_update) pre_x0[rr] = row0[jp * ld];
#pragma unroll
      for (int k = 0; k < LU_LEAF; ++k)
        if (k < nlive) pre_a[rr][k] = row0[(j + k) * ld];
    }
  }
  if (do_update) {
    // (1) redundant, fixed-order reduction of the previous column's partial arg-maxes
    double best = -1.0;
    int64_t besti = -1;
    int bblk = -1;
    for (int p = tid; p < nblocks_prev; p += 256) {
      const double ov = pval_in[p];
      const int64_t oi = pidx_in[p];
      if (lu_better(ov, oi, best, besti)) { best = ov; besti = oi; bblk = p; }
    }
    // speculative: every thread fetches the candidate row of ITS best workgroup and row j-1's saved
    // values now, so these global-load latencies overlap the reduction instead of following it
    const int nl1 = nlive + 1;
    double spec[LU_LEAF + 1];
#pragma unroll
    for (int k = 0; k < LU_LEAF + 1; ++k)
      spec[k] = (bblk >= 0 && k < nl1) ? cand_in[(int64_t)bblk * (LU_LEAF + 1) + k] : 0.0;
    double oldv = (tid < nl1) ? rowsave_in[tid] : 0.0;
    s_val[tid] = best; s_idx[tid] = besti; s_blk[tid] = bblk;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st && lu_better(s_val[tid + st], s_idx[tid + st], s_val[tid], s_idx[tid])) {
        s_val[tid] = s_val[tid + st]; s_idx[tid] = s_idx[tid + st]; s_blk[tid] = s_blk[tid + st];
      }
      __syncthreads();
    }
    r = s_idx[0];
    const int wblk = s_blk[0];
    const bool valid = (r >= jp && r < m && wblk >= 0);
    if (!valid) r = jp;   // all-NaN column: no interchange
    if (tid < nl1) s_old[tid] = oldv;
    if (valid && bblk == wblk) {          // the (unique) thread whose local best is the global winner
#pragma unroll
      for (int k = 0; k < LU_LEAF + 1; ++k)
        if (k < nl1) s_u[k] = spec[k];
    }
    const double bestv = s_val[0];
    __syncthreads();
    if (!valid && tid < nl1) s_u[tid] = s_old[tid];
    __syncthreads();
    const double piv = s_u[0];
    rpiv = (piv != 0.0) ? 1.0 / piv : 0.0;
    if (blockIdx.x == 0) {
      if (tid == 0) {
        ipiv[jp] = (int32_t)r;
        if (!(bestv > 0.0) && *info == 0) *info = (int32_t)(jp + 1);
      }
      if (r != jp) {
        for (int64_t c = tid; c < l; c += 256) {
          if (c < jp || c >= jb + b) {          // not live: plain interchange
            const double a0 = Y[jp + c * ld];
            const double a1 = Y[r + c * ld];
            Y[jp + c * ld] = a1;
            Y[r + c * ld] = a0;
          }
        }
        if (tid < nl1) Y[jp + (jp + tid) * ld] = s_u[tid];   // pivot row moves up into row j-1
      }
    }
  }
  // (2) + (3)
  double best = -1.0;
  int64_t besti = -1;
  double besta[LU_LEAF];
#pragma unroll
  for (int k = 0; k < LU_LEAF; ++k) besta[k] = 0.0;
  auto do_row = [&](int rr, bool have_pre, double px0, const double* pa) {
    const int64_t i = base + tid + 256 * (int64_t)rr;
    if (i < m) {
      double* row = Y + i;
      double a[LU_LEAF];
      if (do_update) {
        const bool moved = (i == r);           // this position receives the old row j-1
        const double x0 = moved ? s_old[0] : (have_pre ? px0 : row[jp * ld]);
        const double lij = (rpiv != 0.0) ? x0 * rpiv : x0;
        row[jp * ld] = lij;
#pragma unroll
        for (int k = 0; k < LU_LEAF; ++k) {
          if (k < nlive) {
            const double xk = moved ? s_old[1 + k] : (have_pre ? pa[k] : row[(j + k) * ld]);
            a[k] = xk - lij * s_u[1 + k];
            row[(j + k) * ld] = a[k];
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < LU_LEAF; ++k)
          if (k < nlive) a[k] = have_pre ? pa[k] : row[(j + k) * ld];
      }
      if (do_argmax) {
        if (i == j) {
#pragma unroll
          for (int k = 0; k < LU_LEAF; ++k)
            if (k < nlive) rowsave_out[k] = a[k];
        }
        const double av = fabs(a[0]);
        if (av > best) {
          best = av; besti = i;
#pragma unroll
          for (int k = 0; k < LU_LEAF; ++k) besta[k] = (k < nlive) ? a[k] : 0.0;
        }
      }
    }
  };
#pragma unroll
  for (int rr = 0; rr < LU_PRE; ++rr)
    if (rr < rows_per_thread) do_row(rr, true, pre_x0[rr], pre_a[rr]);
  for (int rr = LU_PRE; rr < rows_per_thread; ++rr) do_row(rr, false, 0.0, nullptr);
  if (!do_argmax) return;
  __syncthreads();   // s_val / s_idx reuse
  s_val[tid] = best; s_idx[tid] = besti;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st && lu_better(s_val[tid + st], s_idx[tid + st], s_val[tid], s_idx[tid])) {
      s_val[tid] = s_val[tid + st]; s_idx[tid] = s_idx[tid + st];
    }
    __syncthreads();
  }
  const int64_t wi = s_idx[0];
  if (tid == 0) { pval_out[blockIdx.x] = s_val[0]; pidx_out[blockIdx.x] = wi; }
  if (wi >= 0 && wi == besti) {       // the (unique) thread that owns the workgroup's candidate row
#pragma unroll
    for (int k = 0; k < LU_LEAF; ++k)
      if (k < nlive) cand_out[(int64_t)blockIdx.x * (LU_LEAF + 1) + k] = besta[k];
  }
}

// U12 = L11^-1 * A12 for rows jb..jb+b of the columns [c_begin, c_end); thread = one column
template <int NBT>
__global__ __launch_bounds__(256) void lu_trsm_kernel(double* __restrict__ Y, int64_t ld, int64_t c_begin,
                                                      int64_t c_end, int64_t jb, int b) {
  __shared__ double L11[NBT * NBT];
  const int tid = threadIdx.x;
  for (int e = tid; e < b * b; e += 256) {
    const int r = e % b, c = e / b;
    L11[r + c * NBT] = Y[(jb + r) + (jb + c) * ld];
  }
  __syncthreads();
  for (int64_t c = c_begin + (int64_t)blockIdx.x * 256 + tid; c < c_end; c += (int64_t)gridDim.x * 256) {
    double x[NBT];
    double* col = Y + jb + c * ld;
#pragma unroll
    for (int r = 0; r < NBT; ++r) x[r] = (r < b) ? col[r] : 0.0;
#pragma unroll
    for (int r = 0; r < NBT; ++r) {
      if (r < b) {
        double v = x[r];
#pragma unroll
        for (int rp = 0; rp < NBT; ++rp)
          if (rp < r) v -= L11[r + rp * NBT] * x[rp];
        x[r] = v;
      }
    }
#pragma unroll
    for (int r = 0; r < NBT; ++r)
      if (r < b) col[r] = x[r];
  }
}

// top l x l: unit diagonal, zero strict upper triangle (what Julia's F.L returns)
__global__ void lu_extract_L_kernel(double* __restrict__ Y, int64_t ld, int64_t l) {
  const int64_t total = l * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % l, c = e / l;
    if (r == c) Y[r + c * ld] = 1.0;
    else if (r < c) Y[r + c * ld] = 0.0;
  }
}

namespace {
struct LuCtx {
  hipStream_t st; double* Y; int64_t m, l, ld; const LuWork* w; double* gemm_ws; int rpt; int64_t rows_per_block; int nb;
};

// per-column sweeps over the leaf [jb, jb+b): one launch per column (+1 to finish the last one)
void lu_leaf(const LuCtx& c, int64_t jb, int b) {
  int nblocks_prev = 0;
  for (int64_t j = jb; j <= jb + b; ++j) {
    const int do_update = (j > jb) ? 1 : 0;
    const int do_argmax = (j < jb + b) ? 1 : 0;
    const int64_t rows = c.m - j;
    int64_t nblocks = (rows > 0) ? (rows + c.rows_per_block - 1) / c.rows_per_block : 1;   // >= 1: workgroup 0 finishes column j-1
    const int pin = (int)((j - 1) & 1), pout = (int)(j & 1);
    hipLaunchKernelGGL(lu_sweep_kernel, dim3((unsigned)nblocks), dim3(256), 0, c.st, c.Y, c.ld, c.m, c.l, jb, b, j,
                       do_update, do_argmax, c.rpt, c.w->pval[pin & 1], c.w->pidx[pin & 1], c.w->cand[pin & 1],
                       c.w->rowsave[pin & 1], nblocks_prev, c.w->pval[pout], c.w->pidx[pout], c.w->cand[pout],
                       c.w->rowsave[pout], c.w->ipiv, c.w->info);
    nblocks_prev = (int)nblocks;
  }
}

// columns [c0, c1) right of the factored block [jb, jb+b): U12 = L11^-1 A12, A22 -= L21 U12
void lu_update_right(const LuCtx& c, int64_t jb, int b, int64_t c0, int64_t c1) {
  const int64_t t = c1 - c0;
  if (t <= 0) return;
  const int tb = (int)((t + 255) / 256);
  if (b <= 32) hipLaunchKernelGGL(lu_trsm_kernel<32>, dim3(tb), dim3(256), 0, c.st, c.Y, c.ld, c0, c1, jb, b);
  else hipLaunchKernelGGL(lu_trsm_kernel<LU_NB>, dim3(tb), dim3(256), 0, c.st, c.Y, c.ld, c0, c1, jb, b);
  const int64_t mr = c.m - jb - b;
  if (mr > 0)
    gemm_f64(c.st, false, mr, t, b, -1.0, c.Y + (jb + b) + jb * c.ld, c.ld, c.Y + jb + c0 * c.ld, c.ld, 1.0,
             c.Y + (jb + b) + c0 * c.ld, c.ld, c.gemm_ws);
}

// recursive halving (dgetrf2 style) of the block [j0, j0+w), w <= LU_NB: the per-column sweeps only
// ever touch <= LU_LEAF live columns, the rest of the block is brought up to date by small GEMMs
void lu_rec(const LuCtx& c, int64_t j0, int w) {
  if (w <= LU_LEAF) { lu_leaf(c, j0, w); return; }
  const int w1 = ((w / 2 + LU_LEAF - 1) / LU_LEAF) * LU_LEAF >= w ? w / 2 : ((w / 2 + LU_LEAF - 1) / LU_LEAF) * LU_LEAF;
  lu_rec(c, j0, w1);
  lu_update_right(c, j0, w1, j0 + w1, j0 + w);
  lu_rec(c, j0 + w1, w - w1);
}
}  // namespace

void lu_L(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, const LuWork& w, double* gemm_ws) {
  LuCtx c;
  c.st = st; c.Y = Y; c.m = m; c.l = l; c.ld = ld; c.w = &w; c.gemm_ws = gemm_ws;
  c.rpt = (int)((m + 256 * 1024 - 1) / (256 * 1024));
  if (c.rpt < 1) c.rpt = 1;
  c.rows_per_block = 256 * (int64_t)c.rpt;
  // outer block: 32 columns while the panel is cache-sized (the sweeps are latency-bound and narrow trailing
  // updates are cheapest), 64 once it is not (n = 10^6: the trailing matrix is re-read l/NB times from HBM;
  // measured LU 22 -> 19 ms at l = 320, while 64 costs 0.6 ms per LU at n = 65536)
  c.nb = ((double)m * (double)l * 8.0 > 512e6) ? LU_NB : 32;
  for (int64_t jb = 0; jb < l; jb += c.nb) {
    const int b = (int)((l - jb < c.nb) ? (l - jb) : c.nb);
    lu_rec(c, jb, b);
    lu_update_right(c, jb, b, jb + b, l);
  }
  int eb = (int)((l * l + 255) / 256);
  if (eb > 1024) eb = 1024;
  hipLaunchKernelGGL(lu_extract_L_kernel, dim3(eb), dim3(256), 0, st, Y, ld, l);
}

}}  // namespace gsi::hipk
