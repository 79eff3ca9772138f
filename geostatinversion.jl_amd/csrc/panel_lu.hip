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
//     lu_step   : apply the rank-1 update of column j-1 to the live panel columns, then the
//                 per-workgroup arg-max of |Y[j:m, j]|                       (HBM/L2-bound)
//     lu_pivot  : one workgroup: fixed-order reduction of the partial arg-maxes, the row
//                 interchange across all l columns, and the pivot row for the next step
//   per block: lu_trsm (U12 = L11^-1 A12) and the trailing update A22 -= L21*U12 through the
//   MFMA gemm kernel.  Blocks of LU_NB columns are split recursively down to LU_LEAF columns
//   (dgetrf2 style), so a sweep touches at most LU_LEAF live columns: the HBM traffic of the
//   sweeps is n*l*LU_LEAF*8 B instead of n*l*LU_NB*8 B.
// Everything is stream-ordered; the host never looks at a pivot.  An exactly zero pivot
// (Julia: SingularException) is recorded in the sticky *info flag, which the backend reads
// and clears at the end of the entry point.
#include "hip_common.hpp"

namespace gsi { namespace hipk {

constexpr int LU_ROWS_PER_BLOCK_MIN = 256;

int64_t lu_max_blocks(int64_t m) {
  int64_t rpt = (m + 256 * 1024 - 1) / (256 * 1024);
  if (rpt < 1) rpt = 1;
  return (m + 256 * rpt - 1) / (256 * rpt) + 1;
}

// rows [j, m): update with column j-1 (if do_update), then arg-max over column j (if do_argmax)
__global__ __launch_bounds__(256) void lu_step_kernel(double* __restrict__ Y, int64_t ld, int64_t m,
                                                      int64_t jb, int b, int64_t j, int do_update,
                                                      int do_argmax, const double* __restrict__ urow,
                                                      int rows_per_thread, double* __restrict__ pval,
                                                      int64_t* __restrict__ pidx) {
  __shared__ double s_val[4];
  __shared__ int64_t s_idx[4];
  const int tid = threadIdx.x;
  const int nlive = (int)(jb + b - j);  // columns j .. jb+b-1
  double u[LU_LEAF];
  double rpiv = 0.0;
  if (do_update) {
    const double piv = urow[j - 1 - jb];
    rpiv = (piv != 0.0) ? 1.0 / piv : 0.0;
#pragma unroll
    for (int k = 0; k < LU_LEAF; ++k) u[k] = (k < nlive) ? urow[j - jb + k] : 0.0;
  }
  double best = -1.0;
  int64_t besti = -1;
  const int64_t base = j + (int64_t)blockIdx.x * 256 * rows_per_thread;
  for (int rr = 0; rr < rows_per_thread; ++rr) {
    const int64_t i = base + tid + 256 * (int64_t)rr;
    if (i < m) {
      double yj = 0.0;
      if (do_update) {
        double* row = Y + i;
        const double lij = (rpiv != 0.0) ? row[(j - 1) * ld] * rpiv : row[(j - 1) * ld];
        row[(j - 1) * ld] = lij;
#pragma unroll
        for (int k = 0; k < LU_LEAF; ++k) {
          if (k < nlive) {
            const double v = row[(j + k) * ld] - lij * u[k];
            row[(j + k) * ld] = v;
            if (k == 0) yj = v;
          }
        }
      } else if (do_argmax) {
        yj = Y[i + j * ld];
      }
      if (do_argmax) {
        const double a = fabs(yj);
        if (a > best) { best = a; besti = i; }
      }
    }
  }
  if (!do_argmax) return;
  // wave reduction: larger value wins, smaller row index on ties (idamax: first maximum)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(best, off, 64);
    const int64_t oi = __shfl_down(besti, off, 64);
    if (ov > best || (ov == best && oi >= 0 && (besti < 0 || oi < besti))) { best = ov; besti = oi; }
  }
  if ((tid & 63) == 0) { s_val[tid >> 6] = best; s_idx[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      const double ov = s_val[w];
      const int64_t oi = s_idx[w];
      if (ov > best || (ov == best && oi >= 0 && (besti < 0 || oi < besti))) { best = ov; besti = oi; }
    }
    pval[blockIdx.x] = best;
    pidx[blockIdx.x] = besti;
  }
}

// one workgroup: final arg-max, row interchange j <-> r over all l columns, pivot row out
__global__ __launch_bounds__(256) void lu_pivot_kernel(double* __restrict__ Y, int64_t ld, int64_t m,
                                                       int64_t l, int64_t jb, int b, int64_t j,
                                                       const double* __restrict__ pval,
                                                       const int64_t* __restrict__ pidx, int nblocks,
                                                       double* __restrict__ urow, int32_t* __restrict__ ipiv,
                                                       int32_t* __restrict__ info) {
  __shared__ double s_val[256];
  __shared__ int64_t s_idx[256];
  const int tid = threadIdx.x;
  double best = -1.0;
  int64_t besti = -1;
  for (int p = tid; p < nblocks; p += 256) {
    const double ov = pval[p];
    const int64_t oi = pidx[p];
    if (ov > best || (ov == best && oi >= 0 && (besti < 0 || oi < besti))) { best = ov; besti = oi; }
  }
  s_val[tid] = best;
  s_idx[tid] = besti;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const double ov = s_val[tid + s];
      const int64_t oi = s_idx[tid + s];
      if (ov > s_val[tid] || (ov == s_val[tid] && oi >= 0 && (s_idx[tid] < 0 || oi < s_idx[tid]))) {
        s_val[tid] = ov;
        s_idx[tid] = oi;
      }
    }
    __syncthreads();
  }
  int64_t r = s_idx[0];
  if (r < 0 || r >= m) r = j;  // all-NaN / empty column: no interchange
  if (tid == 0) {
    ipiv[j] = (int32_t)r;
    if (!(s_val[0] > 0.0) && *info == 0) *info = (int32_t)(j + 1);
  }
  if (r != j) {
    for (int64_t c = tid; c < l; c += 256) {
      const double a = Y[j + c * ld];
      const double bb = Y[r + c * ld];
      Y[j + c * ld] = bb;
      Y[r + c * ld] = a;
    }
  }
  __syncthreads();
  for (int k = tid; k < b; k += 256) urow[k] = Y[j + (jb + k) * ld];
}

// U12 = L11^-1 * A12 for rows jb..jb+b of the columns [c_begin, c_end); thread = one column
__global__ __launch_bounds__(256) void lu_trsm_kernel(double* __restrict__ Y, int64_t ld, int64_t c_begin,
                                                      int64_t c_end, int64_t jb, int b) {
  __shared__ double L11[LU_NB * LU_NB];
  const int tid = threadIdx.x;
  for (int e = tid; e < b * b; e += 256) {
    const int r = e % b, c = e / b;
    L11[r + c * LU_NB] = Y[(jb + r) + (jb + c) * ld];
  }
  __syncthreads();
  for (int64_t c = c_begin + (int64_t)blockIdx.x * 256 + tid; c < c_end; c += (int64_t)gridDim.x * 256) {
    double x[LU_NB];
    double* col = Y + jb + c * ld;
#pragma unroll
    for (int r = 0; r < LU_NB; ++r) x[r] = (r < b) ? col[r] : 0.0;
#pragma unroll
    for (int r = 0; r < LU_NB; ++r) {
      if (r < b) {
        double v = x[r];
#pragma unroll
        for (int rp = 0; rp < LU_NB; ++rp)
          if (rp < r) v -= L11[r + rp * LU_NB] * x[rp];
        x[r] = v;
      }
    }
#pragma unroll
    for (int r = 0; r < LU_NB; ++r)
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
  hipStream_t st; double* Y; int64_t m, l, ld; const LuWork* w; double* gemm_ws; int rpt; int64_t rows_per_block;
};

// per-column sweeps over the leaf [jb, jb+b): 2 launches per column
void lu_leaf(const LuCtx& c, int64_t jb, int b) {
  for (int64_t j = jb; j <= jb + b; ++j) {
    const int do_update = (j > jb) ? 1 : 0;
    const int do_argmax = (j < jb + b) ? 1 : 0;
    const int64_t rows = c.m - j;
    if (rows <= 0) break;
    const int64_t nblocks = (rows + c.rows_per_block - 1) / c.rows_per_block;
    hipLaunchKernelGGL(lu_step_kernel, dim3((unsigned)nblocks), dim3(256), 0, c.st, c.Y, c.ld, c.m, jb, b, j,
                       do_update, do_argmax, c.w->urow, c.rpt, c.w->pval, c.w->pidx);
    if (do_argmax)
      hipLaunchKernelGGL(lu_pivot_kernel, dim3(1), dim3(256), 0, c.st, c.Y, c.ld, c.m, c.l, jb, b, j, c.w->pval,
                         c.w->pidx, (int)nblocks, c.w->urow, c.w->ipiv, c.w->info);
  }
}

// columns [c0, c1) right of the factored block [jb, jb+b): U12 = L11^-1 A12, A22 -= L21 U12
void lu_update_right(const LuCtx& c, int64_t jb, int b, int64_t c0, int64_t c1) {
  const int64_t t = c1 - c0;
  if (t <= 0) return;
  const int tb = (int)((t + 255) / 256);
  hipLaunchKernelGGL(lu_trsm_kernel, dim3(tb), dim3(256), 0, c.st, c.Y, c.ld, c0, c1, jb, b);
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
  for (int64_t jb = 0; jb < l; jb += LU_NB) {
    const int b = (int)((l - jb < LU_NB) ? (l - jb) : LU_NB);
    lu_rec(c, jb, b);
    lu_update_right(c, jb, b, jb + b, l);
  }
  int eb = (int)((l * l + 255) / 256);
  if (eb > 1024) eb = 1024;
  hipLaunchKernelGGL(lu_extract_L_kernel, dim3(eb), dim3(256), 0, st, Y, ld, l);
}

}}  // namespace gsi::hipk
