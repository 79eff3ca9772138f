// panel_qr.hip -- `F = qr(Y, Val(true)); Matrix(F.Q)` (RandMatFact.jl:57-58, 75-76) and the QR
// half of `svd(B)` (RandMatFact.jl:86) for a tall-skinny m x l panel on gfx950.
//
// Householder QR (unconditionally orthonormal Q, safe for the rank-deficient sketches the
// reference's own tests produce), blocked with compact-WY panels of width
// QR_NB, explicit thin Q formed dorgqr-style.  Column pivoting is not reproduced: the
// reference discards R and the permutation, only range(Q) reaches the output (singular
// values and right singular vectors of Q'A are invariant under Q -> QW), and Householder
// without pivoting spans the same range; parity is therefore on subspace / singular values.
//
// Panel factorization, per column j (thread = one row; coalesced column segments):
//   qr_step  : apply reflector j-1 to the live panel columns and, in the same sweep,
//              accumulate sigma = |x|^2 and d_k = x . a_k for column j.  The reflector
//              v = x - beta*e1 differs from x only in its first element, so the products
//              v'a_k follow from d_k once beta is known -- one sweep per column, not two.
//   qr_house : one workgroup: fixed-order sum of the partials, beta/tau, row j of R, and
//              the coefficients the next sweep applies.
// Per panel: V copied out with its unit diagonal, G = V'V and the trailing update
// (I - V T' V') through the MFMA gemm kernels (V'C is computed as C'V so the big
// dimension stays on the grid).
#include "hip_common.hpp"

namespace gsi { namespace hipk {

int64_t qr_max_blocks(int64_t m) {
  int64_t rpt = (m + 256 * 1024 - 1) / (256 * 1024);
  if (rpt < 1) rpt = 1;
  return (m + 256 * rpt - 1) / (256 * rpt) + 1;
}

// coef layout: [0] = scale (1/(alpha-beta)) of column j-1, [1] = tau, [2+k] = tau*w_k for panel column k
__global__ __launch_bounds__(256) void qr_step_kernel(double* __restrict__ Y, int64_t ld, int64_t m,
                                                      int64_t jb, int b, int64_t j, int do_update,
                                                      int do_reduce, const double* __restrict__ coef,
                                                      int rows_per_thread, double* __restrict__ part) {
  __shared__ double s_red[4][QR_NB];
  const int tid = threadIdx.x;
  const int nlive = (int)(jb + b - j);  // columns j .. jb+b-1
  double tw[QR_NB];
  double scale = 0.0;
  if (do_update) {
    scale = coef[0];
#pragma unroll
    for (int k = 0; k < QR_NB; ++k) tw[k] = (k < nlive) ? coef[2 + (j - jb) + k] : 0.0;
  }
  double acc[QR_NB];
#pragma unroll
  for (int k = 0; k < QR_NB; ++k) acc[k] = 0.0;

  const int64_t base = j + (int64_t)blockIdx.x * 256 * rows_per_thread;
  for (int rr = 0; rr < rows_per_thread; ++rr) {
    const int64_t i = base + tid + 256 * (int64_t)rr;
    if (i < m) {
      double* row = Y + i;
      double a[QR_NB];
      if (do_update) {
        const double v = row[(j - 1) * ld] * scale;
        row[(j - 1) * ld] = v;
#pragma unroll
        for (int k = 0; k < QR_NB; ++k) {
          if (k < nlive) {
            a[k] = row[(j + k) * ld] - v * tw[k];
            row[(j + k) * ld] = a[k];
          }
        }
      } else if (do_reduce) {
#pragma unroll
        for (int k = 0; k < QR_NB; ++k)
          if (k < nlive) a[k] = row[(j + k) * ld];
      }
      if (do_reduce && i > j) {
        const double x = a[0];
#pragma unroll
        for (int k = 0; k < QR_NB; ++k)
          if (k < nlive) acc[k] += x * a[k];
      }
    }
  }
  if (!do_reduce) return;
#pragma unroll
  for (int k = 0; k < QR_NB; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid < QR_NB) {
    const double v = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
    part[(int64_t)blockIdx.x * QR_NB + tid] = v;
  }
}

// one workgroup (256 threads): reduce partials, form reflector j (LAPACK dlarfg), update row j
__global__ __launch_bounds__(256) void qr_house_kernel(double* __restrict__ Y, int64_t ld, int64_t jb, int b,
                                                       int64_t j, const double* __restrict__ part,
                                                       int nblocks, double* __restrict__ coef,
                                                       double* __restrict__ tau_out) {
  __shared__ double s_part[16][QR_NB];
  __shared__ double s_sum[QR_NB];
  __shared__ double s_hh[3];  // scale, tau, beta
  const int tid = threadIdx.x;
  const int nlive = (int)(jb + b - j);
  // thread (k = tid % 16, g = tid / 16): fixed-order partial sums over blocks g, g+16, ...
  const int k = tid & (QR_NB - 1), g = tid >> 4;
  double v = 0.0;
  if (k < nlive)
    for (int p = g; p < nblocks; p += 16) v += part[(int64_t)p * QR_NB + k];
  s_part[g][k] = v;
  __syncthreads();
  if (tid < QR_NB) {
    double s = 0.0;
#pragma unroll
    for (int gg = 0; gg < 16; ++gg) s += s_part[gg][tid];
    s_sum[tid] = s;
  }
  __syncthreads();
  if (tid == 0) {
    const double alpha = Y[j + j * ld];
    const double sigma = s_sum[0];
    double beta, tau, scale;
    if (sigma == 0.0) {  // dlarfg: H = I
      beta = alpha; tau = 0.0; scale = 0.0;
    } else {
      beta = -copysign(sqrt(alpha * alpha + sigma), alpha);
      tau = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    s_hh[0] = scale; s_hh[1] = tau; s_hh[2] = beta;
    Y[j + j * ld] = beta;
    coef[0] = scale;
    coef[1] = tau;
    tau_out[j] = tau;
  }
  __syncthreads();
  if (tid >= 1 && tid < nlive) {
    const double scale = s_hh[0], tau = s_hh[1];
    const int64_t c = j + tid;
    const double yjc = Y[j + c * ld];
    const double wk = yjc + s_sum[tid] * scale;   // v' a_c  with v[0] = 1
    Y[j + c * ld] = yjc - tau * wk;
    coef[2 + (j - jb) + tid] = tau * wk;
  }
}

// Vbuf (mr x b, ld mr) <- unit-lower-trapezoidal V of the panel starting at (jb, jb)
__global__ void qr_copyV_kernel(const double* __restrict__ Y, int64_t ld, int64_t m, int64_t jb, int b,
                                double* __restrict__ V) {
  const int64_t mr = m - jb;
  const int64_t total = mr * b;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % mr;
    const int c = (int)(e / mr);
    double v;
    if (r < c) v = 0.0;
    else if (r == c) v = 1.0;
    else v = Y[(jb + r) + (jb + c) * ld];
    V[e] = v;
  }
}

// T (b x b upper, ld QR_NB) from G = V'V and tau (dlarft, forward columnwise); one workgroup
__global__ __launch_bounds__(64) void qr_buildT_kernel(const double* __restrict__ G, int b,
                                                       const double* __restrict__ tau,
                                                       double* __restrict__ T) {
  __shared__ double Ts[QR_NB * QR_NB];
  const int tid = threadIdx.x;
  for (int e = tid; e < QR_NB * QR_NB; e += 64) Ts[e] = 0.0;
  __syncthreads();
  for (int i = 0; i < b; ++i) {
    const double ti = tau[i];
    // T[0:i, i] = -tau_i * T[0:i,0:i] * G[0:i, i]
    if (tid < i) {
      double s = 0.0;
      for (int c = tid; c < i; ++c) s += Ts[tid + c * QR_NB] * G[c + i * b];
      Ts[tid + i * QR_NB] = -ti * s;
    }
    if (tid == 0) Ts[i + i * QR_NB] = ti;
    __syncthreads();
  }
  for (int e = tid; e < QR_NB * QR_NB; e += 64) T[e] = Ts[e];
}

// W2 (b x t, ld b) = op(T) * Wt'   where Wt is t x b (ld t);  transT: use T' (factorization) or T (form Q)
__global__ void qr_applyT_kernel(const double* __restrict__ T, int b, int transT,
                                 const double* __restrict__ Wt, int64_t t, double* __restrict__ W2) {
  const int64_t total = (int64_t)b * t;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % b);
    const int64_t c = e / b;
    double s = 0.0;
    for (int k = 0; k < b; ++k) {
      const double tv = transT ? T[k + r * QR_NB] : T[r + k * QR_NB];
      s += tv * Wt[c + (int64_t)k * t];
    }
    W2[r + c * b] = s;
  }
}

__global__ void qr_init_Q_kernel(double* __restrict__ Q, int64_t m, int64_t l) {
  const int64_t total = m * l;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e % m, c = e / m;
    Q[e] = (r == c) ? 1.0 : 0.0;
  }
}

static inline int grid_for(int64_t total, int cap = 2048) {
  int64_t g = (total + 255) / 256;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// C (mr x t, ld) <- (I - V op(T) V') C
static void apply_block_reflector(hipStream_t st, const QrWork& w, int64_t mr, int b, const double* T,
                                  int transT, double* C, int64_t ldc, int64_t t, double* gemm_ws) {
  if (t <= 0 || mr <= 0) return;
  // Wt (t x b) = C' V
  gemm_f64(st, true, t, b, mr, 1.0, C, ldc, w.Vbuf, mr, 0.0, w.Wt, t, gemm_ws);
  hipLaunchKernelGGL(qr_applyT_kernel, dim3(grid_for((int64_t)b * t)), dim3(256), 0, st, T, b, transT, w.Wt,
                     t, w.W2);
  // C -= V W2
  gemm_f64(st, false, mr, t, b, -1.0, w.Vbuf, mr, w.W2, b, 1.0, C, ldc, gemm_ws);
}

void qr_thinQ(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, double* R, const QrWork& w,
              double* gemm_ws) {
  int rpt = (int)((m + 256 * 1024 - 1) / (256 * 1024));
  if (rpt < 1) rpt = 1;
  const int64_t rows_per_block = 256 * (int64_t)rpt;
  const int64_t npanels = (l + QR_NB - 1) / QR_NB;
  // ---- factorization ----
  for (int64_t pk = 0; pk < npanels; ++pk) {
    const int64_t jb = pk * QR_NB;
    const int b = (int)((l - jb < QR_NB) ? (l - jb) : QR_NB);
    for (int64_t j = jb; j <= jb + b; ++j) {
      const int do_update = (j > jb) ? 1 : 0;
      const int do_reduce = (j < jb + b) ? 1 : 0;
      const int64_t rows = m - j;
      if (rows <= 0) break;
      const int64_t nblocks = (rows + rows_per_block - 1) / rows_per_block;
      hipLaunchKernelGGL(qr_step_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, Y, ld, m, jb, b, j,
                         do_update, do_reduce, w.coef, rpt, w.part);
      if (do_reduce)
        hipLaunchKernelGGL(qr_house_kernel, dim3(1), dim3(256), 0, st, Y, ld, jb, b, j, w.part, (int)nblocks,
                           w.coef, w.tau);
    }
    const int64_t mr = m - jb;
    hipLaunchKernelGGL(qr_copyV_kernel, dim3(grid_for(mr * b)), dim3(256), 0, st, Y, ld, m, jb, b, w.Vbuf);
    gemm_f64(st, true, b, b, mr, 1.0, w.Vbuf, mr, w.Vbuf, mr, 0.0, w.G, b, gemm_ws);
    double* T = w.T + pk * QR_NB * QR_NB;
    hipLaunchKernelGGL(qr_buildT_kernel, dim3(1), dim3(64), 0, st, w.G, b, w.tau + jb, T);
    const int64_t t = l - jb - b;
    if (t > 0) apply_block_reflector(st, w, mr, b, T, 1, Y + jb + (jb + b) * ld, ld, t, gemm_ws);
  }
  if (R != nullptr) extract_upper(st, Y, ld, l, R);
  // ---- explicit thin Q = H_1 ... H_l [I; 0], last panel first ----
  hipLaunchKernelGGL(qr_init_Q_kernel, dim3(grid_for(m * l)), dim3(256), 0, st, w.Qo, m, l);
  for (int64_t pk = npanels - 1; pk >= 0; --pk) {
    const int64_t jb = pk * QR_NB;
    const int b = (int)((l - jb < QR_NB) ? (l - jb) : QR_NB);
    const int64_t mr = m - jb;
    hipLaunchKernelGGL(qr_copyV_kernel, dim3(grid_for(mr * b)), dim3(256), 0, st, Y, ld, m, jb, b, w.Vbuf);
    double* T = w.T + pk * QR_NB * QR_NB;
    apply_block_reflector(st, w, mr, b, T, 0, w.Qo + jb + jb * m, m, l - jb, gemm_ws);
  }
  hipMemcpy2DAsync(Y, ld * sizeof(double), w.Qo, m * sizeof(double), m * sizeof(double), l,
                   hipMemcpyDeviceToDevice, st);
}

}}  // namespace gsi::hipk
