// cpu_backend.cpp -- TEST INFRASTRUCTURE ONLY.  A gsi::Backend over the C restatement
// (gsi_oracle.c) so that the SAME pipeline.cpp / api.cpp that ship in libgsi_hip.so can be run
// on a machine without a GPU: the C ABI surface, the row-sharded order of operations and the
// collective sequence are then testable under `pytest -m "not gpu"` (world_size-2 gloo).
// Built by oracle/Makefile into oracle/_build/libgsi_cpuref.so.  The product package never
// loads this library (geostatinversion.jl_amd/_lib.py opens libgsi_hip.so only).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../include/gsi_hip.h"
#include "../geostatinversion.jl_amd/csrc/backend.hpp"
#include "../geostatinversion.jl_amd/csrc/lsqr_state.hpp"
#include "../geostatinversion.jl_amd/csrc/pointcov.hpp"

extern "C" {
void gsio_gemm_nn(int64_t, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t, double, double*, int64_t);
void gsio_gemm_tn(int64_t, int64_t, int64_t, double, const double*, int64_t, const double*, int64_t, double, double*, int64_t);
int gsio_lu_L(double*, int64_t, int64_t, int64_t, int32_t*);
void gsio_qr_thinQ(double*, int64_t, int64_t, int64_t, int, double*, int32_t*);
int gsio_svd_tall(double*, int64_t, int64_t, int64_t, double*);
int gsio_chol_upper(double*, int64_t);
void gsio_trsm_right_upper(double*, int64_t, int64_t, int64_t, const double*);
void gsio_center_rows(double*, int64_t, int64_t, int64_t);

// collectives supplied by the test harness (torch.distributed/gloo through ctypes callbacks)
typedef void (*gsi_cpuref_allreduce_fn)(double* buf, int64_t count);
typedef void (*gsi_cpuref_allgather_fn)(const double* send, double* recv, int64_t count);
static gsi_cpuref_allreduce_fn g_allreduce = nullptr;
static gsi_cpuref_allgather_fn g_allgather = nullptr;
void gsi_cpuref_set_collectives(gsi_cpuref_allreduce_fn ar, gsi_cpuref_allgather_fn ag) {
  g_allreduce = ar;
  g_allgather = ag;
}
}

namespace gsi {
namespace {

class CpuBackend : public Backend {
 public:
  const char* name() const override { return "cpu-reference (test only)"; }
  double* alloc(size_t count) override {
    if (count == 0) count = 1;
    double* p = (double*)std::malloc(count * sizeof(double));
    if (!p) throw Error(GSI_ERR_OOM, "malloc failed");
    in_use_ += (int64_t)(count * sizeof(double));
    sizes_.push_back({p, count * sizeof(double)});
    return p;
  }
  void release(double* p) override {
    if (!p) return;
    for (size_t i = 0; i < sizes_.size(); ++i)
      if (sizes_[i].first == p) { in_use_ -= (int64_t)sizes_[i].second; sizes_[i] = sizes_.back(); sizes_.pop_back(); break; }
    std::free(p);
  }
  int64_t bytes_in_use() const override { return in_use_; }
  void upload2d(double* d, int64_t ldd, const double* h, int64_t ldh, int64_t r, int64_t c) override { copy2d(d, ldd, h, ldh, r, c); }
  void download2d(double* h, int64_t ldh, const double* s, int64_t lds, int64_t r, int64_t c) override { copy2d(h, ldh, s, lds, r, c); }
  void copy2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t r, int64_t c) override {
    for (int64_t j = 0; j < c; ++j) std::memmove(d + j * ldd, s + j * lds, sizeof(double) * (size_t)r);
  }
  void fill_zero(double* p, size_t count) override { std::memset(p, 0, count * sizeof(double)); }
  void sync() override {}
  void gemm_nn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda, const double* B,
               int64_t ldb, double beta, double* C, int64_t ldc) override {
    gsio_gemm_nn(m, l, k, alpha, A, lda, B, ldb, beta, C, ldc);
  }
  void gemm_tn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda, const double* B,
               int64_t ldb, double beta, double* C, int64_t ldc) override {
    gsio_gemm_tn(m, l, k, alpha, A, lda, B, ldb, beta, C, ldc);
  }
  void gemm_nn_gridcov(int64_t m, int64_t l, int64_t k, const double* tab, int64_t nx, int64_t ny, int64_t roff,
                       int64_t koff, const double* B, int64_t ldb, double* C, int64_t ldc) override {
    (void)nx;
    for (int64_t c = 0; c < l; ++c)
      for (int64_t r = 0; r < m; ++r) {
        const int64_t gi = roff + r;
        double s = 0.0;
        for (int64_t kk = 0; kk < k; ++kk) {
          const int64_t gj = koff + kk;
          s += tab[std::llabs(gi / ny - gj / ny) * ny + std::llabs(gi % ny - gj % ny)] * B[kk + c * ldb];
        }
        C[r + c * ldc] = s;
      }
  }
  void gemm_nn_pointcov(int64_t m, int64_t l, int64_t k, const double* pts, int d, int kind, double ell, double sigma2,
                        double nugget, int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C,
                        int64_t ldc) override {
    const pointcov::Params prm{d, kind, 1.0 / ell, sigma2, nugget};
    for (int64_t c = 0; c < l; ++c)
      for (int64_t r = 0; r < m; ++r) {
        const int64_t gi = roff + r;
        double s = 0.0;
        for (int64_t kk = 0; kk < k; ++kk) {
          const int64_t gj = koff + kk;
          double d2 = 0.0;
          for (int a = 0; a < d; ++a) { const double t = pts[gi * d + a] - pts[gj * d + a]; d2 += t * t; }
          s += pointcov::kernel(prm, d2, gi == gj) * B[kk + c * ldb];
        }
        C[r + c * ldc] = s;
      }
  }
  // circulant-embedding covariance with a naive O(M * sum M_a) separable DFT (test sizes only)
  struct FftCov { int64_t N[3], M[3]; std::vector<double> lam; };
  static void dft_axis(std::vector<std::complex<double>>& w, const int64_t M[3], int axis, double sign) {
    const int64_t Ma = M[axis];
    if (Ma == 1) return;
    int64_t es = 1;
    for (int a = 0; a < axis; ++a) es *= M[a];
    const int64_t os = es * Ma, total = M[0] * M[1] * M[2];
    std::vector<std::complex<double>> line((size_t)Ma), out((size_t)Ma);
    for (int64_t o = 0; o < total / os; ++o)
      for (int64_t i = 0; i < es; ++i) {
        for (int64_t k = 0; k < Ma; ++k) line[(size_t)k] = w[(size_t)(i + es * k + os * o)];
        for (int64_t k = 0; k < Ma; ++k) {
          std::complex<double> s = 0.0;
          for (int64_t j = 0; j < Ma; ++j)
            s += line[(size_t)j] * std::polar(1.0, sign * 2.0 * M_PI * (double)((j * k) % Ma) / (double)Ma);
          out[(size_t)k] = s;
        }
        for (int64_t k = 0; k < Ma; ++k) w[(size_t)(i + es * k + os * o)] = out[(size_t)k];
      }
  }
  void* fftcov_create(const int64_t N[3], double beta, int fftrf) override {
    FftCov* p = new FftCov();
    int64_t Mtot = 1;
    for (int a = 0; a < 3; ++a) {
      p->N[a] = N[a];
      int64_t m = 1;
      while (m < 2 * N[a]) m <<= 1;
      if (fftrf) m = 2 * N[a];           // FFTRF's own embedding; the naive DFT here takes any length
      p->M[a] = (N[a] == 1) ? 1 : m;
      Mtot *= p->M[a];
    }
    p->lam.resize((size_t)Mtot);
    double tot = 0.0;
    for (int64_t e = 0; e < Mtot; ++e) {
      const int64_t k0 = e % p->M[0], r = e / p->M[0], k1 = r % p->M[1], k2 = r / p->M[1];
      const int64_t kk[3] = {k0, k1, k2};
      double s = 0.0;
      for (int a = 0; a < 3; ++a) {
        const double f = (double)std::min(kk[a], p->M[a] - kk[a]) / (fftrf ? 1.0 : (double)p->M[a]);
        s += f * f;
      }
      p->lam[(size_t)e] = s > 0.0 ? std::pow(s, 0.5 * beta) : 0.0;
      tot += p->lam[(size_t)e];
    }
    for (double& v : p->lam) v /= tot;
    return p;
  }
  void fftcov_destroy(void* plan) override { delete static_cast<FftCov*>(plan); }
  void fftcov_apply(void* plan, int64_t l, const double* X, int64_t ldx, double* Y, int64_t ldy) override {
    FftCov* p = static_cast<FftCov*>(plan);
    const int64_t* N = p->N; const int64_t* M = p->M;
    const int64_t Mtot = M[0] * M[1] * M[2], n = N[0] * N[1] * N[2];
    std::vector<std::complex<double>> w((size_t)Mtot);
    for (int64_t c = 0; c < l; ++c) {
      std::fill(w.begin(), w.end(), std::complex<double>(0.0, 0.0));
      for (int64_t i = 0; i < n; ++i) {
        const int64_t i0 = i % N[0], r = i / N[0], i1 = r % N[1], i2 = r / N[1];
        w[(size_t)(i0 + M[0] * (i1 + M[1] * i2))] = X[i + c * ldx];
      }
      for (int a = 0; a < 3; ++a) dft_axis(w, M, a, -1.0);
      for (int64_t e = 0; e < Mtot; ++e) w[(size_t)e] *= p->lam[(size_t)e];
      for (int a = 2; a >= 0; --a) dft_axis(w, M, a, +1.0);
      for (int64_t i = 0; i < n; ++i) {
        const int64_t i0 = i % N[0], r = i / N[0], i1 = r % N[1], i2 = r / N[1];
        Y[i + c * ldy] = w[(size_t)(i0 + M[0] * (i1 + M[1] * i2))].real();
      }
    }
  }
  void lu_L(double* Y, int64_t m, int64_t l, int64_t ld, int32_t* ipiv) override {
    const int info = gsio_lu_L(Y, m, l, ld, ipiv);
    if (info && !lu_info_) lu_info_ = info;
  }
  // ---- row-sharded LU primitives: the arithmetic of gsio_lu_L (unblocked right-looking, multiplier = a * (1/pivot),
  //      a -= l * u in ascending column order), regrouped by leaves and blocks -- per element the same operations in the
  //      same order, so the sharded factorization equals the single-rank one bit for bit ----
  int lus_block() const override { return 32; }
  void lus_u12_leaf(const double* Y, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w, double* U12) override {
    const int64_t kp = j0 - jb;
    for (int v = 0; v < 8; ++v)
      for (int64_t c = 0; c < kp; ++c) {
        double x = (v < w) ? Y[(jb - row0 + c) + (j0 + v) * ld] : 0.0;
        for (int64_t cp = 0; cp < c; ++cp) x -= Y[(jb - row0 + c) + (jb + cp) * ld] * U12[cp * 8 + v];
        U12[c * 8 + v] = x;
      }
  }
  void lus_pending(double* Y, int64_t mloc, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w,
                   const double* U12) override {
    const int64_t kp = j0 - jb;
    for (int64_t li = 0; li < mloc; ++li) {
      if (row0 + li < j0) continue;
      for (int k = 0; k < w; ++k) {
        double a = Y[li + (j0 + k) * ld];
        for (int64_t c = 0; c < kp; ++c) a -= Y[li + (jb + c) * ld] * U12[c * 8 + k];
        Y[li + (j0 + k) * ld] = a;
      }
    }
  }
  void lus_candidate(const double* Y, int64_t mloc, int64_t ld, int64_t row0, int64_t l, int64_t j, double* rec) override {
    double best = -1.0;
    int64_t bi = -1;
    for (int64_t li = 0; li < mloc; ++li)
      if (row0 + li >= j) {
        const double av = std::fabs(Y[li + j * ld]);
        if (av > best) { best = av; bi = row0 + li; }
      }
    const bool has_j = (j >= row0 && j < row0 + mloc);
    rec[0] = best; rec[1] = (double)bi; rec[2] = has_j ? 1.0 : 0.0; rec[3] = 0.0;
    for (int64_t c = 0; c < l; ++c) {
      rec[4 + c] = (bi >= 0) ? Y[(bi - row0) + c * ld] : 0.0;
      rec[4 + l + c] = has_j ? Y[(j - row0) + c * ld] : 0.0;
    }
  }
  void lus_apply(double* Y, int64_t mloc, int64_t ld, int64_t row0, int64_t m, int64_t l, int64_t j0, int s, int w,
                 const double* recs, int nranks) override {
    const int64_t j = j0 + s, reclen = 4 + 2 * l;
    double best = -1.0;
    int64_t bi = -1;
    int gw = -1, go = -1;
    for (int g = 0; g < nranks; ++g) {
      const double v = recs[g * reclen];
      const int64_t i = (int64_t)recs[g * reclen + 1];
      if (i >= 0 && (v > best || (v == best && i < bi))) { best = v; bi = i; gw = g; }
      if (recs[g * reclen + 2] != 0.0) go = g;
    }
    const bool valid = (bi >= j && bi < m && gw >= 0);
    const int64_t r = valid ? bi : j;
    const double* prow = valid ? recs + gw * reclen + 4 : recs + go * reclen + 4 + l;
    const double* orow = recs + go * reclen + 4 + l;
    if ((int64_t)lus_ipiv_.size() < l) lus_ipiv_.resize((size_t)l);
    lus_ipiv_[(size_t)j] = (int32_t)r;
    if (best == 0.0 || !(best > 0.0)) { if (!lu_info_) lu_info_ = (int)(j + 1); }
    const double piv = prow[j];
    const bool singular = !(best > 0.0);                  // gsio_lu_L skips the column on an exactly zero pivot
    const bool has_j = (j >= row0 && j < row0 + mloc), has_r = (r >= row0 && r < row0 + mloc);
    if (!singular && r != j) {
      std::vector<double> pr(prow, prow + l), orw(orow, orow + l);
      if (has_j) for (int64_t c = 0; c < l; ++c) Y[(j - row0) + c * ld] = pr[(size_t)c];
      if (has_r) for (int64_t c = 0; c < l; ++c) Y[(r - row0) + c * ld] = orw[(size_t)c];
    }
    if (singular) return;
    const double rp = 1.0 / piv;
    for (int64_t li = 0; li < mloc; ++li) {
      if (row0 + li <= j) continue;
      Y[li + j * ld] *= rp;
      const double lij = Y[li + j * ld];
      for (int k = s + 1; k < w; ++k) {
        const double u = prow[j0 + k];
        if (u != 0.0) Y[li + (j0 + k) * ld] -= lij * u;
      }
    }
  }
  void lus_u12_block(const double* Y, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t c1,
                     double* U12) override {
    for (int64_t c = c0; c < c1; ++c)
      for (int r = 0; r < b; ++r) {
        double x = Y[(jb - row0 + r) + c * ld];
        for (int p = 0; p < r; ++p) x -= Y[(jb - row0 + r) + (jb + p) * ld] * U12[p + (c - c0) * b];
        U12[r + (c - c0) * b] = x;
      }
  }
  void lus_rankk(double* Y, int64_t mloc, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t t,
                 const double* U12) override {
    for (int64_t li = 0; li < mloc; ++li) {
      if (row0 + li < c0) continue;
      for (int64_t c = 0; c < t; ++c) {
        double a = Y[li + (c0 + c) * ld];
        for (int p = 0; p < b; ++p) {
          const double u = U12[p + c * b];
          if (u != 0.0) a -= Y[li + (jb + p) * ld] * u;
        }
        Y[li + (c0 + c) * ld] = a;
      }
    }
  }
  void lus_finish(double* Y, int64_t mloc, int64_t ld, int64_t row0, int64_t l) override {
    for (int64_t c = 0; c < l; ++c)
      for (int64_t r = 0; r <= c; ++r)
        if (r >= row0 && r < row0 + mloc) Y[(r - row0) + c * ld] = (r == c) ? 1.0 : 0.0;
  }
  void lus_pivots(int32_t* host, int64_t l) override {
    for (int64_t i = 0; i < l; ++i) host[i] = (i < (int64_t)lus_ipiv_.size()) ? lus_ipiv_[(size_t)i] : 0;
  }
  void qr_thinQ(double* Y, int64_t m, int64_t l, int64_t ld, double* R, bool /*replicated*/) override {
    gsio_qr_thinQ(Y, m, l, ld, 0, R, nullptr);   // unpivoted, like the HIP backend (same range)
  }
  void svd_small(double* G, int64_t l, double* U, double* S) override {
    gsio_svd_tall(G, l, l, l, S);
    std::memcpy(U, G, sizeof(double) * (size_t)l * (size_t)l);
  }
  void chol_upper(double* B, int64_t j) override {
    const int info = gsio_chol_upper(B, j);
    if (info && !chol_info_) chol_info_ = info;
  }
  void trsm_right_upper(double* F, int64_t m, int64_t j, int64_t ldf, const double* C) override {
    gsio_trsm_right_upper(F, m, j, ldf, C);
  }
  void scale_cols_sqrt(double* U, int64_t l, const double* S, int64_t K) override {
    for (int64_t c = 0; c < l; ++c) {
      const double s = c < K ? std::sqrt(S[c]) : 0.0;
      for (int64_t i = 0; i < l; ++i) U[i + c * l] *= s;
    }
  }
  void center_rows(double* S, int64_t n, int64_t N, int64_t ld) override { gsio_center_rows(S, n, N, ld); }
  void randn(double* p, size_t count, uint64_t seed) override {
    std::mt19937_64 g(seed);
    std::normal_distribution<double> d;
    for (size_t i = 0; i < count; ++i) p[i] = d(g);
  }
  void fill_gridcov(double* A, int64_t lda, int64_t nx, int64_t ny, double ell, int kind, int64_t row0,
                    int64_t mloc) override {
    const int64_t n = nx * ny;
    for (int64_t c = 0; c < n; ++c)
      for (int64_t r = 0; r < mloc; ++r) {
        const int64_t gi = row0 + r;
        const double dx = (double)(gi / ny) - (double)(c / ny), dy = (double)(gi % ny) - (double)(c % ny);
        const double d2 = dx * dx + dy * dy;
        A[r + c * lda] = kind == 0 ? std::exp(-d2 / (2 * ell * ell)) : std::exp(-std::sqrt(d2) / ell);
      }
  }
  void fill_lowrank_samples(double* S, int64_t ld, int64_t nloc, int64_t N, int64_t row0, uint64_t seed,
                            double decay) override {
    // any generator addressed by the global (row, sample) index will do here (not the HIP backend's stream)
    for (int64_t j = 0; j < N; ++j)
      for (int64_t r = 0; r < nloc; ++r) {
        std::mt19937_64 g(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(row0 + r + 1)) ^ (0xC2B2AE3D27D4EB4Full * (uint64_t)(j + 1)));
        std::normal_distribution<double> d;
        S[r + j * ld] = d(g) * std::pow((double)(j + 1), -decay);
      }
  }
  void colnorms(const double* Y, int64_t m, int64_t c, int64_t ld, double* out) override {
    for (int64_t j = 0; j < c; ++j) {
      double s = 0.0;
      for (int64_t i = 0; i < m; ++i) s += Y[i + j * ld] * Y[i + j * ld];
      out[j] = std::sqrt(s);
    }
  }
  void axpy(int64_t n, double a, const double* x, double* y) override { for (int64_t i = 0; i < n; ++i) y[i] += a * x[i]; }
  double dot(int64_t n, const double* x, const double* y) override { double s = 0; for (int64_t i = 0; i < n; ++i) s += x[i] * y[i]; return s; }
  double nrm2(int64_t n, const double* x) override { return std::sqrt(dot(n, x, x)); }
  void scal_copy(int64_t n, double a, const double* x, double* y) override { for (int64_t i = 0; i < n; ++i) y[i] = a * x[i]; }
  void pcga_params(const double* Z, int64_t n, int64_t K, const double* s, const double* X, double delta,
                   double* out) override {
    for (int64_t c = 0; c < K + 3; ++c)
      for (int64_t i = 0; i < n; ++i) {
        const double d = c < K ? Z[i + c * n] : (c == K ? X[i] : (c == K + 1 ? s[i] : 0.0));
        out[i + c * n] = s[i] + delta * d;
      }
  }
  void gemv_n(int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double beta,
              double* y) override {
    gsio_gemm_nn(m, 1, k, alpha, A, lda, x, k > 0 ? k : 1, beta, y, m > 0 ? m : 1);
  }
  void gemv_t(int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double* y) override {
    gsio_gemm_tn(k, 1, m, alpha, A, lda, x, m > 0 ? m : 1, 0.0, y, k > 0 ? k : 1);
  }
  void project_out(int64_t m, int64_t ncols, const double* q, double* Y, int64_t ld) override {
    for (int64_t c = 0; c < ncols; ++c) {
      const double d = dot(m, q, Y + c * ld);
      axpy(m, -d, q, Y + c * ld);
    }
  }
  void scal(int64_t n, double a, double* x) override { for (int64_t i = 0; i < n; ++i) x[i] *= a; }
  void diag_mul_add(int64_t n, const double* d, const double* x, double* y) override { for (int64_t i = 0; i < n; ++i) y[i] += d[i] * x[i]; }
  // lsqr with the scalars in the state block (lsqr_state.hpp): the same functions the device kernels call
  size_t lsqr_work_doubles() override { return (size_t)lsqrst::COUNT + 3; }
  void lsqr_begin(int64_t n, const double* w, double* work) override { work[lsqrst::COUNT + 2] = dot(n, w, w); }
  void lsqr_step_u(int64_t m, const double* t, double* u, double* work) override {
    using namespace lsqrst;
    double* s = work;
    double usq = 0.0;
    if (s[STOPPED] == 0.0 && s[ITERS] < s[MAXITER]) {
      for (int64_t i = 0; i < m; ++i) { u[i] = t[i] - s[ALPHA] * u[i]; usq += u[i] * u[i]; }
    }
    after_u(s, usq);
    if (s[APPLY] != 0.0 && s[BETA] > 0.0) { const double inv = 1.0 / s[BETA]; for (int64_t i = 0; i < m; ++i) u[i] *= inv; }
  }
  void lsqr_step_v(int64_t n, const double* t, double* v, double* w, double* x, double* work) override {
    using namespace lsqrst;
    double* s = work;
    double vsq = 0.0;
    if (s[APPLY] != 0.0 && s[BETA] > 0.0)
      for (int64_t i = 0; i < n; ++i) { v[i] = t[i] - s[BETA] * v[i]; vsq += v[i] * v[i]; }
    after_v(s, vsq, work[COUNT + 2]);
    if (s[APPLY] == 0.0) return;
    const bool rescale = (s[BETA] > 0.0 && s[ALPHA] > 0.0);
    const double inv = rescale ? 1.0 / s[ALPHA] : 1.0;
    double wsq = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      if (rescale) v[i] *= inv;
      x[i] += s[T1] * w[i];
      w[i] = v[i] + s[T2] * w[i];
      wsq += w[i] * w[i];
    }
    work[COUNT + 2] = wsq;
  }
  void f64_to_f32(const double* src, void* dst32, size_t count) override {
    float* d = (float*)dst32;
    for (size_t i = 0; i < count; ++i) d[i] = (float)src[i];
  }
  void pcga_params_f32(const void* Z32, int64_t n, int64_t K, const double* s, const double* X, double delta,
                       double* out) override {
    const float* Z = (const float*)Z32;
    for (int64_t c = 0; c < K + 3; ++c)
      for (int64_t i = 0; i < n; ++i) {
        const double d = c < K ? (double)Z[i + c * n] : (c == K ? X[i] : (c == K + 1 ? s[i] : 0.0));
        out[i + c * n] = s[i] + delta * d;
      }
  }
  void basis_gemv_f32(const void* Z32, int64_t n, int64_t K, const double* w, double beta, const double* X,
                      double* y) override {
    const float* Z = (const float*)Z32;
    for (int64_t i = 0; i < n; ++i) {
      double acc = beta * X[i];
      for (int64_t k = 0; k < K; ++k) acc += (double)Z[i + k * n] * w[k];
      y[i] = acc;
    }
  }
  int take_error(std::string* msg) override {
    if (lu_info_) {
      if (msg) *msg = "SingularException(" + std::to_string(lu_info_) + "): exactly zero pivot in lu()";
      lu_info_ = chol_info_ = 0;
      return GSI_ERR_SINGULAR;
    }
    if (chol_info_) {
      if (msg) *msg = "PosDefException: matrix is not positive definite; Cholesky failed at " + std::to_string(chol_info_);
      chol_info_ = 0;
      return GSI_ERR_NOT_POSDEF;
    }
    return 0;
  }
  void profile(bool) override {}
  void phase_begin(Phase) override {}
  void phase_end(Phase p) override { counts_[p] += 1; }
  void phase_reset() override { for (auto& c : counts_) c = 0; }
  void phase_times(double* ms, int64_t* counts) override { for (int i = 0; i < PH_COUNT; ++i) { ms[i] = 0.0; counts[i] = counts_[i]; } }

 private:
  std::vector<std::pair<double*, size_t>> sizes_;
  int64_t in_use_ = 0;
  int lu_info_ = 0, chol_info_ = 0;
  std::vector<int32_t> lus_ipiv_;
  int64_t counts_[PH_COUNT] = {0};
};

class CallbackComm : public Comm {
 public:
  CallbackComm(int n, int r) { nranks = n; rank = r; }
  void do_allreduce_sum(double* buf, size_t count) override {
    if (!g_allreduce) throw Error(GSI_ERR_RCCL, "cpuref: collectives not registered");
    g_allreduce(buf, (int64_t)count);
  }
  void do_allgather(const double* send, double* recv, size_t count) override {
    if (!g_allgather) throw Error(GSI_ERR_RCCL, "cpuref: collectives not registered");
    g_allgather(send, recv, (int64_t)count);
  }
  void do_reduce_scatter_sum(const double* send, double* recv, size_t count) override {   // all-reduce, keep own block
    if (!g_allreduce) throw Error(GSI_ERR_RCCL, "cpuref: collectives not registered");
    std::vector<double> tmp(send, send + count * (size_t)nranks);
    g_allreduce(tmp.data(), (int64_t)tmp.size());
    std::memcpy(recv, tmp.data() + count * (size_t)rank, count * sizeof(double));
  }
  void do_alltoall(const double* send, double* recv, size_t count) override {             // all-gather, keep the blocks meant for me
    if (!g_allgather) throw Error(GSI_ERR_RCCL, "cpuref: collectives not registered");
    std::vector<double> all(count * (size_t)nranks * (size_t)nranks);
    g_allgather(send, all.data(), (int64_t)(count * (size_t)nranks));
    for (int s = 0; s < nranks; ++s)
      std::memcpy(recv + count * (size_t)s, all.data() + count * ((size_t)s * nranks + (size_t)rank), count * sizeof(double));
  }
};
}  // namespace

Backend* make_backend(int) { return new CpuBackend(); }
Comm* make_comm(Backend*, int nranks, int rank, const void*) { return new CallbackComm(nranks, rank); }
void comm_unique_id(void* id_out) { std::memset(id_out, 0, GSI_UNIQUE_ID_BYTES); }
const char* backend_name() { return "cpu-reference (test only)"; }
}  // namespace gsi
