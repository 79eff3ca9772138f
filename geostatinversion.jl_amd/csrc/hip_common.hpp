// hip_common.hpp -- declarations shared by the gfx950 kernel files and hip_backend.hip
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstddef>
#include "pointcov.hpp"

namespace gsi { namespace hipk {

// The > 64 KB dynamic-LDS opt-in (hipFuncSetAttribute) is a property of a kernel ON ONE DEVICE.  A process may
// hold one context per GPU (one thread per GPU, include/gsi_hip.h), so "done once" is tracked per device: one
// bit per device ordinal, one mask per kernel instantiation.  Racing first calls both set the attribute (idempotent).
inline bool first_use_on_this_device(std::atomic<uint64_t>& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;   // unknown: set the attribute again
  const uint64_t bit = (uint64_t)1 << dev;
  return (mask.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
}

// ---- gemm_f64.hip ----
size_t gemm_workspace_doubles(int64_t M, int64_t L, int64_t K);
void gemm_f64(hipStream_t st, bool transA, int64_t M, int64_t L, int64_t K, double alpha, const double* A,
              int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, double* ws);
// rows [r0, r0 + mb) of the NN product of an M_full-row launch, with that launch's K split (bit-identical blocks)
size_t gemm_rowblock_workspace_doubles(int64_t M_full, int64_t mb, int64_t L, int64_t K);
void gemm_f64_nn_rowblock(hipStream_t st, int64_t M_full, int64_t r0, int64_t mb, int64_t L, int64_t K, const double* A,
                          int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double* ws);
// C (l x l) = A'A, upper-triangle tiles only; C = A * B with B upper triangular (CholeskyQR: half the flops each)
size_t gemm_syrk_workspace_doubles(int64_t l, int64_t m);
// syrk_f64.hip: G = A'A (both triangles) for l <= 320 by the register-resident kernel; false = shape not covered
size_t syrk_upper_workspace_doubles(int64_t l, int64_t m);
bool syrk_full_from_upper(hipStream_t st, int64_t l, int64_t m, const double* Y, int64_t ld, double* G, int64_t ldg, double* ws);
// C (m x l) = A X, X (l x l) upper triangular, l <= 320, C not aliasing A; false = shape not covered
bool trmm_upper_tall(hipStream_t st, int64_t m, int64_t l, const double* A, int64_t lda, const double* X, int64_t ldx, double* C,
                     int64_t ldc);
void gemm_f64_syrk_upper(hipStream_t st, int64_t l, int64_t m, const double* A, int64_t lda, double* C, int64_t ldc,
                         double* ws);
void gemm_f64_trmm_upper(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* A, int64_t lda, const double* B,
                         int64_t ldb, double* C, int64_t ldc, double* ws);
// C = G * B, G(i,k) = tab[|x_i-x_k| * ny + |y_i-y_k|] generated in registers (tab: nx * ny kernel table)
void gemm_f64_gridcov(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* tab, int64_t nx, int64_t ny,
                      int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C, int64_t ldc, double* ws);

// ---- the scattered-point covariance (gemm_f64.hip GEN 2, pointcov_gemm.hip) ----
// the scattered-point covariance generated inside the contraction's tile loader (gemm_f64.hip, GEN 2); pts4: 32-byte records
void gemm_f64_pointcov(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* pts4, int64_t npts, int d, int kind,
                       double sigma2, double nugget, int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C,
                       int64_t ldc, double* ws, double* xpack);
double pointcov_point_scale(int kind, double inv_ell);
// pointcov_gemm.hip: the same product with 96-row x 320-column tiles (every entry generated once for sketches of up to 320
// columns); false = not applicable (L <= 160, GSI_POINTCOV_WIDE=0, X beyond 32-bit tile offsets)
bool gemm_f64_pointcov_wide(hipStream_t st, int64_t M, int64_t L, int64_t K, const double* pts4, int64_t npts, int d, int kind,
                            double sigma2, double nugget, int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C,
                            int64_t ldc, double* ws, double* xpack);
size_t gemm_pointcov_workspace_doubles(int64_t M, int64_t L, int64_t K);   // covers both tilings
size_t gemm_pointcov_pack_doubles(int64_t M, int64_t L, int64_t K);        // the wide kernel's packed copy of X (0: not its product)
int gemm_choose_split(int64_t nwg, int64_t K);
// C = sum over the nsplit slabs (M x L each, leading dimension M), fixed order
void gemm_splitk_reduce(hipStream_t st, int64_t M, int64_t L, int nsplit, const double* slabs, double* C, int64_t ldc);
void pointcov_pad_points(hipStream_t st, const double* pts, int d, int64_t n, double scale, double* out4);
// ---- fft_cov.hip ----
int64_t fft_embed_size(int64_t N);   // next power of two >= 2 N (1 for a singleton axis)
size_t fft_plan_doubles(const int64_t M[3]);   // spectrum + scratch + twiddle table
// lam (fft_plan_doubles) <- |k|^beta / sum, scratch part64 = lam + Mtot, twiddles behind it
void fft_spectrum(hipStream_t st, double* lam, double* part64, const int64_t M[3], double beta, int fftrf);
// FFTRF's 2 N embedding on grids that are not powers of two (fft_cov.hip): pieces of the re-embedded spectrum
void fft_cos_matrix(hipStream_t st, double* out, int64_t rows, int64_t cols, int64_t period, bool weighted);
void fft_lines_layout(hipStream_t st, const double* nat, double* out, const int64_t M[3]);
void fft_spectrum_natural(hipStream_t st, double* lam, const int64_t M[3], double beta, int fftrf);
void fft_finish_plan(hipStream_t st, double* lam, double* part64, const int64_t M[3]);
void fft_cov_apply(hipStream_t st, const int64_t N[3], const int64_t M[3], const double* lam, double2* W, int nb_max,
                   int64_t l, const double* X, int64_t ldx, double* Y, int64_t ldy);

// ---- panel_lu_leaf.hip: register-resident leaves + streaming rank-K updates (panels of <= 4096 rows per CU) ----
constexpr int LU2_LEAF = 8;           // leaf width: columns a thread keeps in registers
constexpr int LU2_NB = 64;            // widest block (left-looking leaves inside, one rank-NB update per block)
constexpr int LU2_RES_COPIES = 8;     // copies of the per-step result record (one per group of pollers)
constexpr int LU2_REC_GRANULES = 64;  // one published record per workgroup and pivot step: 64 8-byte granules (512 B)
struct Lu2Work {
  unsigned long long* recs;   // [2][grid][LU2_REC_GRANULES] candidate records, then [2][LU2_RES_COPIES][LU2_REC_GRANULES] results
  double* u12;      // [nb * l]
  int32_t* ipiv;    // [l]
  int32_t* info;    // [1] first exactly-zero pivot (1-based); -1: the exchange between workgroups timed out
  int bs, rpt, grid, nb;
  int poll_limit = 0;          // polls before a workgroup gives up on a record (0: the default, ~ seconds)
  uint32_t mute_epoch = 0;     // tests only: the last workgroup publishes nothing at this pivot step (1-based)
  bool cooperative = false;    // launch the leaves with hipLaunchCooperativeKernel (co-residency guaranteed by the runtime)
  bool ov = false;             // the panel is taller than grid x 4096 rows: <512, 8> leaves with lazily evaluated overflow rows
};
int lu2_resident_per_cu_ov();
// workgroups of the (bs, rpt) leaf kernel that fit one CU (occupancy query); 0 if the query fails
int lu2_resident_per_cu(int bs, int rpt);
// the same leaf kernel across RANKS (one launch per rank, records written into every rank's peer-mapped buffer)
constexpr int LU2_MAX_RANKS = 16;
constexpr int LU2_MR_MAXL = 8192;        // widest panel of the multi-rank persistent-leaf path (row boxes)
struct Lu2MrWork {
  unsigned long long* peer[LU2_MAX_RANKS];   // every rank's record buffer: [2][nranks * grid][LU2_REC_GRANULES]
  int32_t* ipiv;                             // [l] this rank's copy of the pivot rows (identical on every rank)
  int32_t* info;
  int rank, nranks, bs, rpt, grid;
  int hier = 0;                              // two-hop exchange (ranks reduce among their own workgroups first)
  int ov = 0;                                // the shard is taller than grid x 4096 rows: overflow rows evaluated lazily
  int poll_limit = 0;
};
bool lu2_mr_config(int64_t pad, int nranks, int ncus, int* bs, int* rpt, int* grid, int* hier, int* ov, int force = 0);
int lu2_mr_resident_per_cu(int bs, int rpt);
int lu2_mr_resident_per_cu_ov();
size_t lu2_mr_record_granules(int nranks, int grid);
void lu2_leaf_mr(hipStream_t st, const Lu2MrWork& w, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t m, int64_t l,
                 int64_t jb, int64_t j0, int wd, const double* us, uint32_t epoch_base);
void lus_swap_peer(hipStream_t st, const Lu2MrWork& w, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t m, int64_t l,
                   int64_t j0, int wd, uint32_t epoch_base);
void lus_swap_pack(hipStream_t st, const double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l, int64_t j0, int w,
                   const int32_t* ipiv, double* table);
void lus_swap_apply(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l, int64_t j0, int w,
                    const int32_t* ipiv, const double* table);
// launch geometry for an m-row panel on a chip with `ncus` CUs; false: the panel does not fit the register file
bool lu2_config(int64_t m, int ncus, int* bs, int* rpt, int* grid);
void lu2_L(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, const Lu2Work& w);

// panels taller than the register file: streamed leaves, lazily evaluated (no spin-waits); `work` = lu3_work_bytes(l) bytes
size_t lu3_work_bytes(int64_t l);
void lu3_L(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, void* work, int32_t* info, int32_t** ipiv_out);

// row-sharded form (the exchange between ranks is pipeline.cpp's): primitives on this rank's rows [row0, row0 + mloc)
int lus_grid(int64_t mloc);
void lus_candidate(hipStream_t st, const double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l, int64_t j, double* rec,
                   double* pval, int64_t* pidx, bool partials_ready);
void lus_apply(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t m, int64_t l, int64_t j0, int s, int w,
               const double* recs, int nranks, int32_t* ipiv, int32_t* info, double* next_pval, int64_t* next_pidx);
void lus_u12_leaf(hipStream_t st, const double* Y, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w, double* U12);
void lus_pending(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t jb, int64_t j0, int w,
                 const double* U12);
void lus_u12_block(hipStream_t st, const double* Y, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t c1,
                   double* U12);
void lus_rankk(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t jb, int b, int64_t c0, int64_t t,
               const double* U12);
void lus_finish(hipStream_t st, double* Y, int64_t ld, int64_t mloc, int64_t row0, int64_t l);
// the factorization's time-out flag (info < 0) as a double (1.0 / 0.0) for an all-reduce on the stream, and back: raise
// info = -1 here if any rank had it up
void lu_flag_export(hipStream_t st, const int32_t* info, double* flag);
void lu_flag_import(hipStream_t st, int32_t* info, const double* flag);

// ---- lsqr_dev.hip: IterativeSolvers.lsqr's vector updates with device-resident scalars ----
size_t lsqr_work_doubles();
void lsqr_begin(hipStream_t st, int64_t n, const double* w, double* work);
void lsqr_step_u(hipStream_t st, int64_t m, const double* t, double* u, double* work);
void lsqr_step_v(hipStream_t st, int64_t n, const double* t, double* v, double* w, double* x, double* work);

// ---- panel_qr.hip ----
constexpr int QR_NB = 16;
struct QrWork {
  double* coef;     // [2 + NB]: scale, tau, tw[NB]
  double* part;     // [maxblocks * NB] partial sums
  double* tau;      // [l]
  double* T;        // [l * NB]   T factors, panel k at T + k*NB*NB
  double* G;        // [NB*NB]    V'V
  double* Vbuf;     // [m * NB]
  double* Wt;       // [l * NB]
  double* W2;       // [NB * l]
  double* Qo;       // [m * l]
  int64_t maxblocks;
};
int64_t qr_max_blocks(int64_t m);
void qr_thinQ(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, double* R, const QrWork& w,
              double* gemm_ws);

// ---- cholqr.hip ----
size_t cholqr_small_doubles(int64_t l);
void cholqr2_factor(hipStream_t st, const double* Y, int64_t m, int64_t l, int64_t ld, double* T, int64_t ldt,
                    double* small_ws, int32_t* flag, double* gemm_ws);
void cholqr2_apply(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, const double* T, int64_t ldt,
                   double* R, double* small_ws, double* gemm_ws);
void cholqr2_R(hipStream_t st, int64_t l, double* small_ws, double* R);         // R = R2 R1 after cholqr2_factor
const double* cholqr2_X2(const double* small_ws, int64_t l);                    // R2^-1 (l x l) after cholqr2_factor
void scholqr3_factor(hipStream_t st, const double* Y, int64_t m, int64_t l, int64_t ld, double* T, int64_t ldt,
                     double* S, int64_t lds, double* small_ws, int32_t* flag, double* gemm_ws);
void scholqr3_apply(hipStream_t st, double* Y, int64_t m, int64_t l, int64_t ld, const double* S, int64_t lds,
                    double* R, double* small_ws, double* gemm_ws);

// ---- jacobi_svd.hip ----
struct SvdWork {
  int32_t* rotcount;  // [1]
  double* norms;      // [l]
  int32_t* pairs;     // [SVD_SCHED_INTS] block-pair activity flags, then the sparse sweep's schedule (null: plain sweeps only)
};
constexpr int SVD_SCHED_INTS = 4096;
// returns number of sweeps used
int svd_small(hipStream_t st, double* G, int64_t l, double* U, double* S, const SvdWork& w);

// ---- misc.hip ----
void center_rows(hipStream_t st, double* S, int64_t n, int64_t N, int64_t ld);
void scale_cols_sqrt(hipStream_t st, double* U, int64_t l, const double* S, int64_t K);
void randn_fill(hipStream_t st, double* p, size_t count, uint64_t seed);
void fill_gridcov(hipStream_t st, double* A, int64_t lda, int64_t nx, int64_t ny, double ell, int kind,
                  int64_t row0, int64_t mloc);
void fill_lowrank_samples(hipStream_t st, double* S, int64_t ld, int64_t nloc, int64_t N, int64_t row0, uint64_t seed,
                          double decay);
void colnorms_sq(hipStream_t st, const double* Y, int64_t m, int64_t c, int64_t ld, double* out_dev);
void chol_upper(hipStream_t st, double* B, int64_t j, int32_t* info);
void trsm_right_upper(hipStream_t st, double* F, int64_t m, int64_t j, int64_t ldf, const double* C);
void axpy(hipStream_t st, int64_t n, double a, const double* x, double* y);
void scal_copy(hipStream_t st, int64_t n, double a, const double* x, double* y);
void dot_dev(hipStream_t st, int64_t n, const double* x, const double* y, double* out_dev);   // out_dev: >= 8 + 256 doubles
void scal(hipStream_t st, int64_t n, double a, double* x);
void gemv_n(hipStream_t st, int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double beta,
            double* y);
size_t gemv_t_workspace_doubles(int64_t k);
void gemv_t(hipStream_t st, int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double* y,
            double* part);
void project_out(hipStream_t st, int64_t m, int64_t ncols, const double* q, double* Y, int64_t ld, double* part);
void diag_mul_add(hipStream_t st, int64_t n, const double* d, const double* x, double* y);
void f64_to_f32(hipStream_t st, const double* src, float* dst, size_t count);
void pcga_params_f32(hipStream_t st, const float* Z, int64_t n, int64_t K, const double* s, const double* X, double delta,
                     double* out);
void basis_gemv_f32(hipStream_t st, const float* Z, int64_t n, int64_t K, const double* w, double beta, const double* X,
                    double* y);
void extract_upper(hipStream_t st, const double* Y, int64_t ld, int64_t l, double* R);
void pcga_params(hipStream_t st, const double* Z, int64_t n, int64_t K, const double* s, const double* X,
                 double delta, double* out);

}}  // namespace gsi::hipk
