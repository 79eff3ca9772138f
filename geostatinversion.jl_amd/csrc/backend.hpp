// backend.hpp -- the seam between the host-side algorithm (pipeline.cpp: the order of
// operations of RandMatFact.jl, row-sharded) and the code that executes each operation.
// The product library links exactly one implementation, the HIP/gfx950 one
// (hip_backend.hip).  A second implementation over the C oracle exists ONLY in the test
// tree (oracle/cpu_backend.cpp -> oracle/_build/libgsi_cpuref.so) so that the sharded
// pipeline and the C ABI can be exercised on machines without a GPU; the product loader
// never opens it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

namespace gsi {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

enum Phase : int {
  PH_GEMM_N = 0, PH_GEMM_T = 1, PH_LU = 2, PH_QR = 3, PH_SVD = 4, PH_SMALL_GEMM = 5,
  PH_COMM = 6, PH_OTHER = 7, PH_COMM_WAIT = 8, PH_COUNT = 9
};

// All pointers below are "backend memory" (HBM for the HIP backend), column-major fp64.
class Backend {
 public:
  virtual ~Backend() {}
  virtual const char* name() const = 0;

  // ---- memory ----
  virtual double* alloc(size_t count) = 0;  // throws Error(GSI_ERR_OOM)
  virtual void release(double* p) = 0;
  virtual int64_t bytes_in_use() const = 0;
  virtual void release_cache() {}            // return cached (released) blocks and idle workspaces to the driver
  virtual void upload2d(double* dst, int64_t ldd, const double* host, int64_t ldh, int64_t rows, int64_t cols) = 0;
  virtual void download2d(double* host, int64_t ldh, const double* src, int64_t lds, int64_t rows, int64_t cols) = 0;
  virtual void copy2d(double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols) = 0;
  virtual void fill_zero(double* p, size_t count) = 0;
  virtual void sync() = 0;
  // The same upload in ROW BLOCKS of upload_block_rows(rows, cols) rows, in the background: upload2d_begin returns at once
  // (a handle, or null when the backend uploaded synchronously), upload2d_wait_block makes the context's stream wait for
  // block b (rows [b * mb, (b + 1) * mb)), upload2d_end returns when the host buffer is no longer read.  What lets the
  // first pass over a dense host matrix run under its own upload (gsi_randsvd_dense_host; getxis(Q::Matrix, ...),
  // GeostatInversion.jl:63-70).  The default implementation is the synchronous upload.
  virtual int64_t upload_block_rows(int64_t rows, int64_t cols) { (void)cols; return rows; }
  virtual void* upload2d_begin(double* dst, int64_t ldd, const double* host, int64_t ldh, int64_t rows, int64_t cols,
                               int64_t block_rows) {
    (void)block_rows;
    upload2d(dst, ldd, host, ldh, rows, cols);
    return nullptr;
  }
  // GB/s of one plain copy of `bytes` from / to pinned host memory on this context's device: the ceiling the staged
  // transfers are measured against (gsi_ctx_pinned_copy_rate).  0: not measurable on this backend.
  virtual void pinned_copy_rate(int64_t bytes, double* h2d_gbs, double* d2h_gbs) { (void)bytes; *h2d_gbs = 0.0; *d2h_gbs = 0.0; }
  virtual void upload2d_wait_block(void* handle, int64_t b) { (void)handle; (void)b; }
  virtual void upload2d_end(void* handle) { (void)handle; }

  // ---- products ----
  // C(m x l) = alpha * A(m x k) * B(k x l) + beta * C      beta in {0, 1}
  virtual void gemm_nn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double beta, double* C, int64_t ldc) = 0;
  // rows [r0, r0 + mb) of the product C(m_full x l) = A(m_full x k) * B that gemm_nn would compute -- the same reduction
  // order per element (the K split is the one the FULL shape gets), so the blocks put together are bit-identical to the one
  // launch.  A, C: the full matrices (element (0, 0)); r0 a multiple of 128.
  virtual void gemm_nn_rowblock(int64_t m_full, int64_t r0, int64_t mb, int64_t l, int64_t k, const double* A, int64_t lda,
                                const double* B, int64_t ldb, double* C, int64_t ldc) {
    (void)m_full;
    gemm_nn(mb, l, k, 1.0, A + r0, lda, B, ldb, 0.0, C + r0, ldc);
  }
  // C(m x l) = alpha * A'(m x k) * B(k x l) + beta * C,  A stored k x m
  virtual void gemm_tn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda,
                       const double* B, int64_t ldb, double beta, double* C, int64_t ldc) = 0;

  // C(m x l) = G * B(k x l) for a stationary grid covariance G(i, j) = tab[|x_i - x_j| * ny + |y_i - y_j|],
  // grid point i = (i / ny, i % ny), rows roff.., reduction indices koff..; tab = the nx * ny table of the kernel over
  // grid offsets.  G is generated, never stored (the "implicit" operator).
  virtual void gemm_nn_gridcov(int64_t m, int64_t l, int64_t k, const double* tab, int64_t nx, int64_t ny,
                               int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C,
                               int64_t ldc) = 0;

  // C (m x l) = G * B(k x l) for the covariance of SCATTERED points, G(i, j) = sigma2 kfun(|p_i - p_j| / ell) (+ nugget if
  // i == j), rows roff.., reduction indices koff..; pts = d x npts coordinates (point i = column i) in backend memory;
  // kind / params: pointcov.hpp.  G is generated panel by panel, never stored whole (the "row-streamed" operator).
  virtual void gemm_nn_pointcov(int64_t m, int64_t l, int64_t k, const double* pts, int d, int kind, double ell,
                                double sigma2, double nugget, int64_t roff, int64_t koff, const double* B, int64_t ldb,
                                double* C, int64_t ldc) = 0;

  // Matrix-free stationary covariance on an N[0] x N[1] x N[2] grid (column-major point index) by circulant
  // embedding: A = R F^-1 diag(lambda) F R', lambda(k) = |k|^beta, unit diagonal (fft_cov.hip).  Opaque plan.
  // fftrf != 0: FFTRF.jl's own convention -- embedding of exactly 2 N[a] points, integer wavenumbers (FFTRF.jl:83-90)
  virtual void* fftcov_create(const int64_t N[3], double beta, int fftrf) = 0;
  virtual void fftcov_destroy(void* plan) = 0;
  // Y (n x l, ld ldy) = A X (n x l, ld ldx)
  virtual void fftcov_apply(void* plan, int64_t l, const double* X, int64_t ldx, double* Y, int64_t ldy) = 0;

  // ---- panel factorizations, in place ----
  // Y (m x l, ld) <- L of lu(Y) in pivoted row order; ipiv (device int32[l]) may be null.
  // Sets *singular_flag (backend int, see flags()) to j+1 on an exactly zero pivot.
  virtual void lu_L(double* Y, int64_t m, int64_t l, int64_t ld, int32_t* ipiv_host_or_null) = 0;
  // ---- the same factorization ROW-SHARDED (SURVEY.md 8e "sharded alternative"): primitives on this rank's rows
  //      [row0, row0 + mloc) of the m x l panel, ld = leading dimension of the local block; the exchange between ranks is
  //      pipeline.cpp:lu_panel_sharded.  Blocks of `lus_block()` columns, leaves of 8; per element the arithmetic of lu_L.
  //      Record (4 + 2 l doubles): [0] max |value| (-1: none), [1] global row, [2] 1 if this rank holds row j,
  //      [4, 4+l) that row, [4+l, 4+2l) row j.
  virtual int lus_block() const = 0;
  virtual void lus_u12_leaf(const double* Yloc, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w, double* U12) = 0;
  virtual void lus_pending(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w,
                           const double* U12) = 0;
  virtual void lus_candidate(const double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t l, int64_t j,
                             double* rec) = 0;
  virtual void lus_apply(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t m, int64_t l, int64_t j0, int s,
                         int w, const double* recs, int nranks) = 0;
  virtual void lus_u12_block(const double* Yloc, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t c1,
                             double* U12) = 0;
  virtual void lus_rankk(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t t,
                         const double* U12) = 0;
  virtual void lus_finish(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t l) = 0;
  virtual void lus_pivots(int32_t* host, int64_t l) = 0;      // the pivot rows of the last sharded factorization
  // ---- the sharded factorization with ONE persistent launch per 8-column leaf and rank: the 8 pivot exchanges of a leaf
  //      run inside the kernels, through records every rank writes into every rank's peer-mapped buffer (no collective per
  //      pivot step).  lus_mr_begin: can this backend do it over this communicator for an m x l panel (sets up and exchanges
  //      the record buffers on first use)?  false: pipeline.cpp falls back to the per-step form above.  U12 (kp x 8) of the
  //      leaf's pending update comes from lus_u12_leaf + all-reduce as before; lus_swap_pack / _apply move the rows the
  //      leaf's pivots exchange in the columns OUTSIDE the leaf (table: 16 x l doubles, all-reduced in between).
  virtual bool lus_mr_begin(class Comm* comm, int64_t m, int64_t l) { (void)comm; (void)m; (void)l; return false; }
  // The exchange has three forms -- 0: one hop, 1: two hops, 2: two hops with lazily evaluated overflow rows -- chosen by the
  // shard height.  lus_mr_mode: the form the last successful lus_mr_begin chose; lus_mr_force (self-test): the form the next
  // ones must take (1, 2) or the natural choice again (0).
  virtual int lus_mr_mode() { return 0; }
  virtual void lus_mr_force(int mode) { (void)mode; }
  // Launch geometry of the form the last successful lus_mr_begin chose, packed into one number (block size, rows per
  // thread, grid, hops, overflow rows): it depends on rank-local facts (CUs, ranks sharing the device), and ranks with
  // different geometries would address different record slots -- pipeline.cpp lets the path run only if all ranks report
  // the same number.  lus_mr_reason: why the last lus_mr_begin said no (for the error text of GSI_LU_MR_REQUIRE).
  virtual int64_t lus_mr_signature() { return 0; }
  virtual const char* lus_mr_reason() { return ""; }
  // Bumped whenever something that lus_mr_begin's answer depends on changes on this rank (a time-out switched the path
  // off, ...): agreements the ranks reached before are void and must be reached again.  Time-outs are made global
  // (lu_flag_export / _import below), so the ranks' counters move together.
  virtual int64_t lus_mr_generation() { return 0; }
  // Before the first kernel that spins for its peers: launch every kernel the persistent-leaf factorization uses once on
  // dummies, so that no code object is loaded (a runtime call that may wait for the device) while peers spin.
  virtual void lus_mr_warmup() {}
  // The LU's asynchronous time-out flag (info = -1) made GLOBAL without a host round trip: export writes 1.0 / 0.0 into a
  // backend double, the caller all-reduces it on the stream, import raises the flag on this rank if any rank had it up.
  virtual void lu_flag_export(double* flag) { (void)flag; }
  virtual void lu_flag_import(const double* flag) { (void)flag; }
  virtual void lus_leaf_mr(double*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, const double*) {}
  virtual bool lus_mr_swaps_done() { return false; }   // lus_leaf_mr also moved the rows in the other columns (peer pushes)
  virtual void lus_swap_pack(const double*, int64_t, int64_t, int64_t, int64_t, int64_t, int, double*) {}
  virtual void lus_swap_apply(double*, int64_t, int64_t, int64_t, int64_t, int64_t, int, const double*) {}
  // Y (m x l) <- thin Q; R (l x l, ld l) <- upper triangular factor if R != null.
  // replicated: the same panel is factored by every rank and the results must be bit-identical everywhere
  // (stacked R factors of the TSQR, gathered panels): the backend must then not let anything rank-local
  // (e.g. which algorithm tier an earlier rank-local panel needed) steer the choice of algorithm.
  virtual void qr_thinQ(double* Y, int64_t m, int64_t l, int64_t ld, double* R, bool replicated = false) = 0;
  // The thin Q of Y UP TO A RIGHT FACTOR: on success *Q1 (m x l, ld *ldq1, in backend workspace: valid until this
  // backend's next factorization call) and X2 (l x l upper triangular, caller's buffer) with Q = Q1 X2 orthonormal and
  // range(Q) = range(Y); Y is untouched.  What CholeskyQR2 has after its second Gram matrix, one tall product short of
  // Q.  A caller that only ever multiplies Q from the left (B = Q'A in randsvd) folds X2 into the l x l factor behind
  // it and never pays that product.  false: not available for this panel (the caller runs qr_thinQ).
  virtual bool qr_thinQ_deferred(const double* Y, int64_t m, int64_t l, int64_t ld, const double** Q1, int64_t* ldq1,
                                 double* X2) { (void)Y; (void)m; (void)l; (void)ld; (void)Q1; (void)ldq1; (void)X2; return false; }
  // thin SVD of a tall W (n x l, ld) in one go, W untouched: V (n x l, ldv) = left singular vectors (scaled by sqrt(S_i)
  // for i < K_scale, zero beyond, when K_scale >= 0), S (l).  A backend may decline (false): the caller then runs
  // qr_thinQ + svd_small + the l x l product itself.  The HIP backend fuses the last product of CholeskyQR2 with the
  // product by the small factor: Z = T (R2^-1 U sqrt(S)) -- one tall product instead of two.
  // Xr (may be null): an l x l right factor -- the SVD is that of W Xr (the X2 a deferred thin Q left behind).
  virtual bool svd_tall_fused(const double*, int64_t, int64_t, int64_t, int64_t, double*, int64_t, double*, const double* Xr = nullptr) {
    (void)Xr;
    return false;
  }
  // G (l x l, ld l), columns orthogonalised in place by one-sided Jacobi; on return
  // U (l x l) = left singular vectors sorted by descending S, S (l) singular values.
  virtual void svd_small(double* G, int64_t l, double* U, double* S) = 0;
  // upper Cholesky factor of the symmetric j x j B (upper triangle read), in place
  virtual void chol_upper(double* B, int64_t j) = 0;
  // F (m x j) <- F * inv(C), C upper triangular j x j
  virtual void trsm_right_upper(double* F, int64_t m, int64_t j, int64_t ldf, const double* C) = 0;

  // ---- small fused helpers ----
  // U (l x l) <- U * diag(sqrt(S_i) for i < K, 0 otherwise)
  virtual void scale_cols_sqrt(double* U, int64_t l, const double* S, int64_t K) = 0;
  // rows of S (n x N, ld): subtract the mean over the N columns (lowrank.jl:17-27)
  virtual void center_rows(double* S, int64_t n, int64_t N, int64_t ld) = 0;
  virtual void randn(double* p, size_t count, uint64_t seed) = 0;
  // synthetic covariance rows [row0,row0+mloc) of an (nx*ny)^2 grid covariance
  virtual void fill_gridcov(double* A, int64_t lda, int64_t nx, int64_t ny, double ell, int kind,
                            int64_t row0, int64_t mloc) = 0;
  // synthetic sample fields (benchmark / test input): S(i, j) = g(row0 + i, j) (j+1)^-decay, g iid N(0,1) addressed
  // by the global index (identical for every rank layout)
  virtual void fill_lowrank_samples(double* S, int64_t ld, int64_t nloc, int64_t N, int64_t row0, uint64_t seed,
                                    double decay) = 0;
  // column norms of Y (m x c) -> host array
  virtual void colnorms(const double* Y, int64_t m, int64_t c, int64_t ld, double* host_out) = 0;
  // y <- y - Q (Q' y) for Q m x j (classical Gram-Schmidt step of Alg 4.2), y length m
  // and out = y/||y|| if normalize_into != null
  virtual void axpy(int64_t n, double a, const double* x, double* y) = 0;
  virtual double dot(int64_t n, const double* x, const double* y) = 0;
  virtual double nrm2(int64_t n, const double* x) = 0;
  virtual void scal_copy(int64_t n, double a, const double* x, double* y) = 0;  // y = a*x
  // out (n x (K+3)) = [s + delta*Z[:,i] (i<K), s + delta*X, s + delta*s, s]   direct.jl:39-45
  virtual void pcga_params(const double* Z, int64_t n, int64_t K, const double* s, const double* X, double delta,
                           double* out) = 0;

  // BLAS-2 (adaptive range finder, saddle-point products): y (m) = alpha A x + beta y; y (k) = alpha A' x (A m x k);
  // Y[:, c] -= dot(q, Y[:, c]) q for c < ncols <= 64  (RandMatFact.jl:42-45) -- no scalar leaves the backend
  virtual void gemv_n(int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double beta,
                      double* y) = 0;
  virtual void gemv_t(int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double* y) = 0;
  virtual void project_out(int64_t m, int64_t ncols, const double* q, double* Y, int64_t ld) = 0;
  virtual void scal(int64_t n, double a, double* x) = 0;                          // x *= a
  virtual void diag_mul_add(int64_t n, const double* d, const double* x, double* y) = 0;   // y += d .* x

  // ---- IterativeSolvers.lsqr with the scalar recurrences resident in backend memory (lsqr_state.hpp): `work` holds
  //      lsqr_work_doubles() doubles = [state | partial sums]; pipeline.cpp:lsqr sequences the two operator products
  //      around these and polls the state every few iterations ----
  virtual size_t lsqr_work_doubles() = 0;
  virtual void lsqr_begin(int64_t n, const double* w, double* work) = 0;                       // |w|^2 of the initial w
  virtual void lsqr_step_u(int64_t m, const double* t, double* u, double* work) = 0;           // u = t - alpha u; beta; u /= beta
  virtual void lsqr_step_v(int64_t n, const double* t, double* v, double* w, double* x, double* work) = 0;
  //                                     v = t - beta v; alpha; the recurrences and stopping rules; v /= alpha; x += t1 w; w = v + t2 w

  // ---- fp32-STORED xi-basis (BASELINE configs[4], "fp32 mixed precision"): the n x K basis is the one big HBM stream
  //      of the PCGA iteration's own algebra; stored in fp32 it is half the bytes, every sum stays in fp64 ----
  virtual void f64_to_f32(const double* src, void* dst32, size_t count) = 0;
  // as pcga_params with Z given in fp32 (n x K, ld n); out is fp64
  virtual void pcga_params_f32(const void* Z32, int64_t n, int64_t K, const double* s, const double* X, double delta,
                               double* out) = 0;
  // y (n) = beta * X + Z32 (n x K, fp32) * w (K): fp64 accumulation   (direct.jl:59-65 with the basis in fp32)
  virtual void basis_gemv_f32(const void* Z32, int64_t n, int64_t K, const double* w, double beta, const double* X,
                              double* y) = 0;

  // error flags raised asynchronously by kernels (zero pivot, non-posdef); checked and
  // cleared by the pipeline at the end of each entry point. Returns GSI_* code or 0.
  virtual int take_error(std::string* msg) = 0;
  // true (once) if the last error was a lost resource rather than a property of the input, and the backend has switched
  // to a path that does not need it: the caller may run the same entry point again on its intact inputs
  virtual bool retryable_failure() { return false; }
  virtual void forgive_lost_coresidency() {}     // undo what a time-out inside a self-test switched off

  // ---- profiling ----
  virtual void profile(bool on) = 0;
  virtual void phase_begin(Phase p) = 0;
  virtual void phase_end(Phase p) = 0;
  virtual void phase_reset() = 0;
  virtual void phase_times(double* ms, int64_t* counts) = 0;
  // [0] CholeskyQR2 factorizations, [1] Householder factorizations (incl. fallbacks), [2] Jacobi sweeps of
  // the last small SVD, [3] shifted CholeskyQR3 factorizations
  virtual void counters(int64_t* out4) { out4[0] = out4[1] = out4[2] = out4[3] = 0; }
  // [0] LU pivot-exchange time-outs seen by this backend since creation (take_error mapped info = -1)
  virtual int64_t lu_timeouts() { return 0; }
  // small SVDs that used up their sweep budget with rotatable column pairs left (the result is still the best available)
  virtual int64_t svd_cap_hits() { return 0; }
  // how many ranks of the context's communicator run on this backend's device (Comm::ranks_on_my_device, told once the
  // communicator exists): kernels that need all their workgroups resident at once must not assume the whole chip
  virtual void set_ranks_sharing_device(int n) { (void)n; }
};

class Comm {
 public:
  int rank = 0, nranks = 1;
  int64_t ncollectives = 0;       // collectives entered since creation (gsi_ctx_path_info reports the count per timed region)
  virtual ~Comm() {}
  // The four collectives.  Callers use these; implementations override the do_* hooks below.
  void allreduce_sum(double* buf, size_t count) { ++ncollectives; do_allreduce_sum(buf, count); }
  // every rank contributes `count` doubles; recv holds nranks*count, rank-major
  void allgather(const double* send, double* recv, size_t count) { ++ncollectives; do_allgather(send, recv, count); }
  // send holds nranks blocks of `count` doubles (block g is destined for rank g); recv (count) = sum over ranks
  // of their block `rank`
  void reduce_scatter_sum(const double* send, double* recv, size_t count) { ++ncollectives; do_reduce_scatter_sum(send, recv, count); }
  // send holds nranks blocks of `count` doubles, block g for rank g; recv block s = what rank s sent to this rank
  void alltoall(const double* send, double* recv, size_t count) { ++ncollectives; do_alltoall(send, recv, count); }
  // every rank hands in a device buffer; all[g] = rank g's buffer as THIS rank can address it (same process: the pointer
  // itself; other processes: an IPC mapping).  false: not supported by this communicator.  Collective.
  virtual bool share_pointers(void* mine, size_t bytes, void** all) { (void)mine; (void)bytes; (void)all; return false; }
  // ranks of this communicator that run on the SAME device as this one (itself included).  1 with RCCL (it refuses two ranks
  // on one device); the RCCL-free communicators allow any placement, and kernels that need all their workgroups resident
  // at once (the persistent LU leaves) must share the chip between that many launches.
  virtual int ranks_on_my_device() { return 1; }
  // Ranks that are THREADS of one process share one HIP runtime: a call that waits for the device (hipFree does, and a
  // hipMalloc that has to map new memory was seen to) waits for the other ranks' kernels too.  Before a rank launches a
  // kernel that spins for its peers, all ranks meet here with their allocations done.  No-op for ranks in processes.
  virtual void host_barrier() {}

 protected:
  virtual void do_allreduce_sum(double* buf, size_t count) = 0;
  virtual void do_allgather(const double* send, double* recv, size_t count) = 0;
  virtual void do_reduce_scatter_sum(const double* send, double* recv, size_t count) = 0;
  virtual void do_alltoall(const double* send, double* recv, size_t count) = 0;
};

// Provided by whichever backend is linked into the library.
Backend* make_backend(int device_id);
Comm* make_comm(Backend* be, int nranks, int rank, const void* unique_id);
void comm_unique_id(void* id_out);
const char* backend_name();

struct ScopedPhase {
  Backend* be; Phase p;
  ScopedPhase(Backend* b, Phase ph) : be(b), p(ph) { be->phase_begin(p); }
  ~ScopedPhase() { be->phase_end(p); }
};

}  // namespace gsi
