// pipeline.hpp -- host-side order of operations of RandMatFact.jl over a gsi::Backend,
// row-sharded over gsi::Comm ranks.  No kernel code here.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <vector>
#include "backend.hpp"

namespace gsi {

// RAII panel in backend memory
struct Buf {
  Backend* be = nullptr;
  double* p = nullptr;
  size_t count = 0;
  Buf() {}
  Buf(Backend* b, size_t n) : be(b), p(b->alloc(n)), count(n) {}
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  Buf(Buf&& o) noexcept : be(o.be), p(o.p), count(o.count) { o.p = nullptr; }
  Buf& operator=(Buf&& o) noexcept {
    if (this != &o) { reset(); be = o.be; p = o.p; count = o.count; o.p = nullptr; }
    return *this;
  }
  ~Buf() { reset(); }
  void reset() { if (p) { be->release(p); p = nullptr; } }
};

struct Context {
  std::unique_ptr<Backend> be;
  std::unique_ptr<Comm> comm;  // null when single rank
  std::map<int64_t, bool> lus_mr_ok;   // panel height -> all ranks can run the sharded LU with in-kernel pivot exchange
  int lus_mr_selftest = -1;            // -1 not run yet; else bit f set: form f of the in-kernel exchange (0 one hop, 1 two hops,
                                       // 2 two hops + overflow rows) reproduced the per-step factors on this communicator
  int64_t lus_mr_gen = -1;             // Backend::lus_mr_generation() the entries of lus_mr_ok were agreed under
  // ---- which path ran (gsi_ctx_path_info) ----
  // form of a panel LU under a communicator: 0 none yet, 1 replicated (gathered panel, every rank factors it), 2 row-sharded
  // with one collective per pivot step, 3 / 4 / 5 row-sharded persistent leaves with the in-kernel exchange in one hop / two
  // hops / two hops + overflow rows
  enum LuForm { LU_NONE = 0, LU_REPLICATED = 1, LU_PER_STEP = 2, LU_MR_1HOP = 3, LU_MR_2HOP = 4, LU_MR_OV = 5, LU_FORMS = 6 };
  int64_t lu_form_last = LU_NONE;
  int64_t lu_form_count[LU_FORMS] = {0, 0, 0, 0, 0, 0};
  int64_t ranks_seen = 1;              // sum over the communicator of one per rank, taken when it was created
  int profile_level = 0;               // gsi_ctx_profile: 2 = skew barriers in front of collectives / sharded LUs (PH_COMM_WAIT)
  int64_t lu_timeouts_recovered = 0;   // entry points re-run transparently after a lost co-residency (api.cpp:with_retry)
  int rank() const { return comm ? comm->rank : 0; }
  int nranks() const { return comm ? comm->nranks : 1; }
};

enum OpKind { OP_DENSE = 0, OP_LOWRANK = 1, OP_GRIDCOV_IMPLICIT = 2, OP_FFT_COV = 3, OP_POINTCOV = 4 };

// A linear operator m x n; this rank holds rows [row0, row0 + mloc).
struct Operator {
  Context* ctx = nullptr;
  OpKind kind = OP_DENSE;
  int64_t m = 0, n = 0, row0 = 0, mloc = 0;
  Buf data;        // dense: mloc x n (ld mloc).  lowrank: samples shard mloc x N (ld mloc)
  int64_t ld = 0;
  int64_t N = 0;   // lowrank: number of samples
  int64_t gx = 0, gy = 0;   // implicit grid covariance: data = the gx * gy table of the kernel over grid offsets; never stored
  // OP_POINTCOV (scattered points): data = the d x n coordinates; entries generated panel by panel, never stored whole
  int pc_d = 0, pc_kind = 0;
  double pc_ell = 1.0, pc_sigma2 = 1.0, pc_nugget = 0.0;
  void* plan = nullptr;     // OP_FFT_COV: the backend's circulant-embedding plan (owned; single rank)
  // OP_DENSE whose rows are still crossing PCIe (gsi_randsvd_dense_host / gsi_rangefinder_dense_host): the handle of
  // Backend::upload2d_begin and its row-block height.  The FIRST product A*X consumes it -- block by block as the rows land,
  // bit-identical to the product of the resident matrix -- and clears it; the entry point's guard ends the upload on every
  // other exit path.  Null everywhere else.
  mutable void* pending_upload = nullptr;
  mutable int64_t pending_block_rows = 0;
  Operator() = default;
  Operator(const Operator&) = delete;
  Operator& operator=(const Operator&) = delete;
  ~Operator();
};

// block-row layout used whenever a caller does not supply one
void default_shard(int64_t m, int nranks, int rank, int64_t* row0, int64_t* mloc);

// Y_loc (mloc x l, ld ldy) = rows [row0,row0+mloc) of A * X, X replicated n x l (ld ldx)
void op_mul(const Operator& A, const double* X, int64_t ldx, int64_t l, double* Yloc, int64_t ldy);
// Z (n x l, ld ldz, replicated) = A' * X where Xloc (mloc x l, ld ldx) are this rank's rows of X
void op_mul_t(const Operator& A, const double* Xloc, int64_t ldx, int64_t l, double* Z, int64_t ldz);
// Y (m x l, ld m, replicated) <- all ranks' row shards
void gather_rows(Context& c, const Operator& A, const double* Yloc, int64_t ldy, int64_t l, double* Yfull);

// lu(Y).L of a ROW-SHARDED m x l panel (this rank: rows [row0, row0 + mloc), block layout of default_shard(m)), in
// place; bit-identical to the single-rank factorization.  One small all-gather per pivot step, one all-reduce per
// leaf / block for the U12 rows of rank 0; nothing of size m x l is communicated.  Needs l <= rows of rank 0.
void lu_panel_sharded(Context& c, double* Yloc, int64_t m, int64_t row0, int64_t mloc, int64_t l);
// the same with G virtual ranks on ONE device (Yloc[g]: shard g, ld = max(mloc_g, 1), block layout of default_shard(m)):
// every primitive runs with real row offsets, the exchanges are device copies -- the one-GPU check of the multi-rank kernels
void lu_panel_sharded_virtual(Context& c, double* const* Yloc, int64_t m, int64_t l, int G);
bool use_sharded_lu(Context& c, const Operator& A, int64_t rows, int64_t l);
// rangefinder(A, l, numiterations)  RandMatFact.jl:50-80.  Omega replicated n x l (ld n).
// Returns this rank's rows of Q (mloc x l, ld mloc).
Buf rangefinder(const Operator& A, const double* Omega, int64_t l, int64_t q);
// randsvd(A, K, p, q)  RandMatFact.jl:83-90.  Z (n x (K+p), ld n) and S (K+p) replicated.
void randsvd(const Operator& A, const double* Omega, int64_t K, int64_t p, int64_t q, double* Z, double* S);
// The same with Omega given and Z returned as this rank's ROWS (block layout of default_shard(n), ld = max(nloc, 1)):
// nothing n x l is gathered for LowRankCovMatrix / FFT operators.  S replicated.
void randsvd_rows(const Operator& A, const double* Omega_loc, int64_t K, int64_t p, int64_t q, double* Zloc, double* S);
// layout changes of an n x l panel over the ranks: ROWS (own rows, all columns, ld ldr) <-> COLS (all rows, own columns, ld n)
void rows_to_cols(Context& c, int64_t n, int64_t l, const double* Rloc, int64_t ldr, double* Cloc);
void cols_to_rows(Context& c, int64_t n, int64_t l, const double* Cloc, double* Rloc, int64_t ldr);
// eig_nystrom(A, Q)  RandMatFact.jl:92-102.  Q replicated n x j; U (m x j), Sigma (j) replicated.
void eig_nystrom(const Operator& A, const double* Q, int64_t j, double* U, double* Sigma);
// thin SVD of a replicated tall W (n x l, destroyed): V (n x l, may alias W) = left singular
// vectors scaled by `scale_K` rule (K < 0: plain V; else V*sqrt(S) for i<K, 0 otherwise)
// Xr (may be null): l x l right factor, the SVD is that of W Xr (a thin Q left one product short: Backend::qr_thinQ_deferred)
void svd_tall(Context& c, double* W, int64_t n, int64_t l, int64_t K_scale, double* V, double* S, const double* Xr = nullptr);
// rangefinder(A; epsilon, r)  RandMatFact.jl:15-48 (dense operator, single rank)
typedef void (*randn_fn)(void* user, double* buf, int64_t count);
int64_t rangefinder_adaptive(const Operator& A, randn_fn rn, void* user, double epsilon, int64_t r,
                             double* Q_host);
// IterativeSolvers.jl 0.9 `lsqr(A, b; maxiter)` (Paige & Saunders 1982) with that package's defaults
// (atol = btol = sqrt(eps), conlim = 1e8) on backend-resident vectors.  Call sites in the reference: lsqr.jl:54,
// lowrank.jl:142.  mul(x, y): y = A x (x: ncols, y: nrows); mul_t(x, y): y = A' x.  Returns the iteration count.
struct LsqrOperator {
  int64_t nrows = 0, ncols = 0;
  std::function<void(const double*, double*)> mul, mul_t;
};
int64_t lsqr(Context& c, const LsqrOperator& A, const double* b, double* x, int64_t maxiter);
// x = A \ b for a LowRankCovMatrix: lsqr with maxiter = number of samples  (lowrank.jl:141-144); b, x replicated (n)
int64_t lowrank_solve(const Operator& A, const double* b, double* x);
// PCGALowRankMatrix(etas, HX, R)  (lowrank.jl:32-36, 83-97): [(HQH + R) HX; HX' 0], HQH = sum eta_i eta_i' kept implicit.
struct PcgaLowRank {
  Context* ctx = nullptr;
  int64_t nobs = 0, K = 0;
  Buf E, HX, R;        // E: nobs x K (eta_i as columns); R: nobs x nobs dense, or nobs values when r_diag
  bool r_diag = false;
  void mul(const double* x, double* y) const;   // x, y: nobs + 1
};
// throws Error if a kernel raised an asynchronous flag
void check_async_errors(Context& c);

}  // namespace gsi
