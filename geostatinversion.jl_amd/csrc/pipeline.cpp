// pipeline.cpp -- RandMatFact.jl's order of operations, row-sharded.
// Each function cites the reference lines it follows.  All arithmetic happens in the
// Backend; this file only sequences products, panel factorizations and collectives.
//
// Data distribution.  The operator A (m x n) is cut into contiguous row blocks, one per
// rank.  Tall panels are either REPLICATED (every rank holds all rows; needed by the
// partial-pivot LU, whose pivot sequence must equal LAPACK's over the whole panel) or
// ROW-SHARDED (each rank its rows; the final TSQR).  Exchange steps:
//   Y = A*X      local rows, then AllGather of the row shards when the LU follows
//   Z = A'*Y     local partial product over the rank's rows, then AllReduce(sum)
//   final Q      TSQR: local Householder QR, AllGather of the l x l R factors, replicated
//                QR of the stack, local fix-up Q_g <- Q_g * Q2_g
#include "pipeline.hpp"
#include "lsqr_state.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include "../../include/gsi_hip.h"

namespace gsi {

void default_shard(int64_t m, int nranks, int rank, int64_t* row0, int64_t* mloc) {
  const int64_t pad = (m + nranks - 1) / nranks;
  int64_t r0 = std::min<int64_t>((int64_t)rank * pad, m);
  *row0 = r0;
  *mloc = std::min<int64_t>(pad, m - r0);
}

void check_async_errors(Context& c) {
  std::string msg;
  const int code = c.be->take_error(&msg);
  if (code != 0) throw Error(code, msg);
}

Operator::~Operator() {
  if (plan && ctx && ctx->be) ctx->be->fftcov_destroy(plan);
}

// C (m x l) = G[roff .. roff + m, koff .. koff + k) * B for the two generated (never stored) symmetric covariances: a table
// over grid offsets (OP_GRIDCOV_IMPLICIT) or a kernel function of scattered coordinates (OP_POINTCOV)
static void implicit_mul(Backend* be, const Operator& A, int64_t m, int64_t l, int64_t k, int64_t roff, int64_t koff,
                         const double* B, int64_t ldb, double* C, int64_t ldc) {
  if (A.kind == OP_GRIDCOV_IMPLICIT) be->gemm_nn_gridcov(m, l, k, A.data.p, A.gx, A.gy, roff, koff, B, ldb, C, ldc);
  else be->gemm_nn_pointcov(m, l, k, A.data.p, A.pc_d, A.pc_kind, A.pc_ell, A.pc_sigma2, A.pc_nugget, roff, koff, B, ldb, C, ldc);
}
// With profile level 2 the ranks meet in a one-double all-reduce before every collective and every row-sharded LU: what a
// rank then waits for its peers (load skew, a slower GPU) is timed as PH_COMM_WAIT instead of inflating the phase that follows
// (profiles/r04_rehearsal_2ranks_one_gpu.json: "lu" 2445 ms where one rank takes 31 -- nobody could tell waiting from work).
static void skew_barrier(Context& c) {
  if (!c.comm || c.profile_level < 2) return;
  Backend* be = c.be.get();
  Buf one(be, 1);
  be->fill_zero(one.p, 1);
  ScopedPhase ph(be, PH_COMM_WAIT);
  c.comm->allreduce_sum(one.p, 1);
}
struct CommPhase {
  Backend* be;
  explicit CommPhase(Context& c) : be(c.be.get()) { skew_barrier(c); be->phase_begin(PH_COMM); }
  ~CommPhase() { be->phase_end(PH_COMM); }
};
static bool is_implicit(const Operator& A) { return A.kind == OP_GRIDCOV_IMPLICIT || A.kind == OP_POINTCOV; }

void op_mul(const Operator& A, const double* X, int64_t ldx, int64_t l, double* Yloc, int64_t ldy) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  // a rank without rows (m < rank * ceil(m / G)) has nothing to compute, but it still takes part in every
  // collective below: only the collective-free operator kinds may leave early
  if (A.mloc == 0 && A.kind != OP_LOWRANK && !(A.kind == OP_FFT_COV && c.comm)) return;
  if (A.kind == OP_FFT_COV) {     // matrix-free: pad, FFT passes, spectrum, inverse passes, restrict
    if (!c.comm) {
      ScopedPhase ph(be, PH_GEMM_N);
      be->fftcov_apply(A.plan, l, X, ldx, Yloc, ldy);
      return;
    }
    // several ranks, X replicated: every rank transforms its own columns, then columns -> rows
    int64_t c0, lloc;
    default_shard(l, c.nranks(), c.rank(), &c0, &lloc);
    Buf YC(be, (size_t)A.n * std::max<int64_t>(lloc, 1));
    if (lloc > 0) {
      ScopedPhase ph(be, PH_GEMM_N);
      be->fftcov_apply(A.plan, lloc, X + c0 * ldx, ldx, YC.p, A.n);
    }
    cols_to_rows(c, A.n, l, YC.p, Yloc, ldy);
    return;
  }
  if (A.kind == OP_DENSE && A.pending_upload != nullptr) {
    // the matrix is still being uploaded in row blocks: Y = A*Omega (RandMatFact.jl:55) block by block as the rows land, each
    // block with the K split of the whole product (bit-identical to the product of the resident matrix)
    void* h = A.pending_upload;
    const int64_t mb = A.pending_block_rows;
    A.pending_upload = nullptr;                              // (the guard of the entry point still ends the transfer if we throw)
    {
      ScopedPhase ph(be, PH_GEMM_N);
      for (int64_t r0 = 0, b = 0; r0 < A.mloc; r0 += mb, ++b) {
        be->upload2d_wait_block(h, b);
        be->gemm_nn_rowblock(A.mloc, r0, std::min(mb, A.mloc - r0), l, A.n, A.data.p, A.ld, X, ldx, Yloc, ldy);
      }
    }
    be->upload2d_end(h);
    return;
  }
  if (A.kind == OP_DENSE) {
    ScopedPhase ph(be, PH_GEMM_N);
    be->gemm_nn(A.mloc, l, A.n, 1.0, A.data.p, A.ld, X, ldx, 0.0, Yloc, ldy);   // RandMatFact.jl:55,70
    return;
  }
  if (is_implicit(A)) {
    ScopedPhase ph(be, PH_GEMM_N);
    implicit_mul(be, A, A.mloc, l, A.n, A.row0, 0, X, ldx, Yloc, ldy);
    return;
  }
  // LowRankCovMatrix: A*X = S (S'X) / (N-1)   (lowrank.jl:115-121 as two tall-skinny products)
  Buf T(be, (size_t)A.N * l);
  {
    ScopedPhase ph(be, PH_GEMM_T);
    if (A.mloc > 0) be->gemm_tn(A.N, l, A.mloc, 1.0, A.data.p, A.ld, X + A.row0, ldx, 0.0, T.p, A.N);
    else be->fill_zero(T.p, (size_t)A.N * l);
  }
  if (c.comm) {
    CommPhase ph(c);
    c.comm->allreduce_sum(T.p, (size_t)A.N * l);
  }
  if (A.mloc > 0) {
    ScopedPhase ph(be, PH_GEMM_N);
    be->gemm_nn(A.mloc, l, A.N, 1.0 / (double)(A.N - 1), A.data.p, A.ld, T.p, A.N, 0.0, Yloc, ldy);
  }
}

void gather_rows(Context& c, const Operator& A, const double* Yloc, int64_t ldy, int64_t l, double* Yfull) {
  Backend* be = c.be.get();
  if (!c.comm) {
    if (Yloc != Yfull) be->copy2d(Yfull, A.m, Yloc, ldy, A.m, l);
    return;
  }
  const int G = c.nranks();
  const int64_t pad = (A.m + G - 1) / G;
  Buf send(be, (size_t)pad * l), recv(be, (size_t)pad * l * G);
  if (A.mloc < pad) be->fill_zero(send.p, (size_t)pad * l);
  be->copy2d(send.p, pad, Yloc, ldy, A.mloc, l);
  {
    CommPhase ph(c);
    c.comm->allgather(send.p, recv.p, (size_t)pad * l);
  }
  for (int g = 0; g < G; ++g) {
    int64_t r0, ml;
    default_shard(A.m, G, g, &r0, &ml);
    be->copy2d(Yfull + r0, A.m, recv.p + (size_t)g * pad * l, pad, ml, l);
  }
}

// ---- panel layouts over the ranks --------------------------------------------------------------------------------------
// ROWS: this rank's rows [r0, r0 + nloc) of default_shard(n), all l columns (ld given).  COLS: all n rows, this rank's
// columns [c0, c0 + lloc) of default_shard(l), ld = n.  A matrix-free operator that acts column by column (the FFT
// covariance) wants COLS, the panel factorizations (row-sharded LU, TSQR) want ROWS; the change of layout is an
// all-to-all of rows_s x cols_d blocks -- on xGMI every pair of GPUs has its own link, so all of them carry traffic at
// once.  Chunked over columns so that the staging buffers stay ~1 GB whatever n is (512^3: one column per chunk).
static int64_t a2a_chunk_cols(int64_t n, int64_t pad_l) {
  int64_t cc = ((int64_t)1 << 27) / std::max<int64_t>(n, 1);
  return std::max<int64_t>(1, std::min(cc, pad_l));
}
void rows_to_cols(Context& c, int64_t n, int64_t l, const double* Rloc, int64_t ldr, double* Cloc) {
  Backend* be = c.be.get();
  const int G = c.nranks(), rank = c.rank();
  if (!c.comm) { be->copy2d(Cloc, n, Rloc, ldr, n, l); return; }   // (a 1-rank communicator still goes through the all-to-all)
  const int64_t pad_n = (n + G - 1) / G, pad_l = (l + G - 1) / G;
  int64_t r0, nloc, c0, lloc;
  default_shard(n, G, rank, &r0, &nloc);
  default_shard(l, G, rank, &c0, &lloc);
  const int64_t cc = a2a_chunk_cols(n, pad_l), blk = pad_n * cc;
  Buf send(be, (size_t)blk * G), recv(be, (size_t)blk * G);
  be->fill_zero(send.p, (size_t)blk * G);
  for (int64_t j = 0; j < pad_l; j += cc) {
    const int64_t w = std::min(cc, pad_l - j);
    for (int d = 0; d < G; ++d) {                        // to rank d: my rows of ITS columns j .. j + w
      int64_t c0d, ld_;
      default_shard(l, G, d, &c0d, &ld_);
      const int64_t valid = std::max<int64_t>(0, std::min(w, ld_ - j));
      if (valid > 0 && nloc > 0) be->copy2d(send.p + (size_t)d * blk, pad_n, Rloc + (c0d + j) * ldr, ldr, nloc, valid);
    }
    {
      CommPhase ph(c);
      c.comm->alltoall(send.p, recv.p, (size_t)blk);
    }
    const int64_t mine = std::max<int64_t>(0, std::min(w, lloc - j));
    for (int s2 = 0; s2 < G && mine > 0; ++s2) {          // from rank s2: its rows of my columns
      int64_t r0s, ns;
      default_shard(n, G, s2, &r0s, &ns);
      if (ns > 0) be->copy2d(Cloc + r0s + j * n, n, recv.p + (size_t)s2 * blk, pad_n, ns, mine);
    }
  }
}
void cols_to_rows(Context& c, int64_t n, int64_t l, const double* Cloc, double* Rloc, int64_t ldr) {
  Backend* be = c.be.get();
  const int G = c.nranks(), rank = c.rank();
  if (!c.comm) { be->copy2d(Rloc, ldr, Cloc, n, n, l); return; }
  const int64_t pad_n = (n + G - 1) / G, pad_l = (l + G - 1) / G;
  int64_t r0, nloc, c0, lloc;
  default_shard(n, G, rank, &r0, &nloc);
  default_shard(l, G, rank, &c0, &lloc);
  const int64_t cc = a2a_chunk_cols(n, pad_l), blk = pad_n * cc;
  Buf send(be, (size_t)blk * G), recv(be, (size_t)blk * G);
  be->fill_zero(send.p, (size_t)blk * G);
  for (int64_t j = 0; j < pad_l; j += cc) {
    const int64_t w = std::min(cc, pad_l - j);
    const int64_t mine = std::max<int64_t>(0, std::min(w, lloc - j));
    for (int d = 0; d < G && mine > 0; ++d) {             // to rank d: ITS rows of my columns j .. j + w
      int64_t r0d, nd;
      default_shard(n, G, d, &r0d, &nd);
      if (nd > 0) be->copy2d(send.p + (size_t)d * blk, pad_n, Cloc + r0d + j * n, n, nd, mine);
    }
    {
      CommPhase ph(c);
      c.comm->alltoall(send.p, recv.p, (size_t)blk);
    }
    for (int s2 = 0; s2 < G; ++s2) {                      // from rank s2: my rows of its columns
      int64_t c0s, ls;
      default_shard(l, G, s2, &c0s, &ls);
      const int64_t valid = std::max<int64_t>(0, std::min(w, ls - j));
      if (valid > 0 && nloc > 0) be->copy2d(Rloc + (c0s + j) * ldr, ldr, recv.p + (size_t)s2 * blk, pad_n, nloc, valid);
    }
  }
}

// The matrix-free FFT covariance on several ranks: every rank holds the (replicated) plan and transforms ITS columns.
// Rows in, rows out: rows -> columns (all-to-all), the column-wise transforms, columns -> rows (all-to-all).
static void fft_mul_rows(const Operator& A, const double* Xloc, int64_t ldx, int64_t l, double* Yloc, int64_t ldy) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  int64_t c0, lloc;
  default_shard(l, c.nranks(), c.rank(), &c0, &lloc);
  Buf XC(be, (size_t)A.n * std::max<int64_t>(lloc, 1));
  rows_to_cols(c, A.n, l, Xloc, ldx, XC.p);
  Buf YC(be, (size_t)A.n * std::max<int64_t>(lloc, 1));
  if (lloc > 0) {
    ScopedPhase ph(be, PH_GEMM_N);
    be->fftcov_apply(A.plan, lloc, XC.p, A.n, YC.p, A.n);
  }
  XC.reset();
  cols_to_rows(c, A.n, l, YC.p, Yloc, ldy);
}

void op_mul_t(const Operator& A, const double* Xloc, int64_t ldx, int64_t l, double* Z, int64_t ldz) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  if (A.kind == OP_FFT_COV) {     // symmetric: A' X = A X
    if (!c.comm) {
      ScopedPhase ph(be, PH_GEMM_T);
      be->fftcov_apply(A.plan, l, Xloc, ldx, Z, ldz);
      return;
    }
    // several ranks: rows -> columns, transform the own columns, all-gather the column blocks into the replicated Z
    if (ldz != A.n) throw Error(GSI_ERR_INTERNAL, "op_mul_t: strided output with a communicator");
    const int G = c.nranks();
    const int64_t pad_l = (l + G - 1) / G;
    int64_t c0, lloc;
    default_shard(l, G, c.rank(), &c0, &lloc);
    Buf XC(be, (size_t)A.n * pad_l), YC(be, (size_t)A.n * pad_l), all(be, (size_t)A.n * pad_l * G);
    rows_to_cols(c, A.n, l, Xloc, ldx, XC.p);
    be->fill_zero(YC.p, (size_t)A.n * pad_l);
    if (lloc > 0) {
      ScopedPhase ph(be, PH_GEMM_T);
      be->fftcov_apply(A.plan, lloc, XC.p, A.n, YC.p, A.n);
    }
    {
      CommPhase ph(c);
      c.comm->allgather(YC.p, all.p, (size_t)A.n * pad_l);
    }
    for (int g = 0; g < G; ++g) {
      int64_t c0g, lg;
      default_shard(l, G, g, &c0g, &lg);
      if (lg > 0) be->copy2d(Z + c0g * ldz, ldz, all.p + (size_t)g * A.n * pad_l, A.n, A.n, lg);
    }
    return;
  }
  if (A.kind == OP_DENSE) {
    {
      ScopedPhase ph(be, PH_GEMM_T);
      if (A.mloc > 0)
        be->gemm_tn(A.n, l, A.mloc, 1.0, A.data.p, A.ld, Xloc, ldx, 0.0, Z, ldz);   // RandMatFact.jl:67,85
      else
        for (int64_t cidx = 0; cidx < l; ++cidx) be->fill_zero(Z + cidx * ldz, (size_t)A.n);
    }
    if (c.comm) {
      CommPhase ph(c);
      if (ldz == A.n) c.comm->allreduce_sum(Z, (size_t)A.n * l);
      else throw Error(GSI_ERR_INTERNAL, "op_mul_t: strided output with a communicator");
    }
    return;
  }
  if (is_implicit(A)) {
    // A symmetric: (A_loc)' X_loc = G[:, row0 .. row0+mloc) * X_loc -- the same generated NN product with
    // the roles of the row and reduction offsets exchanged; partial sums over the ranks' row blocks
    {
      ScopedPhase ph(be, PH_GEMM_T);
      if (A.mloc > 0)
        implicit_mul(be, A, A.n, l, A.mloc, 0, A.row0, Xloc, ldx, Z, ldz);
      else
        for (int64_t cidx = 0; cidx < l; ++cidx) be->fill_zero(Z + cidx * ldz, (size_t)A.n);
    }
    if (c.comm) {
      CommPhase ph(c);
      if (ldz == A.n) c.comm->allreduce_sum(Z, (size_t)A.n * l);
      else throw Error(GSI_ERR_INTERNAL, "op_mul_t: strided output with a communicator");
    }
    return;
  }
  // adjoint(A::LowRankCovMatrix) === A  (lowrank.jl:38-40): A'*X = A*X, rows then gathered
  Buf T(be, (size_t)A.N * l);
  {
    ScopedPhase ph(be, PH_GEMM_T);
    be->gemm_tn(A.N, l, A.mloc, 1.0, A.data.p, A.ld, Xloc, ldx, 0.0, T.p, A.N);
  }
  if (c.comm) {
    CommPhase ph(c);
    c.comm->allreduce_sum(T.p, (size_t)A.N * l);
  }
  if (!c.comm && ldz == A.m) {
    ScopedPhase ph(be, PH_GEMM_N);
    be->gemm_nn(A.mloc, l, A.N, 1.0 / (double)(A.N - 1), A.data.p, A.ld, T.p, A.N, 0.0, Z, ldz);
    return;
  }
  Buf Yloc(be, (size_t)std::max<int64_t>(A.mloc, 1) * l);
  {
    ScopedPhase ph(be, PH_GEMM_N);
    be->gemm_nn(A.mloc, l, A.N, 1.0 / (double)(A.N - 1), A.data.p, A.ld, T.p, A.N, 0.0, Yloc.p, A.mloc);
  }
  if (ldz != A.m) throw Error(GSI_ERR_INTERNAL, "op_mul_t: strided output for a sharded LowRankCovMatrix");
  gather_rows(c, A, Yloc.p, A.mloc, l, Z);
}

// Rows [r0n, r0n + nloc) (block layout of default_shard(A.n)) of W = A' * X, X given by this rank's rows Xloc.
// Multi-rank only.  Dense / implicit: local partial over the rank's rows of A, packed by destination rank,
// reduce-scattered (half the traffic of the all-reduce of op_mul_t, and nothing n x l is replicated).
// LowRankCovMatrix: adjoint(A) === A and its rows are sharded like A's, so the local rows come out directly.
static Buf op_mul_t_sharded(const Operator& A, const double* Xloc, int64_t ldx, int64_t l, int64_t nloc) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  const int G = c.nranks();
  const int64_t n = A.n, pad = (n + G - 1) / G;
  if (A.kind == OP_LOWRANK) {
    Buf T(be, (size_t)A.N * l);
    {
      ScopedPhase ph(be, PH_GEMM_T);
      be->gemm_tn(A.N, l, A.mloc, 1.0, A.data.p, A.ld, Xloc, ldx, 0.0, T.p, A.N);
    }
    {
      CommPhase ph(c);
      c.comm->allreduce_sum(T.p, (size_t)A.N * l);
    }
    Buf Wloc(be, (size_t)std::max<int64_t>(A.mloc, 1) * l);
    ScopedPhase ph(be, PH_GEMM_N);
    be->gemm_nn(A.mloc, l, A.N, 1.0 / (double)(A.N - 1), A.data.p, A.ld, T.p, A.N, 0.0, Wloc.p, A.mloc);
    return Wloc;
  }
  if (A.kind == OP_FFT_COV) {              // symmetric, row shards in and out: nothing n x l exists on any rank
    Buf Wloc(be, (size_t)std::max<int64_t>(nloc, 1) * l);
    fft_mul_rows(A, Xloc, ldx, l, Wloc.p, std::max<int64_t>(nloc, 1));
    return Wloc;
  }
  Buf P(be, (size_t)n * l);
  {
    ScopedPhase ph(be, PH_GEMM_T);
    if (A.mloc == 0)
      be->fill_zero(P.p, (size_t)n * l);
    else if (A.kind == OP_DENSE)
      be->gemm_tn(n, l, A.mloc, 1.0, A.data.p, A.ld, Xloc, ldx, 0.0, P.p, n);       // RandMatFact.jl:85
    else
      implicit_mul(be, A, n, l, A.mloc, 0, A.row0, Xloc, ldx, P.p, n);
  }
  Buf send(be, (size_t)pad * l * G), recv(be, (size_t)pad * l);
  if (pad * G != n) be->fill_zero(send.p, (size_t)pad * l * G);
  for (int g = 0; g < G; ++g) {
    int64_t r0, ml;
    default_shard(n, G, g, &r0, &ml);
    if (ml > 0) be->copy2d(send.p + (size_t)g * pad * l, pad, P.p + r0, n, ml, l);
  }
  P.reset();
  {
    CommPhase ph(c);
    c.comm->reduce_scatter_sum(send.p, recv.p, (size_t)pad * l);
  }
  if (nloc == pad) return recv;
  Buf Wloc(be, (size_t)std::max<int64_t>(nloc, 1) * l);
  be->copy2d(Wloc.p, nloc, recv.p, pad, nloc, l);
  return Wloc;
}

static void note_lu_form(Context& c, Context::LuForm f) {
  c.lu_form_last = f;
  c.lu_form_count[f] += 1;
}

static void lu_panel(Context& c, double* Y, int64_t rows, int64_t l) {
  skew_barrier(c);                         // profile level 2: the replicated factorizations start together on every rank
  ScopedPhase ph(c.be.get(), PH_LU);
  if (c.comm) note_lu_form(c, Context::LU_REPLICATED);
  c.be->lu_L(Y, rows, l, rows, nullptr);   // F = lu(Y); Q = F.L   RandMatFact.jl:60-61,68-69,72-73
}

static bool all_shards_tall(int64_t m, int G, int64_t l);

// Partial-pivot LU of a row-sharded panel ("sharded alternative": local arg-max -> exchange -> pivot row
// to everyone).  Per pivot step ONE all-gather of a (4 + 2 l)-double record per rank {local max, its row, row j};
// every rank then picks the same winner (largest value, lowest global row: LAPACK's idamax).  Left-looking leaves of 8
// inside blocks of lus_block() columns, exactly like the single-rank kernels; the U12 rows (rank 0 holds the first l
// rows) reach the other ranks by an all-reduce with zeros.
static void lu_panel_sharded_impl(Context& c, double* Yloc, int64_t m, int64_t row0, int64_t mloc, int64_t l, bool mr);

// do ALL ranks answer yes -- and, when they do, with the same `sig`?  (one all-gathered pair per rank: a rank that cannot
// take a path, or would take it with another launch geometry, would leave the others polling)
static bool all_ranks_agree(Context& c, bool mine, int64_t sig = 0) {
  Backend* be = c.be.get();
  const int G = c.nranks();
  const double v[2] = {mine ? 1.0 : 0.0, (double)sig};
  Buf send(be, 2), recv(be, (size_t)2 * G);
  be->upload2d(send.p, 2, v, 2, 2, 1);
  c.comm->allgather(send.p, recv.p, 2);
  std::vector<double> all((size_t)2 * G);
  be->download2d(all.data(), 2, recv.p, 2, 2, G);
  for (int g = 0; g < G; ++g)
    if (all[(size_t)2 * g] != 1.0 || all[(size_t)2 * g + 1] != all[1]) return false;
  return true;
}

// The in-kernel pivot exchange relies on things a build box with one GPU cannot prove about the machine it later runs on
// (peer mappings, visibility of system-scope stores across the fabric).  So, once per communicator, it has to EARN its
// place: a small panel (2048 rows per rank x 24 columns: three leaves of one block -- records, U mailbox, row boxes) is
// factored by the per-step form and by the persistent leaves; the fast path is used only if every rank got bit-identical
// factors and no kernel raised a flag.  Anything else -- a mapping that failed, a poll that timed out -- leaves the per-step
// form in charge, silently and on all ranks.
static int lus_mr_selftest(Context& c) {
  Backend* be = c.be.get();
  const int G = c.nranks(), rank = c.rank();
  int forms = 0;
  for (int form = 0; form < 3; ++form) {          // every form the exchange can take has to prove itself (see Backend::lus_mr_mode)
    const int64_t rows = (form == 2) ? 12288 : 2048, lt = 24, mt = rows * G;
    be->lus_mr_force(form);
    bool mine = false;
    const bool can = be->lus_mr_begin(c.comm.get(), mt, lt) && be->lus_mr_mode() == form;
    if (all_ranks_agree(c, can, can ? be->lus_mr_signature() : 0)) {
      Buf A(be, (size_t)rows * lt), B(be, (size_t)rows * lt);
      be->randn(A.p, (size_t)rows * lt, 0x5e1f7e57ull + (uint64_t)rank + 131ull * (uint64_t)form);
      be->copy2d(B.p, rows, A.p, rows, rows, lt);
      mine = true;
      try {
        lu_panel_sharded_impl(c, A.p, mt, rows * rank, rows, lt, false);
        lu_panel_sharded_impl(c, B.p, mt, rows * rank, rows, lt, true);
        be->axpy(rows * lt, -1.0, A.p, B.p);
        mine = (be->nrm2(rows * lt, B.p) == 0.0);
        std::string msg;
        if (be->take_error(&msg) != 0) mine = false;
      } catch (const Error&) {
        mine = false;
      }
      be->forgive_lost_coresidency();        // a time-out in here must not cost the context its single-GPU fast path
      if (all_ranks_agree(c, mine)) forms |= 1 << form;
      else if (form == 0) break;             // the plain form failed: nothing to build the others on
    }
  }
  be->lus_mr_force(0);
  return forms;
}

void lu_panel_sharded(Context& c, double* Yloc, int64_t m, int64_t row0, int64_t mloc, int64_t l) {
  Backend* be = c.be.get();
  const int G = c.nranks();
  skew_barrier(c);                           // profile level 2: arrival skew is PH_COMM_WAIT, not LU time
  if (G > 1 && l > (m + G - 1) / G)
    throw Error(GSI_ERR_INTERNAL, "lu_panel_sharded: the first rank must hold the first l rows");
  // One persistent launch per leaf and rank, pivot exchange inside the kernels (peer-written records), when the backend
  // and the communicator can do it -- decided by ALL ranks together, after the self-test above has passed on this
  // communicator.  The agreement for a panel height is cached, and it is void as soon as anything it rested on changes:
  // Backend::lus_mr_generation() moves on every rank at once (a time-out is made global below), and a moved generation
  // empties the cache, so the next call re-agrees through a collective instead of trusting a rank-local answer.
  bool mr = false;
  if (c.comm) {
    if (c.lus_mr_selftest < 0) c.lus_mr_selftest = lus_mr_selftest(c);
    if (c.lus_mr_gen != be->lus_mr_generation()) {
      c.lus_mr_ok.clear();
      c.lus_mr_gen = be->lus_mr_generation();
    }
    if (c.lus_mr_selftest != 0) {
      auto it = c.lus_mr_ok.find(m);
      if (it == c.lus_mr_ok.end()) {
        const bool can = be->lus_mr_begin(c.comm.get(), m, l) && ((c.lus_mr_selftest >> be->lus_mr_mode()) & 1) != 0;
        it = c.lus_mr_ok.emplace(m, all_ranks_agree(c, can, can ? be->lus_mr_signature() : 0)).first;
      }
      mr = it->second;
    }
    if (!mr && getenv("GSI_LU_MR_REQUIRE") != nullptr)      // tests: the in-kernel exchange must be what runs
      throw Error(GSI_ERR_INTERNAL, std::string("lu_panel_sharded: the multi-rank persistent leaf path is not available "
                                                "(GSI_LU_MR_REQUIRE): ") + be->lus_mr_reason() +
                                        " [self-test mask " + std::to_string(c.lus_mr_selftest) + "]");
  }
  lu_panel_sharded_impl(c, Yloc, m, row0, mloc, l, mr);
  if (c.comm) note_lu_form(c, !mr ? Context::LU_PER_STEP
                                  : (Context::LuForm)((int)Context::LU_MR_1HOP + be->lus_mr_mode()));
}

static void lu_panel_sharded_impl(Context& c, double* Yloc, int64_t m, int64_t row0, int64_t mloc, int64_t l, bool mr) {
  Backend* be = c.be.get();
  const int G = c.nranks(), rank = c.rank();
  ScopedPhase ph(be, PH_LU);
  const int64_t reclen = 4 + 2 * l, ld = std::max<int64_t>(mloc, 1);
  const int nb = be->lus_block();
  // (a rank that refuses here although the ranks agreed would leave its peers polling until their time-out, which the flag
  // exchange at the end then reports on every rank; it cannot happen while the generation rule of lu_panel_sharded holds)
  if (mr && !be->lus_mr_begin(c.comm.get(), m, l))
    throw Error(GSI_ERR_INTERNAL, std::string("lu_panel_sharded: persistent leaves refused: ") + be->lus_mr_reason());
  Buf rec(be, (size_t)reclen), recs(be, (size_t)reclen * G), u12leaf(be, (size_t)nb * 8), u12blk(be, (size_t)nb * l);
  Buf swaps, eflag;
  if (mr) {
    swaps = Buf(be, (size_t)16 * l);
    eflag = Buf(be, 1);
    // Everything this factorization allocates exists now (lus_mr_begin sized the backend's own workspaces) and every
    // kernel it launches has run once (lus_mr_warmup: code objects loaded): from here on kernels spin for their peers,
    // and with ranks as threads of one process no rank may be inside a runtime call that waits for the device.
    // DESIGN.md section 6 lists what is reachable between this barrier and the last leaf.
    be->lus_mr_warmup();
    c.comm->host_barrier();
  }
  for (int64_t jb = 0; jb < l; jb += nb) {
    const int b = (int)std::min<int64_t>(nb, l - jb);
    for (int64_t j0 = jb; j0 < jb + b; j0 += 8) {
      const int w = (int)std::min<int64_t>(8, jb + b - j0);
      const int64_t kp = j0 - jb;
      if (kp > 0 && !mr) {
        if (rank == 0) be->lus_u12_leaf(Yloc, ld, row0, jb, j0, w, u12leaf.p);
        else be->fill_zero(u12leaf.p, (size_t)kp * 8);
        if (c.comm) c.comm->allreduce_sum(u12leaf.p, (size_t)kp * 8);
        be->lus_pending(Yloc, mloc, ld, row0, jb, j0, w, u12leaf.p);
      }
      if (mr) {
        // pending update (U12 solved by rank 0 inside its kernel and pushed to the others) + the 8 pivot steps
        be->lus_leaf_mr(Yloc, mloc, ld, row0, m, l, jb, j0, w, nullptr);
        if (!be->lus_mr_swaps_done()) {                                             // rows the pivots exchange, other columns
          be->lus_swap_pack(Yloc, mloc, ld, row0, l, j0, w, swaps.p);
          c.comm->allreduce_sum(swaps.p, (size_t)16 * l);
          be->lus_swap_apply(Yloc, mloc, ld, row0, l, j0, w, swaps.p);
        }
        continue;
      }
      for (int s = 0; s < w; ++s) {
        be->lus_candidate(Yloc, mloc, ld, row0, l, j0 + s, rec.p);
        if (c.comm) c.comm->allgather(rec.p, recs.p, (size_t)reclen);
        else be->copy2d(recs.p, reclen, rec.p, reclen, reclen, 1);
        be->lus_apply(Yloc, mloc, ld, row0, m, l, j0, s, w, recs.p, G);
      }
    }
    const int64_t c0 = jb + b, t = l - c0;
    if (t > 0) {
      if (rank == 0) be->lus_u12_block(Yloc, ld, row0, jb, b, c0, l, u12blk.p);
      else be->fill_zero(u12blk.p, (size_t)b * t);
      if (c.comm) c.comm->allreduce_sum(u12blk.p, (size_t)b * t);
      be->lus_rankk(Yloc, mloc, ld, row0, jb, b, c0, t, u12blk.p);
    }
  }
  be->lus_finish(Yloc, mloc, ld, row0, l);
  if (mr) {
    // A workgroup that gave up polling raised info = -1 on ITS rank only, while its peers may have finished on records it
    // kept publishing: the time-out is made global on the stream (no host round trip), so that every rank's take_error
    // reports it, every rank switches the path off and bumps its generation in the same call.
    be->lu_flag_export(eflag.p);
    c.comm->allreduce_sum(eflag.p, 1);
    be->lu_flag_import(eflag.p);
  }
}

// The same factorization with G VIRTUAL ranks on one device: shard g = rows [row0_g, row0_g + mloc_g) of the panel in a
// buffer of its own, every lus_* primitive called with that shard's real row0 / mloc, the all-gather and the all-reduces
// with zeros replaced by the device copies they amount to.  This is how the row0 != 0 branches of the HIP primitives are
// exercised on a single GPU (gsi_lu_L_sharded_virtual; the multi-rank run itself needs G devices).
void lu_panel_sharded_virtual(Context& c, double* const* Yloc, int64_t m, int64_t l, int G) {
  Backend* be = c.be.get();
  if (G < 1) throw Error(GSI_ERR_ARG, "lu_panel_sharded_virtual: need at least one shard");
  if (G > 1 && l > (m + G - 1) / G)
    throw Error(GSI_ERR_ARG, "lu_panel_sharded: the first rank must hold the first l rows");
  ScopedPhase ph(be, PH_LU);
  const int64_t reclen = 4 + 2 * l;
  const int nb = be->lus_block();
  std::vector<int64_t> r0((size_t)G), ml((size_t)G);
  for (int g = 0; g < G; ++g) default_shard(m, G, g, &r0[(size_t)g], &ml[(size_t)g]);
  auto ldg = [&](int g) { return std::max<int64_t>(ml[(size_t)g], 1); };
  Buf recs(be, (size_t)reclen * G), u12leaf(be, (size_t)nb * 8), u12blk(be, (size_t)nb * l);
  for (int64_t jb = 0; jb < l; jb += nb) {
    const int b = (int)std::min<int64_t>(nb, l - jb);
    for (int64_t j0 = jb; j0 < jb + b; j0 += 8) {
      const int w = (int)std::min<int64_t>(8, jb + b - j0);
      if (j0 > jb) {
        be->lus_u12_leaf(Yloc[0], ldg(0), r0[0], jb, j0, w, u12leaf.p);           // rank 0 holds rows jb .. j0
        for (int g = 0; g < G; ++g) be->lus_pending(Yloc[g], ml[(size_t)g], ldg(g), r0[(size_t)g], jb, j0, w, u12leaf.p);
      }
      for (int s = 0; s < w; ++s) {
        for (int g = 0; g < G; ++g)                                               // = the all-gather of the records
          be->lus_candidate(Yloc[g], ml[(size_t)g], ldg(g), r0[(size_t)g], l, j0 + s, recs.p + (size_t)g * reclen);
        for (int g = 0; g < G; ++g)
          be->lus_apply(Yloc[g], ml[(size_t)g], ldg(g), r0[(size_t)g], m, l, j0, s, w, recs.p, G);
      }
    }
    const int64_t c0 = jb + b, t = l - c0;
    if (t > 0) {
      be->lus_u12_block(Yloc[0], ldg(0), r0[0], jb, b, c0, l, u12blk.p);
      for (int g = 0; g < G; ++g) be->lus_rankk(Yloc[g], ml[(size_t)g], ldg(g), r0[(size_t)g], jb, b, c0, t, u12blk.p);
    }
  }
  for (int g = 0; g < G; ++g) be->lus_finish(Yloc[g], ml[(size_t)g], ldg(g), r0[(size_t)g], l);
}

// When is the sharded form used?  It trades the all-gather of the m x l panel plus a replicated factorization for ~l
// latency-bound collectives (DESIGN.md section 6 has the numbers).  A LowRankCovMatrix never needs the panel whole
// (its products take and give row shards), so with the sharded LU its range finder moves nothing of size n x l at
// all; for the other operators the replicated form stays the default.  GSI_LU_SHARDED=1 / =0 forces either.
bool use_sharded_lu(Context& c, const Operator& A, int64_t rows, int64_t l) {
  if (!c.comm || !all_shards_tall(rows, c.nranks(), l)) return false;
  static const char* env = getenv("GSI_LU_SHARDED");
  if (env != nullptr) return env[0] == '1';
  return A.kind == OP_LOWRANK && c.nranks() > 1;
}

static bool all_shards_tall(int64_t m, int G, int64_t l) {
  for (int g = 0; g < G; ++g) {
    int64_t r0, ml;
    default_shard(m, G, g, &r0, &ml);
    if (ml < l) return false;
  }
  return true;
}

// thin orthonormal basis of a row-sharded panel: this rank holds rows [row0, row0 + mloc) of the m x l panel in
// Yloc (ld mloc), block layout of default_shard(m).  In place / swapped.  Rout (may be null): the l x l
// triangular factor of the WHOLE panel, replicated.
static void tsqr(Context& c, int64_t m, int64_t row0, int64_t mloc, Buf& Yloc, int64_t l, double* Rout) {
  Backend* be = c.be.get();
  const int G = c.nranks();
  if (G == 1) {
    ScopedPhase ph(be, PH_QR);
    be->qr_thinQ(Yloc.p, mloc, l, mloc, Rout);   // qr(Y, Val(true)) -> Matrix(F.Q)  :57-58,75-76
    return;
  }
  if (!all_shards_tall(m, G, l)) {  // a shard shorter than the sketch width: factor the gathered panel everywhere
    Operator shape;
    shape.m = m; shape.row0 = row0; shape.mloc = mloc;
    Buf Yfull(be, (size_t)m * l);
    gather_rows(c, shape, Yloc.p, mloc, l, Yfull.p);
    {
      ScopedPhase ph(be, PH_QR);
      be->qr_thinQ(Yfull.p, m, l, m, Rout, true);
    }
    be->copy2d(Yloc.p, mloc, Yfull.p + row0, m, mloc, l);
    return;
  }
  Buf R(be, (size_t)l * l), Rall(be, (size_t)l * l * G), stack(be, (size_t)l * l * G);
  {
    ScopedPhase ph(be, PH_QR);
    be->qr_thinQ(Yloc.p, mloc, l, mloc, R.p);
  }
  {
    CommPhase ph(c);
    c.comm->allgather(R.p, Rall.p, (size_t)l * l);
  }
  const int64_t sl = (int64_t)G * l;
  for (int g = 0; g < G; ++g) be->copy2d(stack.p + (int64_t)g * l, sl, Rall.p + (size_t)g * l * l, l, l, l);
  {
    ScopedPhase ph(be, PH_QR);
    be->qr_thinQ(stack.p, sl, l, sl, Rout, true);
  }
  Buf Qn(be, (size_t)mloc * l);
  {
    ScopedPhase ph(be, PH_SMALL_GEMM);
    be->gemm_nn(mloc, l, l, 1.0, Yloc.p, mloc, stack.p + (int64_t)c.rank() * l, sl, 0.0, Qn.p, mloc);
  }
  Yloc = std::move(Qn);
}
static void tsqr(Context& c, const Operator& A, Buf& Yloc, int64_t l) { tsqr(c, A.m, A.row0, A.mloc, Yloc, l, nullptr); }

// The range finder with EVERY panel a row shard from the sketch to the TSQR (square operators whose adjoint is the
// operator itself and whose products take and give row shards): LowRankCovMatrix -- S (S'X), only the N x l sums cross
// ranks, adjoint(A) === A (lowrank.jl:38-40) -- and the FFT covariance -- rows -> columns, column-wise transforms,
// columns -> rows.  LU row-sharded (bit-identical to the single-rank factorization), final Q by TSQR.  Nothing of size
// n x l exists on any rank.  Omega_loc: this rank's rows of Omega (ld ldo).
static bool rows_path_ok(Context& c, const Operator& A, int64_t l) {
  return c.comm && (A.kind == OP_LOWRANK || A.kind == OP_FFT_COV) && A.m == A.n && all_shards_tall(A.m, c.nranks(), l);
}
static Buf rangefinder_rows(const Operator& A, const double* Omega_loc, int64_t ldo, int64_t l, int64_t q) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  const int64_t m = A.m, ldl = std::max<int64_t>(A.mloc, 1);
  Buf Yloc(be, (size_t)ldl * l), Zloc;
  auto mul_rows = [&](const double* Xloc, int64_t ldx, double* Out) {
    if (A.kind == OP_FFT_COV) fft_mul_rows(A, Xloc, ldx, l, Out, ldl);
    else op_mul(A, Xloc - A.row0, ldx, l, Out, A.mloc);     // op_mul reads rows [row0, row0 + mloc) of its X argument only
  };
  mul_rows(Omega_loc, ldo, Yloc.p);                         // Y = A*Omega            :55
  if (q > 0) {
    Zloc = Buf(be, (size_t)ldl * l);
    lu_panel_sharded(c, Yloc.p, m, A.row0, A.mloc, l);      // Q = lu(Y).L            :60-61
  }
  for (int64_t i = 1; i <= q; ++i) {
    mul_rows(Yloc.p, ldl, Zloc.p);                          // Q = A'*Q               :67
    lu_panel_sharded(c, Zloc.p, m, A.row0, A.mloc, l);      //                        :68-69
    mul_rows(Zloc.p, ldl, Yloc.p);                          // Q = A*Q                :70
    if (i < q) lu_panel_sharded(c, Yloc.p, m, A.row0, A.mloc, l);   //                :72-73
  }
  Zloc.reset();
  tsqr(c, A, Yloc, l);                                      //                        :57-58, 75-76
  return Yloc;
}

// `defer` (randsvd on one rank): the final thin Q may come back one tall product short -- Q = Q1 X2 with Q1 in backend
// workspace (Backend::qr_thinQ_deferred); the returned Buf is then empty and defer->Q1 / ldq / X2 are set.
struct DeferredQ { const double* Q1 = nullptr; int64_t ldq = 0; Buf X2; };
static Buf rangefinder_impl(const Operator& A, const double* Omega, int64_t l, int64_t q, DeferredQ* defer);
Buf rangefinder(const Operator& A, const double* Omega, int64_t l, int64_t q) { return rangefinder_impl(A, Omega, l, q, nullptr); }

static Buf rangefinder_impl(const Operator& A, const double* Omega, int64_t l, int64_t q, DeferredQ* defer) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  if (q < 0)   // RandMatFact.jl:62-64
    throw Error(GSI_ERR_NEG_ITERS,
                "parameter numiterations should be positive, but numiterations=" + std::to_string(q));
  if (l < 1 || l > A.m || l > A.n) throw Error(GSI_ERR_ARG, "rangefinder: need 1 <= l <= min(size(A))");
  const bool single = (c.nranks() == 1);
  const int64_t m = A.m, n = A.n;
  auto final_q_single = [&](Buf& Y) -> bool {               // true: deferred, Y is no longer needed
    if (defer == nullptr || !single) return false;
    ScopedPhase ph(be, PH_QR);
    defer->X2 = Buf(be, (size_t)l * l);
    if (be->qr_thinQ_deferred(Y.p, m, l, m, &defer->Q1, &defer->ldq, defer->X2.p)) return true;
    defer->X2.reset();
    defer->Q1 = nullptr;
    return false;
  };
  if (q == 0) {
    Buf Yloc(be, (size_t)std::max<int64_t>(A.mloc, 1) * l);
    op_mul(A, Omega, n, l, Yloc.p, A.mloc);                 // Y = A*Omega            :55
    if (final_q_single(Yloc)) return Buf();
    tsqr(c, A, Yloc, l);                                    //                        :57-58
    return Yloc;
  }
  if (!single && A.m == A.n && ((A.kind == OP_LOWRANK && use_sharded_lu(c, A, m, l)) ||
                                (A.kind == OP_FFT_COV && rows_path_ok(c, A, l))))
    return rangefinder_rows(A, Omega + A.row0, n, l, q);    // every panel stays a row shard (Omega is replicated here)
  Buf Yfull(be, (size_t)m * l);                              // replicated m x l
  Buf Z(be, (size_t)n * l);                                  // replicated n x l
  Buf Yloc;                                                  // this rank's rows (multi-rank only)
  if (!single) Yloc = Buf(be, (size_t)std::max<int64_t>(A.mloc, 1) * l);
  double* yl = single ? Yfull.p : Yloc.p;
  const int64_t ldyl = single ? m : A.mloc;
  // a sketch panel comes out of A*X as row shards: factor it where it lies when the sharded LU is on (the all-gather
  // of the panel and the replicated factorization go away; A'*L only reads the local rows anyway)
  const bool shard_y = !single && use_sharded_lu(c, A, m, l);
  auto lu_y = [&]() {
    if (shard_y) { lu_panel_sharded(c, yl, m, A.row0, A.mloc, l); return; }
    gather_rows(c, A, yl, ldyl, l, Yfull.p);
    lu_panel(c, Yfull.p, m, l);
  };
  op_mul(A, Omega, n, l, yl, ldyl);                         // Y = A*Omega            :55
  lu_y();                                                   // Q = lu(Y).L            :60-61
  for (int64_t i = 1; i <= q; ++i) {                        //                        :66
    if (shard_y) op_mul_t(A, yl, ldyl, l, Z.p, n);          // Q = A'*Q               :67
    else op_mul_t(A, Yfull.p + A.row0, m, l, Z.p, n);
    lu_panel(c, Z.p, n, l);                                 // Q = lu(Q).L            :68-69 (Z is replicated by the all-reduce)
    op_mul(A, Z.p, n, l, yl, ldyl);                         // Q = A*Q                :70
    if (i < q) lu_y();                                      //                        :72-73
  }
  Z.reset();                                                // the QR below wants a panel of its own (512^3: each is tens of GB)
  if (single) {
    if (final_q_single(Yfull)) return Buf();
    tsqr(c, A, Yfull, l);                                   // pivoted-QR range       :75-76
    return Yfull;
  }
  Yfull.reset();
  tsqr(c, A, Yloc, l);
  return Yloc;
}

void svd_tall(Context& c, double* W, int64_t n, int64_t l, int64_t K_scale, double* V, double* S, const double* Xr) {
  Backend* be = c.be.get();
  if (!c.comm && V != W && be->svd_tall_fused(W, n, l, n, K_scale, V, n, S, Xr)) return;   // :86-88 in one pass (single rank)
  Buf WX;
  if (Xr != nullptr) {                                       // the fused path declined: form W Xr after all
    ScopedPhase ph(be, PH_SMALL_GEMM);
    WX = Buf(be, (size_t)n * l);
    be->gemm_nn(n, l, l, 1.0, W, n, Xr, l, 0.0, WX.p, n);
    W = WX.p;
  }
  Buf R(be, (size_t)l * l), U(be, (size_t)l * l);
  {
    ScopedPhase ph(be, PH_QR);
    be->qr_thinQ(W, n, l, n, R.p, c.comm != nullptr);        // B' = Q_B R (W replicated)
  }
  {
    ScopedPhase ph(be, PH_SVD);
    be->svd_small(R.p, l, U.p, S);                          // R = U_R S V_R'
    if (K_scale >= 0) be->scale_cols_sqrt(U.p, l, S, K_scale);   // Sh = sqrt([S[1:K]; zeros(p)])  :87
  }
  {
    ScopedPhase ph(be, PH_SMALL_GEMM);
    const int64_t lz = (K_scale >= 0 && K_scale < l) ? K_scale : l;      // the last p columns of Z are zero by definition (:87)
    be->gemm_nn(n, lz, l, 1.0, W, n, U.p, l, 0.0, V, n);    // Z = V*Sh = Q_B (U_R Sh)         :88
    if (lz < l) be->fill_zero(V + (size_t)lz * n, (size_t)(l - lz) * n);
  }
}

// (), S, V = svd(B), Z = V*Sh with B = Q'A never formed whole (RandMatFact.jl:85-88): this rank's rows of W = B' = A'Q,
// TSQR of the row blocks (the l x l factor everywhere), replicated small SVD, the local rows of Z.  Needs every row block
// of n to be at least l tall.  Zloc: nloc x l, ld max(nloc, 1).
static void svd_rows(const Operator& A, Buf& Q, int64_t K, int64_t l, double* Zloc, double* S) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  const int G = c.nranks();
  int64_t r0n, nloc;
  default_shard(A.n, G, c.rank(), &r0n, &nloc);
  Buf Wloc = op_mul_t_sharded(A, Q.p, std::max<int64_t>(A.mloc, 1), l, nloc);   // B = Q'*A (rows of B')          :85
  Q.reset();
  Buf R(be, (size_t)l * l), U(be, (size_t)l * l);
  tsqr(c, A.n, r0n, nloc, Wloc, l, R.p);
  skew_barrier(c);
  {
    ScopedPhase ph(be, PH_SVD);
    be->svd_small(R.p, l, U.p, S);                         // (), S, V = svd(B)                    :86
    be->scale_cols_sqrt(U.p, l, S, K);                     // Sh = sqrt([S[1:K]; zeros(p)])        :87
  }
  ScopedPhase ph(be, PH_SMALL_GEMM);
  const int64_t ldz = std::max<int64_t>(nloc, 1), lz = (K < l) ? K : l;      // the last p columns of Z are zero by definition (:87)
  be->gemm_nn(nloc, lz, l, 1.0, Wloc.p, ldz, U.p, l, 0.0, Zloc, ldz);       // Z = V*Sh  :88
  if (lz < l) be->fill_zero(Zloc + (size_t)lz * ldz, (size_t)(l - lz) * ldz);
}

void randsvd(const Operator& A, const double* Omega, int64_t K, int64_t p, int64_t q, double* Z, double* S) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  if (K < 0 || p < 0 || K + p < 1) throw Error(GSI_ERR_ARG, "randsvd: need K >= 0, p >= 0, K + p >= 1");
  const int64_t l = K + p;
  // One rank, no communicator: the thin Q is only ever used as B = Q'A (:85), so its last tall product is deferred into the
  // l x l factor of svd(B) -- Q = Q1 X2, W = A'Q1, svd(W X2) -- when CholeskyQR2 applies (the usual case).
  DeferredQ dq;
  Buf Q = rangefinder_impl(A, Omega, l, q, c.comm ? nullptr : &dq);   // Q = rangefinder(A, K+p, q)     :84
  if (dq.Q1 != nullptr) {
    Buf W(be, (size_t)A.n * l);
    op_mul_t(A, dq.Q1, dq.ldq, l, W.p, A.n);                // B = Q'*A  (held as A'Q1; X2 follows in svd_tall)   :85
    svd_tall(c, W.p, A.n, l, K, Z, S, dq.X2.p);             // (), S, V = svd(B); Z = V*Sh    :86-88
    return;
  }
  const int G = c.nranks();
  if (c.comm && all_shards_tall(A.n, G, l) && (A.kind != OP_LOWRANK || A.m == A.n)) {   // any communicator, also 1 rank
    int64_t r0n, nloc;
    default_shard(A.n, G, c.rank(), &r0n, &nloc);
    Buf Zloc(be, (size_t)std::max<int64_t>(nloc, 1) * l);
    svd_rows(A, Q, K, l, Zloc.p, S);
    Operator shape;
    shape.m = A.n; shape.row0 = r0n; shape.mloc = nloc;
    gather_rows(c, shape, Zloc.p, std::max<int64_t>(nloc, 1), l, Z);
    return;
  }
  Buf W(be, (size_t)A.n * l);
  op_mul_t(A, Q.p, A.mloc, l, W.p, A.n);                    // B = Q'*A  (held as B' = A'Q)   :85
  Q.reset();
  svd_tall(c, W.p, A.n, l, K, Z, S);                        // (), S, V = svd(B); Z = V*Sh    :86-88
}

// randsvd with Omega given and Z returned as ROW SHARDS (this rank's rows of default_shard(n); ld = max(nloc, 1)): the
// form for panels that must never exist whole on one GPU (BASELINE configs[2] at 512^3 / rank 256: 275 GB per panel) and
// for consumers that keep the xi-basis sharded (configs[4]).  LowRankCovMatrix and FFT operators never gather anything of
// size n x l; dense / implicit operators need X replicated for their products, so Omega is all-gathered for them (their
// n is bounded by the stored / generated operator anyway) and only the output stays sharded.
void randsvd_rows(const Operator& A, const double* Omega_loc, int64_t K, int64_t p, int64_t q, double* Zloc, double* S) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  if (K < 0 || p < 0 || K + p < 1) throw Error(GSI_ERR_ARG, "randsvd: need K >= 0, p >= 0, K + p >= 1");
  const int64_t l = K + p, n = A.n;
  if (!c.comm) { randsvd(A, Omega_loc, K, p, q, Zloc, S); return; }
  if (q < 0)
    throw Error(GSI_ERR_NEG_ITERS, "parameter numiterations should be positive, but numiterations=" + std::to_string(q));
  if (l > A.m || l > A.n) throw Error(GSI_ERR_ARG, "rangefinder: need 1 <= l <= min(size(A))");
  const int G = c.nranks();
  if (!all_shards_tall(n, G, l))
    throw Error(GSI_ERR_ARG, "randsvd_rows: every rank's row block must be at least K + p rows tall");
  int64_t r0n, nloc;
  default_shard(n, G, c.rank(), &r0n, &nloc);
  Buf Q;
  if (rows_path_ok(c, A, l)) {
    Q = rangefinder_rows(A, Omega_loc, std::max<int64_t>(nloc, 1), l, q);
  } else {
    Buf Om(be, (size_t)n * l);
    Operator shape;
    shape.m = n; shape.row0 = r0n; shape.mloc = nloc;
    gather_rows(c, shape, Omega_loc, std::max<int64_t>(nloc, 1), l, Om.p);
    Q = rangefinder(A, Om.p, l, q);
  }
  svd_rows(A, Q, K, l, Zloc, S);
}

void eig_nystrom(const Operator& A, const double* Q, int64_t j, double* U, double* Sigma) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  if (A.m != A.n) throw Error(GSI_ERR_ARG, "eig_nystrom: A must be square");
  if (j < 1 || j > A.n) throw Error(GSI_ERR_ARG, "eig_nystrom: need 1 <= size(Q,2) <= n");
  const int64_t n = A.n;
  Buf B1loc(be, (size_t)std::max<int64_t>(A.mloc, 1) * j), B1(be, (size_t)n * j), B2(be, (size_t)j * j);
  op_mul(A, Q, n, j, B1loc.p, A.mloc);                      // B1 = A*Q                       :93
  gather_rows(c, A, B1loc.p, A.mloc, j, B1.p);
  {
    ScopedPhase ph(be, PH_SMALL_GEMM);
    be->gemm_tn(j, j, n, 1.0, Q, n, B1.p, n, 0.0, B2.p, j); // B2 = Q'*B1                     :94
  }
  {
    ScopedPhase ph(be, PH_OTHER);
    be->chol_upper(B2.p, j);                                // C = cholesky(Hermitian(B2)).U  :95
    be->trsm_right_upper(B1.p, n, j, n, B2.p);              // F = B1*inv(C)                  :96
  }
  svd_tall(c, B1.p, n, j, -1, U, Sigma);                    // U, Sigmavec, V = svd(F)        :97
}

// ---- IterativeSolvers.lsqr (third-party, Project.toml:21; not under the reference tree): Paige & Saunders' LSQR
//      with that package's defaults, restated from the published algorithm exactly as oracle/oracle.py:lsqr does
//      (same order of operations, same stopping rules).  Vectors AND the scalar recurrences live in backend memory
//      (lsqr_state.hpp): an iteration is the two operator products plus two fused launch groups, nothing waits for the
//      host; the host polls {stopped, iterations} every LSQR_POLL iterations (iterations enqueued past the stopping point
//      are no-ops on the device, so x and the iteration count are exactly those of the one-scalar-at-a-time loop).
int64_t lsqr(Context& c, const LsqrOperator& A, const double* b, double* x, int64_t maxiter) {
  using namespace lsqrst;
  Backend* be = c.be.get();
  const int64_t m = A.nrows, n = A.ncols;
  const double tol = std::sqrt(2.220446049250313e-16);
  const double conlim = 1e8;
  if (maxiter < 0) maxiter = std::max(m, n);
  Buf u(be, (size_t)m), v(be, (size_t)n), w(be, (size_t)n), t(be, (size_t)std::max(m, n));
  be->fill_zero(x, (size_t)n);
  be->copy2d(u.p, m, b, m, m, 1);
  const double beta = be->nrm2(m, u.p);                // the two start-up norms are read by the host: once per solve
  if (beta == 0.0) return 0;
  be->scal(m, 1.0 / beta, u.p);
  A.mul_t(u.p, v.p);
  const double alpha = be->nrm2(n, v.p);
  if (alpha == 0.0) return 0;
  be->scal(n, 1.0 / alpha, v.p);
  be->copy2d(w.p, n, v.p, n, n, 1);
  double st[COUNT] = {0.0};
  st[ALPHA] = alpha; st[BETA] = beta; st[RHOBAR] = alpha; st[PHIBAR] = beta; st[BNORM] = beta;
  st[CS2] = -1.0; st[SN2] = 0.0; st[MAXITER] = (double)maxiter;
  st[ATOL] = tol; st[BTOL] = tol; st[CTOL] = 1.0 / conlim;
  Buf work(be, be->lsqr_work_doubles());
  be->upload2d(work.p, COUNT, st, COUNT, COUNT, 1);
  be->lsqr_begin(n, w.p, work.p);
  static const int64_t poll = getenv("GSI_LSQR_POLL") ? std::max(1, atoi(getenv("GSI_LSQR_POLL"))) : 8;
  int64_t enq = 0;
  while (enq < maxiter) {
    const int64_t burst = std::min<int64_t>(poll, maxiter - enq);
    for (int64_t k = 0; k < burst; ++k) {
      A.mul(v.p, t.p);                                 // u = A v - alpha u; beta = |u|; u /= beta
      be->lsqr_step_u(m, t.p, u.p, work.p);
      A.mul_t(u.p, t.p);                               // v = A' u - beta v; alpha = |v|; v /= alpha; x, w updates
      be->lsqr_step_v(n, t.p, v.p, w.p, x, work.p);
    }
    enq += burst;
    be->download2d(st, COUNT, work.p, COUNT, COUNT, 1);
    if (st[STOPPED] != 0.0) break;
  }
  return (int64_t)st[ITERS];
}

int64_t lowrank_solve(const Operator& A, const double* b, double* x) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  if (A.kind != OP_LOWRANK) throw Error(GSI_ERR_ARG, "lowrank_solve: the operator is not a LowRankCovMatrix");
  LsqrOperator L;
  L.nrows = L.ncols = A.n;
  Buf yloc(be, (size_t)std::max<int64_t>(A.mloc, 1));
  L.mul = [&](const double* xin, double* y) {          // adjoint(A) === A (lowrank.jl:38-40)
    op_mul(A, xin, A.n, 1, yloc.p, A.mloc);
    gather_rows(c, A, yloc.p, A.mloc, 1, y);
  };
  L.mul_t = L.mul;
  return lsqr(c, L, b, x, A.N);                        // maxiter = length(A.samples)   lowrank.jl:142
}

// v[1:end-1] = R xs + sum_i eta_i dot(eta_i, xs) + HX x[end];  v[end] = dot(HX, xs)      lowrank.jl:83-97
void PcgaLowRank::mul(const double* x, double* y) const {
  Backend* be = ctx->be.get();
  Buf t(be, (size_t)K);
  be->gemv_t(nobs, K, 1.0, E.p, nobs, x, t.p);                                    // t_i = dot(eta_i, xs)
  be->gemv_n(nobs, K, 1.0, E.p, nobs, t.p, 0.0, y);                               // sum_i eta_i t_i
  if (r_diag) be->diag_mul_add(nobs, R.p, x, y);                                  // + R xs
  else be->gemv_n(nobs, nobs, 1.0, R.p, nobs, x, 1.0, y);
  be->gemv_n(nobs, 1, 1.0, HX.p, nobs, x + nobs, 1.0, y);                         // + HX x[end]
  be->gemv_t(nobs, 1, 1.0, HX.p, nobs, x, y + nobs);                              // dot(HX, xs)
}

int64_t rangefinder_adaptive(const Operator& A, randn_fn rn, void* user, double epsilon, int64_t r,
                             double* Q_host) {
  Context& c = *A.ctx;
  Backend* be = c.be.get();
  if (c.nranks() != 1) throw Error(GSI_ERR_ARG, "adaptive rangefinder: single rank only");
  if (A.kind != OP_DENSE) throw Error(GSI_ERR_ARG, "adaptive rangefinder needs a dense (strided) A, as the reference's gemv! does");
  if (A.m != A.n) throw Error(GSI_ERR_ARG, "adaptive rangefinder: the reference's Yfull allocation assumes a square A (RandMatFact.jl:18)");
  if (r < 1 || r > 64) throw Error(GSI_ERR_ARG, "adaptive rangefinder: need 1 <= r <= 64");
  const int64_t m = A.m, n = A.n, kmax = std::min(m, n);
  std::vector<double> host((size_t)n * r);
  Buf Yfull(be, (size_t)n * (r + kmax)), Qfull(be, (size_t)m * kmax), om(be, (size_t)n * r),
      Aom(be, (size_t)m), tmp(be, (size_t)std::max<int64_t>(kmax, 1));
  be->fill_zero(Yfull.p, (size_t)n * (r + kmax));           //                                :18
  be->fill_zero(Qfull.p, (size_t)m * kmax);                 //                                :23
  rn(user, host.data(), n * r);                             // randn(n, r)                    :20
  be->upload2d(om.p, n, host.data(), n, n, r);
  be->gemm_nn(m, r, n, 1.0, A.data.p, A.ld, om.p, n, 0.0, Yfull.p, n);   // gemm!('N','N',1,A,randn,0,Y)
  const double thresh = epsilon / std::sqrt(200.0 / M_PI);
  std::vector<double> norms((size_t)r);
  int64_t j = 0;
  for (;;) {
    be->colnorms(Yfull.p + j * n, n, r, n, norms.data());   // maximum(colnorms(view(Yfull,:,j+1:j+r)))   :26
    double mx = 0.0;
    for (double v : norms) mx = std::max(mx, v);
    if (!(mx > thresh)) break;
    if (j >= kmax) break;   // the reference would raise a BoundsError here
    j += 1;
    double* Yj = Yfull.p + (j - 1) * n;
    double* Qj = Qfull.p + (j - 1) * m;
    if (j > 1) {
      be->gemv_t(m, j - 1, 1.0, Qfull.p, m, Yj, tmp.p);                        // QtYj = gemv('T',1,Q,Yj)   :30
      // the reference's `Yj -= Q*QtYj` rebinds Yj to a new vector; the view in Yfull keeps the old
      // values.  Column j of Yfull is never read again, so updating it in place is equivalent.
      be->gemv_n(m, j - 1, -1.0, Qfull.p, m, tmp.p, 1.0, Yj);                  // Yj - Q*QtYj               :31
    }
    const double nrm = be->nrm2(m, Yj);
    be->axpy(m, 1.0 / nrm, Yj, Qj);                         // axpy!(1/norm(Yj), Yj, Qj)      :32-34
    rn(user, host.data(), n);                               // randn!(omega)                  :36
    be->upload2d(om.p, n, host.data(), n, n, 1);
    be->gemv_n(m, n, 1.0, A.data.p, A.ld, om.p, 0.0, Aom.p);                    // gemv!('N',1,A,omega,0,Aomega) :37
    be->gemv_t(m, j, 1.0, Qfull.p, m, Aom.p, tmp.p);                            // QtAomega                 :38
    double* ynew = Yfull.p + (r + j - 1) * n;
    be->scal_copy(m, 1.0, Aom.p, ynew);
    be->gemv_n(m, j, -1.0, Qfull.p, m, tmp.p, 1.0, ynew);                       // ynew = Aomega - Q*QtAomega :39-40
    be->project_out(m, r - 1, Qj, Yfull.p + j * n, n);      // Yi -= dot(Qj, Yi) Qj, i = j+1 .. j+r-1   :42-45
  }
  if (j > 0) be->download2d(Q_host, m, Qfull.p, m, m, j);   // return Qfull[:, 1:j]           :47
  return j;
}

}  // namespace gsi
