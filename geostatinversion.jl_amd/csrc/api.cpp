// api.cpp -- the C ABI of include/gsi_hip.h: argument checking, host<->device staging and
// error translation around pipeline.cpp.  No C++ exception crosses the boundary.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>
#include "../../include/gsi_hip.h"
#include "pipeline.hpp"
#include "pointcov.hpp"

using namespace gsi;

struct gsi_ctx {
  Context c;
};
struct gsi_op {
  Operator op;
};
struct gsi_mat {
  gsi_ctx* ctx;
  Buf buf;
  int64_t rows, cols;
  int refs = 1;            // the caller's handle + one per gsi_basis that points into buf (precision 64: no copy)
};
static void mat_unref(gsi_mat* m) {
  if (m && --m->refs <= 0) delete m;
}
struct gsi_pcgamat {
  gsi_ctx* ctx;
  PcgaLowRank A;
};
struct gsi_basis {
  gsi_ctx* ctx;
  int64_t n, K;
  int precision;          // 64: Z64 points into the gsi_mat; 32: own fp32 copy in buf32
  const double* Z64;
  Buf buf32;
  gsi_mat* owner = nullptr;   // precision 64: the matrix whose buffer Z64 points into, kept alive by this basis
};

namespace {
thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

template <class F>
int guarded(F&& f) {
  try {
    f();
    return GSI_OK;
  } catch (const Error& e) {
    return fail(e.code, e.what());
  } catch (const std::bad_alloc&) {
    return fail(GSI_ERR_OOM, "host allocation failed");
  } catch (const std::exception& e) {
    return fail(GSI_ERR_INTERNAL, e.what());
  } catch (...) {
    return fail(GSI_ERR_INTERNAL, "unknown error");
  }
}

// Runs f; if it fails because the backend lost a resource it no longer depends on (Backend::retryable_failure: the
// persistent LU kernel's co-residency on a shared GPU), runs it once more -- the entry points below never modify
// their inputs (operator, Omega, the host panel), so a second run starts from the same state.
template <class F>
void with_retry(Context& c, F&& f) {
  try {
    f();
  } catch (const Error&) {
    // GSI_NO_RETRY: tests of the error path.  With a communicator the ranks must agree on what runs next: no silent re-run
    // (every rank saw the same time-out and has switched paths; the caller repeats the collective call)
    if (!c.be->retryable_failure() || c.comm || getenv("GSI_NO_RETRY") != nullptr) throw;
    c.lu_timeouts_recovered += 1;      // visible in gsi_ctx_path_info: the re-run must not hide that a time-out happened
    f();
  }
}

#define REQUIRE(cond, msg) \
  do { if (!(cond)) throw Error(GSI_ERR_ARG, msg); } while (0)

void check_shard(Context& c, int64_t m, int64_t row0, int64_t mloc) {
  REQUIRE(row0 >= 0 && mloc >= 0 && row0 + mloc <= m, "row shard out of range");
  if (c.nranks() == 1) {
    REQUIRE(row0 == 0 && mloc == m, "single-rank context: the operator must hold all rows");
  } else {
    int64_t r0, ml;
    default_shard(m, c.nranks(), c.rank(), &r0, &ml);
    REQUIRE(r0 == row0 && ml == mloc,
            "multi-rank operators use the block-row layout: pad = ceil(m/nranks), row0 = rank*pad");
  }
}
}  // namespace

extern "C" {

int gsi_version(void) { return GSI_VERSION; }
const char* gsi_last_error(void) { return g_last_error.c_str(); }
const char* gsi_backend_name(void) { return backend_name(); }

int gsi_ctx_create(gsi_ctx** ctx, int device_id) {
  return guarded([&] {
    REQUIRE(ctx != nullptr, "ctx is NULL");
    *ctx = nullptr;
    gsi_ctx* c = new gsi_ctx();
    try {
      c->c.be.reset(make_backend(device_id));
    } catch (...) {
      delete c;
      throw;
    }
    *ctx = c;
  });
}

int gsi_ctx_destroy(gsi_ctx* ctx) {
  return guarded([&] {
    if (!ctx) return;
    ctx->c.comm.reset();
    ctx->c.be.reset();
    delete ctx;
  });
}

int gsi_ctx_sync(gsi_ctx* ctx) {
  return guarded([&] {
    REQUIRE(ctx, "ctx is NULL");
    ctx->c.be->sync();
  });
}

int gsi_comm_unique_id(void* id_out) {
  return guarded([&] {
    REQUIRE(id_out, "id_out is NULL");
    comm_unique_id(id_out);
  });
}

int gsi_ctx_comm_init(gsi_ctx* ctx, int nranks, int rank, const void* id) {
  return guarded([&] {
    REQUIRE(ctx && id, "NULL argument");
    REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank / nranks");
    REQUIRE(!ctx->c.comm, "communicator already initialised");
    // a 1-rank job needs no communicator; GSI_FORCE_COMM=1 creates one anyway so that the RCCL
    // path (library loading, stream-ordered collectives) can be exercised on a single GPU
    if (nranks == 1 && getenv("GSI_FORCE_COMM") == nullptr) return;
    ctx->c.comm.reset(make_comm(ctx->c.be.get(), nranks, rank, id));
    // how many ranks the communicator really joins: one per rank, summed through it
    Context& c = ctx->c;
    const double one = 1.0;
    double seen = 0.0;
    Buf b(c.be.get(), 1);
    c.be->upload2d(b.p, 1, &one, 1, 1, 1);
    c.comm->allreduce_sum(b.p, 1);
    c.be->download2d(&seen, 1, b.p, 1, 1, 1);
    c.ranks_seen = (int64_t)(seen + 0.5);
    c.be->set_ranks_sharing_device(c.comm->ranks_on_my_device());
    c.comm->ncollectives = 0;
  });
}

int gsi_ctx_host_allgather(gsi_ctx* ctx, const double* mine, int64_t count, double* all_out) {
  return guarded([&] {
    REQUIRE(ctx && mine && all_out && count >= 1, "bad argument");
    Context& c = ctx->c;
    c.be->sync();
    if (!c.comm) {
      std::memcpy(all_out, mine, sizeof(double) * (size_t)count);
      return;
    }
    const int G = c.nranks();
    Buf send(c.be.get(), (size_t)count), recv(c.be.get(), (size_t)count * G);
    c.be->upload2d(send.p, count, mine, count, count, 1);
    const int64_t before = c.comm->ncollectives;
    c.comm->allgather(send.p, recv.p, (size_t)count);
    c.comm->ncollectives = before;          // a host-side barrier / exchange of the caller's is not part of the data path
    c.be->download2d(all_out, count, recv.p, count, count, G);
  });
}

int gsi_ctx_path_info(gsi_ctx* ctx, int64_t* out, int64_t n_out) {
  return guarded([&] {
    REQUIRE(ctx && out && n_out >= 1, "bad argument");
    Context& c = ctx->c;
    int64_t v[GSI_PATH_INFO_COUNT] = {0};
    v[GSI_PATH_LU_FORM_LAST] = c.lu_form_last;
    v[GSI_PATH_LU_SELFTEST_MASK] = c.lus_mr_selftest;
    v[GSI_PATH_COLLECTIVES] = c.comm ? c.comm->ncollectives : 0;
    v[GSI_PATH_RANKS_SEEN] = c.ranks_seen;
    v[GSI_PATH_LU_TIMEOUTS] = c.be->lu_timeouts();
    v[GSI_PATH_LU_TIMEOUTS_RECOVERED] = c.lu_timeouts_recovered;
    for (int f = 0; f < (int)Context::LU_FORMS; ++f) v[GSI_PATH_LU_FORM_COUNTS + f] = c.lu_form_count[f];
    v[GSI_PATH_SVD_CAP_HITS] = c.be->svd_cap_hits();
    for (int64_t i = 0; i < n_out; ++i) out[i] = i < GSI_PATH_INFO_COUNT ? v[i] : 0;
  });
}

int gsi_ctx_pinned_copy_rate(gsi_ctx* ctx, int64_t bytes, double* h2d_gbs, double* d2h_gbs) {
  return guarded([&] {
    REQUIRE(ctx && h2d_gbs && d2h_gbs, "NULL argument");
    REQUIRE(bytes >= 0 && bytes <= ((int64_t)64 << 30), "bad size");
    ctx->c.be->pinned_copy_rate(bytes, h2d_gbs, d2h_gbs);
  });
}

int gsi_ctx_rank(gsi_ctx* ctx, int* rank, int* nranks) {
  return guarded([&] {
    REQUIRE(ctx, "ctx is NULL");
    if (rank) *rank = ctx->c.rank();
    if (nranks) *nranks = ctx->c.nranks();
  });
}

int gsi_op_dense(gsi_ctx* ctx, gsi_op** op, const double* A_rows, int64_t m, int64_t n, int64_t lda,
                 int64_t row0, int64_t m_local) {
  return guarded([&] {
    REQUIRE(ctx && op && A_rows, "NULL argument");
    *op = nullptr;
    REQUIRE(m >= 1 && n >= 1 && lda >= m_local && lda >= 1, "bad matrix shape");
    check_shard(ctx->c, m, row0, m_local);
    std::unique_ptr<gsi_op> o(new gsi_op());
    Operator& A = o->op;
    A.ctx = &ctx->c; A.kind = OP_DENSE; A.m = m; A.n = n; A.row0 = row0; A.mloc = m_local;
    A.ld = m_local > 0 ? ((m_local + 15) / 16) * 16 : 16;   // padded: 16-byte aligned column starts
    A.data = Buf(ctx->c.be.get(), (size_t)A.ld * n);
    ctx->c.be->upload2d(A.data.p, A.ld, A_rows, lda, m_local, n);
    *op = o.release();
  });
}

int gsi_op_lowrank(gsi_ctx* ctx, gsi_op** op, const double* samples, int64_t n, int64_t N, int64_t lds,
                   int center, int64_t row0, int64_t n_local) {
  return guarded([&] {
    REQUIRE(ctx && op && samples, "NULL argument");
    *op = nullptr;
    REQUIRE(n >= 1 && N >= 2 && lds >= n_local && lds >= 1, "bad sample matrix shape (need N >= 2 samples)");
    check_shard(ctx->c, n, row0, n_local);
    std::unique_ptr<gsi_op> o(new gsi_op());
    Operator& A = o->op;
    A.ctx = &ctx->c; A.kind = OP_LOWRANK; A.m = n; A.n = n; A.row0 = row0; A.mloc = n_local; A.N = N;
    A.ld = n_local > 0 ? ((n_local + 15) / 16) * 16 : 16;
    A.data = Buf(ctx->c.be.get(), (size_t)A.ld * N);
    ctx->c.be->upload2d(A.data.p, A.ld, samples, lds, n_local, N);
    if (center) ctx->c.be->center_rows(A.data.p, n_local, N, A.ld);   // lowrank.jl:17-27
    *op = o.release();
  });
}

int gsi_op_lowrank_synthetic(gsi_ctx* ctx, gsi_op** op, int64_t n, int64_t N, uint64_t seed, double decay,
                             int64_t row0, int64_t n_local) {
  return guarded([&] {
    REQUIRE(ctx && op, "NULL argument");
    *op = nullptr;
    REQUIRE(n >= 1 && N >= 2 && decay >= 0.0, "bad synthetic sample parameters (need N >= 2 samples, decay >= 0)");
    check_shard(ctx->c, n, row0, n_local);
    std::unique_ptr<gsi_op> o(new gsi_op());
    Operator& A = o->op;
    A.ctx = &ctx->c; A.kind = OP_LOWRANK; A.m = n; A.n = n; A.row0 = row0; A.mloc = n_local; A.N = N;
    A.ld = n_local > 0 ? ((n_local + 15) / 16) * 16 : 16;
    A.data = Buf(ctx->c.be.get(), (size_t)A.ld * N);
    ctx->c.be->fill_lowrank_samples(A.data.p, A.ld, n_local, N, row0, seed, decay);
    ctx->c.be->center_rows(A.data.p, n_local, N, A.ld);   // lowrank.jl:17-27
    *op = o.release();
  });
}

int gsi_op_lowrank_samples(gsi_ctx* ctx, const gsi_op* op, double* samples_out, int64_t ld) {
  return guarded([&] {
    REQUIRE(ctx && op && samples_out, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c, "operator belongs to another context");
    const Operator& A = op->op;
    REQUIRE(A.kind == OP_LOWRANK, "the operator is not a LowRankCovMatrix");
    REQUIRE(ld >= A.mloc && ld >= 1, "leading dimension too small");
    if (A.mloc > 0) ctx->c.be->download2d(samples_out, ld, A.data.p, A.ld, A.mloc, A.N);
  });
}

int gsi_op_dense_gridcov(gsi_ctx* ctx, gsi_op** op, int64_t nx, int64_t ny, double ell, int kind,
                         int64_t row0, int64_t m_local) {
  return guarded([&] {
    REQUIRE(ctx && op, "NULL argument");
    *op = nullptr;
    REQUIRE(nx >= 1 && ny >= 1 && ell > 0 && (kind == 0 || kind == 1), "bad grid covariance parameters");
    const int64_t n = nx * ny;
    check_shard(ctx->c, n, row0, m_local);
    std::unique_ptr<gsi_op> o(new gsi_op());
    Operator& A = o->op;
    A.ctx = &ctx->c; A.kind = OP_DENSE; A.m = n; A.n = n; A.row0 = row0; A.mloc = m_local;
    A.ld = m_local > 0 ? ((m_local + 15) / 16) * 16 : 16;
    A.data = Buf(ctx->c.be.get(), (size_t)A.ld * n);
    ctx->c.be->fill_gridcov(A.data.p, A.ld, nx, ny, ell, kind, row0, m_local);
    *op = o.release();
  });
}

// table[dx * ny + dy] = k(dx, dy): any stationary kernel on the regular grid
static void make_gridcov_table(gsi_ctx* ctx, gsi_op** op, int64_t nx, int64_t ny, const double* table, int64_t row0,
                               int64_t m_local) {
  REQUIRE(nx >= 1 && ny >= 1, "bad grid");
  REQUIRE(nx * ny < ((int64_t)1 << 31), "implicit grid covariance: more than 2^31 points");
  const int64_t n = nx * ny;
  check_shard(ctx->c, n, row0, m_local);
  std::unique_ptr<gsi_op> o(new gsi_op());
  Operator& A = o->op;
  A.ctx = &ctx->c; A.kind = OP_GRIDCOV_IMPLICIT; A.m = n; A.n = n; A.row0 = row0; A.mloc = m_local;
  A.gx = nx; A.gy = ny; A.ld = 0;
  A.data = Buf(ctx->c.be.get(), (size_t)n);
  ctx->c.be->upload2d(A.data.p, n, table, n, n, 1);
  *op = o.release();
}

int gsi_op_gridcov_implicit_kind(gsi_ctx* ctx, gsi_op** op, int64_t nx, int64_t ny, double ell, int kind, int64_t row0,
                                 int64_t m_local) {
  return guarded([&] {
    REQUIRE(ctx && op, "NULL argument");
    *op = nullptr;
    REQUIRE(nx >= 1 && ny >= 1 && ell > 0 && (kind == 0 || kind == 1),
            "bad grid covariance parameters (kind 0: Gaussian, 1: exponential)");
    std::vector<double> tab((size_t)(nx * ny));
    if (kind == 0) {     // exp(-(dx^2 + dy^2) / (2 ell^2)) as the product of its two 1-D factors
      std::vector<double> ex((size_t)nx), ey((size_t)ny);
      for (int64_t d = 0; d < nx; ++d) ex[(size_t)d] = std::exp(-(double)(d * d) / (2.0 * ell * ell));
      for (int64_t d = 0; d < ny; ++d) ey[(size_t)d] = std::exp(-(double)(d * d) / (2.0 * ell * ell));
      for (int64_t dx = 0; dx < nx; ++dx)
        for (int64_t dy = 0; dy < ny; ++dy) tab[(size_t)(dx * ny + dy)] = ex[(size_t)dx] * ey[(size_t)dy];
    } else {
      for (int64_t dx = 0; dx < nx; ++dx)
        for (int64_t dy = 0; dy < ny; ++dy)
          tab[(size_t)(dx * ny + dy)] = std::exp(-std::sqrt((double)(dx * dx + dy * dy)) / ell);
    }
    make_gridcov_table(ctx, op, nx, ny, tab.data(), row0, m_local);
  });
}

int gsi_op_gridcov_implicit(gsi_ctx* ctx, gsi_op** op, int64_t nx, int64_t ny, double ell, int64_t row0,
                            int64_t m_local) {
  return gsi_op_gridcov_implicit_kind(ctx, op, nx, ny, ell, 0, row0, m_local);
}

int gsi_op_gridcov_implicit_table(gsi_ctx* ctx, gsi_op** op, int64_t nx, int64_t ny, const double* table, int64_t row0,
                                  int64_t m_local) {
  return guarded([&] {
    REQUIRE(ctx && op && table, "NULL argument");
    *op = nullptr;
    make_gridcov_table(ctx, op, nx, ny, table, row0, m_local);
  });
}

int gsi_op_pointcov_implicit(gsi_ctx* ctx, gsi_op** op, const double* points, int64_t n, int d, int kind, double ell,
                             double sigma2, double nugget, int64_t row0, int64_t m_local) {
  return guarded([&] {
    REQUIRE(ctx && op && points, "NULL argument");
    *op = nullptr;
    REQUIRE(n >= 1 && d >= 1 && d <= 3, "point covariance: need n >= 1 points in 1, 2 or 3 dimensions");
    REQUIRE(kind >= 0 && kind < pointcov::NUM_KINDS, "point covariance: kind 0 Gaussian, 1 exponential, 2 Matern 3/2, 3 Matern 5/2");
    REQUIRE(ell > 0.0 && sigma2 > 0.0 && nugget >= 0.0, "point covariance: need ell > 0, sigma2 > 0, nugget >= 0");
    REQUIRE(std::isfinite(ell) && std::isfinite(sigma2) && std::isfinite(nugget), "point covariance: parameters must be finite");
    // The in-loader generator has no clamp in front of its exponential (every vector instruction there is matrix time lost):
    // it saturates to 0 by itself for arguments up to ~1e77 and would return inf / NaN beyond.  The accepted range
    // is checked here instead, once, on the host: finite coordinates within 1e29 correlation lengths of the first point
    // (squared scaled distances <= 3e58 x c1^2: 19 orders of magnitude inside what the generator handles).
    for (int64_t i = 0; i < n; ++i)
      for (int a = 0; a < d; ++a) {
        const double x = points[i * d + a], dx = (x - points[a]) / ell;
        REQUIRE(std::isfinite(x) && std::fabs(dx) <= 1e29,
                "point covariance: coordinates must be finite and within 1e29 correlation lengths of the first point");
      }
    check_shard(ctx->c, n, row0, m_local);
    std::unique_ptr<gsi_op> o(new gsi_op());
    Operator& A = o->op;
    A.ctx = &ctx->c; A.kind = OP_POINTCOV; A.m = n; A.n = n; A.row0 = row0; A.mloc = m_local; A.ld = 0;
    A.pc_d = d; A.pc_kind = kind; A.pc_ell = ell; A.pc_sigma2 = sigma2; A.pc_nugget = nugget;
    A.data = Buf(ctx->c.be.get(), (size_t)n * d);
    ctx->c.be->upload2d(A.data.p, d, points, d, d, n);       // every rank holds all coordinates (8 d n bytes)
    *op = o.release();
  });
}

static void make_fft_powerlaw(gsi_ctx* ctx, gsi_op** op, int ndims, const int64_t* N, double beta, int fftrf) {
  REQUIRE(ctx && op && N, "NULL argument");
  *op = nullptr;
  REQUIRE(ndims >= 1 && ndims <= 3, "fft covariance: 1, 2 or 3 grid dimensions");
  int64_t N3[3] = {1, 1, 1};
  int64_t n = 1;
  for (int a = 0; a < ndims; ++a) {
    REQUIRE(N[a] >= 1, "fft covariance: grid dimensions must be >= 1");
    N3[a] = N[a];
    n *= N[a];
  }
  REQUIRE(n >= 2, "fft covariance: need at least two grid points");
  std::unique_ptr<gsi_op> o(new gsi_op());
  Operator& A = o->op;
  A.ctx = &ctx->c; A.kind = OP_FFT_COV; A.m = n; A.n = n; A.ld = 0;
  // several ranks: every rank holds the plan (spectrum + work array) and transforms ITS columns of a panel; panels are
  // row-sharded between the products (pipeline.cpp: rows_to_cols / cols_to_rows), so row0 / mloc = the block-row layout
  default_shard(n, ctx->c.nranks(), ctx->c.rank(), &A.row0, &A.mloc);
  A.plan = ctx->c.be->fftcov_create(N3, beta, fftrf);
  *op = o.release();
}

int gsi_op_fft_powerlaw(gsi_ctx* ctx, gsi_op** op, int ndims, const int64_t* N, double beta) {
  return guarded([&] { make_fft_powerlaw(ctx, op, ndims, N, beta, 0); });
}

int gsi_op_fft_powerlaw_fftrf(gsi_ctx* ctx, gsi_op** op, int ndims, const int64_t* N, double beta) {
  return guarded([&] { make_fft_powerlaw(ctx, op, ndims, N, beta, 1); });
}

int gsi_op_destroy(gsi_op* op) {
  return guarded([&] { delete op; });
}

int gsi_op_size(const gsi_op* op, int64_t* m, int64_t* n, int64_t* row0, int64_t* m_local) {
  return guarded([&] {
    REQUIRE(op, "op is NULL");
    if (m) *m = op->op.m;
    if (n) *n = op->op.n;
    if (row0) *row0 = op->op.row0;
    if (m_local) *m_local = op->op.mloc;
  });
}

int gsi_op_mul(gsi_ctx* ctx, const gsi_op* op, int trans, const double* X, int64_t ldx, int64_t l, double* Y,
               int64_t ldy) {
  return guarded([&] {
    REQUIRE(ctx && op && X && Y, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c, "operator belongs to another context");
    const Operator& A = op->op;
    Backend* be = ctx->c.be.get();
    REQUIRE(l >= 1, "need at least one column");
    const int64_t rows_in = trans ? A.m : A.n, rows_out = trans ? A.n : A.m;
    REQUIRE(ldx >= rows_in && ldy >= rows_out, "leading dimension too small");
    Buf Xd(be, (size_t)rows_in * l), Yd(be, (size_t)rows_out * l);
    be->upload2d(Xd.p, rows_in, X, ldx, rows_in, l);
    if (!trans) {
      Buf Yloc(be, (size_t)(A.mloc > 0 ? A.mloc : 1) * l);
      op_mul(A, Xd.p, A.n, l, Yloc.p, A.mloc);
      gather_rows(ctx->c, A, Yloc.p, A.mloc, l, Yd.p);
    } else {
      op_mul_t(A, Xd.p + A.row0, A.m, l, Yd.p, A.n);
    }
    be->download2d(Y, ldy, Yd.p, rows_out, rows_out, l);
    check_async_errors(ctx->c);
  });
}

int gsi_rangefinder(gsi_ctx* ctx, const gsi_op* op, const double* Omega, int64_t l, int64_t numiterations,
                    double* Q_out) {
  return guarded([&] {
    REQUIRE(ctx && op && Omega && Q_out, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c, "operator belongs to another context");
    const Operator& A = op->op;
    Backend* be = ctx->c.be.get();
    REQUIRE(l >= 1, "l must be positive");
    Buf Om(be, (size_t)A.n * l);
    be->upload2d(Om.p, A.n, Omega, A.n, A.n, l);
    with_retry(ctx->c, [&] {
      Buf Qloc = rangefinder(A, Om.p, l, numiterations);
      Buf Qfull(be, (size_t)A.m * l);
      gather_rows(ctx->c, A, Qloc.p, A.mloc, l, Qfull.p);
      check_async_errors(ctx->c);
      be->download2d(Q_out, A.m, Qfull.p, A.m, A.m, l);
    });
  });
}

int gsi_randsvd(gsi_ctx* ctx, const gsi_op* op, const double* Omega, int64_t K, int64_t p, int64_t q,
                double* Z_out, double* S_out) {
  return guarded([&] {
    REQUIRE(ctx && op && Omega && Z_out, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c, "operator belongs to another context");
    REQUIRE(K >= 0 && p >= 0 && K + p >= 1, "need K >= 0, p >= 0, K + p >= 1");
    const Operator& A = op->op;
    Backend* be = ctx->c.be.get();
    const int64_t l = K + p;
    Buf Om(be, (size_t)A.n * l), Z(be, (size_t)A.n * l), S(be, (size_t)l);
    be->upload2d(Om.p, A.n, Omega, A.n, A.n, l);
    with_retry(ctx->c, [&] {
      randsvd(A, Om.p, K, p, q, Z.p, S.p);
      check_async_errors(ctx->c);
    });
    be->download2d(Z_out, A.n, Z.p, A.n, A.n, l);
    if (S_out) be->download2d(S_out, l, S.p, l, l, 1);
  });
}

// ---- the dense operator still in HOST memory: what getxis(Q::Matrix, ...) is called with ----------------------------
}  // extern "C" (the helpers below are C++)
namespace {
// A dense operator whose rows are on their way: allocated, the upload started in row blocks (single rank, large matrices;
// otherwise uploaded synchronously), `pending_upload` set for the first product.  The guard ends the transfer on every path.
struct UploadGuard {
  Backend* be = nullptr;
  void* h = nullptr;
  ~UploadGuard() {
    if (!h) return;
    try { be->upload2d_end(h); } catch (...) {}
  }
};
std::unique_ptr<gsi_op> dense_op_streaming(gsi_ctx* ctx, const double* A_host, int64_t m, int64_t n, int64_t lda, UploadGuard& g) {
  REQUIRE(m >= 1 && n >= 1 && lda >= m, "bad matrix shape");
  REQUIRE(ctx->c.nranks() == 1, "the host-matrix entry points are single-rank: with a communicator upload this rank's rows with gsi_op_dense");
  std::unique_ptr<gsi_op> o(new gsi_op());
  Operator& A = o->op;
  Backend* be = ctx->c.be.get();
  A.ctx = &ctx->c; A.kind = OP_DENSE; A.m = m; A.n = n; A.row0 = 0; A.mloc = m;
  A.ld = ((m + 15) / 16) * 16;
  A.data = Buf(be, (size_t)A.ld * n);
  const int64_t mb = be->upload_block_rows(m, n);
  g.be = be;
  g.h = be->upload2d_begin(A.data.p, A.ld, A_host, lda, m, n, mb);
  if (g.h != nullptr) { A.pending_upload = g.h; A.pending_block_rows = mb; }
  return o;
}
}  // namespace
extern "C" {

int gsi_randsvd_dense_host(gsi_ctx* ctx, const double* A_host, int64_t m, int64_t n, int64_t lda, const double* Omega,
                           int64_t K, int64_t p, int64_t q, double* Z_out, double* S_out, gsi_op** op_out) {
  return guarded([&] {
    REQUIRE(ctx && A_host && Omega && Z_out, "NULL argument");
    if (op_out) *op_out = nullptr;
    REQUIRE(K >= 0 && p >= 0 && K + p >= 1, "need K >= 0, p >= 0, K + p >= 1");
    Backend* be = ctx->c.be.get();
    const int64_t l = K + p;
    REQUIRE(m >= 1 && n >= 1 && lda >= m, "bad matrix shape");
    Buf Om(be, (size_t)n * l), Z(be, (size_t)n * l), S(be, (size_t)l);
    be->upload2d(Om.p, n, Omega, n, n, l);                  // Omega first: the first block's product needs all of it
    std::unique_ptr<gsi_op> o;                              // (declared before the guard: the transfer ends before the buffer goes)
    UploadGuard g;
    o = dense_op_streaming(ctx, A_host, m, n, lda, g);
    const Operator& A = o->op;
    with_retry(ctx->c, [&] {
      randsvd(A, Om.p, K, p, q, Z.p, S.p);
      check_async_errors(ctx->c);
    });
    be->download2d(Z_out, n, Z.p, n, n, l);
    if (S_out) be->download2d(S_out, l, S.p, l, l, 1);
    if (op_out) *op_out = o.release();
  });
}

int gsi_rangefinder_dense_host(gsi_ctx* ctx, const double* A_host, int64_t m, int64_t n, int64_t lda, const double* Omega,
                               int64_t l, int64_t numiterations, double* Q_out, gsi_op** op_out) {
  return guarded([&] {
    REQUIRE(ctx && A_host && Omega && Q_out, "NULL argument");
    if (op_out) *op_out = nullptr;
    REQUIRE(l >= 1, "l must be positive");
    REQUIRE(m >= 1 && n >= 1 && lda >= m, "bad matrix shape");
    Backend* be = ctx->c.be.get();
    Buf Om(be, (size_t)n * l);
    be->upload2d(Om.p, n, Omega, n, n, l);
    std::unique_ptr<gsi_op> o;                              // (declared before the guard: the transfer ends before the buffer goes)
    UploadGuard g;
    o = dense_op_streaming(ctx, A_host, m, n, lda, g);
    const Operator& A = o->op;
    with_retry(ctx->c, [&] {
      Buf Q = rangefinder(A, Om.p, l, numiterations);
      check_async_errors(ctx->c);
      be->download2d(Q_out, m, Q.p, m, m, l);
    });
    if (op_out) *op_out = o.release();
  });
}

int gsi_eig_nystrom(gsi_ctx* ctx, const gsi_op* op, const double* Q, int64_t j, double* U_out,
                    double* Sigma_out) {
  return guarded([&] {
    REQUIRE(ctx && op && Q && U_out && Sigma_out, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c, "operator belongs to another context");
    const Operator& A = op->op;
    Backend* be = ctx->c.be.get();
    REQUIRE(j >= 1 && j <= A.n, "bad number of columns in Q");
    Buf Qd(be, (size_t)A.n * j), U(be, (size_t)A.m * j), S(be, (size_t)j);
    be->upload2d(Qd.p, A.n, Q, A.n, A.n, j);
    eig_nystrom(A, Qd.p, j, U.p, S.p);
    be->download2d(U_out, A.m, U.p, A.m, A.m, j);
    be->download2d(Sigma_out, j, S.p, j, j, 1);
    check_async_errors(ctx->c);
  });
}

int gsi_rangefinder_adaptive(gsi_ctx* ctx, const gsi_op* op, gsi_randn_fn randn, void* user, double epsilon,
                             int64_t r, double* Q_out, int64_t* ncols_out) {
  return guarded([&] {
    REQUIRE(ctx && op && randn && Q_out && ncols_out, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c, "operator belongs to another context");
    *ncols_out = rangefinder_adaptive(op->op, randn, user, epsilon, r, Q_out);
    check_async_errors(ctx->c);
  });
}

// ---- device-resident matrices ------------------------------------------------------
int gsi_mat_create(gsi_ctx* ctx, gsi_mat** mat, int64_t rows, int64_t cols) {
  return guarded([&] {
    REQUIRE(ctx && mat, "NULL argument");
    *mat = nullptr;
    REQUIRE(rows >= 1 && cols >= 1, "bad shape");
    std::unique_ptr<gsi_mat> m(new gsi_mat());
    m->ctx = ctx; m->rows = rows; m->cols = cols;
    m->buf = Buf(ctx->c.be.get(), (size_t)rows * cols);
    *mat = m.release();
  });
}
int gsi_mat_destroy(gsi_mat* mat) {
  return guarded([&] { mat_unref(mat); });     // the buffer lives on while a 64-bit gsi_basis still points into it
}
int gsi_mat_upload(gsi_ctx* ctx, gsi_mat* mat, const double* host, int64_t ldh) {
  return guarded([&] {
    REQUIRE(ctx && mat && host && mat->ctx == ctx, "bad argument");
    REQUIRE(ldh >= mat->rows, "leading dimension too small");
    ctx->c.be->upload2d(mat->buf.p, mat->rows, host, ldh, mat->rows, mat->cols);
  });
}
int gsi_mat_download(gsi_ctx* ctx, const gsi_mat* mat, double* host, int64_t ldh) {
  return guarded([&] {
    REQUIRE(ctx && mat && host && mat->ctx == ctx, "bad argument");
    REQUIRE(ldh >= mat->rows, "leading dimension too small");
    ctx->c.be->download2d(host, ldh, mat->buf.p, mat->rows, mat->rows, mat->cols);
  });
}
int gsi_mat_randn(gsi_ctx* ctx, gsi_mat* mat, uint64_t seed) {
  return guarded([&] {
    REQUIRE(ctx && mat && mat->ctx == ctx, "bad argument");
    ctx->c.be->randn(mat->buf.p, (size_t)mat->rows * mat->cols, seed);
  });
}

int gsi_op_mul_dev(gsi_ctx* ctx, const gsi_op* op, int trans, const gsi_mat* X, gsi_mat* Y) {
  return guarded([&] {
    REQUIRE(ctx && op && X && Y, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c && X->ctx == ctx && Y->ctx == ctx, "objects belong to another context");
    const Operator& A = op->op;
    const int64_t l = X->cols;
    const int64_t rows_in = trans ? A.m : A.n, rows_out = trans ? A.n : A.m;
    REQUIRE(X->rows == rows_in && Y->rows == rows_out && Y->cols == l, "shape mismatch");
    if (!trans) {
      if (ctx->c.nranks() == 1) {
        op_mul(A, X->buf.p, A.n, l, Y->buf.p, A.m);
      } else {
        Buf Yloc(ctx->c.be.get(), (size_t)(A.mloc > 0 ? A.mloc : 1) * l);
        op_mul(A, X->buf.p, A.n, l, Yloc.p, A.mloc);
        gather_rows(ctx->c, A, Yloc.p, A.mloc, l, Y->buf.p);
      }
    } else {
      op_mul_t(A, X->buf.p + A.row0, A.m, l, Y->buf.p, A.n);
    }
  });
}

int gsi_rangefinder_dev(gsi_ctx* ctx, const gsi_op* op, const gsi_mat* Omega, int64_t numiterations,
                        gsi_mat* Q) {
  return guarded([&] {
    REQUIRE(ctx && op && Omega && Q, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c && Omega->ctx == ctx && Q->ctx == ctx, "objects belong to another context");
    const Operator& A = op->op;
    const int64_t l = Omega->cols;
    REQUIRE(Omega->rows == A.n, "Omega must have size(A,2) rows");
    REQUIRE(Q->rows == A.m && Q->cols == l, "Q must be size(A,1) x l");
    with_retry(ctx->c, [&] {
      Buf Qloc = rangefinder(A, Omega->buf.p, l, numiterations);
      gather_rows(ctx->c, A, Qloc.p, A.mloc, l, Q->buf.p);
      check_async_errors(ctx->c);
    });
  });
}

int gsi_randsvd_dev(gsi_ctx* ctx, const gsi_op* op, const gsi_mat* Omega, int64_t K, int64_t p, int64_t q,
                    gsi_mat* Z, gsi_mat* S) {
  return guarded([&] {
    REQUIRE(ctx && op && Omega && Z, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c && Omega->ctx == ctx && Z->ctx == ctx, "objects belong to another context");
    REQUIRE(K >= 0 && p >= 0 && K + p >= 1, "need K >= 0, p >= 0, K + p >= 1");
    const Operator& A = op->op;
    const int64_t l = K + p;
    REQUIRE(Omega->rows == A.n && Omega->cols == l, "Omega must be size(A,2) x (K+p)");
    REQUIRE(Z->rows == A.n && Z->cols == l, "Z must be size(A,2) x (K+p)");
    Backend* be = ctx->c.be.get();
    Buf Stmp;
    double* Sp;
    if (S) {
      REQUIRE(S->ctx == ctx && S->rows * S->cols == l, "S must hold K+p values");
      Sp = S->buf.p;
    } else {
      Stmp = Buf(be, (size_t)l);
      Sp = Stmp.p;
    }
    with_retry(ctx->c, [&] {
      randsvd(A, Omega->buf.p, K, p, q, Z->buf.p, Sp);
      check_async_errors(ctx->c);
    });
  });
}

int gsi_randsvd_rows(gsi_ctx* ctx, const gsi_op* op, const gsi_mat* Omega_rows, int64_t K, int64_t p, int64_t q,
                     gsi_mat* Z_rows, gsi_mat* S) {
  return guarded([&] {
    REQUIRE(ctx && op && Omega_rows && Z_rows, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c && Omega_rows->ctx == ctx && Z_rows->ctx == ctx, "objects belong to another context");
    REQUIRE(K >= 0 && p >= 0 && K + p >= 1, "need K >= 0, p >= 0, K + p >= 1");
    const Operator& A = op->op;
    const int64_t l = K + p;
    int64_t r0, nloc;
    default_shard(A.n, ctx->c.nranks(), ctx->c.rank(), &r0, &nloc);
    REQUIRE(nloc >= 1, "this rank holds no rows of the panel");
    REQUIRE(Omega_rows->rows == nloc && Omega_rows->cols == l, "Omega_rows must be (this rank's rows of size(A,2)) x (K+p)");
    REQUIRE(Z_rows->rows == nloc && Z_rows->cols == l, "Z_rows must be (this rank's rows of size(A,2)) x (K+p)");
    Backend* be = ctx->c.be.get();
    Buf Stmp;
    double* Sp;
    if (S) {
      REQUIRE(S->ctx == ctx && S->rows * S->cols == l, "S must hold K+p values");
      Sp = S->buf.p;
    } else {
      Stmp = Buf(be, (size_t)l);
      Sp = Stmp.p;
    }
    with_retry(ctx->c, [&] {
      randsvd_rows(A, Omega_rows->buf.p, K, p, q, Z_rows->buf.p, Sp);
      check_async_errors(ctx->c);
    });
  });
}

// ---- panel primitives ----------------------------------------------------------------
int gsi_lu_L(gsi_ctx* ctx, const double* Y, int64_t m, int64_t l, double* L_out, int32_t* ipiv_out) {
  return guarded([&] {
    REQUIRE(ctx && Y && L_out, "NULL argument");
    REQUIRE(m >= 1 && l >= 1 && l <= m, "lu_L: need 1 <= l <= m (tall panel)");
    Backend* be = ctx->c.be.get();
    Buf P(be, (size_t)m * l);
    with_retry(ctx->c, [&] {
      be->upload2d(P.p, m, Y, m, m, l);
      be->lu_L(P.p, m, l, m, ipiv_out);
      check_async_errors(ctx->c);
    });
    be->download2d(L_out, m, P.p, m, m, l);
  });
}

int gsi_lu_L_dev(gsi_ctx* ctx, gsi_mat* Y, int32_t* ipiv_out) {
  return guarded([&] {
    REQUIRE(ctx && Y, "NULL argument");
    REQUIRE(Y->ctx == ctx, "matrix belongs to another context");
    REQUIRE(Y->rows >= 1 && Y->cols >= 1 && Y->cols <= Y->rows, "lu_L: need 1 <= l <= m (tall panel)");
    Backend* be = ctx->c.be.get();
    {
      ScopedPhase ph(be, PH_LU);
      be->lu_L(Y->buf.p, Y->rows, Y->cols, Y->rows, ipiv_out);     // in place: a lost co-residency leaves no input to re-run from
    }
    check_async_errors(ctx->c);
  });
}

int gsi_lu_L_sharded(gsi_ctx* ctx, const double* Y, int64_t m, int64_t l, double* L_out, int32_t* ipiv_out) {
  return guarded([&] {
    REQUIRE(ctx && Y && L_out, "NULL argument");
    REQUIRE(m >= 1 && l >= 1 && l <= m, "lu_L: need 1 <= l <= m (tall panel)");
    Context& c = ctx->c;
    Backend* be = c.be.get();
    int64_t r0, ml;
    default_shard(m, c.nranks(), c.rank(), &r0, &ml);
    int64_t r00, ml0;
    default_shard(m, c.nranks(), 0, &r00, &ml0);
    REQUIRE(l <= ml0, "sharded lu: the first rank must hold the first l rows (l <= ceil(m / nranks))");
    Buf P(be, (size_t)std::max<int64_t>(ml, 1) * l);
    if (ml > 0) be->upload2d(P.p, ml, Y + r0, m, ml, l);
    lu_panel_sharded(c, P.p, m, r0, ml, l);
    Buf Full(be, (size_t)m * l);
    Operator shape;
    shape.m = m; shape.row0 = r0; shape.mloc = ml;
    gather_rows(c, shape, P.p, std::max<int64_t>(ml, 1), l, Full.p);
    be->download2d(L_out, m, Full.p, m, m, l);
    if (ipiv_out) be->lus_pivots(ipiv_out, l);
    check_async_errors(c);
  });
}

int gsi_lu_L_sharded_virtual(gsi_ctx* ctx, const double* Y, int64_t m, int64_t l, int nshards, double* L_out,
                             int32_t* ipiv_out) {
  return guarded([&] {
    REQUIRE(ctx && Y && L_out, "NULL argument");
    REQUIRE(m >= 1 && l >= 1 && l <= m, "lu_L: need 1 <= l <= m (tall panel)");
    REQUIRE(nshards >= 1 && nshards <= 64, "need 1 <= nshards <= 64");
    REQUIRE(nshards == 1 || l <= (m + nshards - 1) / nshards,
            "sharded lu: the first shard must hold the first l rows (l <= ceil(m / nshards))");
    Context& c = ctx->c;
    Backend* be = c.be.get();
    std::vector<Buf> shards;
    std::vector<double*> ptrs;
    for (int g = 0; g < nshards; ++g) {
      int64_t r0, ml;
      default_shard(m, nshards, g, &r0, &ml);
      shards.emplace_back(be, (size_t)std::max<int64_t>(ml, 1) * l);
      if (ml > 0) be->upload2d(shards.back().p, ml, Y + r0, m, ml, l);
      ptrs.push_back(shards.back().p);
    }
    lu_panel_sharded_virtual(c, ptrs.data(), m, l, nshards);
    check_async_errors(c);
    for (int g = 0; g < nshards; ++g) {
      int64_t r0, ml;
      default_shard(m, nshards, g, &r0, &ml);
      if (ml > 0) be->download2d(L_out + r0, m, ptrs[(size_t)g], ml, ml, l);
    }
    if (ipiv_out) be->lus_pivots(ipiv_out, l);
  });
}

int gsi_qr_thinQ(gsi_ctx* ctx, const double* Y, int64_t m, int64_t l, double* Q_out, double* R_out) {
  return guarded([&] {
    REQUIRE(ctx && Y && Q_out, "NULL argument");
    REQUIRE(m >= 1 && l >= 1 && l <= m, "qr_thinQ: need 1 <= l <= m (tall panel)");
    Backend* be = ctx->c.be.get();
    Buf P(be, (size_t)m * l), R(be, (size_t)l * l);
    be->upload2d(P.p, m, Y, m, m, l);
    {
      ScopedPhase ph(be, PH_QR);
      be->qr_thinQ(P.p, m, l, m, R.p);
    }
    be->download2d(Q_out, m, P.p, m, m, l);
    if (R_out) be->download2d(R_out, l, R.p, l, l, l);
    check_async_errors(ctx->c);
  });
}

int gsi_svd_tall(gsi_ctx* ctx, const double* W, int64_t n, int64_t l, double* V_out, double* S_out) {
  return guarded([&] {
    REQUIRE(ctx && W && V_out && S_out, "NULL argument");
    REQUIRE(n >= 1 && l >= 1 && l <= n, "svd_tall: need 1 <= l <= n");
    Backend* be = ctx->c.be.get();
    Buf Wd(be, (size_t)n * l), V(be, (size_t)n * l), S(be, (size_t)l);
    be->upload2d(Wd.p, n, W, n, n, l);
    svd_tall(ctx->c, Wd.p, n, l, -1, V.p, S.p);
    be->download2d(V_out, n, V.p, n, n, l);
    be->download2d(S_out, l, S.p, l, l, 1);
    check_async_errors(ctx->c);
  });
}

int gsi_gemm(gsi_ctx* ctx, int trans, int64_t m, int64_t l, int64_t k, double alpha, const double* A,
             int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc) {
  return guarded([&] {
    REQUIRE(ctx && A && B && C, "NULL argument");
    REQUIRE(m >= 1 && l >= 1 && k >= 1, "bad shape");
    const int64_t ar = trans ? k : m, ac = trans ? m : k;
    REQUIRE(lda >= ar && ldb >= k && ldc >= m, "leading dimension too small");
    Backend* be = ctx->c.be.get();
    Buf Ad(be, (size_t)ar * ac), Bd(be, (size_t)k * l), Cd(be, (size_t)m * l);
    be->upload2d(Ad.p, ar, A, lda, ar, ac);
    be->upload2d(Bd.p, k, B, ldb, k, l);
    if (trans) be->gemm_tn(m, l, k, alpha, Ad.p, ar, Bd.p, k, 0.0, Cd.p, m);
    else be->gemm_nn(m, l, k, alpha, Ad.p, ar, Bd.p, k, 0.0, Cd.p, m);
    be->download2d(C, ldc, Cd.p, m, m, l);
    check_async_errors(ctx->c);
  });
}

// ---- consumers -----------------------------------------------------------------------
namespace {
void pcga_params_impl(Context& c, const double* Zdev, int64_t n, int64_t K, const double* s, const double* X,
                      double delta, double* out) {
  Backend* be = c.be.get();
  Buf O(be, (size_t)n * (K + 3)), sv(be, (size_t)n), Xv(be, (size_t)n);
  be->upload2d(sv.p, n, s, n, n, 1);
  be->upload2d(Xv.p, n, X, n, n, 1);
  be->pcga_params(Zdev, n, K, sv.p, Xv.p, delta, O.p);           // direct.jl:39-45 / lsqr.jl:37-43
  be->download2d(out, n, O.p, n, n, K + 3);
}
void pcga_update_impl(Context& c, const double* Zdev, int64_t n, int64_t K, const double* X, double beta_bar,
                      const double* etas, int64_t nobs, const double* xi_bar, double* s_out) {
  Backend* be = c.be.get();
  Buf E(be, (size_t)nobs * K), xb(be, (size_t)nobs), w(be, (size_t)K), sd(be, (size_t)n), Xd(be, (size_t)n);
  be->upload2d(E.p, nobs, etas, nobs, nobs, K);
  be->upload2d(xb.p, nobs, xi_bar, nobs, nobs, 1);
  be->upload2d(Xd.p, n, X, n, n, 1);
  be->gemm_tn(K, 1, nobs, 1.0, E.p, nobs, xb.p, nobs, 0.0, w.p, K);     // w_i = dot(eta_i, xi_bar)
  be->scal_copy(n, beta_bar, Xd.p, sd.p);                               // s = X * beta_bar      direct.jl:61
  be->gemm_nn(n, 1, K, 1.0, Zdev, n, w.p, K, 1.0, sd.p, n);             // s += sum_i xis[i]*w_i  direct.jl:62-65
  be->download2d(s_out, n, sd.p, n, n, 1);
}
}  // namespace

int gsi_pcga_params(gsi_ctx* ctx, const double* Z, int64_t n, int64_t K, const double* s, const double* X,
                    double delta, double* out) {
  return guarded([&] {
    REQUIRE(ctx && Z && s && X && out, "NULL argument");
    REQUIRE(n >= 1 && K >= 1, "bad shape");
    Buf Zd(ctx->c.be.get(), (size_t)n * K);
    ctx->c.be->upload2d(Zd.p, n, Z, n, n, K);
    pcga_params_impl(ctx->c, Zd.p, n, K, s, X, delta, out);
  });
}

int gsi_pcga_update(gsi_ctx* ctx, const double* Z, int64_t n, int64_t K, const double* X, double beta_bar,
                    const double* etas, int64_t nobs, const double* xi_bar, double* s_out) {
  return guarded([&] {
    REQUIRE(ctx && Z && X && etas && xi_bar && s_out, "NULL argument");
    REQUIRE(n >= 1 && K >= 1 && nobs >= 1, "bad shape");
    Buf Zd(ctx->c.be.get(), (size_t)n * K);
    ctx->c.be->upload2d(Zd.p, n, Z, n, n, K);
    pcga_update_impl(ctx->c, Zd.p, n, K, X, beta_bar, etas, nobs, xi_bar, s_out);
  });
}

int gsi_pcga_params_dev(gsi_ctx* ctx, const gsi_mat* basis, int64_t K, const double* s, const double* X,
                        double delta, double* out) {
  return guarded([&] {
    REQUIRE(ctx && basis && s && X && out, "NULL argument");
    REQUIRE(basis->ctx == ctx && K >= 1 && K <= basis->cols, "bad basis / K");
    pcga_params_impl(ctx->c, basis->buf.p, basis->rows, K, s, X, delta, out);
  });
}

int gsi_pcga_update_dev(gsi_ctx* ctx, const gsi_mat* basis, int64_t K, const double* X, double beta_bar,
                        const double* etas, int64_t nobs, const double* xi_bar, double* s_out) {
  return guarded([&] {
    REQUIRE(ctx && basis && X && etas && xi_bar && s_out, "NULL argument");
    REQUIRE(basis->ctx == ctx && K >= 1 && K <= basis->cols && nobs >= 1, "bad basis / K / nobs");
    pcga_update_impl(ctx->c, basis->buf.p, basis->rows, K, X, beta_bar, etas, nobs, xi_bar, s_out);
  });
}

int gsi_mat_download_col(gsi_ctx* ctx, const gsi_mat* mat, int64_t col, double* host) {
  return guarded([&] {
    REQUIRE(ctx && mat && host && mat->ctx == ctx, "bad argument");
    REQUIRE(col >= 0 && col < mat->cols, "column out of range");
    ctx->c.be->download2d(host, mat->rows, mat->buf.p + (size_t)col * mat->rows, mat->rows, mat->rows, 1);
  });
}

// ---- LSQR consumers ------------------------------------------------------------------
int gsi_op_lowrank_solve(gsi_ctx* ctx, const gsi_op* op, const double* b, double* x_out, int64_t* iters_out) {
  return guarded([&] {
    REQUIRE(ctx && op && b && x_out, "NULL argument");
    REQUIRE(op->op.ctx == &ctx->c, "operator belongs to another context");
    const Operator& A = op->op;
    Backend* be = ctx->c.be.get();
    Buf bd(be, (size_t)A.n), xd(be, (size_t)A.n);
    be->upload2d(bd.p, A.n, b, A.n, A.n, 1);
    const int64_t it = lowrank_solve(A, bd.p, xd.p);
    be->download2d(x_out, A.n, xd.p, A.n, A.n, 1);
    if (iters_out) *iters_out = it;
    check_async_errors(ctx->c);
  });
}

int gsi_pcgamat_create(gsi_ctx* ctx, gsi_pcgamat** mat, const double* etas, int64_t nobs, int64_t K, const double* HX,
                       const double* R, int r_is_diag) {
  return guarded([&] {
    REQUIRE(ctx && mat && etas && HX && R, "NULL argument");
    *mat = nullptr;
    REQUIRE(nobs >= 1 && K >= 1, "bad shape");
    Backend* be = ctx->c.be.get();
    std::unique_ptr<gsi_pcgamat> m(new gsi_pcgamat());
    m->ctx = ctx;
    PcgaLowRank& A = m->A;
    A.ctx = &ctx->c; A.nobs = nobs; A.K = K; A.r_diag = (r_is_diag != 0);
    A.E = Buf(be, (size_t)nobs * K);
    A.HX = Buf(be, (size_t)nobs);
    A.R = Buf(be, A.r_diag ? (size_t)nobs : (size_t)nobs * nobs);
    be->upload2d(A.E.p, nobs, etas, nobs, nobs, K);
    be->upload2d(A.HX.p, nobs, HX, nobs, nobs, 1);
    if (A.r_diag) be->upload2d(A.R.p, nobs, R, nobs, nobs, 1);
    else be->upload2d(A.R.p, nobs, R, nobs, nobs, nobs);
    *mat = m.release();
  });
}
int gsi_pcgamat_destroy(gsi_pcgamat* mat) {
  return guarded([&] { delete mat; });
}
int gsi_pcgamat_mul(gsi_ctx* ctx, const gsi_pcgamat* mat, const double* x, double* y) {
  return guarded([&] {
    REQUIRE(ctx && mat && x && y && mat->ctx == ctx, "bad argument");
    Backend* be = ctx->c.be.get();
    const int64_t n1 = mat->A.nobs + 1;
    Buf xd(be, (size_t)n1), yd(be, (size_t)n1);
    be->upload2d(xd.p, n1, x, n1, n1, 1);
    mat->A.mul(xd.p, yd.p);
    be->download2d(y, n1, yd.p, n1, n1, 1);
    check_async_errors(ctx->c);
  });
}
int gsi_pcgamat_lsqr(gsi_ctx* ctx, const gsi_pcgamat* mat, const double* b, double* x_out, int64_t* iters_out) {
  return guarded([&] {
    REQUIRE(ctx && mat && b && x_out && mat->ctx == ctx, "bad argument");
    Backend* be = ctx->c.be.get();
    const int64_t n1 = mat->A.nobs + 1;
    Buf bd(be, (size_t)n1), xd(be, (size_t)n1);
    be->upload2d(bd.p, n1, b, n1, n1, 1);
    LsqrOperator L;
    L.nrows = L.ncols = n1;
    const PcgaLowRank* A = &mat->A;
    L.mul = [A](const double* xin, double* y) { A->mul(xin, y); };
    L.mul_t = L.mul;                                     // the saddle-point matrix is symmetric
    const int64_t it = lsqr(ctx->c, L, bd.p, xd.p, -1);  // maxiter = max(size(A))   IterativeSolvers default
    be->download2d(x_out, n1, xd.p, n1, n1, 1);
    if (iters_out) *iters_out = it;
    check_async_errors(ctx->c);
  });
}

// ---- xi-basis objects ------------------------------------------------------------------
int gsi_basis_create(gsi_ctx* ctx, gsi_basis** basis, const gsi_mat* Z, int64_t K, int precision) {
  return guarded([&] {
    REQUIRE(ctx && basis && Z && Z->ctx == ctx, "bad argument");
    *basis = nullptr;
    REQUIRE(K >= 1 && K <= Z->cols, "K out of range for this matrix");
    REQUIRE(precision == 64 || precision == 32, "precision must be 64 or 32");
    std::unique_ptr<gsi_basis> b(new gsi_basis());
    b->ctx = ctx; b->n = Z->rows; b->K = K; b->precision = precision; b->Z64 = Z->buf.p;
    if (precision == 32) {
      const size_t count = (size_t)b->n * (size_t)K;
      b->buf32 = Buf(ctx->c.be.get(), (count + 1) / 2);               // fp32 elements in a double-typed allocation
      ctx->c.be->f64_to_f32(Z->buf.p, b->buf32.p, count);
      b->Z64 = nullptr;
    } else {
      b->owner = const_cast<gsi_mat*>(Z);      // shared ownership: gsi_mat_destroy before gsi_basis_destroy is safe
      b->owner->refs += 1;
    }
    *basis = b.release();
  });
}
int gsi_basis_destroy(gsi_basis* basis) {
  return guarded([&] {
    if (!basis) return;
    gsi_mat* owner = basis->owner;
    delete basis;
    mat_unref(owner);
  });
}
int gsi_pcga_params_basis(gsi_ctx* ctx, const gsi_basis* basis, const double* s, const double* X, double delta,
                          double* out) {
  return guarded([&] {
    REQUIRE(ctx && basis && s && X && out && basis->ctx == ctx, "bad argument");
    if (basis->precision == 64) {
      pcga_params_impl(ctx->c, basis->Z64, basis->n, basis->K, s, X, delta, out);
      return;
    }
    Backend* be = ctx->c.be.get();
    const int64_t n = basis->n, K = basis->K;
    Buf O(be, (size_t)n * (K + 3)), sv(be, (size_t)n), Xv(be, (size_t)n);
    be->upload2d(sv.p, n, s, n, n, 1);
    be->upload2d(Xv.p, n, X, n, n, 1);
    be->pcga_params_f32(basis->buf32.p, n, K, sv.p, Xv.p, delta, O.p);
    be->download2d(out, n, O.p, n, n, K + 3);
  });
}
int gsi_pcga_update_basis(gsi_ctx* ctx, const gsi_basis* basis, const double* X, double beta_bar, const double* etas,
                          int64_t nobs, const double* xi_bar, double* s_out) {
  return guarded([&] {
    REQUIRE(ctx && basis && X && etas && xi_bar && s_out && basis->ctx == ctx && nobs >= 1, "bad argument");
    if (basis->precision == 64) {
      pcga_update_impl(ctx->c, basis->Z64, basis->n, basis->K, X, beta_bar, etas, nobs, xi_bar, s_out);
      return;
    }
    Backend* be = ctx->c.be.get();
    const int64_t n = basis->n, K = basis->K;
    Buf E(be, (size_t)nobs * K), xb(be, (size_t)nobs), w(be, (size_t)K), sd(be, (size_t)n), Xd(be, (size_t)n);
    be->upload2d(E.p, nobs, etas, nobs, nobs, K);
    be->upload2d(xb.p, nobs, xi_bar, nobs, nobs, 1);
    be->upload2d(Xd.p, n, X, n, n, 1);
    be->gemm_tn(K, 1, nobs, 1.0, E.p, nobs, xb.p, nobs, 0.0, w.p, K);          // w_i = dot(eta_i, xi_bar)
    be->basis_gemv_f32(basis->buf32.p, n, K, w.p, beta_bar, Xd.p, sd.p);      // s = X beta_bar + sum_i xis[i] w_i
    be->download2d(s_out, n, sd.p, n, n, 1);
  });
}
int gsi_basis_download_col(gsi_ctx* ctx, const gsi_basis* basis, int64_t col, double* host) {
  return guarded([&] {
    REQUIRE(ctx && basis && host && basis->ctx == ctx, "bad argument");
    REQUIRE(col >= 0 && col < basis->K, "column out of range");
    Backend* be = ctx->c.be.get();
    const int64_t n = basis->n;
    if (basis->precision == 64) {
      be->download2d(host, n, basis->Z64 + (size_t)col * n, n, n, 1);
      return;
    }
    // unit weight on column `col`: the fp32 column widened exactly
    std::vector<double> w((size_t)basis->K, 0.0), zero((size_t)n, 0.0);
    w[(size_t)col] = 1.0;
    Buf wd(be, (size_t)basis->K), Xd(be, (size_t)n), sd(be, (size_t)n);
    be->upload2d(wd.p, basis->K, w.data(), basis->K, basis->K, 1);
    be->fill_zero(Xd.p, (size_t)n);
    be->basis_gemv_f32(basis->buf32.p, n, basis->K, wd.p, 0.0, Xd.p, sd.p);
    be->download2d(host, n, sd.p, n, n, 1);
  });
}

// ---- measurement ---------------------------------------------------------------------
int gsi_ctx_profile(gsi_ctx* ctx, int enable) {
  return guarded([&] {
    REQUIRE(ctx, "ctx is NULL");
    ctx->c.be->profile(enable != 0);
    ctx->c.profile_level = enable;
  });
}
int gsi_ctx_phase_reset(gsi_ctx* ctx) {
  return guarded([&] {
    REQUIRE(ctx, "ctx is NULL");
    ctx->c.be->phase_reset();
    if (ctx->c.comm) ctx->c.comm->ncollectives = 0;
    for (int f = 0; f < (int)Context::LU_FORMS; ++f) ctx->c.lu_form_count[f] = 0;
  });
}
int gsi_ctx_phase_times(gsi_ctx* ctx, double* ms_out, int64_t* count_out) {
  return guarded([&] {
    REQUIRE(ctx && ms_out && count_out, "NULL argument");
    static_assert((int)PH_COUNT == GSI_NUM_PHASES, "phase tables out of sync");
    ctx->c.be->phase_times(ms_out, count_out);
  });
}
int gsi_ctx_counters(gsi_ctx* ctx, int64_t* out4) {
  return guarded([&] {
    REQUIRE(ctx && out4, "NULL argument");
    ctx->c.be->counters(out4);
  });
}
int gsi_ctx_release_cache(gsi_ctx* ctx) {
  return guarded([&] {
    REQUIRE(ctx, "ctx is NULL");
    ctx->c.be->release_cache();
  });
}
int gsi_ctx_device_bytes(gsi_ctx* ctx, int64_t* bytes) {
  return guarded([&] {
    REQUIRE(ctx && bytes, "NULL argument");
    *bytes = ctx->c.be->bytes_in_use();
  });
}

}  // extern "C"
