// hip_backend.hip -- the gfx950 implementation of gsi::Backend / gsi::Comm: one HIP stream
// per context, every operation enqueued on it (no host round trips except where the
// algorithm needs a scalar), workspaces grown on demand, RCCL for the collectives.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <atomic>
#include <mutex>
#include <string>
#include <chrono>
#include <map>
#include <memory>
#include <set>
#include <vector>
#include "../../include/gsi_hip.h"
#include "backend.hpp"
#include "hip_common.hpp"
#include "host_staging.hpp"

namespace gsi {

#define HIP_CHECK(expr)                                                                            \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      throw Error(_e == hipErrorOutOfMemory ? GSI_ERR_OOM : GSI_ERR_HIP,                            \
                  std::string(#expr) + ": " + hipGetErrorString(_e));                               \
  } while (0)

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool pooled = false;      // a panel-sized workspace taken from (and returned to) the block cache instead of hipMalloc / hipFree
};

// roctx ranges around the phases of the hot path: bound lazily, so the library has no hard
// dependency on the profiler's marker library; visible with `rocprofv3 --marker-trace`.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
static RoctxApi& roctx() { static RoctxApi api; return api; }
static const char* const kPhaseNames[PH_COUNT] = {"gsi:A*X", "gsi:A'*X", "gsi:lu(Y).L", "gsi:thin QR", "gsi:svd(l x l)",
                                                 "gsi:panel x (l x l)", "gsi:collective", "gsi:other", "gsi:waiting for peers"};

class HipBackend : public Backend {
 public:
  explicit HipBackend(int device) : device_(device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      throw Error(GSI_ERR_HIP, "no HIP device visible (libgsi_hip needs an AMD GPU; there is no CPU fallback)");
    if (device < 0 || device >= count) throw Error(GSI_ERR_ARG, "device id out of range");
    HIP_CHECK(hipSetDevice(device_));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device_));
    arch_ = prop.gcnArchName;
    ncus_ = prop.multiProcessorCount;
    if (arch_.find("gfx950") == std::string::npos)
      throw Error(GSI_ERR_HIP, "libgsi_hip is built for gfx950 (MI355X); device reports " + arch_);
    // non-blocking: nothing of this library uses the NULL stream, and a runtime-internal or foreign NULL-stream operation
    // must never wait for (or be waited for by) a persistent kernel of ours that spins for its peers
    HIP_CHECK(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    HIP_CHECK(hipMalloc(&flags_, 16 * sizeof(int32_t)));
    HIP_CHECK(hipMemsetAsync(flags_, 0, 16 * sizeof(int32_t), st_));
    HIP_CHECK(hipMalloc(&scal_, (64 + 8 + 256) * sizeof(double)));   // scalars + partial sums of long dot products
  }
  ~HipBackend() override {
    hipSetDevice(device_);
    hipStreamSynchronize(st_);
    stager_.reset();
    if (ev_stage_) hipEventDestroy(ev_stage_);
    for (auto& b : {&ws_gemm_, &ws_lu_, &ws_qr_, &ws_svd_, &ws_blas2_, &ws_lus_, &ws_svdf_, &ws_qr_hh_}) free_ws(*b);
    collect_garbage();
    for (auto& b : pool_) hipFree(b.p);
    for (auto& ev : ev_pool_) hipEventDestroy(ev);
    for (auto& r : records_) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    hipFree(flags_);
    hipFree(scal_);
    if (mr_recs_) hipFree(mr_recs_);
    hipStreamDestroy(st_);
  }
  const char* name() const override { return "hip-gfx950"; }
  hipStream_t stream() const { return st_; }
  void bind() { hipSetDevice(device_); }

  // ---- memory ----
  // Panel-sized temporaries come and go inside every entry point; everything is ordered on the one
  // stream, so a released block can be handed to the next request without synchronising or going back
  // to hipFree/hipMalloc.  Exact-size free lists (the sizes repeat from call to call), trimmed when the
  // cache exceeds half of what is in use.
  double* alloc(size_t count) override {
    bind();
    if (count == 0) count = 1;
    size_t bytes = count * sizeof(double);
    // Panel-sized requests are rounded up to 256 MiB: the n x l panel, the (n + 1) x l workspace of the out-of-place QR and the
    // even-ld copy of the thin SVD then have the SAME size and hand one cached block to each other instead of going through
    // hipFree / hipMalloc (at 512^3, l = 48 -- 51.5 GB blocks -- that was 4.9 s of an 11.4 s randsvd).
    if (bytes > ((size_t)1 << 30)) bytes = (bytes + (((size_t)256 << 20) - 1)) & ~(((size_t)256 << 20) - 1);
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t i = 0; i < pool_.size(); ++i)
        if (pool_[i].bytes == bytes) {
          void* p = pool_[i].p;
          pool_[i] = pool_.back();
          pool_.pop_back();
          pooled_ -= (int64_t)bytes;
          sizes_.push_back({p, bytes});
          in_use_ += (int64_t)bytes;
          return (double*)p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {   // give the cache back and retry once
      (void)hipGetLastError();
      trim_pool(0);
      e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      throw Error(GSI_ERR_OOM, "hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> g(mu_);
    sizes_.push_back({p, bytes});
    in_use_ += (int64_t)bytes;
    return (double*)p;
  }
  void release(double* p) override {
    if (!p) return;
    bind();
    size_t bytes = 0;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t i = 0; i < sizes_.size(); ++i)
        if (sizes_[i].p == p) {
          bytes = sizes_[i].bytes;
          in_use_ -= (int64_t)bytes;
          sizes_[i] = sizes_.back();
          sizes_.pop_back();
          break;
        }
      if (bytes != 0) {
        pool_.push_back({p, bytes});
        pooled_ += (int64_t)bytes;
        p = nullptr;
      }
    }
    if (p) {
      hipStreamSynchronize(st_);
      hipFree(p);
    }
    // Panels of the tall problems are GBs each and come back every pass: keep up to 160 GB of them (of 288 GB; a
    // failed hipMalloc empties the cache and retries, so nothing is ever refused because of it).  A tighter policy
    // cost 0.7 s of hipFree/hipMalloc per randsvd at n = 1.7e7 and 4.9 s at n = 1.3e8.
    // (ranks that share this device -- one-GPU rehearsals -- share the budget: a co-tenant cannot reach into this cache, it
    // would just see an out-of-memory error)
    const int64_t cap = ((int64_t)160 << 30) / std::max(1, ranks_sharing_device_);
    if (pooled_ > cap) trim_pool(cap / 3 * 2);
  }
  void trim_pool(int64_t keep_bytes) {
    hipStreamSynchronize(st_);
    std::lock_guard<std::mutex> g(mu_);
    while (!pool_.empty() && pooled_ > keep_bytes) {
      hipFree(pool_.back().p);
      pooled_ -= (int64_t)pool_.back().bytes;
      pool_.pop_back();
    }
  }
  void release_cache() override {
    bind();
    for (DevBuf* b : {&ws_gemm_, &ws_lu_, &ws_qr_, &ws_svd_, &ws_blas2_, &ws_lus_, &ws_svdf_, &ws_qr_hh_})
      if (b->bytes > ((size_t)64 << 20)) free_ws(*b);
    (void)hipStreamSynchronize(st_);
    collect_garbage();
    trim_pool(0);
  }
  int64_t bytes_in_use() const override {
    int64_t ws = 0;
    for (const DevBuf* b : {&ws_gemm_, &ws_lu_, &ws_qr_, &ws_svd_, &ws_blas2_, &ws_svdf_, &ws_qr_hh_})
      if (!b->pooled) ws += (int64_t)b->bytes;          // pooled workspaces are counted by alloc()
    return in_use_ + pooled_ + ws + (int64_t)garbage_bytes_;
  }
  // ---- the host boundary (host_staging.hpp): caller memory is pageable; transfers of GSI_STAGE_MIN_MB (default 16) MiB and
  // more go through the pinned staging ring (GSI_STAGE_THREADS workers, default 6 -- four reach the link rate on contiguous
  // sources, the strided row-block reads want the margin --, GSI_STAGE_CHUNK_MB MiB chunks, default 16: 0.98 of this box's
  // pinned-copy rate in both directions, profiles/r05_h2d_rates.log); smaller ones straight from the
  // caller's pages (54 us for C1's 768 KB Omega).  If the ring cannot be allocated the direct path serves everything.
  HostStager* stager() {
    if (stager_ || stager_failed_) return stager_.get();
    static const int threads = getenv("GSI_STAGE_THREADS") ? std::max(1, std::min(32, atoi(getenv("GSI_STAGE_THREADS")))) : 6;
    static const size_t chunk = (size_t)(getenv("GSI_STAGE_CHUNK_MB") ? std::max(1, std::min(1024, atoi(getenv("GSI_STAGE_CHUNK_MB")))) : 16) << 20;
    try {
      stager_.reset(new HostStager(device_, threads, chunk));
      HIP_CHECK(hipEventCreateWithFlags(&ev_stage_, hipEventDisableTiming));
    } catch (const std::exception&) {
      (void)hipGetLastError();
      stager_.reset();
      stager_failed_ = true;
    }
    return stager_.get();
  }
  static size_t stage_min_bytes() {
    const char* e = getenv("GSI_STAGE_MIN_MB");            // read per call (a getenv against a transfer of MiBs; tests switch it)
    return (size_t)(e ? std::max(0ll, atoll(e)) : 16) << 20;
  }
  bool staged(bool h2d, double* dev, int64_t ldd, double* host, int64_t ldh, int64_t rows, int64_t cols) {
    if (sizeof(double) * (size_t)rows * (size_t)cols < std::max<size_t>(stage_min_bytes(), 1)) return false;
    HostStager* s = stager();
    if (!s) return false;
    HIP_CHECK(hipEventRecord(ev_stage_, st_));       // copies start after everything queued so far (pooled destination / source in the making)
    try {
      // mid-size transfers (C1's 32 MB matrix): smaller rectangles, so that every worker has several in flight
      const size_t bytes = sizeof(double) * (size_t)rows * (size_t)cols;
      size_t chunk = std::min(s->chunk_bytes(), std::max<size_t>((size_t)1 << 20, (bytes / (4 * (size_t)s->threads())) & ~(((size_t)1 << 18) - 1)));
      s->begin(h2d, dev, ldd, host, ldh, stage_plan_whole(rows, cols, chunk), 1, ev_stage_);
      s->end();
    } catch (const Error&) {
      throw;
    } catch (const std::exception& e) {
      throw Error(GSI_ERR_HIP, e.what());
    }
    return true;
  }
  void upload2d(double* dst, int64_t ldd, const double* host, int64_t ldh, int64_t rows, int64_t cols) override {
    if (rows <= 0 || cols <= 0) return;
    bind();
    if (staged(true, dst, ldd, const_cast<double*>(host), ldh, rows, cols)) return;
    HIP_CHECK(hipMemcpy2DAsync(dst, ldd * sizeof(double), host, ldh * sizeof(double), rows * sizeof(double),
                               cols, hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));  // the caller may reuse / free the host buffer on return
  }
  void download2d(double* host, int64_t ldh, const double* src, int64_t lds, int64_t rows, int64_t cols) override {
    if (rows <= 0 || cols <= 0) return;
    bind();
    if (staged(false, const_cast<double*>(src), lds, host, ldh, rows, cols)) return;
    HIP_CHECK(hipMemcpy2DAsync(host, ldh * sizeof(double), src, lds * sizeof(double), rows * sizeof(double),
                               cols, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
  }
  // Row-block upload in the background (Backend::upload2d_begin): matrices of 256 MiB and more, in blocks of up to 32768 rows
  // = 256 output tiles, one full round of the contraction (what stays exposed after the upload is the LAST block's product,
  // ~10 ms at C2 whatever the block height, so tall blocks lose nothing) and 256 KB per column segment on the host side:
  // the workers' strided reads of the caller's matrix ran at 0.24 s per 8.6 GB against 0.35 s with 32 KB segments, and
  // with short segments the upload was sometimes bound by them (profiles/r05_boundary_probe.log).  At least two blocks.
  // GSI_STAGE_BLOCK_ROWS overrides (tests: many small blocks).
  int64_t upload_block_rows(int64_t rows, int64_t cols) override {
    const int64_t forced = getenv("GSI_STAGE_BLOCK_ROWS") ? atoll(getenv("GSI_STAGE_BLOCK_ROWS")) : 0;
    if (forced >= 128) return std::min(rows, (forced / 128) * 128);
    if (sizeof(double) * (size_t)rows * (size_t)cols < ((size_t)256 << 20) || rows < 2 * 4096) return rows;
    const int64_t half = (((rows + 1) / 2 + 127) / 128) * 128;
    return std::max<int64_t>(4096, std::min<int64_t>(32768, half));
  }
  void* upload2d_begin(double* dst, int64_t ldd, const double* host, int64_t ldh, int64_t rows, int64_t cols,
                       int64_t block_rows) override {
    if (rows <= 0 || cols <= 0) return nullptr;
    bind();
    HostStager* s = (block_rows < rows) ? stager() : nullptr;
    if (!s) { upload2d(dst, ldd, host, ldh, rows, cols); return nullptr; }
    HIP_CHECK(hipEventRecord(ev_stage_, st_));
    int nblocks = 0;
    std::vector<StageRect> plan = stage_plan_rowblocks(rows, cols, block_rows, s->chunk_bytes(), &nblocks);
    try {
      s->begin(true, dst, ldd, const_cast<double*>(host), ldh, std::move(plan), nblocks, ev_stage_);
    } catch (const std::exception& e) {
      throw Error(GSI_ERR_HIP, e.what());
    }
    return s;
  }
  void upload2d_wait_block(void* handle, int64_t b) override {
    if (!handle) return;
    bind();
    try {
      static_cast<HostStager*>(handle)->wait_block((int)b, st_);
    } catch (const std::exception& e) {
      throw Error(GSI_ERR_HIP, e.what());
    }
  }
  void pinned_copy_rate(int64_t bytes, double* h2d_gbs, double* d2h_gbs) override {
    bind();
    if (bytes < (1 << 20)) bytes = 1 << 20;
    void* pin = nullptr;
    HIP_CHECK(hipHostMalloc(&pin, (size_t)bytes, hipHostMallocDefault));
    struct FreePin { void* p; ~FreePin() { hipHostFree(p); } } fp{pin};
    memset(pin, 1, (size_t)bytes);
    Scratch dev(this, (size_t)bytes / sizeof(double) + 1);
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    struct FreeEv { hipEvent_t a, b; ~FreeEv() { hipEventDestroy(a); hipEventDestroy(b); } } fe{e0, e1};
    double best[2] = {0.0, 0.0};
    for (int dir = 0; dir < 2; ++dir)
      for (int rep = 0; rep < 3; ++rep) {
        HIP_CHECK(hipEventRecord(e0, st_));
        if (dir == 0) HIP_CHECK(hipMemcpyAsync(dev.p, pin, (size_t)bytes, hipMemcpyHostToDevice, st_));
        else HIP_CHECK(hipMemcpyAsync(pin, dev.p, (size_t)bytes, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipEventRecord(e1, st_));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms > 0.f) best[dir] = std::max(best[dir], (double)bytes / (ms * 1e-3) / 1e9);
      }
    *h2d_gbs = best[0];
    *d2h_gbs = best[1];
  }
  void upload2d_end(void* handle) override {
    if (!handle) return;
    try {
      static_cast<HostStager*>(handle)->end();
    } catch (const std::exception& e) {
      throw Error(GSI_ERR_HIP, e.what());
    }
  }
  void copy2d(double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols) override {
    if (rows <= 0 || cols <= 0) return;
    bind();
    HIP_CHECK(hipMemcpy2DAsync(dst, ldd * sizeof(double), src, lds * sizeof(double), rows * sizeof(double),
                               cols, hipMemcpyDeviceToDevice, st_));
  }
  void fill_zero(double* p, size_t count) override {
    bind();
    HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(double), st_));
  }
  void sync() override {
    bind();
    HIP_CHECK(hipStreamSynchronize(st_));
  }

  // ---- products ----
  void gemm_nn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda, const double* B,
               int64_t ldb, double beta, double* C, int64_t ldc) override {
    bind();
    double* ws = gemm_ws(hipk::gemm_workspace_doubles(m, l, k));
    hipk::gemm_f64(st_, false, m, l, k, alpha, A, lda, B, ldb, beta, C, ldc, ws);
    check_launch("gemm_nn");
  }
  void gemm_nn_rowblock(int64_t m_full, int64_t r0, int64_t mb, int64_t l, int64_t k, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double* C, int64_t ldc) override {
    bind();
    double* ws = gemm_ws(hipk::gemm_rowblock_workspace_doubles(m_full, mb, l, k));
    hipk::gemm_f64_nn_rowblock(st_, m_full, r0, mb, l, k, A, lda, B, ldb, C, ldc, ws);
    check_launch("gemm_nn_rowblock");
  }
  void gemm_tn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda, const double* B,
               int64_t ldb, double beta, double* C, int64_t ldc) override {
    bind();
    double* ws = gemm_ws(hipk::gemm_workspace_doubles(m, l, k));
    hipk::gemm_f64(st_, true, m, l, k, alpha, A, lda, B, ldb, beta, C, ldc, ws);
    check_launch("gemm_tn");
  }
  void gemm_nn_gridcov(int64_t m, int64_t l, int64_t k, const double* tab, int64_t nx, int64_t ny, int64_t roff,
                       int64_t koff, const double* B, int64_t ldb, double* C, int64_t ldc) override {
    bind();
    double* ws = gemm_ws(hipk::gemm_workspace_doubles(m, l, k));
    hipk::gemm_f64_gridcov(st_, m, l, k, tab, nx, ny, roff, koff, B, ldb, C, ldc, ws);
    check_launch("gemm_nn_gridcov");
  }

  // Scattered-point covariance, row-streamed: the entries are generated inside the contraction's tile loader (gemm_f64.hip
  // GEN 2: 128 x 160 tiles; pointcov_gemm.hip: 96 x 320) -- nothing of A ever exists in HBM.  (Round 3's form, row panels of A
  // generated into HBM on a second stream, is kept as tools/rejected_kernels/pointcov_round3_panels.hip.txt: 39 TFLOP/s
  // against 59.)
  void gemm_nn_pointcov(int64_t m, int64_t l, int64_t k, const double* pts, int d, int kind, double ell, double sigma2,
                        double nugget, int64_t roff, int64_t koff, const double* B, int64_t ldb, double* C,
                        int64_t ldc) override {
    bind();
    if (m <= 0 || l <= 0 || k <= 0) return;
    const int64_t npts = std::max(roff + m, koff + k);
    Scratch p4(this, (size_t)4 * npts);                  // the points as 32-byte records (x, y, z, 0): one scalar load each
    hipk::pointcov_pad_points(st_, pts, d, npts, hipk::pointcov_point_scale(kind, 1.0 / ell), p4.p);
    double* ws = gemm_ws(hipk::gemm_pointcov_workspace_doubles(m, l, k));
    // the wide kernel streams a tile-ordered copy of the sketch panel (k x l doubles); memory is never a reason to fail: without
    // the copy the product runs on the 128 x 160 kernel, which reads X where it lies
    double* xpack = nullptr;
    const size_t xdoubles = hipk::gemm_pointcov_pack_doubles(m, l, k);          // 0: not the wide kernel's product
    if (xdoubles > 0) { try { xpack = alloc(xdoubles); } catch (const Error&) { xpack = nullptr; } }
    struct Free { HipBackend* be; double* p; ~Free() { if (p) be->release(p); } } xfree{this, xpack};
    hipk::gemm_f64_pointcov(st_, m, l, k, p4.p, npts, d, kind, sigma2, nugget, roff, koff, B, ldb, C, ldc, ws, xpack);
    check_launch("gemm_nn_pointcov (in-loader generator)");
  }

  // ---- matrix-free FFT covariance ----
  struct FftCov { int64_t N[3], M[3]; double* lam; double* W; int nb_max; };
  void* fftcov_create(const int64_t N[3], double beta, int fftrf) override {
    bind();
    std::unique_ptr<FftCov> p(new FftCov());
    int64_t Mtot = 1;
    int d = 0;
    for (int a = 0; a < 3; ++a) { p->N[a] = 1; p->M[a] = 1; }
    bool reembed = false;                // FFTRF's exact 2 N embedding is not a power of two on some axis
    for (int a = 0; a < 3; ++a) {        // squeeze singleton axes: vec() of a 1 x 50 field is a 50-point line
      if (N[a] < 1) throw Error(GSI_ERR_ARG, "fft covariance: grid dimensions must be >= 1");
      if (N[a] == 1) continue;
      p->N[d] = N[a]; p->M[d] = hipk::fft_embed_size(N[a]);
      if (fftrf && p->M[d] != 2 * N[a]) reembed = true;
      if (p->M[d] > 8192) throw Error(GSI_ERR_ARG, "fft covariance: at most 4096 grid points per axis (a line must fit LDS)");
      Mtot *= p->M[d];
      ++d;
    }
    // element offsets inside one column pair's array are 32-bit in the pass kernels (fft_cov.hip): refuse such grids HERE, before
    // the spectrum and the work array (tens of GB) are allocated, not at the first product
    if (Mtot >= ((int64_t)1 << 31))
      throw Error(GSI_ERR_ARG, "fft covariance: the embedding grid must have fewer than 2^31 points (e.g. 1024 x 512 x 512 embeds to 2^31)");
    // work array: as many column pairs at once as fit ~2 GB, at most 64.  (GSI_FFT_W_MB: the experiment of VERDICT r4 item 7 --
    // few enough pairs per batch that the array between the passes stays in the 256 MB Infinity Cache: DESIGN.md 4.6,
    // profiles/r05_fft_pair_major.log)
    // 2-D grids whose pair array is <= 128 MB run 256 MB batches: +4 % at 1000^2 (6.57 -> 6.31 ms per 256 columns), the only
    // place the experiment moved anything; 128 MB and less lose to the shorter launches.
    static const int64_t w_env = getenv("GSI_FFT_W_MB") ? std::max<int64_t>(1, atoll(getenv("GSI_FFT_W_MB"))) : 0;
    const int64_t w_mb = w_env > 0 ? w_env : ((d == 2 && 16 * Mtot <= ((int64_t)128 << 20)) ? 256 : 2048);
    int64_t nb = (w_mb << 20) / (16 * Mtot);
    p->nb_max = (int)std::max<int64_t>(1, std::min<int64_t>(nb, 64));
    p->lam = alloc(hipk::fft_plan_doubles(p->M));
    try { p->W = alloc((size_t)2 * Mtot * p->nb_max); } catch (...) { release(p->lam); throw; }
    if (!reembed) {
      hipk::fft_spectrum(st_, p->lam, p->lam + Mtot, p->M, beta, fftrf);
      check_launch("fft_spectrum");
      return p.release();
    }
    // FFTRF's convention on an arbitrary grid (FFTRF.jl:83-90 embeds on exactly 2 N points for ANY N): the same matrix
    // through the power-of-two passes with a re-embedded spectrum (fft_cov.hip, "FFTRF's convention on a grid ...").
    //   c  = (x_a C_a) lambda_{2N}    C_a[t, k]  = cos(2 pi t k / 2 N_a),            t < N_a, k < 2 N_a     (lags 0 .. N_a - 1)
    //   l' = (x_a D_a) c              D_a[k', t] = w_t cos(2 pi t k' / M'_a), w_0 = 1, w_t = 2              (c' even, 0 beyond)
    // Each factor is applied to the FASTEST axis as cur' = cur^T B (the contraction kernel's A'B form: the array as a
    // k x rest matrix), which also rotates that axis to the back: d applications restore the natural order.
    try {
      int64_t cur[3] = {1, 1, 1};
      for (int a = 0; a < d; ++a) cur[a] = 2 * p->N[a];
      int64_t tot = cur[0] * cur[1] * cur[2];
      // every temporary of the re-embedding goes back to the block cache on every exit path (a throwing HIP_CHECK
      // or allocation between alloc and release leaked pooled blocks)
      struct Hold { HipBackend* be; double* p; ~Hold() { if (p) be->release(p); } } bufh{this, nullptr};
      bufh.p = alloc((size_t)tot);
      double*& buf = bufh.p;
      hipk::fft_spectrum_natural(st_, buf, cur, beta, 1);
      auto rotate_apply = [&](int64_t rows_out, int64_t period, bool weighted) {
        // cur = (k, rest) column-major; B = k x rows_out; result (rest x rows_out) = the array with axis 0 replaced and last
        const int64_t k = cur[0], rest = (tot / k);
        Hold Bh{this, alloc((size_t)k * rows_out)};
        double* B = Bh.p;
        // B[kk + t k] : weighted (the D factors) is indexed [t = row of the INPUT, k' = output]; unweighted [k, t]
        if (!weighted) hipk::fft_cos_matrix(st_, B, k, rows_out, period, false);          // C_a^T: B[k, t]
        else hipk::fft_cos_matrix(st_, B, k, rows_out, period, true);                     // D_a^T: B[t, k'] = w_t cos(.)
        Hold outh{this, alloc((size_t)rest * rows_out)};
        double* out = outh.p;
        double* ws = gemm_ws(hipk::gemm_workspace_doubles(rest, rows_out, k));
        hipk::gemm_f64(st_, true, rest, rows_out, k, 1.0, buf, k, B, k, 0.0, out, rest, ws);
        check_launch("fft re-embedding product");
        release(buf);
        buf = out;
        outh.p = nullptr;
        tot = rest * rows_out;
        // rotate the dimension list: (k, c1, c2) -> (c1, c2, rows_out) over the d real axes
        int64_t nd[3] = {1, 1, 1};
        for (int a = 0; a + 1 < d; ++a) nd[a] = cur[a + 1];
        nd[d - 1] = rows_out;
        for (int a = 0; a < 3; ++a) cur[a] = nd[a];
      };
      for (int a = 0; a < d; ++a) rotate_apply(p->N[a], 2 * p->N[a], false);   // lags 0 .. N_a - 1 of c
      for (int a = 0; a < d; ++a) rotate_apply(p->M[a], p->M[a], true);        // the spectrum of the re-embedded kernel
      hipk::fft_lines_layout(st_, buf, p->lam, p->M);
      hipk::fft_finish_plan(st_, p->lam, p->lam + Mtot, p->M);
      check_launch("fft re-embedded spectrum");
      release(buf);
      buf = nullptr;                                          // (bufh's destructor must not release it a second time)
    } catch (...) {
      release(p->W); release(p->lam);
      throw;
    }
    return p.release();
  }
  void fftcov_destroy(void* plan) override {
    FftCov* p = static_cast<FftCov*>(plan);
    if (!p) return;
    release(p->W); release(p->lam);
    delete p;
  }
  void fftcov_apply(void* plan, int64_t l, const double* X, int64_t ldx, double* Y, int64_t ldy) override {
    bind();
    FftCov* p = static_cast<FftCov*>(plan);
    hipk::fft_cov_apply(st_, p->N, p->M, p->lam, reinterpret_cast<double2*>(p->W), p->nb_max, l, X, ldx, Y, ldy);
    check_launch("fft_cov_apply");
  }

  // ---- panels ----
  void lu_L(double* Y, int64_t m, int64_t l, int64_t ld, int32_t* ipiv_host) override {
    bind();
    // Panels of up to 4096 rows per CU: leaves held in registers, left-looking blocks, streaming rank-64 updates
    // (panel_lu_leaf.hip).  Everything else takes the streamed leaves below.
    hipk::Lu2Work w2;
    static const bool tall_first = (getenv("GSI_LU_TALL") != nullptr && getenv("GSI_LU_TALL")[0] == '1');
    // Panels taller than the register file holds (up to GSI_LU_OV_MAX rows, default 5 x 2^20): the resident kernel on all
    // CUs with the rows beyond its window evaluated lazily out of L2 / Infinity Cache / HBM; beyond that the streamed leaves
    // win (measured cross-over: 6e6 rows at l = 320).  GSI_LU_OV=0 switches it off (A/B).
    static const int64_t ov_max = getenv("GSI_LU_OV_MAX") ? atoll(getenv("GSI_LU_OV_MAX")) : ((int64_t)5 << 20);
    static const bool ov_off = (getenv("GSI_LU_OV") != nullptr && getenv("GSI_LU_OV")[0] == '0');
    // Ranks that SHARE this device (the RCCL-free communicators, one-GPU rehearsals of a multi-GPU job) each factor their
    // replicated panel with a whole-chip grid of their own: the grids cannot all be resident, so the persistent kernel is
    // not a candidate at all (seen: 2 rank processes on one GPU, gathered 10^6 x 320 panel: a poll time-out on some runs).
    const bool resident_ok = !tall_first && !lu2_lost_ && ranks_sharing_device_ <= 1;
    bool fits = resident_ok && hipk::lu2_config(m, ncus_, &w2.bs, &w2.rpt, &w2.grid) && lu2_fits(w2.bs, w2.rpt, w2.grid);
    if (!fits && resident_ok && !ov_off && m <= ov_max && ncus_ >= 1) {
      const int g = std::min(ncus_, 256);
      if (m > (int64_t)g * 4096) {
        if (lu2_ov_resident_ < 0) lu2_ov_resident_ = hipk::lu2_resident_per_cu_ov();
        if ((int64_t)lu2_ov_resident_ * ncus_ >= g) { w2.bs = 512; w2.rpt = 8; w2.grid = g; w2.ov = true; fits = true; }
      }
    }
    if (fits) {
      // test / A-B knobs, read per call: GSI_LU_POLL_LIMIT (polls before a workgroup gives up), GSI_LU_TEST_MUTE_EPOCH
      // (one workgroup stays silent at that pivot step: exercises the info = -1 path), GSI_LU_COOPERATIVE=1
      if (const char* e = getenv("GSI_LU_POLL_LIMIT")) w2.poll_limit = atoi(e);
      if (const char* e = getenv("GSI_LU_TEST_MUTE_EPOCH")) w2.mute_epoch = (uint32_t)atoi(e);
      if (const char* e = getenv("GSI_LU_COOPERATIVE")) w2.cooperative = (e[0] == '1');
      static const int nb_env = getenv("GSI_LU_NB") ? atoi(getenv("GSI_LU_NB")) : 0;
      w2.nb = (nb_env == 32 || nb_env == 64) ? nb_env : hipk::LU2_NB;
      const size_t rec_bytes = sizeof(unsigned long long) * 2 * ((size_t)w2.grid + hipk::LU2_RES_COPIES) * hipk::LU2_REC_GRANULES;
      const size_t u12_bytes = sizeof(double) * (size_t)w2.nb * (size_t)l;
      grow(ws_lu_, rec_bytes + u12_bytes + sizeof(int32_t) * (l + 4) + 256);
      char* base = (char*)ws_lu_.p;
      w2.recs = (unsigned long long*)base; base += rec_bytes;
      w2.u12 = (double*)base; base += u12_bytes;
      w2.ipiv = (int32_t*)base;
      w2.info = flags_ + 0;
      hipk::lu2_L(st_, Y, m, l, ld, w2);
      check_launch("lu2_L");
      if (ipiv_host) {
        HIP_CHECK(hipMemcpyAsync(ipiv_host, w2.ipiv, sizeof(int32_t) * l, hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
      }
      return;
    }
    // Everything the resident kernel does not take -- panels the register file cannot hold (more than 4096 rows per CU), a
    // context that lost co-residency once, a leaf grid that does not fit the chip: streamed leaves with lazily evaluated
    // candidates (same blocks, pivots and arithmetic; no spin-waits between workgroups).  GSI_LU_TALL=1 forces it for any
    // height (tests: bit-identical to the resident kernel).  (Round 1's per-column sweeps, the fall-back behind these through
    // round 4, are kept as tools/rejected_kernels/panel_lu_round1_sweeps.hip.txt.)
    if (m >= ((int64_t)1 << 31)) throw Error(GSI_ERR_ARG, "lu: panels of 2^31 rows and more are not supported");
    grow(ws_lu_, hipk::lu3_work_bytes(l));
    int32_t* ipiv_dev = nullptr;
    hipk::lu3_L(st_, Y, m, l, ld, ws_lu_.p, flags_ + 0, &ipiv_dev);
    check_launch("lu3_L");
    if (ipiv_host) {
      HIP_CHECK(hipMemcpyAsync(ipiv_host, ipiv_dev, sizeof(int32_t) * l, hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipStreamSynchronize(st_));
    }
  }
  // ---- row-sharded LU primitives ----
  int lus_block() const override { return hipk::LU2_NB; }
  struct LusWs { double* pval; int64_t* pidx; int32_t* ipiv; };
  LusWs lus_ws(int64_t, int64_t l) {
    // fixed layout: pivots (l <= 8188), then the per-workgroup partial arg-maxes (<= 1024 workgroups)
    if (l > 8188) throw Error(GSI_ERR_ARG, "sharded lu: sketch width too large");
    grow(ws_lus_, 32768 + 2 * 8 * 1024 + 64);
    char* base = (char*)ws_lus_.p;
    LusWs w;
    w.ipiv = (int32_t*)base;
    w.pval = (double*)(base + 32768);
    w.pidx = (int64_t*)(base + 32768 + 8 * 1024);
    return w;
  }
  void lus_u12_leaf(const double* Yloc, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w, double* U12) override {
    bind();
    hipk::lus_u12_leaf(st_, Yloc, ld, row0, jb, j0, w, U12);
    check_launch("lus_u12_leaf");
  }
  void lus_pending(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t jb, int64_t j0, int w,
                   const double* U12) override {
    bind();
    if (mloc > 0) hipk::lus_pending(st_, Yloc, ld, mloc, row0, jb, j0, w, U12);
    check_launch("lus_pending");
  }
  void lus_candidate(const double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t l, int64_t j, double* rec) override {
    bind();
    LusWs w = lus_ws(mloc, l);
    // the previous pivot step's apply kernel may already have left this column's per-workgroup partials (same panel, same rows)
    const bool ready = (lus_part_col_ == j && lus_part_Y_ == Yloc && lus_part_mloc_ == mloc);
    lus_part_col_ = -1;
    hipk::lus_candidate(st_, Yloc, ld, mloc, row0, l, j, rec, w.pval, w.pidx, ready);
    check_launch("lus_candidate");
  }
  void lus_apply(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t m, int64_t l, int64_t j0, int s, int w,
                 const double* recs, int nranks) override {
    bind();
    LusWs ws = lus_ws(mloc, l);
    static const bool fuse = (getenv("GSI_LUS_NO_FUSE") == nullptr);      // A/B knob
    const bool next = fuse && (s + 1 < w) && mloc > 0;
    hipk::lus_apply(st_, Yloc, ld, mloc, row0, m, l, j0, s, w, recs, nranks, ws.ipiv, flags_ + 0, next ? ws.pval : nullptr,
                    next ? ws.pidx : nullptr);
    check_launch("lus_apply");
    lus_part_col_ = next ? j0 + s + 1 : -1;
    lus_part_Y_ = Yloc;
    lus_part_mloc_ = mloc;
  }
  void lus_u12_block(const double* Yloc, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t c1,
                     double* U12) override {
    bind();
    hipk::lus_u12_block(st_, Yloc, ld, row0, jb, b, c0, c1, U12);
    check_launch("lus_u12_block");
  }
  void lus_rankk(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t jb, int b, int64_t c0, int64_t t,
                 const double* U12) override {
    bind();
    hipk::lus_rankk(st_, Yloc, ld, mloc, row0, jb, b, c0, t, U12);
    check_launch("lus_rankk");
  }
  void lus_finish(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t l) override {
    bind();
    hipk::lus_finish(st_, Yloc, ld, mloc, row0, l);
  }
  bool lus_mr_begin(Comm* comm, int64_t m, int64_t l) override {
    bind();
    auto no = [&](const std::string& why) { mr_reason_ = why; return false; };
    if (comm == nullptr) return no("no communicator");
    // First use over this communicator: the record buffer is created and exchanged BEFORE any rank-local veto below, so
    // that every rank enters the collective inside share_pointers (a rank that skipped it would leave its peers in it).
    if (mr_comm_ != comm && !mr_share_tried_) {
      mr_share_tried_ = true;
      if (comm->nranks <= hipk::LU2_MAX_RANKS) {
        if (mr_recs_ == nullptr) {
          const size_t bytes = sizeof(unsigned long long) * hipk::lu2_mr_record_granules(1, 256);   // 256 records + mailboxes
          if (hipExtMallocWithFlags((void**)&mr_recs_, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_CHECK(hipMalloc((void**)&mr_recs_, bytes));
          }
          HIP_CHECK(hipMemsetAsync(mr_recs_, 0, bytes, st_));
          HIP_CHECK(hipStreamSynchronize(st_));
          mr_epoch_ = 0;
        }
        void* all[hipk::LU2_MAX_RANKS] = {nullptr};
        if (comm->share_pointers(mr_recs_, 0, all)) {
          for (int g = 0; g < comm->nranks; ++g) mr_peer_[g] = (unsigned long long*)all[g];
          mr_comm_ = comm;
        } else {
          mr_disabled_ = true;
          mr_why_disabled_ = "the ranks' record buffers could not be mapped into each other (share_pointers)";
        }
      }
      lus_ws(1, 1);                        // the pivot / partial-arg-max workspace exists before any leaf is launched
    }
    static const bool off = (getenv("GSI_LU_NO_MR") != nullptr);
    if (off) return no("GSI_LU_NO_MR is set");
    if (mr_disabled_) return no("switched off on this context: " + mr_why_disabled_);
    if (mr_comm_ != comm) return no("another communicator owns this context's record buffers");
    if (comm->nranks > hipk::LU2_MAX_RANKS) return no("more than " + std::to_string(hipk::LU2_MAX_RANKS) + " ranks");
    if (m >= ((int64_t)1 << 28) || l > hipk::LU2_MR_MAXL) return no("panel too tall (2^28 rows) or too wide (" + std::to_string(hipk::LU2_MR_MAXL) + " columns)");
    const int G = comm->nranks;
    const int64_t pad = (m + G - 1) / G;
    hipk::Lu2MrWork w{};
    // Ranks that share this device (possible only with the RCCL-free communicators): ALL their leaf grids must be resident
    // together, and residency is decided per SHADER ENGINE (8 CUs; 4 per XCD) -- the workgroups of a launch are dealt
    // round-robin over the 8 XCDs and, inside an XCD, over its 4 engines, so a grid of g one-per-CU workgroups puts
    // ceil(ceil(g / 8) / 4) on the first engines.  Measured with rank processes on one GPU: 2 x 123, 4 x 62, 3 x 40, 2 x 100
    // workgroups run; 3 x 74, 3 x 80 and 3 x 82 time out (3 x 3 = 9 workgroups on engines of 8 CUs) although 222-246 <= 256.
    const int sharing = std::max(comm->ranks_on_my_device(), 1);
    const int se_cus = std::max(ncus_ / 32, 1);                        // CUs per shader engine (8)
    const int per_se = sharing > 1 ? std::max(se_cus / sharing, 0) : se_cus;
    if (per_se < 1)
      return no(std::to_string(sharing) + " ranks share this device, more than the " + std::to_string(se_cus) +
                " CUs of a shader engine: their persistent leaf grids cannot all be resident");
    if (!hipk::lu2_mr_config(pad, G, sharing > 1 ? per_se * 32 : ncus_, &w.bs, &w.rpt, &w.grid, &w.hier, &w.ov, mr_force_))
      return no("no launch geometry for shards of " + std::to_string(pad) + " rows on " + std::to_string(G) + " ranks (" +
                std::to_string(sharing) + " sharing this device)");
    const int key = 1000000 + w.bs * 16 + w.rpt + (w.ov ? 100000 : 0);
    auto it = lu2_resident_.find(key);
    if (it == lu2_resident_.end())
      it = lu2_resident_.emplace(key, w.ov ? hipk::lu2_mr_resident_per_cu_ov() : hipk::lu2_mr_resident_per_cu(w.bs, w.rpt)).first;
    if (sharing > 1 ? (int64_t)sharing * ((((w.grid + 7) / 8) + 3) / 4) > (int64_t)se_cus * it->second
                    : (int64_t)it->second * ncus_ < (int64_t)w.grid)
      return no("the leaf grids (" + std::to_string(w.grid) + " workgroups per rank, " + std::to_string(sharing) +
                " ranks on this device) would not all be resident");
    w.rank = comm->rank; w.nranks = G;
    for (int g = 0; g < G; ++g) w.peer[g] = mr_peer_[g];
    w.info = flags_ + 0;
    if (const char* e = getenv("GSI_LU_POLL_LIMIT")) w.poll_limit = atoi(e);
    mr_work_ = w;
    mr_reason_.clear();
    return true;
  }
  int64_t lus_mr_signature() override {
    return ((((int64_t)mr_work_.bs * 64 + mr_work_.rpt) * 1024 + mr_work_.grid) * 2 + mr_work_.hier) * 2 + mr_work_.ov;
  }
  const char* lus_mr_reason() override { return mr_reason_.c_str(); }
  int64_t lus_mr_generation() override { return mr_gen_; }
  // Every kernel of the persistent-leaf factorization once, on private dummies (a 64-row panel, a record buffer of its
  // own, its own info word), with the geometry flags of the form that is about to run: what a first launch may cost the
  // runtime -- loading the code object, growing the queue's scratch (the overflow-row leaves spill) -- happens here,
  // before the ranks' barrier, and not while a peer's leaf spins for this rank's launch.  Once per instantiation.
  void lus_mr_warmup() override {
    bind();
    const int64_t key = (((int64_t)mr_work_.bs * 64 + mr_work_.rpt) * 2 + mr_work_.hier) * 2 + mr_work_.ov;
    if (mr_warm_.count(key)) return;
    mr_warm_.insert(key);
    const int64_t R = 256, Cc = 128;
    const size_t rec_doubles = hipk::lu2_mr_record_granules(1, 4);
    Scratch D(this, (size_t)R * Cc + 64), U(this, (size_t)64 * 128), recs(this, rec_doubles);
    hipk::randn_fill(st_, D.p, (size_t)R * Cc, 0x77a12eull);
    HIP_CHECK(hipMemsetAsync(D.p + R * Cc, 0, 64 * sizeof(double), st_));
    HIP_CHECK(hipMemsetAsync(recs.p, 0, rec_doubles * sizeof(double), st_));
    hipk::Lu2MrWork w = mr_work_;
    w.rank = 0; w.nranks = 1; w.grid = 1;
    w.peer[0] = (unsigned long long*)recs.p;
    w.info = (int32_t*)(D.p + R * Cc);
    w.ipiv = (int32_t*)(D.p + R * Cc + 16);
    hipk::lu2_leaf_mr(st_, w, D.p, R, 64, 0, 64, 8, 0, 0, 8, nullptr, 0);
    hipk::lus_swap_peer(st_, w, D.p, R, 64, 0, 64, 8, 0, 8, 0);
    hipk::lus_u12_block(st_, D.p, R, 0, 0, 64, 64, 128, U.p);
    hipk::lus_rankk(st_, D.p, R, R, 0, 0, 64, 64, 64, U.p);
    hipk::lus_u12_block(st_, D.p, R, 0, 0, 32, 32, 64, U.p);
    hipk::lus_rankk(st_, D.p, R, R, 0, 0, 32, 32, 32, U.p);
    hipk::lus_finish(st_, D.p, R, R, 0, 8);
    hipk::lu_flag_export(st_, w.info, U.p);
    hipk::lu_flag_import(st_, w.info, U.p);
    check_launch("lus_mr_warmup");
    HIP_CHECK(hipStreamSynchronize(st_));
  }
  void lu_flag_export(double* flag) override {
    bind();
    hipk::lu_flag_export(st_, flags_ + 0, flag);
  }
  void lu_flag_import(const double* flag) override {
    bind();
    hipk::lu_flag_import(st_, flags_ + 0, flag);
    check_launch("lu_flag_import");
  }
  void lus_leaf_mr(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t m, int64_t l, int64_t jb, int64_t j0, int w,
                   const double* U12) override {
    bind();
    mr_work_.ipiv = lus_ws(mloc, l).ipiv;
    hipk::lu2_leaf_mr(st_, mr_work_, Yloc, ld, mloc, row0, m, l, jb, j0, w, U12, mr_epoch_);
    // the rows this leaf's pivots exchange, in the columns outside the leaf: peer pushes + polls, no collective
    static const bool host_swaps = (getenv("GSI_LU_MR_HOST_SWAPS") != nullptr);      // A/B: pack + all-reduce + apply instead
    mr_peer_swaps_ = !host_swaps;
    if (mr_peer_swaps_) hipk::lus_swap_peer(st_, mr_work_, Yloc, ld, mloc, row0, m, l, j0, w, mr_epoch_);
    mr_epoch_ += (uint32_t)hipk::LU2_LEAF;
    mr_in_flight_ = true;
    check_launch("lu2_leaf_mr");
  }
  bool lus_mr_swaps_done() override { return mr_peer_swaps_; }
  int lus_mr_mode() override { return mr_work_.ov ? 2 : (mr_work_.hier ? 1 : 0); }
  void lus_mr_force(int mode) override { mr_force_ = mode; }
  void lus_swap_pack(const double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t l, int64_t j0, int w,
                     double* table) override {
    bind();
    hipk::lus_swap_pack(st_, Yloc, ld, mloc, row0, l, j0, w, lus_ws(mloc, l).ipiv, table);
    check_launch("lus_swap_pack");
  }
  void lus_swap_apply(double* Yloc, int64_t mloc, int64_t ld, int64_t row0, int64_t l, int64_t j0, int w,
                      const double* table) override {
    bind();
    if (mloc > 0) hipk::lus_swap_apply(st_, Yloc, ld, mloc, row0, l, j0, w, lus_ws(mloc, l).ipiv, table);
    check_launch("lus_swap_apply");
  }
  void lus_pivots(int32_t* host, int64_t l) override {
    bind();
    LusWs w = lus_ws(1, l);
    HIP_CHECK(hipMemcpyAsync(host, w.ipiv, sizeof(int32_t) * l, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
  }
  void qr_thinQ(double* Y, int64_t m, int64_t l, int64_t ld, double* R, bool replicated) override {
    bind();
    const int64_t nb = hipk::qr_max_blocks(m);
    const int NB = hipk::QR_NB;
    const int64_t npan = (l + NB - 1) / NB;
    size_t cnt = 0;
    auto take = [&](size_t c) { size_t o = cnt; cnt += (c + 7) & ~(size_t)7; return o; };
    const size_t o_coef = take(2 + NB + 2), o_part = take((size_t)nb * NB), o_tau = take(l),
                 o_T = take((size_t)npan * NB * NB), o_G = take(NB * NB),
                 o_Wt = take((size_t)l * NB), o_W2 = take((size_t)l * NB), o_small = take(hipk::cholqr_small_doubles(l)),
                 o_Q = take((size_t)(m + 1) * l);
    grow(ws_qr_, cnt * sizeof(double));
    TrimGuard trim_qr{this, &ws_qr_}, trim_hh{this, &ws_qr_hh_};
    double* base = (double*)ws_qr_.p;
    hipk::QrWork w;
    w.coef = base + o_coef; w.part = base + o_part; w.tau = base + o_tau; w.T = base + o_T; w.G = base + o_G;
    w.Vbuf = nullptr; w.Wt = base + o_Wt; w.W2 = base + o_W2; w.Qo = base + o_Q; w.maxblocks = nb;
    // gemm shapes inside: (t x NB, K = m), (NB x NB, K = m), (m x t, K = NB), (l x l, K = m), (m x 32, K <= l)
    size_t gmax = hipk::gemm_workspace_doubles(l, NB, m);
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(NB, NB, m));
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(l, l, m));
    gmax = std::max(gmax, hipk::gemm_syrk_workspace_doubles(l, m));     // CholeskyQR: Gram matrix, upper tiles only
    gmax = std::max(gmax, hipk::syrk_upper_workspace_doubles(l, m));    //             : the dedicated kernel's slab partials
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(m, l, l));       // CholeskyQR2: Y R^-1 as one product
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(l, 32, l));      //             : the l x l inverse
    for (int64_t t = NB; t <= l; t += NB) gmax = std::max(gmax, hipk::gemm_workspace_doubles(t, NB, m));
    double* ws = gemm_ws(gmax + 64);
    // Three tiers, each leaving Y untouched until it is known to have worked (one 4-byte flag read):
    //   CholeskyQR2 (a handful of MFMA passes; needs cond(Y) < ~1e7), shifted CholeskyQR3 (cond up to ~1e13: sketches
    //   of fast-decaying covariance spectra), Householder reflectors (anything, including exact rank deficiency).
    // A panel that needed the second tier makes the next few factorizations OF THE SAME HEIGHT AND KIND start there:
    // consecutive panels of one operator are alike, and a failed first attempt costs ~1.2 ms at C2.  Keyed by
    // (height, replicated) so that, with several ranks, the replicated factorizations (stacked R factors, gathered
    // panels: identical input everywhere) never see a hint left by a rank's own row block -- not even when a row
    // block happens to have G*l rows -- and stay bit-identical across ranks.
    static const bool no_cholqr = (getenv("GSI_NO_CHOLQR") != nullptr);
    if (!no_cholqr && l <= 1024 && m >= 2 * l) {
      const int64_t ldt = (m + 1) & ~(int64_t)1;   // even: 16-byte loads in the contraction kernel
      int32_t f = 0;
      int& skip_tier1_ = skip_tier1_by_height_[std::make_pair(m, replicated)];
      if (skip_tier1_ > 0) {
        --skip_tier1_;
      } else {
        // both rounds run out of place (Y -> Qo -> Y); Y is only overwritten once the flag is known to be clear
        HIP_CHECK(hipMemsetAsync(flags_ + 9, 0, sizeof(int32_t), st_));
        hipk::cholqr2_factor(st_, Y, m, l, ld, w.Qo, ldt, base + o_small, flags_ + 9, ws);
        check_launch("cholqr2_factor");
        HIP_CHECK(hipMemcpyAsync(&f, flags_ + 9, sizeof(int32_t), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
        if (f == 0) {
          hipk::cholqr2_apply(st_, Y, m, l, ld, w.Qo, ldt, R, base + o_small, ws);
          check_launch("cholqr2_apply");
          ++n_cholqr_;
          return;
        }
        skip_tier1_ = 8;
      }
      // second tier: shifted CholeskyQR3 with one more transient panel; Y still untouched
      static const bool no_scholqr3 = (getenv("GSI_NO_SCHOLQR3") != nullptr);
      double* S2 = nullptr;
      if (!no_scholqr3) {
        try { S2 = alloc((size_t)ldt * l); } catch (const Error&) { S2 = nullptr; }
      }
      if (S2 != nullptr) {
        struct Guard { HipBackend* be; double* p; ~Guard() { be->release(p); } } guard{this, S2};   // stream-ordered pool
        HIP_CHECK(hipMemsetAsync(flags_ + 9, 0, sizeof(int32_t), st_));
        hipk::scholqr3_factor(st_, Y, m, l, ld, w.Qo, ldt, S2, ldt, base + o_small, flags_ + 9, ws);
        check_launch("scholqr3_factor");
        HIP_CHECK(hipMemcpyAsync(&f, flags_ + 9, sizeof(int32_t), hipMemcpyDeviceToHost, st_));
        HIP_CHECK(hipStreamSynchronize(st_));
        if (f == 0) {
          hipk::scholqr3_apply(st_, Y, m, l, ld, S2, ldt, R, base + o_small, ws);
          check_launch("scholqr3_apply");
        }
        if (f == 0) { ++n_scholqr3_; return; }
      }
      skip_tier1_ = 0;   // neither Cholesky tier applies to this kind of panel: probe from the top next time
    }
    ++n_householder_;
    grow(ws_qr_hh_, (size_t)m * NB * sizeof(double));       // the reflector block: only this tier needs it
    w.Vbuf = (double*)ws_qr_hh_.p;
    hipk::qr_thinQ(st_, Y, m, l, ld, R, w, ws);
    check_launch("qr_thinQ");
  }
  // CholeskyQR2 stopped one tall product short (Backend::qr_thinQ_deferred): T = Y R1^-1 stays in the QR workspace, X2 = R2^-1
  // goes to the caller.  Only the first tier (the guard of qr_thinQ decides: a panel that needs the shifted tier or
  // Householder gets its Q the ordinary way), and only while the workspace is one this backend keeps between calls.
  bool qr_thinQ_deferred(const double* Y, int64_t m, int64_t l, int64_t ld, const double** Q1, int64_t* ldq1, double* X2) override {
    bind();
    static const bool off = (getenv("GSI_NO_CHOLQR") != nullptr) || (getenv("GSI_NO_QR_DEFER") != nullptr);
    if (off || l > 1024 || m < 2 * l) return false;
    const int64_t ldt = (m + 1) & ~(int64_t)1;
    if ((size_t)ldt * l * sizeof(double) > ((size_t)8 << 30)) return false;       // trimmed on return: nothing would be left to hand out
    int& skip_tier1 = skip_tier1_by_height_[std::make_pair(m, false)];
    if (skip_tier1 > 0) return false;
    size_t cnt = 0;
    auto take = [&](size_t c) { size_t o = cnt; cnt += (c + 7) & ~(size_t)7; return o; };
    const size_t o_small = take(hipk::cholqr_small_doubles(l)), o_T = take((size_t)ldt * l);
    grow(ws_qr_, cnt * sizeof(double));
    double* base = (double*)ws_qr_.p;
    size_t gmax = hipk::gemm_workspace_doubles(l, l, m);
    gmax = std::max(gmax, hipk::gemm_syrk_workspace_doubles(l, m));
    gmax = std::max(gmax, hipk::syrk_upper_workspace_doubles(l, m));
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(m, l, l));
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(l, 32, l));
    double* ws = gemm_ws(gmax + 64);
    int32_t f = 0;
    HIP_CHECK(hipMemsetAsync(flags_ + 9, 0, sizeof(int32_t), st_));
    hipk::cholqr2_factor(st_, Y, m, l, ld, base + o_T, ldt, base + o_small, flags_ + 9, ws);
    check_launch("cholqr2_factor");
    HIP_CHECK(hipMemcpyAsync(&f, flags_ + 9, sizeof(int32_t), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    if (f != 0) { skip_tier1 = 8; return false; }
    HIP_CHECK(hipMemcpyAsync(X2, hipk::cholqr2_X2(base + o_small, l), sizeof(double) * (size_t)l * l, hipMemcpyDeviceToDevice, st_));
    *Q1 = base + o_T;
    *ldq1 = ldt;
    ++n_cholqr_;
    return true;
  }
  // svd(B) for B = W' (RandMatFact.jl:86-88) without forming the thin Q of W: CholeskyQR2's first round leaves
  // T = W R1^-1 and R2, R2^-1; R = R2 R1 goes to the Jacobi SVD, and Z = T (R2^-1 (U sqrt(S))) is ONE tall product
  // (the generic path runs Y = T R2^-1 and Z = Y (U sqrt(S)): two).  Declines when CholeskyQR2 does not apply.
  bool svd_tall_fused(const double* W, int64_t m, int64_t l, int64_t ld, int64_t K_scale, double* V, int64_t ldv,
                      double* S, const double* Xr) override {
    bind();
    static const bool off = (getenv("GSI_NO_CHOLQR") != nullptr) || (getenv("GSI_NO_SVD_FUSION") != nullptr);
    if (off || l > 1024 || m < 2 * l) return false;
    int& skip_tier1 = skip_tier1_by_height_[std::make_pair(m, false)];
    if (skip_tier1 > 0) return false;                       // this kind of panel needs the shifted tier: generic path
    const int64_t ldt = (m + 1) & ~(int64_t)1;
    size_t cnt = 0;
    auto take = [&](size_t c) { size_t o = cnt; cnt += (c + 7) & ~(size_t)7; return o; };
    const size_t o_small = take(hipk::cholqr_small_doubles(l)), o_R = take((size_t)l * l), o_U = take((size_t)l * l),
                 o_M = take((size_t)l * l), o_T = take((size_t)ldt * l);
    grow(ws_svdf_, cnt * sizeof(double));
    TrimGuard trim_svdf{this, &ws_svdf_};
    double* base = (double*)ws_svdf_.p;
    size_t gmax = hipk::gemm_workspace_doubles(l, l, m);
    gmax = std::max(gmax, hipk::gemm_syrk_workspace_doubles(l, m));
    gmax = std::max(gmax, hipk::syrk_upper_workspace_doubles(l, m));
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(m, l, l));
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(l, 32, l));
    gmax = std::max(gmax, hipk::gemm_workspace_doubles(l, l, l));
    double* ws = gemm_ws(gmax + 64);
    int32_t f = 0;
    phase_begin(PH_QR);
    HIP_CHECK(hipMemsetAsync(flags_ + 9, 0, sizeof(int32_t), st_));
    hipk::cholqr2_factor(st_, W, m, l, ld, base + o_T, ldt, base + o_small, flags_ + 9, ws);
    check_launch("cholqr2_factor");
    HIP_CHECK(hipMemcpyAsync(&f, flags_ + 9, sizeof(int32_t), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    if (f != 0) {
      phase_end(PH_QR);
      skip_tier1 = 8;
      return false;                                         // W is untouched: the caller's generic path takes over
    }
    hipk::cholqr2_R(st_, l, base + o_small, base + o_R);    // R = R2 R1
    if (Xr != nullptr) {                                    // the SVD wanted is that of W Xr = Q (R Xr): one l x l product
      hipk::gemm_f64(st_, false, l, l, l, 1.0, base + o_R, l, Xr, l, 0.0, base + o_M, l, ws);
      HIP_CHECK(hipMemcpyAsync(base + o_R, base + o_M, sizeof(double) * (size_t)l * l, hipMemcpyDeviceToDevice, st_));
    }
    phase_end(PH_QR);
    ++n_cholqr_;
    phase_begin(PH_SVD);
    svd_small(base + o_R, l, base + o_U, S);
    if (K_scale >= 0) hipk::scale_cols_sqrt(st_, base + o_U, l, S, K_scale);
    phase_end(PH_SVD);
    phase_begin(PH_SMALL_GEMM);
    hipk::gemm_f64(st_, false, l, l, l, 1.0, hipk::cholqr2_X2(base + o_small, l), l, base + o_U, l, 0.0, base + o_M, l, ws);
    // Z = V sqrt(S) has its last p columns ZERO by definition (RandMatFact.jl:87): they are not multiplied out (a fifth of
    // the product at K = 256, p = 64), they are cleared
    const int64_t lz = (K_scale >= 0 && K_scale < l) ? K_scale : l;
    hipk::gemm_f64(st_, false, m, lz, l, 1.0, base + o_T, ldt, base + o_M, l, 0.0, V, ldv, ws);
    if (lz < l) HIP_CHECK(hipMemsetAsync(V + (size_t)lz * ldv, 0, sizeof(double) * ((size_t)(l - lz - 1) * ldv + m), st_));
    check_launch("svd_tall_fused");
    phase_end(PH_SMALL_GEMM);
    return true;
  }
  void svd_small(double* G, int64_t l, double* U, double* S) override {
    bind();
    if (l > 5000) throw Error(GSI_ERR_ARG, "sketch width l = K+p > 5000 is not supported by the LDS-resident block Jacobi SVD");
    grow(ws_svd_, sizeof(double) * (l + 8) + 64 + sizeof(int32_t) * hipk::SVD_SCHED_INTS);
    hipk::SvdWork w;
    w.norms = (double*)ws_svd_.p;
    w.pairs = (int32_t*)((char*)ws_svd_.p + sizeof(double) * (l + 8) + 64);
    w.rotcount = flags_ + 8;
    const int sw = hipk::svd_small(st_, G, l, U, S, w);
    last_svd_sweeps_ = sw < 0 ? -sw : sw;
    if (sw < 0) ++n_svd_cap_hits_;          // 40 sweeps and rotatable pairs left (factors graded over > 1e10): gsi_ctx_path_info reports it
    check_launch("svd_small");
  }
  void chol_upper(double* B, int64_t j) override {
    bind();
    if (j > 2048) throw Error(GSI_ERR_ARG, "chol_upper: j too large");
    hipk::chol_upper(st_, B, j, flags_ + 1);
    check_launch("chol_upper");
  }
  void trsm_right_upper(double* F, int64_t m, int64_t j, int64_t ldf, const double* C) override {
    bind();
    hipk::trsm_right_upper(st_, F, m, j, ldf, C);
    check_launch("trsm_right_upper");
  }
  void scale_cols_sqrt(double* U, int64_t l, const double* S, int64_t K) override {
    bind();
    hipk::scale_cols_sqrt(st_, U, l, S, K);
  }
  void center_rows(double* S, int64_t n, int64_t N, int64_t ld) override {
    bind();
    hipk::center_rows(st_, S, n, N, ld);
  }
  void randn(double* p, size_t count, uint64_t seed) override {
    bind();
    hipk::randn_fill(st_, p, count, seed);
  }
  void fill_gridcov(double* A, int64_t lda, int64_t nx, int64_t ny, double ell, int kind, int64_t row0,
                    int64_t mloc) override {
    bind();
    hipk::fill_gridcov(st_, A, lda, nx, ny, ell, kind, row0, mloc);
    check_launch("fill_gridcov");
  }
  void fill_lowrank_samples(double* S, int64_t ld, int64_t nloc, int64_t N, int64_t row0, uint64_t seed,
                            double decay) override {
    bind();
    hipk::fill_lowrank_samples(st_, S, ld, nloc, N, row0, seed, decay);
    check_launch("fill_lowrank_samples");
  }
  void colnorms(const double* Y, int64_t m, int64_t c, int64_t ld, double* host_out) override {
    bind();
    if (c > 64) throw Error(GSI_ERR_ARG, "colnorms: at most 64 columns at a time");
    hipk::colnorms_sq(st_, Y, m, c, ld, scal_);
    HIP_CHECK(hipMemcpyAsync(host_out, scal_, sizeof(double) * c, hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    for (int64_t i = 0; i < c; ++i) host_out[i] = std::sqrt(host_out[i]);
  }
  void axpy(int64_t n, double a, const double* x, double* y) override { bind(); hipk::axpy(st_, n, a, x, y); }
  double dot(int64_t n, const double* x, const double* y) override {
    bind();
    hipk::dot_dev(st_, n, x, y, scal_);
    double v = 0.0;
    HIP_CHECK(hipMemcpyAsync(&v, scal_, sizeof(double), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    return v;
  }
  double nrm2(int64_t n, const double* x) override { return std::sqrt(dot(n, x, x)); }
  void scal_copy(int64_t n, double a, const double* x, double* y) override {
    bind();
    hipk::scal_copy(st_, n, a, x, y);
  }
  void pcga_params(const double* Z, int64_t n, int64_t K, const double* s, const double* X, double delta,
                   double* out) override {
    bind();
    hipk::pcga_params(st_, Z, n, K, s, X, delta, out);
    check_launch("pcga_params");
  }
  void gemv_n(int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double beta,
              double* y) override {
    bind();
    hipk::gemv_n(st_, m, k, alpha, A, lda, x, beta, y);
    check_launch("gemv_n");
  }
  void gemv_t(int64_t m, int64_t k, double alpha, const double* A, int64_t lda, const double* x, double* y) override {
    bind();
    grow(ws_blas2_, sizeof(double) * hipk::gemv_t_workspace_doubles(k));
    hipk::gemv_t(st_, m, k, alpha, A, lda, x, y, (double*)ws_blas2_.p);
    check_launch("gemv_t");
  }
  void project_out(int64_t m, int64_t ncols, const double* q, double* Y, int64_t ld) override {
    bind();
    if (ncols > 64) throw Error(GSI_ERR_ARG, "project_out: at most 64 columns at a time");
    grow(ws_blas2_, sizeof(double) * hipk::gemv_t_workspace_doubles(ncols));
    hipk::project_out(st_, m, ncols, q, Y, ld, (double*)ws_blas2_.p);
    check_launch("project_out");
  }
  void scal(int64_t n, double a, double* x) override { bind(); hipk::scal(st_, n, a, x); }
  void diag_mul_add(int64_t n, const double* d, const double* x, double* y) override {
    bind();
    hipk::diag_mul_add(st_, n, d, x, y);
  }
  size_t lsqr_work_doubles() override { return hipk::lsqr_work_doubles(); }
  void lsqr_begin(int64_t n, const double* w, double* work) override {
    bind();
    hipk::lsqr_begin(st_, n, w, work);
    check_launch("lsqr_begin");
  }
  void lsqr_step_u(int64_t m, const double* t, double* u, double* work) override {
    bind();
    hipk::lsqr_step_u(st_, m, t, u, work);
    check_launch("lsqr_step_u");
  }
  void lsqr_step_v(int64_t n, const double* t, double* v, double* w, double* x, double* work) override {
    bind();
    hipk::lsqr_step_v(st_, n, t, v, w, x, work);
    check_launch("lsqr_step_v");
  }
  void f64_to_f32(const double* src, void* dst32, size_t count) override {
    bind();
    hipk::f64_to_f32(st_, src, (float*)dst32, count);
    check_launch("f64_to_f32");
  }
  void pcga_params_f32(const void* Z32, int64_t n, int64_t K, const double* s, const double* X, double delta,
                       double* out) override {
    bind();
    hipk::pcga_params_f32(st_, (const float*)Z32, n, K, s, X, delta, out);
    check_launch("pcga_params_f32");
  }
  void basis_gemv_f32(const void* Z32, int64_t n, int64_t K, const double* w, double beta, const double* X,
                      double* y) override {
    bind();
    if (K > 8000) throw Error(GSI_ERR_ARG, "basis_gemv_f32: K too large for the LDS-resident weights");
    hipk::basis_gemv_f32(st_, (const float*)Z32, n, K, w, beta, X, y);
    check_launch("basis_gemv_f32");
  }

  // The persistent leaf kernel spins on records of ALL its workgroups: the grid must fit the chip at this kernel's
  // occupancy (registers, LDS, waves).  Queried once per instantiation; a panel that does not fit takes the streamed leaves.
  bool lu2_fits(int bs, int rpt, int grid) {
    const int key = bs * 16 + rpt;
    auto it = lu2_resident_.find(key);
    if (it == lu2_resident_.end()) it = lu2_resident_.emplace(key, hipk::lu2_resident_per_cu(bs, rpt)).first;
    return (int64_t)it->second * ncus_ >= grid;
  }
  void forgive_lost_coresidency() override {
    bind();
    (void)hipMemsetAsync(flags_, 0, 8 * sizeof(int32_t), st_);
    lu2_lost_ = false;
    lu2_retry_ = false;          // (mr_disabled_ stays as the self-test left it)
  }
  bool retryable_failure() override {
    const bool r = lu2_retry_;
    lu2_retry_ = false;
    return r;
  }

  int take_error(std::string* msg) override {
    bind();
    int32_t h[16];
    HIP_CHECK(hipMemcpyAsync(h, flags_, sizeof(h), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    HIP_CHECK(hipGetLastError());
    if (!garbage_.empty()) collect_garbage();
    if (h[0] >= 0) mr_in_flight_ = false;
    if (h[0] != 0 || h[1] != 0) {
      HIP_CHECK(hipMemsetAsync(flags_, 0, 8 * sizeof(int32_t), st_));
      if (h[0] < 0) {
        // co-residency of the persistent leaf kernel was lost (a workgroup never got a CU: the GPU is shared with
        // another queue).  The panel of that factorization is destroyed; this context factors with
        // streamed leaves (no spin-waits between workgroups) from now on, and the entry point may be re-run on its inputs.
        lu2_lost_ = true;
        lu2_retry_ = true;
        ++n_lu_timeouts_;
        if (mr_in_flight_) {
          // a multi-rank factorization: the flag was made global on the stream (pipeline.cpp: lu_flag_export / _import), so
          // EVERY rank is here in the same call, switches the in-kernel exchange off and voids the ranks' agreements
          mr_disabled_ = true;
          mr_why_disabled_ = "a pivot exchange between the ranks' kernels timed out";
          ++mr_gen_;
        }
        mr_in_flight_ = false;
        if (msg) *msg = "lu(): the pivot exchange between workgroups timed out (GPU shared with another job?); "
                        "this context now streams its leaves (no spin-waits) [waited for: phase " + std::to_string(h[2]) +
                        ", slot " + std::to_string(h[3]) + ", epoch " + std::to_string(h[4]) + ", by rank*1024+workgroup " +
                        std::to_string(h[5]) + "]";
        return GSI_ERR_INTERNAL;
      }
      if (h[0] != 0) {
        if (msg) *msg = "SingularException(" + std::to_string(h[0]) + "): exactly zero pivot in lu()";
        return GSI_ERR_SINGULAR;
      }
      if (msg) *msg = "PosDefException: matrix is not positive definite; Cholesky failed at " + std::to_string(h[1]);
      return GSI_ERR_NOT_POSDEF;
    }
    return 0;
  }

  // ---- profiling ----
  void profile(bool on) override { prof_ = on; }
  void phase_begin(Phase p) override {
    if (!prof_) return;
    bind();
    if (roctx().push) roctx().push(kPhaseNames[p]);
    cur_.phase = p;
    cur_.a = get_event();
    cur_.b = get_event();
    hipEventRecord(cur_.a, st_);
  }
  void phase_end(Phase) override {
    if (!prof_) return;
    if (roctx().pop) roctx().pop();
    hipEventRecord(cur_.b, st_);
    records_.push_back(cur_);
  }
  void phase_reset() override {
    bind();
    hipStreamSynchronize(st_);
    for (auto& r : records_) { ev_pool_.push_back(r.a); ev_pool_.push_back(r.b); }
    records_.clear();
    for (int i = 0; i < PH_COUNT; ++i) { acc_ms_[i] = 0.0; acc_n_[i] = 0; }
  }
  void phase_times(double* ms, int64_t* counts) override {
    bind();
    hipStreamSynchronize(st_);
    for (auto& r : records_) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { acc_ms_[r.phase] += t; acc_n_[r.phase] += 1; }
      ev_pool_.push_back(r.a);
      ev_pool_.push_back(r.b);
    }
    records_.clear();
    for (int i = 0; i < PH_COUNT; ++i) { ms[i] = acc_ms_[i]; counts[i] = acc_n_[i]; }
  }

  void counters(int64_t* out4) override {
    out4[0] = n_cholqr_; out4[1] = n_householder_; out4[2] = last_svd_sweeps_; out4[3] = n_scholqr3_;
  }
  int64_t lu_timeouts() override { return n_lu_timeouts_; }
  int64_t svd_cap_hits() override { return n_svd_cap_hits_; }
  void set_ranks_sharing_device(int n) override { ranks_sharing_device_ = n; }
  int device() const { return device_; }

 private:
  struct Scratch {                 // a pooled temporary that goes back to the block cache on every exit path
    HipBackend* be; double* p;
    Scratch(HipBackend* b, size_t doubles) : be(b), p(doubles ? b->alloc(doubles) : nullptr) {}
    ~Scratch() { if (p) be->release(p); }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
  };
  struct Rec { Phase phase; hipEvent_t a, b; };
  hipEvent_t get_event() {
    if (!ev_pool_.empty()) { hipEvent_t e = ev_pool_.back(); ev_pool_.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
  void check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(GSI_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  }
  // Cached workspaces make repeated factorizations allocation-free; one of panel size at 512^3 (tens of GB) would
  // instead starve the next phase, so anything above 8 GiB is given back when the call that grew it returns.
  // A workspace that is outgrown is not freed on the spot: hipFree waits for the whole DEVICE, and with the ranks of a
  // communicator as threads of one process (GSI_LOCAL_COMM) another rank's persistent LU kernel may at that moment be
  // spinning for a record of the very kernel this thread is about to launch -- a dead-lock that ends in the kernel's poll
  // time-out (seen once in five runs of test_multirank_pipeline_on_one_gpu[3]: rank B grew ws_lu_ inside the sharded LU
  // while rank A's first leaf was already waiting for it).  The block waits in garbage_ until this context's call has
  // finished its own device work (take_error: whatever still spins for us has all it needs by then).
  void free_ws(DevBuf& b) {
    if (!b.p) return;
    if (b.pooled) release((double*)b.p);       // back to the block cache (stream-ordered: no synchronisation)
    else { std::lock_guard<std::mutex> g(mu_); garbage_.push_back(b.p); garbage_bytes_ += b.bytes; }
    b.p = nullptr; b.bytes = 0; b.pooled = false;
  }
  void collect_garbage() {                     // caller: this context's stream is idle
    std::vector<void*> g;
    { std::lock_guard<std::mutex> lk(mu_); g.swap(garbage_); garbage_bytes_ = 0; }
    for (void* p : g) (void)hipFree(p);
  }
  void trim(DevBuf& b) {
    if (b.bytes <= ((size_t)8 << 30) || !b.p) return;
    free_ws(b);
  }
  struct TrimGuard { HipBackend* be; DevBuf* b; ~TrimGuard() { be->trim(*b); } };
  void grow(DevBuf& b, size_t bytes) {
    if (bytes <= b.bytes) return;
    if (bytes > ((size_t)1 << 30)) {             // panel-sized: from the block cache, where a just-released panel usually waits
      free_ws(b);
      b.p = alloc((bytes + 7) / 8);
      b.bytes = bytes;
      b.pooled = true;
      return;
    }
    free_ws(b);
    bytes = (bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {   // give the cache of released panels back and retry once (the other phases' workspaces may be
      (void)hipGetLastError();   // in use by the caller: they stay)
      (void)hipStreamSynchronize(st_);
      collect_garbage();
      trim_pool(0);
      e = hipMalloc(&b.p, bytes);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      b.p = nullptr;
      throw Error(GSI_ERR_OOM, "workspace hipMalloc of " + std::to_string(bytes) + " bytes failed");
    }
    b.bytes = bytes;
  }
  double* gemm_ws(size_t doubles) {
    if (doubles == 0) return (double*)ws_gemm_.p;
    grow(ws_gemm_, doubles * sizeof(double));
    return (double*)ws_gemm_.p;
  }

  int device_;
  int ncus_ = 0;
  std::string arch_;
  hipStream_t st_ = nullptr;
  std::unique_ptr<HostStager> stager_;        // the pinned staging ring of the host boundary (created at the first large transfer)
  bool stager_failed_ = false;
  hipEvent_t ev_stage_ = nullptr;
  int32_t* flags_ = nullptr;  // [0] lu info, [1] chol info, [8] jacobi rotation counter
  double* scal_ = nullptr;
  DevBuf ws_gemm_, ws_lu_, ws_qr_, ws_svd_, ws_blas2_, ws_lus_, ws_svdf_, ws_qr_hh_;
  std::mutex mu_;
  std::vector<DevBuf> sizes_;
  std::vector<DevBuf> pool_;
  std::vector<void*> garbage_;                 // outgrown workspaces, freed when this context's call has finished (free_ws)
  size_t garbage_bytes_ = 0;
  int64_t in_use_ = 0, pooled_ = 0;
  bool prof_ = false;
  Rec cur_{};
  std::vector<Rec> records_;
  std::vector<hipEvent_t> ev_pool_;
  double acc_ms_[PH_COUNT] = {0};
  int64_t acc_n_[PH_COUNT] = {0};
  int last_svd_sweeps_ = 0;
  int64_t n_svd_cap_hits_ = 0;
  int64_t n_cholqr_ = 0, n_householder_ = 0, n_scholqr3_ = 0;
  std::map<std::pair<int64_t, bool>, int> skip_tier1_by_height_;
  std::map<int, int> lu2_resident_;   // (bs, rpt) -> resident workgroups per CU of that leaf instantiation
  bool lu2_lost_ = false, lu2_retry_ = false;
  int lu2_ov_resident_ = -1;
  int mr_force_ = 0;
  // multi-rank persistent leaves: this rank's record buffer, the peers' (as this rank addresses them), the running epoch
  unsigned long long* mr_recs_ = nullptr;
  unsigned long long* mr_peer_[hipk::LU2_MAX_RANKS] = {nullptr};
  Comm* mr_comm_ = nullptr;
  uint32_t mr_epoch_ = 0;
  bool mr_disabled_ = false, mr_peer_swaps_ = true;
  bool mr_share_tried_ = false, mr_in_flight_ = false;
  int ranks_sharing_device_ = 1;
  int64_t mr_gen_ = 0, n_lu_timeouts_ = 0;
  std::string mr_reason_, mr_why_disabled_;
  std::set<int64_t> mr_warm_;
  hipk::Lu2MrWork mr_work_{};
  int64_t lus_part_col_ = -1, lus_part_mloc_ = 0;     // sharded LU: column whose arg-max partials the last apply kernel left
  const double* lus_part_Y_ = nullptr;
};

// ---- RCCL, bound lazily so a single-GPU user never needs librccl to resolve -----------------
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};
RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    api.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!api.h) api.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!api.h) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.h, "ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
    api.ReduceScatter = (decltype(api.ReduceScatter))dlsym(api.h, "ncclReduceScatter");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
    api.Send = (decltype(api.Send))dlsym(api.h, "ncclSend");
    api.Recv = (decltype(api.Recv))dlsym(api.h, "ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))dlsym(api.h, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.h, "ncclGroupEnd");
  });
  if (!api.h || !api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.AllGather || !api.ReduceScatter)
    throw Error(GSI_ERR_RCCL, "librccl.so could not be loaded: multi-GPU needs RCCL");
  return api;
}
#define RCCL_CHECK(expr)                                                                       \
  do {                                                                                         \
    ncclResult_t _r = (expr);                                                                  \
    if (_r != ncclSuccess)                                                                     \
      throw Error(GSI_ERR_RCCL, std::string(#expr) + ": " +                                     \
                                    (rccl().GetErrorString ? rccl().GetErrorString(_r) : "rccl error")); \
  } while (0)

// all[g] = rank g's buffer as this process addresses it: its own pointer, an IPC mapping of everybody else's
static bool ipc_open_all(int nranks, int rank, void* mine, const hipIpcMemHandle_t* hs, void** all, std::vector<void*>* opened) {
  bool ok = true;
  for (int g = 0; g < nranks && ok; ++g) {
    if (g == rank) { all[g] = mine; continue; }
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, hs[g], hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); ok = false; p = nullptr; }
    all[g] = p;
    if (p && opened) opened->push_back(p);
  }
  return ok;
}

class RcclComm : public Comm {
 public:
  RcclComm(HipBackend* be, int n, int r, const void* id) : be_(be) {
    nranks = n;
    rank = r;
    static_assert(sizeof(ncclUniqueId) <= GSI_UNIQUE_ID_BYTES, "unique id does not fit the ABI slot");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    be_->bind();
    RCCL_CHECK(rccl().CommInitRank(&comm_, n, uid, r));
  }
  ~RcclComm() override {
    if (comm_) {
      be_->bind();
      hipStreamSynchronize(be_->stream());
      rccl().CommDestroy(comm_);
    }
  }
  void do_allreduce_sum(double* buf, size_t count) override {
    be_->bind();
    RCCL_CHECK(rccl().AllReduce(buf, buf, count, ncclDouble, ncclSum, comm_, be_->stream()));
  }
  void do_allgather(const double* send, double* recv, size_t count) override {
    be_->bind();
    RCCL_CHECK(rccl().AllGather(send, recv, count, ncclDouble, comm_, be_->stream()));
  }
  void do_reduce_scatter_sum(const double* send, double* recv, size_t count) override {
    be_->bind();
    RCCL_CHECK(rccl().ReduceScatter(send, recv, count, ncclDouble, ncclSum, comm_, be_->stream()));
  }
  // point-to-point xGMI: every pair of GPUs has a direct link, so the grouped sends / receives use all 7 links at once
  void do_alltoall(const double* send, double* recv, size_t count) override {
    be_->bind();
    RcclApi& r = rccl();
    if (!r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) throw Error(GSI_ERR_RCCL, "librccl has no ncclSend / ncclRecv");
    RCCL_CHECK(r.GroupStart());
    for (int g = 0; g < nranks; ++g) {
      RCCL_CHECK(r.Send(send + (size_t)g * count, count, ncclDouble, g, comm_, be_->stream()));
      RCCL_CHECK(r.Recv(recv + (size_t)g * count, count, ncclDouble, g, comm_, be_->stream()));
    }
    RCCL_CHECK(r.GroupEnd());
  }
  // one process per GPU: IPC handles travel through an all-gather, every rank maps its peers' buffers.  The mapping of
  // OTHER processes' buffers cannot be exercised on the one-GPU build box (RCCL refuses two ranks on one device), which is
  // why pipeline.cpp:lus_mr_selftest makes the path prove itself on the machine it runs on before it is used.
  bool share_pointers(void* mine, size_t bytes, void** all) override {
    (void)bytes;
    static const bool on = !(getenv("GSI_LU_PEER") != nullptr && getenv("GSI_LU_PEER")[0] == '0');   // GSI_LU_PEER=0: never
    if (!on) return false;
    be_->bind();
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    bool ok = (hipIpcGetMemHandle(&h, mine) == hipSuccess);     // a failure here must not skip the collective below
    if (!ok) (void)hipGetLastError();
    double* send = be_->alloc(8);
    double* recv = be_->alloc((size_t)8 * nranks);
    HIP_CHECK(hipMemcpyAsync(send, &h, 64, hipMemcpyHostToDevice, be_->stream()));
    allgather(send, recv, 8);
    std::vector<hipIpcMemHandle_t> hs((size_t)nranks);
    HIP_CHECK(hipMemcpyAsync(hs.data(), recv, (size_t)64 * nranks, hipMemcpyDeviceToHost, be_->stream()));
    HIP_CHECK(hipStreamSynchronize(be_->stream()));
    be_->release(send);
    be_->release(recv);
    if (ok) ok = ipc_open_all(nranks, rank, mine, hs.data(), all, nullptr);
    return ok;      // false on this rank alone is fine: the ranks all-reduce their answers before anyone relies on the buffers
  }

 private:
  HipBackend* be_;
  ncclComm_t comm_ = nullptr;
};

// ---- ranks as THREADS of one process (GSI_LOCAL_COMM=1): one context per thread, on different GPUs with peer access or --
//      what makes the whole multi-rank pipeline runnable on a one-GPU box -- on the SAME GPU.  RCCL refuses two ranks on one
//      device; this communicator needs nothing but HIP: every rank's device pointers are valid in every thread of the
//      process, so a collective is "synchronise my stream, meet at a host barrier, read the peers' buffers with kernels /
//      copies on my own stream, synchronise, meet again".  Rank-ordered sums (deterministic).  Not a performance path.
struct LocalGroup {
  int nranks = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  std::vector<const double*> src;
  std::vector<int> device;               // device of every rank's context
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t gen = generation;
    if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); return; }
    // a rank thread that died (an exception on its way out, a caller that returned) must not hang the others for ever
    static const int timeout_s = getenv("GSI_SHM_TIMEOUT_S") ? std::max(1, atoi(getenv("GSI_SHM_TIMEOUT_S"))) : 300;
    if (!cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return generation != gen; })) {
      --arrived;
      throw Error(GSI_ERR_RCCL, "local communicator: a rank did not reach the barrier (GSI_SHM_TIMEOUT_S)");
    }
  }
};
static std::mutex g_local_mu;
static std::map<std::string, std::shared_ptr<LocalGroup>> g_local_groups;

class LocalComm : public Comm {
 public:
  LocalComm(HipBackend* be, int n, int r, const void* id) : be_(be) {
    nranks = n;
    rank = r;
    const std::string key((const char*)id, 32);
    std::unique_lock<std::mutex> lk(g_local_mu);
    auto& grp = g_local_groups[key];
    if (!grp) { grp = std::make_shared<LocalGroup>(); grp->nranks = n; grp->src.assign((size_t)n, nullptr); grp->device.assign((size_t)n, -1); }
    if (grp->nranks != n) throw Error(GSI_ERR_ARG, "local communicator: ranks disagree on nranks");
    grp->device[(size_t)r] = be_->device();
    grp_ = grp;
    lk.unlock();
    grp_->barrier();                      // like ncclCommInitRank: returns when every rank has joined (and registered its device)
    for (int d = 0, cnt = 0; hipGetDeviceCount(&cnt) == hipSuccess && d < cnt; ++d)      // best effort: peers on other GPUs
      if (d != be_->device()) { (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }
  }
  // publish my buffer, wait until every rank has published and its producing work is complete
  void publish(const double* p) {
    be_->bind();
    HIP_CHECK(hipStreamSynchronize(be_->stream()));
    { std::lock_guard<std::mutex> g(grp_->mu); grp_->src[(size_t)rank] = p; }
    grp_->barrier();
  }
  void finish() {                       // my reads of the peers' buffers are done; nobody may reuse a buffer before all are
    HIP_CHECK(hipStreamSynchronize(be_->stream()));
    grp_->barrier();
  }
  void do_allreduce_sum(double* buf, size_t count) override {
    publish(buf);
    double* tmp = be_->alloc(count);
    hipStream_t st = be_->stream();
    HIP_CHECK(hipMemcpyAsync(tmp, grp_->src[0], count * sizeof(double), hipMemcpyDeviceToDevice, st));
    for (int g = 1; g < nranks; ++g) hipk::axpy(st, (int64_t)count, 1.0, grp_->src[(size_t)g], tmp);
    finish();                           // every rank has summed the ORIGINAL buffers
    HIP_CHECK(hipMemcpyAsync(buf, tmp, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    be_->release(tmp);
  }
  void do_allgather(const double* send, double* recv, size_t count) override {
    publish(send);
    for (int g = 0; g < nranks; ++g)
      HIP_CHECK(hipMemcpyAsync(recv + (size_t)g * count, grp_->src[(size_t)g], count * sizeof(double), hipMemcpyDeviceToDevice,
                               be_->stream()));
    finish();
  }
  void do_reduce_scatter_sum(const double* send, double* recv, size_t count) override {
    publish(send);
    hipStream_t st = be_->stream();
    HIP_CHECK(hipMemcpyAsync(recv, grp_->src[0] + (size_t)rank * count, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    for (int g = 1; g < nranks; ++g) hipk::axpy(st, (int64_t)count, 1.0, grp_->src[(size_t)g] + (size_t)rank * count, recv);
    finish();
  }
  void do_alltoall(const double* send, double* recv, size_t count) override {
    publish(send);
    for (int g = 0; g < nranks; ++g)
      HIP_CHECK(hipMemcpyAsync(recv + (size_t)g * count, grp_->src[(size_t)g] + (size_t)rank * count, count * sizeof(double),
                               hipMemcpyDeviceToDevice, be_->stream()));
    finish();
  }
  bool share_pointers(void* mine, size_t, void** all) override {       // one process: the pointers themselves
    publish((const double*)mine);
    for (int g = 0; g < nranks; ++g) all[g] = const_cast<double*>(grp_->src[(size_t)g]);
    finish();
    return true;
  }
  void host_barrier() override { grp_->barrier(); }
  int ranks_on_my_device() override {
    std::lock_guard<std::mutex> g(grp_->mu);
    int c = 0;
    for (int q = 0; q < nranks; ++q) c += (grp_->device[(size_t)q] == be_->device()) ? 1 : 0;
    return std::max(c, 1);
  }

 private:
  HipBackend* be_;
  std::shared_ptr<LocalGroup> grp_;
};
static bool local_comm_requested() { return getenv("GSI_LOCAL_COMM") != nullptr; }

// ---- ranks as PROCESSES of one node without RCCL (GSI_SHM_COMM=1): host barriers and IPC handles in a POSIX shared-memory
//      block, payloads through one staging buffer per rank that every peer maps with hipIpcOpenMemHandle.  Two uses: the
//      cross-PROCESS half of the multi-rank code (IPC mapping of the pivot-exchange buffer, persistent kernels of different
//      processes polling each other's memory) runs on a one-GPU box, where RCCL refuses two ranks on one device; and a node
//      without librccl still has a communicator.  Rank-ordered sums (deterministic, identical on every rank).  A collective
//      is: copy into my staging buffer, synchronise, barrier, read the peers' staging buffers, synchronise, barrier -- not
//      a performance path.
struct ShmBlock {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  uint32_t ok[16];
  char busid[16][32];                    // PCI bus id of every rank's device
  hipIpcMemHandle_t stage[16];
  hipIpcMemHandle_t shared[16];
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "process-shared atomics");

class ShmComm : public Comm {
 public:
  ShmComm(HipBackend* be, int n, int r, const void* id) : be_(be) {
    nranks = n;
    rank = r;
    if (n < 1 || n > 16) throw Error(GSI_ERR_ARG, "shared-memory communicator: 1..16 ranks");
    char name[96];
    std::memcpy(name, (const char*)id + 16, 95);
    name[95] = 0;
    if (std::memcmp(id, "gsi-shm-comm", 12) != 0 || name[0] != '/')
      throw Error(GSI_ERR_ARG, "shared-memory communicator: the id does not come from gsi_comm_unique_id() under GSI_SHM_COMM");
    if (const char* e = getenv("GSI_SHM_TIMEOUT_S")) timeout_s_ = std::max(1, atoi(e));
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);      // whoever comes first creates it: zero-filled = initial state
    if (fd < 0) throw Error(GSI_ERR_RCCL, std::string("shm_open(") + name + ") failed");
    if (ftruncate(fd, sizeof(ShmBlock)) != 0) { close(fd); throw Error(GSI_ERR_RCCL, "shared-memory communicator: ftruncate failed"); }
    void* m = mmap(nullptr, sizeof(ShmBlock), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) throw Error(GSI_ERR_RCCL, "shared-memory communicator: mmap failed");
    blk_ = (ShmBlock*)m;
    be_->bind();
    size_t mb = 64;
    if (const char* e = getenv("GSI_SHM_STAGE_MB")) mb = (size_t)std::max(1, atoi(e));
    cap_ = mb * 1024 * 1024 / sizeof(double);
    bool ok = (hipMalloc((void**)&stage_, cap_ * sizeof(double)) == hipSuccess);
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    if (ok) ok = (hipIpcGetMemHandle(&h, stage_) == hipSuccess);
    if (!ok) (void)hipGetLastError();
    blk_->stage[rank] = h;
    blk_->ok[rank] = ok ? 1u : 0u;
    std::memset(blk_->busid[rank], 0, sizeof(blk_->busid[rank]));
    if (hipDeviceGetPCIBusId(blk_->busid[rank], (int)sizeof(blk_->busid[rank]) - 1, be_->device()) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(blk_->busid[rank], sizeof(blk_->busid[rank]), "device-%d", be_->device());
    }
    barrier();
    if (rank == 0) shm_unlink(name);                            // every rank has it mapped: nothing is left behind in /dev/shm
    for (int g = 0; g < nranks; ++g) ok = ok && blk_->ok[g] != 0;
    std::vector<hipIpcMemHandle_t> hs(blk_->stage, blk_->stage + nranks);
    void* all[16] = {nullptr};
    if (ok) ok = ipc_open_all(nranks, rank, stage_, hs.data(), all, &opened_);
    blk_->ok[rank] = ok ? 1u : 0u;
    barrier();
    for (int g = 0; g < nranks; ++g) ok = ok && blk_->ok[g] != 0;
    barrier();                                                  // ok[] is reused by share_pointers
    if (!ok) { cleanup(); throw Error(GSI_ERR_RCCL, "shared-memory communicator: the ranks' staging buffers could not be mapped (hipIpc)"); }
    for (int g = 0; g < nranks; ++g) peer_[g] = (const double*)all[g];
    for (int g = 0; g < nranks; ++g) same_device_ += (std::strncmp(blk_->busid[g], blk_->busid[rank], sizeof(blk_->busid[g])) == 0) ? 1 : 0;
  }
  int ranks_on_my_device() override { return std::max(same_device_, 1); }
  ~ShmComm() override {
    be_->bind();
    (void)hipStreamSynchronize(be_->stream());
    cleanup();
  }
  void do_allreduce_sum(double* buf, size_t count) override {
    hipStream_t st = be_->stream();
    for (size_t off = 0; off < count || off == 0; off += cap_) {
      const size_t c = std::min(cap_, count - off);
      stage_in(buf + off, c);
      if (c) HIP_CHECK(hipMemcpyAsync(buf + off, peer_[0], c * sizeof(double), hipMemcpyDeviceToDevice, st));
      for (int g = 1; g < nranks && c; ++g) hipk::axpy(st, (int64_t)c, 1.0, peer_[g], buf + off);
      done_reading();
      if (count == 0) break;
    }
  }
  void do_allgather(const double* send, double* recv, size_t count) override {
    for (size_t off = 0; off < count || off == 0; off += cap_) {
      const size_t c = std::min(cap_, count - off);
      stage_in(send + off, c);
      for (int g = 0; g < nranks && c; ++g)
        HIP_CHECK(hipMemcpyAsync(recv + (size_t)g * count + off, peer_[g], c * sizeof(double), hipMemcpyDeviceToDevice, be_->stream()));
      done_reading();
      if (count == 0) break;
    }
  }
  // blocks of `count` doubles per destination: the staging buffer holds nranks segments of one chunk
  void do_reduce_scatter_sum(const double* send, double* recv, size_t count) override {
    hipStream_t st = be_->stream();
    const size_t cc = std::max<size_t>(cap_ / (size_t)nranks, 1);
    for (size_t off = 0; off < count || off == 0; off += cc) {
      const size_t c = std::min(cc, count - off);
      stage_blocks(send, count, off, c, cc);
      if (c) HIP_CHECK(hipMemcpyAsync(recv + off, peer_[0] + (size_t)rank * cc, c * sizeof(double), hipMemcpyDeviceToDevice, st));
      for (int g = 1; g < nranks && c; ++g) hipk::axpy(st, (int64_t)c, 1.0, peer_[g] + (size_t)rank * cc, recv + off);
      done_reading();
      if (count == 0) break;
    }
  }
  void do_alltoall(const double* send, double* recv, size_t count) override {
    const size_t cc = std::max<size_t>(cap_ / (size_t)nranks, 1);
    for (size_t off = 0; off < count || off == 0; off += cc) {
      const size_t c = std::min(cc, count - off);
      stage_blocks(send, count, off, c, cc);
      for (int g = 0; g < nranks && c; ++g)
        HIP_CHECK(hipMemcpyAsync(recv + (size_t)g * count + off, peer_[g] + (size_t)rank * cc, c * sizeof(double), hipMemcpyDeviceToDevice,
                                 be_->stream()));
      done_reading();
      if (count == 0) break;
    }
  }
  // the buffers are mapped exactly as RcclComm maps them (the handles travel through the shared block instead of an all-gather)
  bool share_pointers(void* mine, size_t, void** all) override {
    static const bool on = !(getenv("GSI_LU_PEER") != nullptr && getenv("GSI_LU_PEER")[0] == '0');
    if (!on) return false;
    be_->bind();
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof(h));
    bool ok = (hipIpcGetMemHandle(&h, mine) == hipSuccess);
    if (!ok) (void)hipGetLastError();
    blk_->shared[rank] = h;
    blk_->ok[rank] = ok ? 1u : 0u;
    barrier();
    for (int g = 0; g < nranks; ++g) ok = ok && blk_->ok[g] != 0;
    std::vector<hipIpcMemHandle_t> hs(blk_->shared, blk_->shared + nranks);
    barrier();                                                  // everybody has read the slots
    if (ok) ok = ipc_open_all(nranks, rank, mine, hs.data(), all, &opened_);
    return ok;      // as with RCCL: the caller all-reduces the ranks' answers before anyone relies on the buffers
  }

 private:
  void barrier() {
    const uint32_t gen = blk_->generation.load(std::memory_order_acquire);
    if (blk_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) {
      blk_->arrived.store(0, std::memory_order_relaxed);
      blk_->generation.fetch_add(1, std::memory_order_release);
      return;
    }
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t it = 0; blk_->generation.load(std::memory_order_acquire) == gen; ++it) {
      if (it < 4096) continue;
      sched_yield();
      if ((it & 1023) == 0) {
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (t1.tv_sec - t0.tv_sec > timeout_s_)
          throw Error(GSI_ERR_RCCL, "shared-memory communicator: a rank did not reach the barrier (GSI_SHM_TIMEOUT_S)");
      }
    }
  }
  void stage_in(const double* src, size_t c) {                 // my chunk -> my staging buffer, visible to every rank
    be_->bind();
    if (c) HIP_CHECK(hipMemcpyAsync(stage_, src, c * sizeof(double), hipMemcpyDeviceToDevice, be_->stream()));
    HIP_CHECK(hipStreamSynchronize(be_->stream()));
    barrier();
  }
  void stage_blocks(const double* send, size_t count, size_t off, size_t c, size_t cc) {
    be_->bind();
    for (int g = 0; g < nranks && c; ++g)
      HIP_CHECK(hipMemcpyAsync(stage_ + (size_t)g * cc, send + (size_t)g * count + off, c * sizeof(double), hipMemcpyDeviceToDevice,
                               be_->stream()));
    HIP_CHECK(hipStreamSynchronize(be_->stream()));
    barrier();
  }
  void done_reading() {                                        // nobody overwrites a staging buffer a peer still reads
    HIP_CHECK(hipStreamSynchronize(be_->stream()));
    barrier();
  }
  void cleanup() {
    for (void* p : opened_) (void)hipIpcCloseMemHandle(p);
    opened_.clear();
    (void)hipGetLastError();
    if (blk_) {
      try { if (!std::uncaught_exceptions()) { const int t = timeout_s_; timeout_s_ = std::min(t, 10); barrier(); timeout_s_ = t; } } catch (...) {}
      munmap(blk_, sizeof(ShmBlock));
      blk_ = nullptr;
    }
    if (stage_) { (void)hipFree(stage_); stage_ = nullptr; }
  }

  HipBackend* be_;
  ShmBlock* blk_ = nullptr;
  double* stage_ = nullptr;
  size_t cap_ = 0;
  const double* peer_[16] = {nullptr};
  std::vector<void*> opened_;
  int same_device_ = 0;
  int timeout_s_ = 300;
};
static bool shm_comm_requested() { return getenv("GSI_SHM_COMM") != nullptr; }

}  // namespace

Backend* make_backend(int device_id) { return new HipBackend(device_id); }
Comm* make_comm(Backend* be, int nranks, int rank, const void* unique_id) {
  if (local_comm_requested()) return new LocalComm(static_cast<HipBackend*>(be), nranks, rank, unique_id);
  if (shm_comm_requested()) return new ShmComm(static_cast<HipBackend*>(be), nranks, rank, unique_id);
  return new RcclComm(static_cast<HipBackend*>(be), nranks, rank, unique_id);
}
void comm_unique_id(void* id_out) {
  std::memset(id_out, 0, GSI_UNIQUE_ID_BYTES);
  if (local_comm_requested()) {         // ranks are threads of this process: any id that is unique within it
    static std::atomic<uint64_t> counter{1};
    const uint64_t c = counter.fetch_add(1);
    std::memcpy(id_out, "gsi-local-comm", 14);
    std::memcpy((char*)id_out + 16, &c, sizeof(c));
    return;
  }
  if (shm_comm_requested()) {           // ranks are processes of this node: the name of a shared-memory block nobody has used
    static std::atomic<uint64_t> counter{1};
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    std::memcpy(id_out, "gsi-shm-comm", 12);
    snprintf((char*)id_out + 16, 96, "/gsi-shm-%ld-%llu-%lld%09ld", (long)getpid(), (unsigned long long)counter.fetch_add(1),
             (long long)t.tv_sec, (long)t.tv_nsec);
    return;
  }
  ncclUniqueId uid;
  RCCL_CHECK(rccl().GetUniqueId(&uid));
  std::memcpy(id_out, &uid, sizeof(uid));
}
const char* backend_name() { return "hip-gfx950"; }

}  // namespace gsi
