// host_staging.hpp -- the host boundary of the C ABI: pageable caller memory (a Julia Matrix{Float64}, a numpy array)
// <-> HBM at the PCIe rate.  What the reference's callers hand over is always host memory (getxis(Q::Matrix, ...),
// GeostatInversion.jl:63-70; LowRankCovMatrix(samples), lowrank.jl:14-30), so for the stored operators this, not any
// kernel, is the time a user sees first.
//
// Measured on the MI355X box (tools/h2d_rates.hip -> profiles/r05_h2d_rates.log, PCIe Gen5 x16): pinned memory moves at
// 57.6 GB/s to the device and 56.9 GB/s back; a first hipMemcpy from freshly written pageable memory reaches 32 GB/s
// (the runtime pins the caller's pages on the fly); hipHostRegister of the caller's buffer costs 28 GB/s before the copy
// starts.  T host threads that copy pageable -> pinned chunks and queue each chunk's DMA on a stream of their own reach
// 56.4 GB/s (T = 4, 16 MiB chunks: 0.98 of the pinned rate) in both directions, with the first byte on the device after
// one chunk instead of after the whole registration -- which is also what lets the first pass over a dense operator
// start while most of it is still on the host (upload in ROW BLOCKS, one event per block: Backend::upload2d_begin).
//
// One transfer at a time per stager (one per context; the ABI allows one thread per context).  Worker t owns stream t
// and two pinned buffers, takes the rectangles t, t + T, ... of the plan in order, and records one event per row block on
// its stream when it has queued its last rectangle of that block.
#pragma once
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace gsi {

struct StageRect {
  int64_t r0, rows, c0, cols;   // rows [r0, r0 + rows) x columns [c0, c0 + cols) of the matrix
  int32_t block;                // the row block this rectangle belongs to (non-decreasing along the plan)
};

// rectangles of at most `chunk` bytes, column-major order: whole columns while a column fits, else row segments
inline std::vector<StageRect> stage_plan_whole(int64_t rows, int64_t cols, size_t chunk) {
  std::vector<StageRect> v;
  const int64_t cap = (int64_t)(chunk / sizeof(double));
  if (rows <= cap) {
    const int64_t cc = std::max<int64_t>(1, cap / rows);
    for (int64_t c0 = 0; c0 < cols; c0 += cc) v.push_back({0, rows, c0, std::min(cc, cols - c0), 0});
  } else {
    for (int64_t c = 0; c < cols; ++c)
      for (int64_t r0 = 0; r0 < rows; r0 += cap) v.push_back({r0, std::min(cap, rows - r0), c, 1, 0});
  }
  return v;
}
// row blocks of `mb` rows, block after block; inside a block column ranges of at most `chunk` bytes
inline std::vector<StageRect> stage_plan_rowblocks(int64_t rows, int64_t cols, int64_t mb, size_t chunk, int* nblocks) {
  std::vector<StageRect> v;
  const int64_t cap = (int64_t)(chunk / sizeof(double));
  int b = 0;
  for (int64_t r0 = 0; r0 < rows; r0 += mb, ++b) {
    const int64_t rr = std::min(mb, rows - r0);
    const int64_t cc = std::max<int64_t>(1, cap / rr);
    for (int64_t c0 = 0; c0 < cols; c0 += cc) v.push_back({r0, rr, c0, std::min(cc, cols - c0), b});
  }
  *nblocks = b;
  return v;
}

// The CPUs next to the GPU (its PCIe root's NUMA node), from sysfs: the staging workers run there and their pinned buffers are
// allocated and first written there.  On the two-socket box this was measured on, a process on the GPU's node moves 54 GB/s in
// both directions, one on the other node 51 up and 42 down (profiles/r05_boundary_numa.log); wherever the CALLER's thread and
// arrays are, the workers' side of the copy is then always the short one.  Best effort: no sysfs entry, no binding.
inline bool gpu_local_cpus(int device, cpu_set_t* set) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return false;
  for (char* p = bdf; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char buf[1024] = {0};
  const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  if (got == 0) return false;
  CPU_ZERO(set);
  int n = 0;
  for (char* p = buf; *p && *p != '\n';) {                   // "64-127,192-255"
    char* e = nullptr;
    const long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, set); ++n; }
    if (*p == ',') ++p;
  }
  return n > 0;
}

class HostStager {
 public:
  HostStager(int device, int threads, size_t chunk_bytes) : device_(device), T_(threads), chunk_(chunk_bytes) {
    pinned_.assign((size_t)2 * T_, nullptr);
    slot_ev_.assign((size_t)2 * T_, nullptr);
    st_.assign((size_t)T_, nullptr);
    static const bool bind_off = (getenv("GSI_STAGE_BIND") != nullptr && getenv("GSI_STAGE_BIND")[0] == '0');
    bind_ = !bind_off && gpu_local_cpus(device_, &cpus_);
    try {
      for (int i = 0; i < 2 * T_; ++i) ck(hipEventCreateWithFlags(&slot_ev_[(size_t)i], hipEventDisableTiming), "hipEventCreate");
      for (int t = 0; t < T_; ++t) ck(hipStreamCreateWithFlags(&st_[(size_t)t], hipStreamNonBlocking), "hipStreamCreate");
    } catch (...) {
      free_all();
      throw;
    }
    blocks_done_.reset(new std::atomic<int>[(size_t)T_]);
    for (int t = 0; t < T_; ++t) blocks_done_[(size_t)t].store(0);
    ready_ = 0;
    for (int t = 0; t < T_; ++t) workers_.emplace_back([this, t] { worker(t); });
    {                                                       // every worker has bound itself and allocated its two buffers
      std::unique_lock<std::mutex> g(mu_);
      done_cv_.wait(g, [this] { return ready_ == T_; });
    }
    if (!init_error_.empty()) {
      { std::lock_guard<std::mutex> g(mu_); quit_ = true; }
      cv_.notify_all();
      for (auto& w : workers_) w.join();
      workers_.clear();
      free_all();
      throw std::runtime_error(init_error_);
    }
  }
  ~HostStager() {
    {
      std::lock_guard<std::mutex> g(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
    free_all();
  }
  HostStager(const HostStager&) = delete;
  HostStager& operator=(const HostStager&) = delete;
  size_t chunk_bytes() const { return chunk_; }
  int threads() const { return T_; }
  bool busy() const { return active_; }

  // Start a transfer.  `start` has been recorded on the context's stream: no copy touches device memory before it (the
  // destination may be a pooled block an earlier kernel still reads; the source of a download is still being produced).
  void begin(bool h2d, double* dev, int64_t ldd, double* host, int64_t ldh, std::vector<StageRect>&& rects, int nblocks,
             hipEvent_t start) {
    if (active_) throw std::runtime_error("host staging: a transfer is already in flight on this context");
    h2d_ = h2d; dev_ = dev; ldd_ = ldd; host_ = host; ldh_ = ldh; rects_ = std::move(rects); nblocks_ = nblocks; start_ = start;
    bool strided = false;                                    // any rectangle that is not contiguous on the device?
    for (const StageRect& r : rects_) if (r.rows != ldd_ && r.cols != 1) { strided = true; break; }
    if (strided && devstage_.empty()) {
      devstage_.assign((size_t)2 * T_, nullptr);
      for (auto& p : devstage_) {
        hipError_t e = hipMalloc((void**)&p, chunk_);
        if (e != hipSuccess) {
          for (auto& q : devstage_) if (q) hipFree(q);
          devstage_.clear();
          rects_.clear();
          ck(e, "hipMalloc (device staging buffer)");
        }
      }
    }
    while ((int)blk_ev_.size() < T_ * nblocks_) {
      hipEvent_t e;
      ck(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
      blk_ev_.push_back(e);
    }
    for (int t = 0; t < T_; ++t) blocks_done_[(size_t)t].store(0, std::memory_order_relaxed);
    failed_.store(false, std::memory_order_relaxed);
    error_.clear();
    trace_ = getenv("GSI_STAGE_TRACE") != nullptr;
    {
      std::lock_guard<std::mutex> g(mu_);
      remaining_ = T_;
      ++gen_;
    }
    active_ = true;
    cv_.notify_all();
  }
  // Make stream `st` wait for every copy of row block b (host-blocking until the workers have queued them).
  void wait_block(int b, hipStream_t st) {
    for (int t = 0; t < T_; ++t) {
      int spins = 0;
      while (blocks_done_[(size_t)t].load(std::memory_order_acquire) <= b) {
        if (failed_.load(std::memory_order_acquire)) { end(); return; }     // end() rethrows the worker's error
        if (++spins > 64) std::this_thread::yield();
      }
      if (failed_.load(std::memory_order_acquire)) { end(); return; }
      ck(hipStreamWaitEvent(st, blk_ev_[(size_t)t * nblocks_ + b], 0), "hipStreamWaitEvent");
    }
  }
  // All copies complete (both directions: the host buffer is final / no longer read).  Rethrows a worker's error.
  void end() {
    if (!active_) return;
    {
      std::unique_lock<std::mutex> g(mu_);
      done_cv_.wait(g, [this] { return remaining_ == 0; });
    }
    active_ = false;
    rects_.clear();
    if (failed_.load(std::memory_order_acquire)) throw std::runtime_error("host staging: " + error_);
  }

 private:
  static void ck(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
  }
  void free_all() {
    for (auto& s : st_) if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    for (auto& e : slot_ev_) if (e) hipEventDestroy(e);
    for (auto& e : blk_ev_) if (e) hipEventDestroy(e);
    for (auto& p : pinned_) if (p) hipHostFree(p);
    for (auto& p : devstage_) if (p) hipFree(p);
  }
  void worker(int t) {
    // bound to the GPU's NUMA node first, then the pinned buffers: allocated and first written from there
    if (bind_) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpus_), &cpus_);
    {
      std::string err;
      if (hipSetDevice(device_) != hipSuccess) err = "hipSetDevice failed in a staging worker";
      for (int i = 0; i < 2 && err.empty(); ++i) {
        hipError_t e = hipHostMalloc((void**)&pinned_[(size_t)(2 * t + i)], chunk_, hipHostMallocDefault);
        if (e != hipSuccess) { err = std::string("hipHostMalloc (staging buffer): ") + hipGetErrorString(e); (void)hipGetLastError(); }
        else memset(pinned_[(size_t)(2 * t + i)], 0, chunk_);
      }
      std::lock_guard<std::mutex> g(mu_);
      if (!err.empty() && init_error_.empty()) init_error_ = err;
      if (++ready_ == T_) done_cv_.notify_all();
    }
    int64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_;
      }
      try {
        run(t);
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(mu_);
        if (!failed_.load()) error_ = e.what();
        failed_.store(true, std::memory_order_release);
      }
      // whatever happened, nobody may wait for this worker's blocks for ever
      blocks_done_[(size_t)t].store(nblocks_ + 1, std::memory_order_release);
      {
        std::lock_guard<std::mutex> g(mu_);
        if (--remaining_ == 0) done_cv_.notify_all();
      }
    }
  }
  void pack(const StageRect& r, double* pb) const {                    // host -> pinned, dense (leading dimension r.rows)
    const double* src = host_ + r.r0 + r.c0 * ldh_;
    if (r.rows == ldh_) { memcpy(pb, src, sizeof(double) * (size_t)r.rows * r.cols); return; }
    for (int64_t c = 0; c < r.cols; ++c) memcpy(pb + c * r.rows, src + c * ldh_, sizeof(double) * (size_t)r.rows);
  }
  void unpack(const StageRect& r, const double* pb) const {            // pinned -> host
    double* dst = host_ + r.r0 + r.c0 * ldh_;
    if (r.rows == ldh_) { memcpy(dst, pb, sizeof(double) * (size_t)r.rows * r.cols); return; }
    for (int64_t c = 0; c < r.cols; ++c) memcpy(dst + c * ldh_, pb + c * r.rows, sizeof(double) * (size_t)r.rows);
  }
  // The PCIe leg is always ONE contiguous copy: a pitched host<->device copy (hipMemcpy2DAsync from pinned memory into a
  // padded / row-block destination) was measured at 42.5 GB/s against 56 for the plain one (profiles/r05_boundary_probe.log).
  // A rectangle that is strided on the device goes through a device-side staging buffer and a device-to-device 2-D copy
  // on the same stream (TB/s: invisible next to the link).
  void dma(const StageRect& r, double* pb, double* ds, hipStream_t s) const {
    double* d = dev_ + r.r0 + r.c0 * ldd_;
    const size_t w = sizeof(double) * (size_t)r.rows;
    const bool flat = (r.rows == ldd_ || r.cols == 1);
    if (h2d_) {
      ck(hipMemcpyAsync(flat ? d : ds, pb, w * r.cols, hipMemcpyHostToDevice, s), "hipMemcpyAsync (staged upload)");
      if (!flat) ck(hipMemcpy2DAsync(d, sizeof(double) * ldd_, ds, w, w, r.cols, hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync (device scatter)");
    } else {
      if (!flat) ck(hipMemcpy2DAsync(ds, w, d, sizeof(double) * ldd_, w, r.cols, hipMemcpyDeviceToDevice, s), "hipMemcpy2DAsync (device gather)");
      ck(hipMemcpyAsync(pb, flat ? d : ds, w * r.cols, hipMemcpyDeviceToHost, s), "hipMemcpyAsync (staged download)");
    }
  }
  static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  void run(int t) {
    const bool trace = trace_;
    double t_copy = 0.0, t_wait = 0.0, t_enq = 0.0, t_start = trace ? now_s() : 0.0, t0 = 0.0;
    size_t nrect = 0;
    ck(hipSetDevice(device_), "hipSetDevice");
    hipStream_t s = st_[(size_t)t];
    ck(hipStreamWaitEvent(s, start_, 0), "hipStreamWaitEvent");
    bool used[2] = {false, false};
    size_t pend[2] = {0, 0};
    int slot = 0, blk = 0;
    auto close_blocks_before = [&](int b) {                            // this worker has queued everything of blocks < b
      for (; blk < b; ++blk) {
        ck(hipEventRecord(blk_ev_[(size_t)t * nblocks_ + blk], s), "hipEventRecord");
        blocks_done_[(size_t)t].store(blk + 1, std::memory_order_release);
      }
    };
    const size_t n = rects_.size();
    for (size_t k = (size_t)t; k < n; k += (size_t)T_) {
      if (failed_.load(std::memory_order_acquire)) break;               // another worker failed: stop queueing
      const StageRect& r = rects_[k];
      close_blocks_before(r.block);
      double* pb = pinned_[(size_t)(2 * t + slot)];
      hipEvent_t e = slot_ev_[(size_t)(2 * t + slot)];
      if (used[slot]) {
        if (trace) t0 = now_s();
        ck(hipEventSynchronize(e), "hipEventSynchronize");
        if (trace) { t_wait += now_s() - t0; t0 = now_s(); }
        if (!h2d_) unpack(rects_[pend[slot]], pb);
        if (trace && !h2d_) t_copy += now_s() - t0;
      }
      if (trace) t0 = now_s();
      if (h2d_) pack(r, pb);
      if (trace) { if (h2d_) t_copy += now_s() - t0; t0 = now_s(); }
      dma(r, pb, devstage_.empty() ? nullptr : devstage_[(size_t)(2 * t + slot)], s);
      ck(hipEventRecord(e, s), "hipEventRecord");
      if (trace) { t_enq += now_s() - t0; ++nrect; }
      used[slot] = true;
      pend[slot] = k;
      slot ^= 1;
    }
    close_blocks_before(nblocks_);
    for (int i = 0; i < 2; ++i, slot ^= 1)                                 // oldest first
      if (used[slot]) {
        ck(hipEventSynchronize(slot_ev_[(size_t)(2 * t + slot)]), "hipEventSynchronize");
        if (!h2d_) unpack(rects_[pend[slot]], pinned_[(size_t)(2 * t + slot)]);
      }
    if (trace)
      fprintf(stderr, "[gsi staging] worker %d %s: %zu rectangles in %.4f s: host copy %.4f s, waiting for DMA %.4f s, queueing %.4f s\n", t,
              h2d_ ? "H2D" : "D2H", nrect, now_s() - t_start, t_copy, t_wait, t_enq);
  }

  int device_, T_;
  size_t chunk_;
  bool bind_ = false;
  cpu_set_t cpus_;
  int ready_ = 0;
  std::string init_error_;
  std::vector<double*> pinned_, devstage_;    // devstage_: device-side staging for rectangles that are strided on the device (lazy)
  std::vector<hipEvent_t> slot_ev_, blk_ev_;
  std::vector<hipStream_t> st_;
  std::vector<std::thread> workers_;
  std::unique_ptr<std::atomic<int>[]> blocks_done_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  int64_t gen_ = 0;
  int remaining_ = 0;
  bool quit_ = false, active_ = false, trace_ = false;
  std::atomic<bool> failed_{false};
  std::string error_;
  // the transfer in flight
  bool h2d_ = true;
  double* dev_ = nullptr;
  double* host_ = nullptr;
  int64_t ldd_ = 0, ldh_ = 0;
  std::vector<StageRect> rects_;
  int nblocks_ = 1;
  hipEvent_t start_ = nullptr;
};

}  // namespace gsi
