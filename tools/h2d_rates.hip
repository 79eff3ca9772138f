// h2d_rates.hip -- what does the host boundary of this box allow?  ("host->device >= 0.8 of the box's
// measured pinned-copy rate".)  Measures, for one large pageable host buffer (what a Julia Matrix{Float64} is):
//   pinned      hipMemcpyAsync from hipHostMalloc memory: the PCIe ceiling of this box, both directions
//   pageable    hipMemcpy straight from / to malloc memory: what libgsi_hip did through round 4
//   register    hipHostRegister + hipMemcpyAsync + hipHostUnregister of the caller's buffer, whole and in slices
//   staged(T,C) T host threads, each with two pinned buffers of C MiB: memcpy pageable -> pinned, hipMemcpyAsync on the
//               thread's own stream (the design of csrc/host_staging.hpp)
// Build: hipcc -O2 --offload-arch=gfx950 -o tools/h2d_rates tools/h2d_rates.hip -lpthread ; run: tools/h2d_rates [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void touch(char* p, size_t bytes, int threads) {
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t)
    th.emplace_back([=] { size_t a = bytes / threads * t, b = (t == threads - 1) ? bytes : bytes / threads * (t + 1); memset(p + a, t + 1, b - a); });
  for (auto& x : th) x.join();
}

// T threads; thread t handles chunks t, t + T, ...; two pinned buffers each
static double staged(char* dev, char* host, size_t bytes, int T, size_t chunk, bool h2d, char** pinned, hipStream_t* st, hipEvent_t* ev) {
  const size_t nchunks = (bytes + chunk - 1) / chunk;
  double t0 = now();
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([=] {
      CK(hipSetDevice(0));
      int slot = 0;
      bool used[2] = {false, false};
      size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};
      for (size_t k = (size_t)t; k < nchunks; k += (size_t)T) {
        const size_t off = k * chunk, len = (off + chunk <= bytes) ? chunk : bytes - off;
        char* pb = pinned[2 * t + slot];
        hipEvent_t e = ev[2 * t + slot];
        if (used[slot]) {
          CK(hipEventSynchronize(e));
          if (!h2d) memcpy(host + pend_off[slot], pb, pend_len[slot]);
        }
        if (h2d) {
          memcpy(pb, host + off, len);
          CK(hipMemcpyAsync(dev + off, pb, len, hipMemcpyHostToDevice, st[t]));
        } else {
          CK(hipMemcpyAsync(pb, dev + off, len, hipMemcpyDeviceToHost, st[t]));
          pend_off[slot] = off; pend_len[slot] = len;
        }
        CK(hipEventRecord(e, st[t]));
        used[slot] = true;
        slot ^= 1;
      }
      for (int s = 0; s < 2; ++s)
        if (used[s]) {
          CK(hipEventSynchronize(ev[2 * t + s]));
          if (!h2d) memcpy(host + pend_off[s], pinned[2 * t + s], pend_len[s]);
        }
    });
  for (auto& x : th) x.join();
  return now() - t0;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;
  const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)4095;
  CK(hipSetDevice(0));
  char* dev;
  CK(hipMalloc(&dev, bytes));
  CK(hipMemset(dev, 0, bytes));
  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  printf("buffer %.2f GiB; host threads available %u\n", gib, std::thread::hardware_concurrency());

  // ---- pinned ceiling ----
  {
    char* pin;
    double t0 = now();
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    double ta = now() - t0;
    touch(pin, bytes, 8);
    for (int dir = 0; dir < 2; ++dir)
      for (int rep = 0; rep < 3; ++rep) {
        t0 = now();
        if (dir == 0) CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s0));
        else CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0));
        double dt = now() - t0;
        printf("pinned    %s rep %d: %7.2f GB/s\n", dir == 0 ? "H2D" : "D2H", rep, bytes / dt / 1e9);
      }
    // two streams, halves (does a second SDMA engine add anything in one direction?)
    hipStream_t s1;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    t0 = now();
    CK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, s0));
    CK(hipMemcpyAsync(dev + bytes / 2, pin + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s0));
    CK(hipStreamSynchronize(s1));
    printf("pinned    H2D two streams: %7.2f GB/s\n", bytes / (now() - t0) / 1e9);
    // both directions at once
    char* dev2;
    CK(hipMalloc(&dev2, bytes / 2));
    t0 = now();
    CK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, s0));
    CK(hipMemcpyAsync(pin + bytes / 2, dev2, bytes / 2, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s0));
    CK(hipStreamSynchronize(s1));
    printf("pinned    H2D + D2H at once: %7.2f GB/s total\n", bytes / (now() - t0) / 1e9);
    CK(hipFree(dev2));
    CK(hipStreamDestroy(s1));
    t0 = now();
    CK(hipHostFree(pin));
    printf("hipHostMalloc %.3f s, hipHostFree %.3f s for %.2f GiB\n", ta, now() - t0, gib);
  }

  char* host = (char*)aligned_alloc(4096, bytes);
  touch(host, bytes, 8);

  // ---- host memcpy rates (what the staging threads can do) ----
  {
    char* other = (char*)aligned_alloc(4096, bytes);
    touch(other, bytes, 8);
    for (int T : {1, 2, 4, 8, 16}) {
      double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([=] { size_t a = bytes / T * t, b = (t == T - 1) ? bytes : bytes / T * (t + 1); memcpy(other + a, host + a, b - a); });
      for (auto& x : th) x.join();
      printf("host memcpy %2d threads: %7.2f GB/s\n", T, bytes / (now() - t0) / 1e9);
    }
    free(other);
  }

  // ---- pageable ----
  for (int dir = 0; dir < 2; ++dir)
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      if (dir == 0) CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
      else CK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
      double dt = now() - t0;
      printf("pageable  %s rep %d: %7.2f GB/s\n", dir == 0 ? "H2D" : "D2H", rep, bytes / dt / 1e9);
    }

  // ---- register the caller's buffer ----
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
    double tr = now() - t0;
    t0 = now();
    CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    double tc = now() - t0;
    t0 = now();
    CK(hipHostUnregister(host));
    double tu = now() - t0;
    printf("register  H2D rep %d: register %.3f s (%.2f GB/s) + copy %.3f s (%.2f GB/s) + unregister %.3f s = %7.2f GB/s overall\n", rep, tr,
           bytes / tr / 1e9, tc, bytes / tc / 1e9, tu, bytes / (tr + tc + tu) / 1e9);
  }
  {  // sliced: register slice k+1 on a helper thread while slice k copies
    const size_t slice = (size_t)256 << 20;
    const size_t ns = (bytes + slice - 1) / slice;
    double t0 = now();
    std::vector<char> done(ns, 0);
    std::thread reg([&] {
      CK(hipSetDevice(0));
      for (size_t k = 0; k < ns; ++k) {
        size_t off = k * slice, len = (off + slice <= bytes) ? slice : bytes - off;
        CK(hipHostRegister(host + off, len, hipHostRegisterDefault));
        __atomic_store_n(&done[k], 1, __ATOMIC_RELEASE);
      }
    });
    for (size_t k = 0; k < ns; ++k) {
      while (!__atomic_load_n(&done[k], __ATOMIC_ACQUIRE)) std::this_thread::yield();
      size_t off = k * slice, len = (off + slice <= bytes) ? slice : bytes - off;
      CK(hipMemcpyAsync(dev + off, host + off, len, hipMemcpyHostToDevice, s0));
    }
    CK(hipStreamSynchronize(s0));
    reg.join();
    double tc = now() - t0;
    t0 = now();
    for (size_t k = 0; k < ns; ++k) CK(hipHostUnregister(host + k * slice));
    printf("register  H2D sliced 256 MiB, pipelined: copy done after %.3f s (%.2f GB/s), unregister %.3f s more\n", tc, bytes / tc / 1e9, now() - t0);
  }

  // ---- staged through pinned buffers ----
  const int TMAX = 16;
  const size_t CMAX = (size_t)64 << 20;
  char* pinned[2 * TMAX];
  hipStream_t st[TMAX];
  hipEvent_t ev[2 * TMAX];
  double t0 = now();
  for (int i = 0; i < 2 * TMAX; ++i) { CK(hipHostMalloc(&pinned[i], CMAX, hipHostMallocDefault)); memset(pinned[i], 0, CMAX); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
  for (int i = 0; i < TMAX; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  printf("staging ring: %d pinned buffers of %zu MiB allocated + touched in %.3f s\n", 2 * TMAX, CMAX >> 20, now() - t0);
  for (int dir = 0; dir < 2; ++dir)
    for (size_t cm : {4, 16, 64})
      for (int T : {1, 2, 4, 8, 12, 16}) {
        double best = 1e30;
        for (int rep = 0; rep < 2; ++rep) {
          double dt = staged(dev, host, bytes, T, cm << 20, dir == 0, pinned, st, ev);
          if (dt < best) best = dt;
        }
        printf("staged    %s T=%2d chunk %2zu MiB: %7.2f GB/s\n", dir == 0 ? "H2D" : "D2H", T, cm, bytes / best / 1e9);
      }
  // small transfers: latency of the paths (32 MB = C1's matrix, 768 KB = C1's Omega)
  for (size_t sb : {(size_t)768 << 10, (size_t)8 << 20, (size_t)32 << 20, (size_t)128 << 20}) {
    double tp = 1e30, ts[3] = {1e30, 1e30, 1e30};
    for (int rep = 0; rep < 5; ++rep) {
      double t1 = now();
      CK(hipMemcpy(dev, host, sb, hipMemcpyHostToDevice));
      tp = std::min(tp, now() - t1);
      int k = 0;
      for (int T : {1, 4, 8}) {
        double dt = staged(dev, host, sb, T, (size_t)4 << 20, true, pinned, st, ev);
        ts[k] = std::min(ts[k], dt);
        ++k;
      }
    }
    printf("small H2D %6zu KiB: pageable %.3f ms (%.1f GB/s); staged 4 MiB chunks T=1 %.3f ms, T=4 %.3f ms, T=8 %.3f ms (%.1f GB/s)\n", sb >> 10, tp * 1e3,
           sb / tp / 1e9, ts[0] * 1e3, ts[1] * 1e3, ts[2] * 1e3, sb / ts[2] / 1e9);
  }
  return 0;
}
