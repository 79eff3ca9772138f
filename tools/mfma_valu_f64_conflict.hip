// mfma_valu_f64_conflict.hip -- do fp64 VALU instructions hide behind fp64 MFMAs of the partner wave on an MI355X SIMD?
// (round 4: the scattered-point covariance generated inside the contraction's tile loader costs ~50 v_fma_f64-class
// instructions per entry; the contraction got exactly that much slower.)  One 512-thread workgroup per CU = two waves per
// SIMD: waves 0-3 issue NM v_mfma_f64_16x16x4_f64 per iteration, waves 4-7 NV v_fma_f64 (8 independent chains).  Three
// launches: MFMA waves alone, VALU waves alone, both.  If the two share the SIMD's fp64 multipliers the third takes the
// SUM of the first two; if they overlap, the max.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_f64_conflict.hip -o tools/mfma_valu_f64_conflict && ./tools/mfma_valu_f64_conflict
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <bool F32>
__global__ __launch_bounds__(512) void conflict(double* out, int iters, int do_mfma, int do_valu, double a0, double b0) {
  const int wave = threadIdx.x >> 6;
  double s = 0.0;
  if (wave < 4) {
    if (do_mfma) {
      double4_t acc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
      double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
      }
      asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
      for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    }
  } else if (do_valu) {
    if (F32) {
      float x[8], m = (float)a0 * 1e-7f + 1.0f, c = (float)b0 * 1e-9f;
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = threadIdx.x * 1e-3f + t;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)                       // 128 v_fma_f32 per iteration
#pragma unroll
          for (int t = 0; t < 8; ++t) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[t]) : "v"(m), "v"(c));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) s += x[t];
    } else {
      double x[8], m = a0 * 1e-7 + 1.0, c = b0 * 1e-9;
#pragma unroll
      for (int t = 0; t < 8; ++t) x[t] = threadIdx.x * 1e-3 + t;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)                       // 128 v_fma_f64 per iteration
#pragma unroll
          for (int t = 0; t < 8; ++t) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[t]) : "v"(m), "v"(c));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) s += x[t];
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool F32>
static float run(double* d, int iters, int m, int v) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  conflict<F32><<<256, 512>>>(d, 50, m, v, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  conflict<F32><<<256, 512>>>(d, iters, m, v, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 256 * 512);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    const float tm = run<false>(d, iters, 1, 0), tv = run<false>(d, iters, 0, 1), tb = run<false>(d, iters, 1, 1);
    printf("fp64 VALU partner: 8 MFMA f64 / iteration alone %.3f ms (%.1f cycles per MFMA at 2.4 GHz), 128 v_fma_f64 / iteration alone %.3f ms "
           "(%.2f cycles per instruction), both %.3f ms  -> sum %.3f, max %.3f\n",
           tm, tm * 1e-3 * 2.4e9 / (iters * 8.0), tv, tv * 1e-3 * 2.4e9 / (iters * 128.0), tb, tm + tv, tm > tv ? tm : tv);
    const float sm = run<true>(d, iters, 1, 0), sv = run<true>(d, iters, 0, 1), sb = run<true>(d, iters, 1, 1);
    printf("fp32 VALU partner: MFMA alone %.3f ms, 128 v_fma_f32 / iteration alone %.3f ms (%.2f cycles per instruction), both %.3f ms  -> sum %.3f, max %.3f\n",
           sm, sv, sv * 1e-3 * 2.4e9 / (iters * 128.0), sb, sm + sv, sm > sv ? sm : sv);
  }
  hipFree(d);
  return 0;
}
