// mfma_f64_peak.hip -- what rate does v_mfma_f64_16x16x4_f64 actually sustain on this MI355X?
// A known-good ceiling for gemm_f64.hip (cdna_hip_programming.md 5.4 rule 10): NACC independent
// accumulators per wave, operands in registers, W waves per SIMD.  Prints TFLOP/s and the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0) {
  double4_t acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t)   // inline asm: hipcc otherwise shuffles the accumulators VGPR<->AGPR every iteration
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[(size_t)gridDim.x * blockDim.x] = (double)(t1 - t0);
    out[(size_t)gridDim.x * blockDim.x + 1] = (double)(r1 - r0);
  }
}

template <int NACC>
void run(int wgs_per_cu, int iters) {
  const int blocks = 256 * wgs_per_cu;
  double* d;
  hipMalloc(&d, sizeof(double) * ((size_t)blocks * 256 + 2));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(d, 100, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(d, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double h[2];
  hipMemcpy(h, d + (size_t)blocks * 256, sizeof(h), hipMemcpyDeviceToHost);
  const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
  const double clk_ghz = h[0] / (h[1] * 10.0) ;   // memrealtime ticks at 100 MHz
  printf("NACC=%2d waves/SIMD=%d iters=%d: %.3f ms  %.2f TFLOP/s  cycles/MFMA/SIMD=%.1f  clock=%.3f GHz\n", NACC,
         wgs_per_cu, iters, ms, flops / (ms * 1e-3) / 1e12, h[0] / ((double)iters * NACC * wgs_per_cu), clk_ghz);
  hipFree(d);
}

int main(int argc, char** argv) {
  if (argc > 1) {  // sustained mode: ~N seconds of back-to-back launches for power/clock sampling
    const int secs = atoi(argv[1]);
    for (int i = 0; i < secs * 28; ++i) run<10>(2, 40000);
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    run<4>(1, 20000);
    run<10>(1, 8000);
    run<10>(2, 8000);
    run<10>(4, 4000);
    run<1>(1, 40000);
    run<2>(1, 40000);
  }
  // sustained: ~2 s back-to-back for the DVFS steady state
  for (int i = 0; i < 12; ++i) run<10>(2, 40000);
  return 0;
}
