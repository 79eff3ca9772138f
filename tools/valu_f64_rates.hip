// valu_f64_rates.hip -- issue cost of the vector instructions the scattered-point generator is made of, on an MI355X SIMD
// (round 4: halving the generator's instruction count took only a quarter off its time -- which of its instructions are the
// expensive ones?).  One wave per SIMD (256-thread workgroups, one per CU), 8 independent chains per thread, every instruction
// through `asm volatile`; reported: SIMD cycles per wave64 instruction at the measured clock (s_memtime is constant-rate, so
// the clock is taken from v_fma_f32 = 4 cycles... no: reported RELATIVE to v_fma_f64 and in ns per instruction).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_f64_rates.hip -o tools/valu_f64_rates && ./tools/valu_f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define OPS(X)                                                                                                   \
  X(0, "v_fma_f64", "v_fma_f64 %0, %0, %1, %2", double)                                                          \
  X(1, "v_mul_f64", "v_mul_f64 %0, %0, %1", double)                                                              \
  X(2, "v_add_f64", "v_add_f64 %0, %0, %2", double)                                                              \
  X(3, "v_rsq_f64", "v_rsq_f64 %0, %0", double)                                                                  \
  X(4, "v_rcp_f64", "v_rcp_f64 %0, %0", double)                                                                  \
  X(5, "v_sqrt_f64", "v_sqrt_f64 %0, %0", double)                                                                \
  X(6, "v_rndne_f64", "v_rndne_f64 %0, %0", double)                                                              \
  X(7, "v_ldexp_f64", "v_ldexp_f64 %0, %0, %3", double)                                                          \
  X(8, "v_max_f64", "v_max_f64 %0, %0, %1", double)                                                              \
  X(9, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2", float)                                                           \
  X(10, "v_rsq_f32", "v_rsq_f32 %0, %0", float)                                                                  \
  X(11, "v_exp_f32", "v_exp_f32 %0, %0", float)                                                                  \
  X(12, "v_mov_b32", "v_mov_b32 %0, %1", float)                                                                  \
  X(13, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %3", int)                                                   \
  X(14, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %3, vcc", int)                                                   \
  X(15, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2", double)                                                   \
  X(16, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %3", int)

template <int OP>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double m0, double c0, int e0) {
  double s = 0.0;
#define X(ID, NAME, ASM, T)                                                                                      \
  if constexpr (OP == ID) {                                                                                      \
    T x[8];                                                                                                      \
    T m = (T)m0, c = (T)c0;                                                                                      \
    for (int t = 0; t < 8; ++t) x[t] = (T)(threadIdx.x + 1 + t);                                                 \
    for (int i = 0; i < iters; ++i) {                                                                            \
      _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                             \
      _Pragma("unroll") for (int t = 0; t < 8; ++t) asm volatile(ASM : "+v"(x[t]) : "v"(m), "v"(c), "v"(e0));    \
    }                                                                                                            \
    for (int t = 0; t < 8; ++t) s += (double)x[t];                                                               \
  }
  OPS(X)
#undef X
  // cvt pair: f64 -> i32 -> f64 (two instructions)
  if constexpr (OP == 100) {
    double x[8]; int k[8];
    for (int t = 0; t < 8; ++t) x[t] = threadIdx.x + t;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(k[t]) : "v"(x[t]));
          asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[t]) : "v"(k[t]));
        }
    }
    for (int t = 0; t < 8; ++t) s += x[t];
  }
  // cvt pair: f64 -> f32 -> f64
  if constexpr (OP == 101) {
    double x[8]; float k[8];
    for (int t = 0; t < 8; ++t) x[t] = threadIdx.x + t;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(k[t]) : "v"(x[t]));
          asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[t]) : "v"(k[t]));
        }
    }
    for (int t = 0; t < 8; ++t) s += x[t];
  }
  // one DEPENDENT chain of v_fma_f64 (latency rather than issue rate)
  if constexpr (OP == 102) {
    double x = threadIdx.x, m = m0, c = c0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 128; ++r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
    }
    s += x;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static double run(const char* name, double* out, int iters, double base_ns) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  rate<OP><<<256, 256>>>(out, 10, 1.0000001, 1e-9, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<OP><<<256, 256>>>(out, iters, 1.0000001, 1e-9, 1);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * 128.0);
  printf("%-34s %8.3f ms  %7.3f ns per wave instruction", name, ms, ns);
  if (base_ns > 0) printf("  = %5.2f x v_fma_f64", ns / base_ns);
  printf("\n");
  return ns;
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 256 * sizeof(double));
  const int iters = 20000;
  double b = 0;
#define X(ID, NAME, ASM, T) { const double ns = run<ID>(NAME, out, iters, b); if (ID == 0) b = ns; }
  OPS(X)
#undef X
  run<100>("v_cvt_i32_f64 + v_cvt_f64_i32 (/2)", out, iters, b);
  run<101>("v_cvt_f32_f64 + v_cvt_f64_f32 (/2)", out, iters, b);
  run<102>("v_fma_f64, ONE dependent chain", out, iters, b);
  hipFree(out);
  return 0;
}
