// pointcov_gen.hpp -- one matrix entry of the scattered-point covariance, as the contraction kernels' tile loaders evaluate it
// (gemm_f64.hip GEN 2: 128 x 160 output tiles; pointcov_gemm.hip: 96 x 320).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include "pointcov.hpp"
namespace gsi { namespace hipk {
// every kind as  v = sigma2 (1 + p1 a + p2 a^2) exp(-arg),  a = c1 r:  Gaussian arg = r^2 / 2 (no square root), the others
// arg = a; (c1, p1, p2) = exponential (1, 0, 0), Matern 3/2 (sqrt 3, 1, 0), Matern 5/2 (sqrt 5, 1, 1/3).  The same
// association as pointcov::kernel: (1 + a) + (a a) / 3.  One straight-line body for all kinds, parameters in SGPRs.
struct GenPointK { double p2; int flags; };   // flags: 1 = third coordinate, 2 = Gaussian (no square root), 4 = Matern polynomial
// Second pass (round 4): every VALU instruction here is matrix time lost (an fp64 MFMA holds the vector ALU),
// so the entry is on a diet -- 27 vector instructions for the exponential kernel in two dimensions where the first version had 55:
//  * the points arrive PRE-SCALED by c1 / ell (1 / (ell sqrt 2) for the Gaussian): the squared distance is the squared
//    argument, no multiply by 1 / ell^2 and none by c1;
//  * r^2 + 1e-280 instead of a select around the square root (coincident points), and no clamp in front of the exponential
//    (the exponent saturates by itself);
//  * sqrt: v_rsq_f64 (2^-23), ONE Goldschmidt step (-> 2^-45) and the residual correction g += (x - g^2) h, whose error is
//    the product of the two (2^-68): 7 instructions;
//  * exp(-a) sigma^2 = 2^n (sigma^2 2^(j/64)) e^r with -a = (64 n + j) ln 2 / 64 + r, |r| <= ln 2 / 128: a 64-entry table in
//    LDS (filled by the first wave at kernel start; the lookup is an LDS instruction, not a vector one), a degree-5 polynomial
//    (r^6 / 720 < 4e-17) instead of degree 12, sigma^2 inside the table: 15 instructions;
//  * both entries of a pair slot side by side (two independent chains);
//  * the Matern polynomial behind a uniform branch, the nugget behind a uniform "does this slot's column fall into the
//    workgroup's rows" test, no edge selects at all (rows beyond M are never stored; reduction indices beyond the range meet
//    zero rows of the X tile, and the generated entry there is finite: it is the clamped index's).
__device__ __forceinline__ double gen_sqrt(double x) {       // x in [1e-280, 1e300]
  const double y0 = __builtin_amdgcn_rsq(x);
  double g = x * y0;
  const double h = 0.5 * y0;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  const double d = fma(-g, g, x);
  return fma(d, h, g);
}
// sigma^2 exp(-a), a >= 0; tab[j] = sigma^2 2^(j/64) / 120 (the polynomial is 120 x Taylor's: every coefficient an integer and
// the only constant of its instruction).  No clamp: beyond a = 745 the exponent 64 n + j saturates in the conversion and
// ldexp returns 0.
__device__ __forceinline__ double gen_exp_neg(double a, const double* tab) {
  const double kd = rint(a * -92.33248261689366);                        // 64 n + j = -a 64 / ln 2, a non-positive integer
  // ONE piece of ln 2 / 64: the error kd (c - ln 2 / 64) <= a 2^-54 is half of what the rounding of `a` itself already put
  // into the exponent (|delta exp(-a)| <= a exp(-a) 2^-53 <= 4e-17 either way)
  const double r = fma(kd, -1.0830424696249145e-02, -a);
  const int ki = (int)kd;
  const double t = tab[ki & 63];
  // (the addends through SGPRs: left alone the compiler rebuilds each one in a vector register pair, two v_mov per constant)
  auto fma_s = [](double x, double y, double c) -> double {
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(y), "s"(c));
    return o;
  };
  double p = r + 5.0;
  p = fma_s(p, r, 20.0);
  p = fma_s(p, r, 60.0);
  p = fma_s(p, r, 120.0);
  p = fma_s(p, r, 120.0);
  return ldexp(t * p, ki >> 6);
}
// the uniform flags of a slot as ONE opaque scalar (left alone they are hoisted as i1 values and rebuilt with two vector
// instructions per use, v_cndmask + v_cmp)
__device__ __forceinline__ int gen_flags(int f) { f = __builtin_amdgcn_readfirstlane(f); asm volatile("" : "+s"(f)); return f; }
// the two entries of a pair slot (same column point, adjacent rows): s = squared scaled distances
__device__ __forceinline__ void gen_point_pair(const GenPointK& q, int fl, double s0, double s1, const double* tab, double& v0, double& v1) {
  double a0 = s0 + 1e-280, a1 = s1 + 1e-280;       // coincident points: rsq(0) = inf
  if (!(fl & 2)) {                                 // uniform
    a0 = gen_sqrt(a0); a1 = gen_sqrt(a1);
    asm volatile("" ::: "memory");                 // (keeps the branch a branch)
  }
  v0 = gen_exp_neg(a0, tab); v1 = gen_exp_neg(a1, tab);
  if (fl & 4) {                                    // uniform; the association of pointcov::kernel
    asm volatile("" ::: "memory");
    v0 *= (1.0 + a0) + (a0 * a0) * q.p2;
    v1 *= (1.0 + a1) + (a1 * a1) * q.p2;
  }
}
// three entries side by side (the 96-row tile of pointcov_gemm.hip: three independent chains per lane)
__device__ __forceinline__ void gen_point_triple(const GenPointK& q, int fl, double s0, double s1, double s2, const double* tab,
                                                 double& v0, double& v1, double& v2) {
  double a0 = s0 + 1e-280, a1 = s1 + 1e-280, a2 = s2 + 1e-280;
  if (!(fl & 2)) {                                 // uniform
    a0 = gen_sqrt(a0); a1 = gen_sqrt(a1); a2 = gen_sqrt(a2);
    asm volatile("" ::: "memory");
  }
  v0 = gen_exp_neg(a0, tab); v1 = gen_exp_neg(a1, tab); v2 = gen_exp_neg(a2, tab);
  if (fl & 4) {                                    // uniform
    asm volatile("" ::: "memory");
    v0 *= (1.0 + a0) + (a0 * a0) * q.p2;
    v1 *= (1.0 + a1) + (a1 * a1) * q.p2;
    v2 *= (1.0 + a2) + (a2 * a2) * q.p2;
  }
}
// the table of gen_exp_neg, filled by the first wave of a workgroup (a barrier must follow)
__device__ __forceinline__ void gen_table_init(double* tab, int tid, double sigma2) {
  if (tid < 64) tab[tid] = (sigma2 / 60.0) * pointcov::exp_nonpos((double)(tid - 64) * 1.0830424696249145e-02);   // 2 sigma^2 / 120 x exp((j - 64) ln 2 / 64)
}
__device__ __forceinline__ GenPointK gen_point_setup(int dim, int kind) {
  GenPointK q;
  q.flags = ((dim > 2) ? 1 : 0) | ((kind == pointcov::GAUSSIAN) ? 2 : 0) | ((kind == pointcov::MATERN32 || kind == pointcov::MATERN52) ? 4 : 0);
  q.p2 = (kind == pointcov::MATERN52) ? (1.0 / 3.0) : 0.0;
  return q;
}
}}  // namespace gsi::hipk
