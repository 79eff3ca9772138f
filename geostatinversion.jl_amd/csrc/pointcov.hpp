// pointcov.hpp -- the stationary covariance kernels of the scattered-point implicit operator (SURVEY.md 8b "kernel-function
// covariance: coords + kernel id + params"), one definition for the device code (the generator table, tests of the in-loader entry) and the CPU reference
// backend of the tests.  r = |x_i - x_j| / ell.
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#define GSI_PC_HD __host__ __device__
#else
#define GSI_PC_HD
#endif
namespace gsi { namespace pointcov {
enum { GAUSSIAN = 0, EXPONENTIAL = 1, MATERN32 = 2, MATERN52 = 3, NUM_KINDS = 4 };
struct Params { int d, kind; double inv_ell, sigma2, nugget; };
// exp(x) for x <= 0.  On the device: n = rint(x log2 e), r = x - n ln 2 (two-piece ln 2), the degree-12 Taylor polynomial
// of exp(r) on |r| <= ln(2)/2 (truncation 1.7e-16 relative) and one ldexp: ~20 instructions against the ~100 of the library
// exp with its special cases -- the generator evaluates it once per matrix entry (1e12 times per pass at n = 1e6).
GSI_PC_HD inline double exp_nonpos(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const bool under = x < -745.2;                   // (a select at the end, not an early return: no divergent branch per entry)
  x = under ? 0.0 : x;
  const double n = rint(x * 1.4426950408889634);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 2.08767569878681e-09;                 // 1/12!
  p = fma(p, r, 2.505210838544172e-08);            // 1/11!
  p = fma(p, r, 2.755731922398589e-07);            // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);           // 1/9!
  p = fma(p, r, 2.48015873015873e-05);             // 1/8!
  p = fma(p, r, 1.984126984126984e-04);            // 1/7!
  p = fma(p, r, 1.3888888888888889e-03);           // 1/6!
  p = fma(p, r, 8.333333333333333e-03);            // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);           // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);           // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double e = ldexp(p, (int)n);
  return under ? 0.0 : e;
#else
  return exp(x);
#endif
}
// d2 = squared distance; same: i and j are the same point (the nugget sits on the diagonal)
GSI_PC_HD inline double kernel(const Params& p, double d2, bool same) {
  const double r2 = d2 * p.inv_ell * p.inv_ell;
  double v;
  if (p.kind == GAUSSIAN) v = exp_nonpos(-0.5 * r2);
  else {
    const double r = sqrt(r2);
    if (p.kind == EXPONENTIAL) v = exp_nonpos(-r);
    else if (p.kind == MATERN32) { const double a = 1.7320508075688772 * r; v = (1.0 + a) * exp_nonpos(-a); }
    else { const double a = 2.23606797749979 * r; v = (1.0 + a + a * a * (1.0 / 3.0)) * exp_nonpos(-a); }
  }
  v *= p.sigma2;
  return same ? v + p.nugget : v;
}
}}  // namespace gsi::pointcov
