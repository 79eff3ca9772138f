// pointcov.hpp -- the stationary covariance kernels of the scattered-point implicit operator (SURVEY.md 8b "kernel-function
// covariance: coords + kernel id + params"), one definition for the device generator (pointcov.hip) and the CPU reference
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
// d2 = squared distance; same: i and j are the same point (the nugget sits on the diagonal)
GSI_PC_HD inline double kernel(const Params& p, double d2, bool same) {
  const double r2 = d2 * p.inv_ell * p.inv_ell;
  double v;
  if (p.kind == GAUSSIAN) v = exp(-0.5 * r2);
  else {
    const double r = sqrt(r2);
    if (p.kind == EXPONENTIAL) v = exp(-r);
    else if (p.kind == MATERN32) { const double a = 1.7320508075688772 * r; v = (1.0 + a) * exp(-a); }
    else { const double a = 2.23606797749979 * r; v = (1.0 + a + a * a * (1.0 / 3.0)) * exp(-a); }
  }
  v *= p.sigma2;
  return same ? v + p.nugget : v;
}
}}  // namespace gsi::pointcov
