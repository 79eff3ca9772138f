// lsqr_state.hpp -- the scalar side of IterativeSolvers.lsqr (Paige & Saunders 1982; IterativeSolvers 0.9 defaults, call
// sites lsqr.jl:54 and lowrank.jl:142 of the reference) as ONE function shared by the device kernel that runs it
// (lsqr_dev.hip) and by the CPU reference backend of the tests: the recurrences live in a small state block in backend
// memory, so an iteration needs no scalar on the host -- the host only polls {stopped, iterations} every few iterations.
// Same operations in the same order as oracle/oracle.py:lsqr.
#pragma once
#include <cmath>
#if defined(__HIPCC__)
#define GSI_HD __host__ __device__
#else
#define GSI_HD
#endif

namespace gsi { namespace lsqrst {

enum {
  ALPHA = 0, BETA, RHOBAR, PHIBAR, ANORM, DDNORM, XXNORM, ZED, CS2, SN2, BNORM, T1, T2,
  STOPPED,      // != 0 once a stopping rule has fired (later iterations are no-ops)
  ITERS,        // iterations applied so far
  APPLY,        // != 0: this iteration's vector updates are to be applied
  MAXITER,      // iteration budget (the device refuses to go beyond it even if the host enqueues more)
  ATOL, BTOL, CTOL,
  COUNT = 24
};

// after u = A v - alpha u:  usq = |u|^2
GSI_HD inline void after_u(double* s, double usq) {
  if (s[STOPPED] != 0.0 || s[ITERS] >= s[MAXITER]) { s[APPLY] = 0.0; return; }
  s[APPLY] = 1.0;
  const double beta = std::sqrt(usq);
  s[BETA] = beta;
  if (beta > 0.0) s[ANORM] = std::sqrt(s[ANORM] * s[ANORM] + s[ALPHA] * s[ALPHA] + beta * beta);
}

// after v = A' u - beta v (only formed when beta > 0):  vsq = |v|^2, wsq = |w|^2 of the CURRENT w
GSI_HD inline void after_v(double* s, double vsq, double wsq) {
  if (s[APPLY] == 0.0) return;
  const double beta = s[BETA];
  if (beta > 0.0) s[ALPHA] = std::sqrt(vsq);
  const double alpha = s[ALPHA];
  const double rhobar1 = s[RHOBAR];                       // damp = 0
  const double rho = std::sqrt(rhobar1 * rhobar1 + beta * beta);
  const double cs = rhobar1 / rho, sn = beta / rho;
  const double theta = sn * alpha;
  s[RHOBAR] = -cs * alpha;
  const double phi = cs * s[PHIBAR];
  s[PHIBAR] = sn * s[PHIBAR];
  const double tau = sn * phi;
  s[T1] = phi / rho;
  s[T2] = -theta / rho;
  const double wn = std::sqrt(wsq);
  s[DDNORM] += (wn / rho) * (wn / rho);
  const double delta = s[SN2] * rho, gambar = -s[CS2] * rho, rhs = phi - delta * s[ZED];
  const double zbar = rhs / gambar;
  const double xnorm = std::sqrt(s[XXNORM] + zbar * zbar);
  const double gamma = std::sqrt(gambar * gambar + theta * theta);
  s[CS2] = gambar / gamma;
  s[SN2] = theta / gamma;
  s[ZED] = rhs / gamma;
  s[XXNORM] += s[ZED] * s[ZED];
  const double Anorm = s[ANORM], bnorm = s[BNORM];
  const double Acond = Anorm * std::sqrt(s[DDNORM]);
  const double rnorm = std::sqrt(s[PHIBAR] * s[PHIBAR]);  // res2 = 0 (damp = 0)
  const double Arnorm = alpha * std::fabs(tau);
  const double test1 = rnorm / bnorm;
  const double test2 = (Anorm * rnorm > 0.0) ? Arnorm / (Anorm * rnorm) : 0.0;
  const double test3 = (Acond > 0.0) ? 1.0 / Acond : 0.0;
  const double t1c = test1 / (1.0 + Anorm * xnorm / bnorm);
  const double rtol = s[BTOL] + s[ATOL] * Anorm * xnorm / bnorm;
  s[ITERS] += 1.0;
  if (1.0 + test3 <= 1.0 || 1.0 + test2 <= 1.0 || 1.0 + t1c <= 1.0) s[STOPPED] = 1.0;
  if (test3 <= s[CTOL] || test2 <= s[ATOL] || test1 <= rtol) s[STOPPED] = 1.0;
}

}}  // namespace gsi::lsqrst
