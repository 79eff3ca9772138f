/*
 * gsi_oracle.c -- self-contained C restatement (no BLAS/LAPACK) of the RandMatFact hot path
 * of GeostatInversion.jl.  TEST INFRASTRUCTURE ONLY: built into oracle/_build/ and used by
 * tests/ (a) as a second, LAPACK-independent checker next to oracle.py and (b) as the compute
 * layer of the CPU reference backend (cpu_backend.cpp) that lets the row-sharded pipeline and
 * the C ABI be exercised without a GPU.  The product library never links or loads this.
 *
 * Pinning: checked in tests/test_oracle_c.py against oracle.py (scipy: the same LAPACK routines
 * Julia calls) and against the reference's own known-answer tests (test/testrmf.jl:21-29 Nystrom
 * eigenvalues; test/testrpcga.jl:46-58 LowRankCovMatrix) -- the reference itself (Julia) cannot
 * run in this image: "pinned by reference KATs/properties only".
 *
 * All matrices column-major double, explicit leading dimensions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define A_(i, j) A[(i) + (int64_t)(j) * lda]

/* C(m x l) = alpha*A(m x k)*B(k x l) + beta*C      (dgemm 'N','N'; RandMatFact.jl:55,70) */
void gsio_gemm_nn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda,
                  const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < l; ++c) {
    double* Cc = C + c * ldc;
    if (beta == 0.0) for (int64_t i = 0; i < m; ++i) Cc[i] = 0.0;
    else if (beta != 1.0) for (int64_t i = 0; i < m; ++i) Cc[i] *= beta;
    for (int64_t p = 0; p < k; ++p) {
      const double b = alpha * B[p + c * ldb];
      const double* Ap = A + p * lda;
      for (int64_t i = 0; i < m; ++i) Cc[i] += Ap[i] * b;
    }
  }
}

/* C(m x l) = alpha*A'(k x m stored)*B(k x l) + beta*C   (dgemm 'T','N'; RandMatFact.jl:67,85) */
void gsio_gemm_tn(int64_t m, int64_t l, int64_t k, double alpha, const double* A, int64_t lda,
                  const double* B, int64_t ldb, double beta, double* C, int64_t ldc) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int64_t c = 0; c < l; ++c)
    for (int64_t i = 0; i < m; ++i) {
      const double* Ai = A + i * lda;
      const double* Bc = B + c * ldb;
      double s = 0.0;
      for (int64_t p = 0; p < k; ++p) s += Ai[p] * Bc[p];
      C[i + c * ldc] = alpha * s + (beta == 0.0 ? 0.0 : beta * C[i + c * ldc]);
    }
}

/* F = lu(Y); F.L  (RandMatFact.jl:60-61): dgetf2 semantics -- partial pivoting, first maximal
 * |entry| (idamax), row interchanges applied across the whole panel, scaling by the reciprocal
 * pivot.  In place: on return Y holds L (unit lower trapezoidal, PIVOTED row order, strict upper
 * part zeroed).  ipiv (may be NULL): 0-based pivot rows.  Returns 0, or j+1 for the first exactly
 * zero pivot (Julia: SingularException). */
int gsio_lu_L(double* A, int64_t m, int64_t l, int64_t lda, int32_t* ipiv) {
  int info = 0;
  const int64_t kmax = m < l ? m : l;
  for (int64_t j = 0; j < kmax; ++j) {
    int64_t r = j;
    double best = fabs(A_(j, j));
    for (int64_t i = j + 1; i < m; ++i) {
      const double a = fabs(A_(i, j));
      if (a > best) { best = a; r = i; }
    }
    if (ipiv) ipiv[j] = (int32_t)r;
    if (best == 0.0) { if (!info) info = (int)(j + 1); continue; }
    if (r != j)
      for (int64_t c = 0; c < l; ++c) { const double t = A_(j, c); A_(j, c) = A_(r, c); A_(r, c) = t; }
    const double rp = 1.0 / A_(j, j);
    for (int64_t i = j + 1; i < m; ++i) A_(i, j) *= rp;
#pragma omp parallel for schedule(static) if ((m - j) * (l - j) > 20000)
    for (int64_t c = j + 1; c < l; ++c) {
      const double u = A_(j, c);
      if (u != 0.0)
        for (int64_t i = j + 1; i < m; ++i) A_(i, c) -= A_(i, j) * u;
    }
  }
  for (int64_t c = 0; c < l; ++c)
    for (int64_t i = 0; i <= c && i < m; ++i) A_(i, c) = (i == c) ? 1.0 : 0.0;
  return info;
}

static double nrm2_(const double* x, int64_t n) {
  double scale = 0.0, ssq = 1.0;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] != 0.0) {
      const double a = fabs(x[i]);
      if (scale < a) { ssq = 1.0 + ssq * (scale / a) * (scale / a); scale = a; }
      else ssq += (a / scale) * (a / scale);
    }
  return scale * sqrt(ssq);
}

/* qr(Y, Val(true)) -> Matrix(F.Q)  (RandMatFact.jl:57-58): Householder QR, with column pivoting
 * (largest remaining column norm first, as dgeqp3 does) when pivot != 0; explicit thin Q
 * (dorg2r).  In place: Y <- Q (m x l).  R (l x l, ld l, may be NULL) <- the triangular factor
 * (of the column-permuted Y when pivot != 0); jpvt (may be NULL) <- 0-based column order. */
void gsio_qr_thinQ(double* A, int64_t m, int64_t l, int64_t lda, int pivot, double* R, int32_t* jpvt) {
  double* tau = (double*)calloc((size_t)l, sizeof(double));
  double* w = (double*)calloc((size_t)l, sizeof(double));
  if (jpvt) for (int64_t c = 0; c < l; ++c) jpvt[c] = (int32_t)c;
  const int64_t kmax = m < l ? m : l;
  for (int64_t j = 0; j < kmax; ++j) {
    if (pivot) {
      int64_t pc = j;
      double best = -1.0;
      for (int64_t c = j; c < l; ++c) {
        const double nc = nrm2_(&A_(j, c), m - j);
        if (nc > best) { best = nc; pc = c; }
      }
      if (pc != j) {
        for (int64_t i = 0; i < m; ++i) { const double t = A_(i, j); A_(i, j) = A_(i, pc); A_(i, pc) = t; }
        if (jpvt) { const int32_t t = jpvt[j]; jpvt[j] = jpvt[pc]; jpvt[pc] = t; }
      }
    }
    /* dlarfg */
    const double alpha = A_(j, j);
    const double xnorm = nrm2_(&A_(j + 1 < m ? j + 1 : j, j), m - j - 1);
    if (xnorm == 0.0) { tau[j] = 0.0; continue; }
    const double beta = -copysign(hypot(alpha, xnorm), alpha);
    tau[j] = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    for (int64_t i = j + 1; i < m; ++i) A_(i, j) *= sc;
    A_(j, j) = beta;
    /* apply H = I - tau v v' to the trailing columns */
#pragma omp parallel for schedule(static) if ((m - j) * (l - j) > 20000)
    for (int64_t c = j + 1; c < l; ++c) {
      double s = A_(j, c);
      for (int64_t i = j + 1; i < m; ++i) s += A_(i, j) * A_(i, c);
      s *= tau[j];
      A_(j, c) -= s;
      for (int64_t i = j + 1; i < m; ++i) A_(i, c) -= s * A_(i, j);
    }
  }
  if (R)
    for (int64_t c = 0; c < l; ++c)
      for (int64_t i = 0; i < l; ++i) R[i + c * l] = (i <= c && i < m) ? A_(i, c) : 0.0;
  /* dorg2r: Q = H_0 ... H_{k-1} [I; 0] */
  for (int64_t j = kmax - 1; j >= 0; --j) {
    /* apply H_j to Q[j:m, j+1:l] */
    for (int64_t c = j + 1; c < l; ++c) {
      double s = A_(j, c);
      for (int64_t i = j + 1; i < m; ++i) s += A_(i, j) * A_(i, c);
      s *= tau[j];
      A_(j, c) -= s;
      for (int64_t i = j + 1; i < m; ++i) A_(i, c) -= s * A_(i, j);
    }
    for (int64_t i = j + 1; i < m; ++i) A_(i, j) *= -tau[j];
    A_(j, j) = 1.0 - tau[j];
    for (int64_t i = 0; i < j; ++i) A_(i, j) = 0.0;
  }
  free(tau);
  free(w);
}

/* Thin SVD of a tall W (n x l) by one-sided Jacobi on its columns: W J = U S.  In place: W <- U
 * (left singular vectors, columns sorted by descending S; a zero column for a zero singular
 * value); S (l).  This is `svd(B)` of RandMatFact.jl:86 for B = W' (S and V = U of W). */
int gsio_svd_tall(double* A, int64_t n, int64_t l, int64_t lda, double* S) {
  const double tol = sqrt((double)(n > l ? n : l)) * 2.220446049250313e-16;
  int sweeps = 0;
  for (; sweeps < 60; ++sweeps) {
    int64_t rot = 0;
    for (int64_t p = 0; p < l - 1; ++p)
      for (int64_t q = p + 1; q < l; ++q) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int64_t i = 0; i < n; ++i) { a += A_(i, p) * A_(i, p); b += A_(i, q) * A_(i, q); c += A_(i, p) * A_(i, q); }
        if (a > 0.0 && b > 0.0 && fabs(c) > tol * sqrt(a * b)) {
          const double zeta = (b - a) / (2.0 * c);
          const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
          for (int64_t i = 0; i < n; ++i) {
            const double x = A_(i, p), y = A_(i, q);
            A_(i, p) = cs * x - sn * y;
            A_(i, q) = sn * x + cs * y;
          }
          ++rot;
        }
      }
    if (rot == 0) break;
  }
  double* nr = (double*)malloc(sizeof(double) * (size_t)l);
  int64_t* rank = (int64_t*)malloc(sizeof(int64_t) * (size_t)l);
  for (int64_t c = 0; c < l; ++c) nr[c] = nrm2_(&A_(0, c), n);
  for (int64_t c = 0; c < l; ++c) {
    int64_t r = 0;
    for (int64_t j = 0; j < l; ++j) if (nr[j] > nr[c] || (nr[j] == nr[c] && j < c)) ++r;
    rank[c] = r;
  }
  double* tmp = (double*)malloc(sizeof(double) * (size_t)n * (size_t)l);
  for (int64_t c = 0; c < l; ++c) {
    const double inv = nr[c] > 0.0 ? 1.0 / nr[c] : 0.0;
    for (int64_t i = 0; i < n; ++i) tmp[i + rank[c] * n] = A_(i, c) * inv;
    S[rank[c]] = nr[c];
  }
  for (int64_t c = 0; c < l; ++c) memcpy(&A_(0, c), tmp + c * n, sizeof(double) * (size_t)n);
  free(tmp); free(nr); free(rank);
  return sweeps;
}

/* cholesky(Hermitian(B)).U  (RandMatFact.jl:95): upper triangle read; in place B <- U (strict
 * lower part zeroed).  Returns 0 or k+1 when the leading minor of order k+1 is not positive. */
int gsio_chol_upper(double* A, int64_t j) {
  const int64_t lda = j;
  for (int64_t k = 0; k < j; ++k) {
    double d = A_(k, k);
    for (int64_t p = 0; p < k; ++p) d -= A_(p, k) * A_(p, k);
    if (!(d > 0.0)) return (int)(k + 1);
    d = sqrt(d);
    A_(k, k) = d;
    for (int64_t c = k + 1; c < j; ++c) {
      double v = A_(k, c);
      for (int64_t p = 0; p < k; ++p) v -= A_(p, k) * A_(p, c);
      A_(k, c) = v / d;
    }
  }
  for (int64_t c = 0; c < j; ++c) for (int64_t i = c + 1; i < j; ++i) A_(i, c) = 0.0;
  return 0;
}

/* F <- F * inv(C), C upper triangular j x j  (RandMatFact.jl:96) */
void gsio_trsm_right_upper(double* F, int64_t m, int64_t j, int64_t ldf, const double* C) {
  for (int64_t c = 0; c < j; ++c) {
    for (int64_t p = 0; p < c; ++p) {
      const double u = C[p + c * j];
      if (u != 0.0) for (int64_t i = 0; i < m; ++i) F[i + c * ldf] -= F[i + p * ldf] * u;
    }
    const double d = 1.0 / C[c + c * j];
    for (int64_t i = 0; i < m; ++i) F[i + c * ldf] *= d;
  }
}

/* rows of S (n x N): subtract the mean over the N columns  (lowrank.jl:17-27) */
void gsio_center_rows(double* S, int64_t n, int64_t N, int64_t ld) {
  for (int64_t r = 0; r < n; ++r) {
    double mean = 0.0;
    for (int64_t i = 0; i < N; ++i) mean += S[r + i * ld];
    mean /= (double)N;
    for (int64_t i = 0; i < N; ++i) S[r + i * ld] -= mean;
  }
}

/* ---- the composed algorithms, dense A (m x n), for the LAPACK-free cross-check -------------- */

/* rangefinder(A, l, numiterations)  RandMatFact.jl:50-80.  Omega n x l.  Q_out m x l.
 * Returns 0, -1 for numiterations < 0 (the reference raises error(...), :62-64), or the LU info. */
int gsio_rangefinder(const double* A, int64_t m, int64_t n, int64_t lda, const double* Omega, int64_t l,
                     int64_t q, double* Q) {
  if (q < 0) return -1;
  double* Z = (double*)malloc(sizeof(double) * (size_t)n * (size_t)l);
  int info = 0;
  gsio_gemm_nn(m, l, n, 1.0, A, lda, Omega, n, 0.0, Q, m);          /* Y = A*Omega           :55 */
  if (q == 0) { gsio_qr_thinQ(Q, m, l, m, 1, NULL, NULL); free(Z); return 0; }   /* :57-58 */
  info = gsio_lu_L(Q, m, l, m, NULL);                                 /* Q = lu(Y).L           :60-61 */
  for (int64_t i = 1; i <= q && !info; ++i) {
    gsio_gemm_tn(n, l, m, 1.0, A, lda, Q, m, 0.0, Z, n);              /* Q = A'*Q              :67 */
    info = gsio_lu_L(Z, n, l, n, NULL);                               /*                       :68-69 */
    if (info) break;
    gsio_gemm_nn(m, l, n, 1.0, A, lda, Z, n, 0.0, Q, m);              /* Q = A*Q               :70 */
    if (i < q) info = gsio_lu_L(Q, m, l, m, NULL);                    /*                       :72-73 */
    else gsio_qr_thinQ(Q, m, l, m, 1, NULL, NULL);                    /*                       :75-76 */
  }
  free(Z);
  return info;
}

/* randsvd(A, K, p, q)  RandMatFact.jl:83-90.  Z_out n x (K+p), S_out K+p. */
int gsio_randsvd(const double* A, int64_t m, int64_t n, int64_t lda, const double* Omega, int64_t K,
                 int64_t p, int64_t q, double* Z, double* S) {
  const int64_t l = K + p;
  double* Q = (double*)malloc(sizeof(double) * (size_t)m * (size_t)l);
  const int info = gsio_rangefinder(A, m, n, lda, Omega, l, q, Q);    /*                       :84 */
  if (info) { free(Q); return info; }
  gsio_gemm_tn(n, l, m, 1.0, A, lda, Q, m, 0.0, Z, n);                /* B' = A'Q              :85 */
  gsio_svd_tall(Z, n, l, n, S);                                       /* (), S, V = svd(B)     :86 */
  for (int64_t c = 0; c < l; ++c) {                                   /* Z = V*Sh              :87-88 */
    const double s = c < K ? sqrt(S[c]) : 0.0;
    for (int64_t i = 0; i < n; ++i) Z[i + c * n] *= s;
  }
  free(Q);
  return 0;
}

/* eig_nystrom(A, Q)  RandMatFact.jl:92-102.  A n x n, Q n x j.  U n x j, Sigma j. */
int gsio_eig_nystrom(const double* A, int64_t n, int64_t lda, const double* Q, int64_t j, double* U,
                     double* Sigma) {
  double* B2 = (double*)malloc(sizeof(double) * (size_t)j * (size_t)j);
  gsio_gemm_nn(n, j, n, 1.0, A, lda, Q, n, 0.0, U, n);                /* B1 = A*Q              :93 */
  gsio_gemm_tn(j, j, n, 1.0, Q, n, U, n, 0.0, B2, j);                 /* B2 = Q'*B1            :94 */
  const int info = gsio_chol_upper(B2, j);                            /* C                     :95 */
  if (info) { free(B2); return info; }
  gsio_trsm_right_upper(U, n, j, n, B2);                              /* F = B1*inv(C)         :96 */
  gsio_svd_tall(U, n, j, n, Sigma);                                   /* U, Sigmavec = svd(F)  :97 */
  free(B2);
  return 0;
}
