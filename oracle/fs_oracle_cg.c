/*
 * fs_oracle_cg.c -- TEST INFRASTRUCTURE ONLY (see fs_oracle.c).
 *
 * CPU restatement of the conjugate-gradient consumers of the A_mul_B path:
 * bsbm_cg (cg.h:25-82), bsbm_cg2 (cg.h:85-187) and the linalg.h reductions they use, over flat
 * CSR arrays of A (N x F) and A' (F x N), both pattern-only.  Pinned bit-for-bit against the real
 * reference (oracle/_ref, built without OpenMP: every reduction is one left-to-right sum) by
 * tests/test_oracle_vs_ref.py and by the golden vectors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FSO_API __attribute__((visibility("default")))

/* y = A x on a pattern-only CSR with k row-major columns, storage-order sums (bsbm_A_mul_B / _B2 of a
 * BlockedSBM add each row's entries in block order, which is the CSR order of fso_coo_to_blocked + stable CSR) */
static void spmm(double *Y, int nrow, const int *rp, const int *cols, const double *X, int k)
{
  for (int r = 0; r < nrow; r++) {
    for (int j = 0; j < k; j++) Y[(int64_t)r * k + j] = 0.0;
    for (int64_t i = rp[r]; i < rp[r + 1]; i++)
      for (int j = 0; j < k; j++) Y[(int64_t)r * k + j] += X[(int64_t)cols[i] * k + j];
  }
}

/* linalg.h:15-22, 51-58 */
static double normsq(const double *x, int n) { double s = 0; for (int i = 0; i < n; i++) s += x[i] * x[i]; return s; }
static double dot(const double *x, const double *y, int n) { double s = 0; for (int i = 0; i < n; i++) s += x[i] * y[i]; return s; }

/* linalg.h:37-49 (pouter2) and :61-73 (pdot2sym): {a'a, b'b, a'b} of row-major 2-column X (with Y) */
static void dot2sym(double *d, const double *X, const double *Y, int n)
{
  double aa = 0, bb = 0, ab = 0;
  for (int i = 0; i < 2 * n; i += 2) { aa += X[i] * Y[i]; bb += X[i + 1] * Y[i + 1]; ab += X[i] * Y[i + 1]; }
  d[0] = aa; d[1] = bb; d[2] = ab;
}

/* linalg.h:77-88 */
static void solve2sym(double *X, const double *A, const double *RHS)
{
  double dinv = 1.0 / (A[0] * A[1] - A[2] * A[2]);
  double i0 = dinv * A[1], i1 = dinv * A[0], i2 = -dinv * A[2];
  X[0] = i0 * RHS[0] + i2 * RHS[1];
  X[1] = i2 * RHS[0] + i1 * RHS[1];
  X[2] = i0 * RHS[2] + i2 * RHS[3];
  X[3] = i2 * RHS[2] + i1 * RHS[3];
}

/* solves (A'A + lambda I) x = b; returns the iteration count the reference reports (cg.h:25-82) */
FSO_API int fso_cg(double *x, int N, int F, const int *a_rp, const int *a_cols, const int *at_rp, const int *at_cols,
                   const double *b, double lambda, double tol)
{
  tol = tol * sqrt(normsq(b, F));
  double *r = (double *)malloc(sizeof(double) * F), *p = (double *)malloc(sizeof(double) * F);
  double *AAp = (double *)malloc(sizeof(double) * F), *tmp = (double *)malloc(sizeof(double) * (N ? N : 1));
  for (int i = 0; i < F; i++) { x[i] = 0.0; r[i] = b[i]; p[i] = b[i]; }
  double rsq_old = normsq(r, F);
  int iter;
  for (iter = 0; iter < F; iter++) {
    spmm(tmp, N, a_rp, a_cols, p, 1);
    spmm(AAp, F, at_rp, at_cols, tmp, 1);
    for (int i = 0; i < F; i++) AAp[i] += lambda * p[i];
    double alpha = rsq_old / dot(AAp, p, F);
    for (int i = 0; i < F; i++) { x[i] += alpha * p[i]; r[i] -= alpha * AAp[i]; }
    double rsq_new = normsq(r, F);
    if (sqrt(rsq_new) <= tol) break;
    double beta = rsq_new / rsq_old;
    for (int i = 0; i < F; i++) p[i] = r[i] + beta * p[i];
    rsq_old = rsq_new;
  }
  free(r); free(p); free(AAp); free(tmp);
  return iter;
}

/* two right-hand sides, row-major X and B (cg.h:85-187) */
FSO_API int fso_cg2(double *X, int N, int F, const int *a_rp, const int *a_cols, const int *at_rp, const int *at_cols,
                    const double *B, double lambda, double tol)
{
  const int F2 = 2 * F;
  const double tolsq = tol * tol;
  double norms[2], inorms[2], t3[3];
  dot2sym(t3, B, B, F);
  norms[0] = sqrt(t3[0]); norms[1] = sqrt(t3[1]);
  inorms[0] = 1.0 / norms[0]; inorms[1] = 1.0 / norms[1];
  double *R = (double *)malloc(sizeof(double) * F2), *P = (double *)malloc(sizeof(double) * F2);
  double *AAP = (double *)malloc(sizeof(double) * F2), *tmp = (double *)malloc(sizeof(double) * 2 * (N ? N : 1));
  for (int i = 0; i < F2; i += 2) {
    X[i] = 0.0; X[i + 1] = 0.0;
    R[i] = B[i] * inorms[0]; R[i + 1] = B[i + 1] * inorms[1];
    P[i] = R[i]; P[i + 1] = R[i + 1];
  }
  double RtR[3], RtR2[3], PtKP[3], Alpha[4], Psi[4];
  dot2sym(RtR, R, R, F);
  int iter;
  for (iter = 0; iter < F; iter++) {
    spmm(tmp, N, a_rp, a_cols, P, 2);
    spmm(AAP, F, at_rp, at_cols, tmp, 2);
    for (int i = 0; i < F2; i++) AAP[i] += lambda * P[i];
    dot2sym(PtKP, P, AAP, F);
    double rhs[4] = {RtR[0], RtR[2], RtR[2], RtR[1]};
    solve2sym(Alpha, PtKP, rhs);
    for (int i = 0; i < F2; i += 2) {
      X[i]     += Alpha[0] * P[i] + Alpha[1] * P[i + 1];
      X[i + 1] += Alpha[2] * P[i] + Alpha[3] * P[i + 1];
      R[i]     -= Alpha[0] * AAP[i] + Alpha[1] * AAP[i + 1];
      R[i + 1] -= Alpha[2] * AAP[i] + Alpha[3] * AAP[i + 1];
    }
    dot2sym(RtR2, R, R, F);
    if (RtR2[0] <= tolsq && RtR2[1] <= tolsq) break;
    double rhs_psi[4] = {RtR2[0], RtR2[2], RtR2[2], RtR2[1]};
    solve2sym(Psi, RtR, rhs_psi);
    for (int i = 0; i < F2; i += 2) {
      double a = P[i], c = P[i + 1];
      P[i]     = R[i]     + Psi[0] * a + Psi[1] * c;
      P[i + 1] = R[i + 1] + Psi[2] * a + Psi[3] * c;
    }
    RtR[0] = RtR2[0]; RtR[1] = RtR2[1]; RtR[2] = RtR2[2];
  }
  for (int i = 0; i < F2; i += 2) { X[i] *= norms[0]; X[i + 1] *= norms[1]; }
  free(R); free(P); free(AAP); free(tmp);
  return iter;
}
