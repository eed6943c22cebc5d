/* A plain C caller of the multi-GPU C-ABI (include/fastsparse_hip.h), as INTEGRATION.md section 3 shows it: a matrix that only
 * exists as per-rank shards (local row_ptr, global columns) goes in through fs_dist_csr_create_from_shards, A' is built on the
 * devices, y = A x and z = A' y are checked against the serial loops of the reference restated here (csr.h:430-437,
 * dsparse.h:54-62; integer-valued data, so every order of additions gives the same bits), then (A'A + lambda I) s = b is solved
 * across the ranks and its residual is recomputed with the same loops.  Usage: dist_shards_caller <ranks> (device 0 listed <ranks>
 * times: virtual ranks).  TEST CODE: compiled by tests/test_gpu_parity.py with gcc, never part of the library. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fastsparse_hip.h"

static uint64_t mix(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, fs_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
  const int ranks = argc > 1 ? atoi(argv[1]) : 3;
  const int nrow = 90000, ncol = 70000;
  int *devs = calloc((size_t)ranks, sizeof(int));                  /* device 0, `ranks` times */
  fs_dist_t D = fs_dist_create(ranks, devs);
  CHECK(D != NULL);
  /* rows of 0 .. 40 entries, a few of 3000; the shards are cut by rows here (the caller's choice) */
  int *len = malloc(sizeof(int) * (size_t)nrow);
  int64_t nnz = 0;
  for (int r = 0; r < nrow; r++) { len[r] = (int)(mix((uint64_t)r) % 41); if (r % 9973 == 5) len[r] = 3000; nnz += len[r]; }
  int *rows_of = calloc((size_t)ranks + 1, sizeof(int));
  for (int k = 1; k <= ranks; k++) rows_of[k] = (int)((int64_t)nrow * k / ranks);
  int **rp = malloc(sizeof(int *) * (size_t)ranks), **cc = malloc(sizeof(int *) * (size_t)ranks);
  double **vv = malloc(sizeof(double *) * (size_t)ranks);
  int *srows = malloc(sizeof(int) * (size_t)ranks);
  int64_t *snnz = malloc(sizeof(int64_t) * (size_t)ranks);
  for (int k = 0; k < ranks; k++) {
    const int lo = rows_of[k], hi = rows_of[k + 1];
    int64_t n = 0;
    for (int r = lo; r < hi; r++) n += len[r];
    srows[k] = hi - lo; snnz[k] = n;
    rp[k] = malloc(sizeof(int) * (size_t)(hi - lo + 1));
    cc[k] = malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    vv[k] = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int64_t at = 0;
    for (int r = lo; r < hi; r++) {
      rp[k][r - lo] = (int)at;
      for (int j = 0; j < len[r]; j++, at++) {
        const uint64_t h = mix(((uint64_t)r << 20) ^ (uint64_t)j);
        cc[k][at] = (int)(h % (uint64_t)ncol);
        vv[k][at] = (double)((int)((h >> 40) % 7) - 3);               /* integer values -3 .. 3 */
      }
    }
    rp[k][hi - lo] = (int)at;
  }
  fs_dist_matrix_t M = fs_dist_csr_create_from_shards(D, nrow, ncol, srows, snnz, (const int *const *)rp, (const int *const *)cc,
                                                      (const double *const *)vv, FS_HOST);
  CHECK(M != NULL);
  CHECK(fs_dist_matrix_nnz(M) == nnz);
  CHECK(fs_dist_matrix_build_transpose_device(M) == FS_OK);
  double *x = malloc(sizeof(double) * (size_t)ncol), *y = malloc(sizeof(double) * (size_t)nrow), *z = malloc(sizeof(double) * (size_t)ncol);
  double *yr = calloc((size_t)nrow, sizeof(double)), *zr = calloc((size_t)ncol, sizeof(double));
  for (int c = 0; c < ncol; c++) x[c] = (double)((int)(mix(77u + (uint64_t)c) % 21) - 10);
  CHECK(fs_dist_spmv(M, y, x) == FS_OK);
  for (int k = 0; k < ranks; k++)
    for (int r = 0; r < srows[k]; r++) {
      double t = 0.0;
      for (int i = rp[k][r]; i < rp[k][r + 1]; i++) t += x[cc[k][i]] * vv[k][i];
      yr[rows_of[k] + r] = t;
    }
  for (int r = 0; r < nrow; r++) CHECK(y[r] == yr[r]);
  for (int r = 0; r < nrow; r++) y[r] = fmod(y[r], 16.0);            /* keep the integers small for the second product */
  CHECK(fs_dist_spmv_t(M, z, y) == FS_OK);
  for (int k = 0; k < ranks; k++)
    for (int r = 0; r < srows[k]; r++)
      for (int i = rp[k][r]; i < rp[k][r + 1]; i++) zr[cc[k][i]] += y[rows_of[k] + r] * vv[k][i];
  for (int c = 0; c < ncol; c++) CHECK(z[c] == zr[c]);
  /* bsbm_cg across the ranks (cg.h:25-82; valued here, which the C-ABI allows) */
  double *b = malloc(sizeof(double) * (size_t)ncol), *s = malloc(sizeof(double) * (size_t)ncol), *t = malloc(sizeof(double) * (size_t)nrow);
  for (int c = 0; c < ncol; c++) b[c] = sin(0.37 * c + 1.0);
  int iters = -1;
  const double lambda = 500.0, tol = 1e-9;
  CHECK(fs_dist_cg(M, s, b, lambda, tol, &iters) == FS_OK);
  double rr = 0.0, bb = 0.0;
  for (int r = 0; r < nrow; r++) t[r] = 0.0;
  for (int k = 0; k < ranks; k++)
    for (int r = 0; r < srows[k]; r++)
      for (int i = rp[k][r]; i < rp[k][r + 1]; i++) t[rows_of[k] + r] += s[cc[k][i]] * vv[k][i];
  for (int c = 0; c < ncol; c++) zr[c] = lambda * s[c];
  for (int k = 0; k < ranks; k++)
    for (int r = 0; r < srows[k]; r++)
      for (int i = rp[k][r]; i < rp[k][r + 1]; i++) zr[cc[k][i]] += t[rows_of[k] + r] * vv[k][i];
  for (int c = 0; c < ncol; c++) { rr += (b[c] - zr[c]) * (b[c] - zr[c]); bb += b[c] * b[c]; }
  CHECK(iters > 0 && sqrt(rr / bb) <= 2.0 * tol);
  printf("OK ranks=%d nnz=%lld cg_iterations=%d relative_residual=%.3g conservative=%d\n", ranks, (long long)nnz, iters, sqrt(rr / bb),
         fs_dist_is_conservative(D));
  fs_dist_matrix_destroy(M);
  fs_dist_destroy(D);
  return 0;
}
