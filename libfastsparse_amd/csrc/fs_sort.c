/*
 * fs_sort.c -- the reference's locality re-orderings behind their original names: Hilbert-curve
 * order of a COO (sort_sbm sparse.h:142, sort_sdm dsparse.h:96), per-row-block Hilbert order (sort_bsbm
 * sparse.h:215, sort_bsdm dsparse.h:193), row-major order inside blocks (sort_bsbm_byrow sparse.h:238) and the
 * curve helpers of hilbert.h.  Host code: these only permute the entries of a matrix (results of every product are
 * unchanged up to rounding, test_sparse.c:275-280).  On the GPU the locality they were written for is provided by
 * the L2-tiled device copy (row panels x column bands, fs_format.hip); they are here so that callers of the
 * reference find the same API, and each one drops the matrix's cached device copy because it changes the arrays.
 *
 * Implementation: one generic "order entries by a 64-bit key" routine (LSD radix sort of (key, position) pairs,
 * stable) instead of the reference's encode / quicksort / decode round trip; the resulting entry order is the same
 * whenever (row, col) pairs are distinct, and equal pairs are interchangeable.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dsparse.h"
#include "fastsparse_hip.h"
#include "hilbert.h"
#include "sparse.h"

#define FS_EXPORT __attribute__((visibility("default")))

/* ---- hilbert.h ---------------------------------------------------------------------------------------------- */
FS_EXPORT int ceilPower2(int x) { return 1 << (int)ceil(log2(x)); }      /* hilbert.h:11 */

/* quadrant fix-up of the curve at scale s (hilbert.h:45): reflect when (rx, ry) = (1, 0), transpose when ry = 0 */
FS_EXPORT void rot(int s, int *x, int *y, int rx, int ry)
{
  if (ry) return;
  if (rx) { *x = s - 1 - *x; *y = s - 1 - *y; }
  int t = *x; *x = *y; *y = t;
}

/* (x, y) in an n x n grid (n a power of two) -> position along the curve (hilbert.h:16) */
FS_EXPORT long xy2d(int n, int x, int y)
{
  long d = 0;
  for (long s = n / 2; s > 0; s /= 2) {
    const int rx = (x & s) != 0, ry = (y & s) != 0;
    d += s * s * ((3 * rx) ^ ry);
    rot((int)s, &x, &y, rx, ry);
  }
  return d;
}

/* position along the curve -> (x, y) (hilbert.h:30) */
FS_EXPORT void d2xy(int n, long d, int *x, int *y)
{
  *x = *y = 0;
  for (int s = 1; s < n; s *= 2) {
    const int rx = (int)(1 & (d / 2)), ry = (int)(1 & (d ^ rx));
    rot(s, x, y, rx, ry);
    *x += s * rx;
    *y += s * ry;
    d /= 4;
  }
}

/* curve over a row block of n rows and any number of columns: n x n squares laid side by side along the columns,
 * (row, col) swapped inside a square (hilbert.h:60, :68) */
FS_EXPORT long row_xy2d(int n, int x, int y)
{
  return xy2d(n, y % n, x) + (long)n * (long)n * (y / n);
}

FS_EXPORT void row_d2xy(int n, long d, int *x, int *y)
{
  const long nsq = (long)n * (long)n;
  d2xy(n, d % nsq, y, x);
  *y += (int)(d / nsq) * n;
}

/* ---- order entries by key -------------------------------------------------------------------------------------- */
/* returns perm with perm[i] = position of the entry that comes i-th when sorted by key; stable LSD radix sort,
 * 16 bits per pass, only over the bits that are set in some key.  The caller frees perm. */
static long *order_by_key(const uint64_t *keys_in, long n)
{
  const size_t m = (size_t)(n ? n : 1);
  uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * m), *kb = (uint64_t *)malloc(sizeof(uint64_t) * m);
  long *pa = (long *)malloc(sizeof(long) * m), *pb = (long *)malloc(sizeof(long) * m);
  size_t *count = (size_t *)malloc(sizeof(size_t) * 65537);
  uint64_t all = 0;
  for (long i = 0; i < n; i++) { ka[i] = keys_in[i]; pa[i] = i; all |= keys_in[i]; }
  for (int shift = 0; shift < 64 && (all >> shift); shift += 16) {
    memset(count, 0, sizeof(size_t) * 65537);
    for (long i = 0; i < n; i++) count[((ka[i] >> shift) & 0xFFFF) + 1]++;
    for (int b = 0; b < 65536; b++) count[b + 1] += count[b];
    for (long i = 0; i < n; i++) {
      const size_t dst = count[(ka[i] >> shift) & 0xFFFF]++;
      kb[dst] = ka[i];
      pb[dst] = pa[i];
    }
    uint64_t *tk = ka; ka = kb; kb = tk;
    long *tp = pa; pa = pb; pb = tp;
  }
  free(ka); free(kb); free(pb); free(count);
  return pa;
}

static void apply_int(int *a, const long *perm, long n)
{
  int *t = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
  for (long i = 0; i < n; i++) t[i] = a[perm[i]];
  memcpy(a, t, sizeof(int) * (size_t)n);
  free(t);
}

static void apply_double(double *a, const long *perm, long n)
{
  double *t = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  for (long i = 0; i < n; i++) t[i] = a[perm[i]];
  memcpy(a, t, sizeof(double) * (size_t)n);
  free(t);
}

enum { KEY_HILBERT, KEY_ROW_HILBERT, KEY_ROW_MAJOR };

/* permute (rows, cols[, vals]) into the order of the chosen key */
static void reorder(int kind, int n, int start_row, long ncol, long nnz, int *rows, int *cols, double *vals)
{
  if (nnz <= 1) return;
  uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nnz);
  for (long j = 0; j < nnz; j++) {
    if (kind == KEY_HILBERT) keys[j] = (uint64_t)xy2d(n, rows[j], cols[j]);
    else if (kind == KEY_ROW_HILBERT) keys[j] = (uint64_t)row_xy2d(n, rows[j] - start_row, cols[j]);
    else keys[j] = (uint64_t)rows[j] * (uint64_t)ncol + (uint64_t)cols[j];
  }
  long *perm = order_by_key(keys, nnz);
  apply_int(rows, perm, nnz);
  apply_int(cols, perm, nnz);
  if (vals) apply_double(vals, perm, nnz);
  free(perm);
  free(keys);
}

static int grid_side(int nrow, int ncol) { return ceilPower2(nrow > ncol ? nrow : ncol); }

/* ---- sparse.h ------------------------------------------------------------------------------------------------ */
FS_EXPORT void sort_sbm(struct SparseBinaryMatrix *A)                     /* sparse.h:142 */
{
  reorder(KEY_HILBERT, grid_side(A->nrow, A->ncol), 0, A->ncol, A->nnz, A->rows, A->cols, NULL);
  fs_invalidate(A);
}

FS_EXPORT void sort_bsbm(struct BlockedSBM *B)                            /* sparse.h:215 */
{
  for (int b = 0; b < B->nblocks; b++)
    reorder(KEY_ROW_HILBERT, ceilPower2(B->start_row[b + 1] - B->start_row[b]), B->start_row[b], B->ncol, B->nnz[b],
            B->rows[b], B->cols[b], NULL);
  fs_invalidate(B);
}

FS_EXPORT void sort_bsbm_byrow(struct BlockedSBM *B)                      /* sparse.h:238 */
{
  for (int b = 0; b < B->nblocks; b++) reorder(KEY_ROW_MAJOR, 0, 0, B->ncol, B->nnz[b], B->rows[b], B->cols[b], NULL);
  fs_invalidate(B);
}

/* ---- dsparse.h ----------------------------------------------------------------------------------------------- */
FS_EXPORT void sort_sdm(struct SparseDoubleMatrix *A)                     /* dsparse.h:96 */
{
  reorder(KEY_HILBERT, grid_side(A->nrow, A->ncol), 0, A->ncol, A->nnz, A->rows, A->cols, A->vals);
  fs_invalidate(A);
}

FS_EXPORT void sort_bsdm(struct BlockedSDM *B)                            /* dsparse.h:193 */
{
  for (int b = 0; b < B->nblocks; b++)
    reorder(KEY_ROW_HILBERT, ceilPower2(B->start_row[b + 1] - B->start_row[b]), B->start_row[b], B->ncol, B->nnz[b],
            B->rows[b], B->cols[b], B->vals[b]);
  fs_invalidate(B);
}

/* ---- quickSort.h / quickSortD.h: ascending sort of a[l..r] (inclusive), optionally carrying v along -------------- */
FS_EXPORT void quickSortD(long a[], long l, long r, double *v)
{
  const long n = r - l + 1;
  if (n <= 1) return;
  /* keys may be negative in principle: flip the sign bit so that unsigned order == signed order */
  uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  for (long i = 0; i < n; i++) keys[i] = (uint64_t)a[l + i] ^ 0x8000000000000000ull;
  long *perm = order_by_key(keys, n);
  long *ta = (long *)malloc(sizeof(long) * (size_t)n);
  for (long i = 0; i < n; i++) ta[i] = a[l + perm[i]];
  memcpy(a + l, ta, sizeof(long) * (size_t)n);
  if (v) apply_double(v + l, perm, n);
  free(ta); free(perm); free(keys);
}

FS_EXPORT void quickSort(long a[], long l, long r) { quickSortD(a, l, r, NULL); }
