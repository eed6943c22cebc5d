/*
 * fs_oracle.c -- TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT.
 *
 * CPU restatement of the A_mul_B / At_mul_B hot path of jaak-s/libfastsparse,
 * written from scratch over flat arrays (no reference structs) so that Python
 * tests can drive it through ctypes.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; nothing under
 * libfastsparse_amd/ or include/ links, loads or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against
 * the real reference compiled from /root/reference (oracle/_ref/libfsref.so,
 * recipe in oracle/Makefile) by tests/test_oracle_vs_ref.py, and against the
 * committed golden vectors in tests/golden/ (generated from that same
 * reference build by tests/golden/make_golden.py) by tests/test_oracle_golden.py.
 *
 * Arithmetic contract (strict build: -O2 -ffp-contract=off, no -ffast-math):
 *   every output element is the left-to-right sum, in storage order, of its
 *   terms; a valued term is one rounded multiply x*v followed by one rounded
 *   add (no FMA).  This is what the reference computes when built without
 *   -ffast-math; see SURVEY.md section 8(a) notes N1/N2.
 *
 * All loops use 64-bit counters (the reference uses int over long nnz, e.g.
 * csr.h:44); results are identical for every size the reference can hold.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define FSO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* Format construction                                                 */
/* ------------------------------------------------------------------ */

/* COO -> CSR, stable in input order within each row; columns are neither
 * sorted nor merged.  vals/out_vals may be NULL (pattern-only matrix).
 * Follows new_bcsr (csr.h:30-67) and new_csr (csr.h:375-422): histogram of
 * rows, exclusive prefix sum, stable scatter. */
FSO_API void fso_coo_to_csr(int64_t nnz, int nrow,
                            const int *rows, const int *cols, const double *vals,
                            int *row_ptr, int *out_cols, double *out_vals)
{
  int64_t *cursor = (int64_t *)calloc((size_t)nrow + 1, sizeof(int64_t));
  for (int64_t k = 0; k < nnz; k++) cursor[rows[k] + 1]++;
  for (int r = 0; r < nrow; r++) cursor[r + 1] += cursor[r];
  for (int r = 0; r <= nrow; r++) row_ptr[r] = (int)cursor[r];
  for (int64_t k = 0; k < nnz; k++) {
    int64_t slot = cursor[rows[k]]++;
    out_cols[slot] = cols[k];
    if (vals) out_vals[slot] = vals[k];
  }
  free(cursor);
}

/* number of row blocks: ceil(nrow / block_size) as computed at sparse.h:179 */
FSO_API int fso_num_blocks(int n, int block_size)
{
  return (int)ceil(n / (double)block_size);
}

/* COO -> row-blocked COO (BlockedSBM sparse.h:175-213, BlockedSDM
 * dsparse.h:132-173).  The per-block arrays of the reference are laid out
 * back to back here: block b owns [blk_off[b], blk_off[b+1]) of
 * out_rows/out_cols/out_vals.  Entry order inside a block = input order. */
FSO_API void fso_coo_to_blocked(int64_t nnz, int nrow, int block_size,
                                const int *rows, const int *cols, const double *vals,
                                int *start_row, int *blk_nnz, int64_t *blk_off,
                                int *out_rows, int *out_cols, double *out_vals)
{
  int nb = fso_num_blocks(nrow, block_size);
  for (int b = 0; b < nb; b++) { start_row[b] = b * block_size; blk_nnz[b] = 0; }
  start_row[nb] = nrow;
  for (int64_t k = 0; k < nnz; k++) blk_nnz[rows[k] / block_size]++;
  blk_off[0] = 0;
  for (int b = 0; b < nb; b++) blk_off[b + 1] = blk_off[b] + blk_nnz[b];
  int64_t *cursor = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nb > 0 ? nb : 1));
  for (int b = 0; b < nb; b++) cursor[b] = blk_off[b];
  for (int64_t k = 0; k < nnz; k++) {
    int64_t slot = cursor[rows[k] / block_size]++;
    out_rows[slot] = rows[k];
    out_cols[slot] = cols[k];
    if (vals) out_vals[slot] = vals[k];
  }
  free(cursor);
}

/* COO -> column-blocked binary CSR (new_cbcsr, cbcsr.h:16-65).  Cell index is
 * block*nrow + row; row_ptr has nblocks*nrow + 1 entries. */
FSO_API void fso_coo_to_cbcsr(int colblocksize, int64_t nnz, int nrow, int ncol,
                              const int *rows, const int *cols,
                              int *row_ptr, int *out_cols)
{
  int nb = fso_num_blocks(ncol, colblocksize);
  int64_t ncell = (int64_t)nb * nrow;
  int64_t *cursor = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
  for (int64_t k = 0; k < nnz; k++)
    cursor[(int64_t)(cols[k] / colblocksize) * nrow + rows[k] + 1]++;
  for (int64_t c = 0; c < ncell; c++) cursor[c + 1] += cursor[c];
  for (int64_t c = 0; c <= ncell; c++) row_ptr[c] = (int)cursor[c];
  for (int64_t k = 0; k < nnz; k++) {
    int64_t cell = (int64_t)(cols[k] / colblocksize) * nrow + rows[k];
    out_cols[cursor[cell]++] = cols[k];
  }
  free(cursor);
}

/* ------------------------------------------------------------------ */
/* CSR kernels                                                         */
/* ------------------------------------------------------------------ */

/* y = A x.  vals == NULL: bcsr_A_mul_B (csr.h:149-161); else csr_A_mul_B
 * (csr.h:425-438).  y is overwritten; empty rows give +0.0.  Rows are
 * independent, so the OpenMP split (same schedule as the reference,
 * csr.h:152/429) does not change any bit of the result. */
FSO_API void fso_csr_mul(double *y, int nrow, const int *row_ptr, const int *cols,
                         const double *vals, const double *x)
{
#pragma omp parallel for schedule(dynamic, 256)
  for (int r = 0; r < nrow; r++) {
    double acc = 0;
    int64_t e = row_ptr[r + 1];
    if (vals) {
      for (int64_t i = row_ptr[r]; i < e; i++) acc += x[cols[i]] * vals[i];
    } else {
      for (int64_t i = row_ptr[r]; i < e; i++) acc += x[cols[i]];
    }
    y[r] = acc;
  }
}

/* Y = A X with k row-major right-hand sides.  vals == NULL covers
 * bcsr_A_mul_B2/_B4/_B8/_B8_auto/_Bn/_B32n (csr.h:164-302), all of which
 * compute the same per-(row, column) left-to-right sum; else csr_A_mul_Bn
 * (csr.h:441-465).  One parallel-for over rows: the nested-parallel
 * redundancy of csr.h:260-263 / 445-448 (SURVEY note N4) is not restated --
 * it repeats identical writes and changes no value. */
FSO_API void fso_csr_mul_n(double *Y, int nrow, const int *row_ptr, const int *cols,
                           const double *vals, const double *X, int k)
{
#pragma omp parallel
  {
    double *acc = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
#pragma omp for schedule(dynamic, 256)
    for (int r = 0; r < nrow; r++) {
      for (int j = 0; j < k; j++) acc[j] = 0;
      int64_t e = row_ptr[r + 1];
      for (int64_t i = row_ptr[r]; i < e; i++) {
        const double *xr = X + (int64_t)cols[i] * k;
        if (vals) {
          double v = vals[i];
          for (int j = 0; j < k; j++) acc[j] += xr[j] * v;
        } else {
          for (int j = 0; j < k; j++) acc[j] += xr[j];
        }
      }
      double *yr = Y + (int64_t)r * k;
      for (int j = 0; j < k; j++) yr[j] = acc[j];
    }
    free(acc);
  }
}

/* The same product with the reference's own OpenMP structure (csr.h:445-448 /
 * 260-263): an `omp parallel for` nested inside an `omp parallel` region.  With
 * nested parallelism off (libgomp's default) the inner region gets a team of
 * one, so EVERY thread of the outer team walks all rows and writes the same
 * values (SURVEY note N4): the results equal fso_csr_mul_n's, the time is what
 * the reference costs as shipped.  Used only by bench.py's cpu_baseline leg,
 * which reports both schedules, labelled. */
FSO_API void fso_csr_mul_n_reference_schedule(double *Y, int nrow, const int *row_ptr, const int *cols,
                                              const double *vals, const double *X, int k)
{
#pragma omp parallel
  {
    double *acc = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
#pragma omp parallel for schedule(dynamic, 256)
    for (int r = 0; r < nrow; r++) {
      for (int j = 0; j < k; j++) acc[j] = 0;
      int64_t e = row_ptr[r + 1];
      for (int64_t i = row_ptr[r]; i < e; i++) {
        const double *xr = X + (int64_t)cols[i] * k;
        if (vals) {
          double v = vals[i];
          for (int j = 0; j < k; j++) acc[j] += xr[j] * v;
        } else {
          for (int j = 0; j < k; j++) acc[j] += xr[j];
        }
      }
      double *yr = Y + (int64_t)r * k;
      for (int j = 0; j < k; j++) yr[j] = acc[j];
    }
    free(acc);
  }
}

/* y = A'A x on a binary CSR, serial (bcsr_AA_mul_B, csr.h:305-319).  With one
 * thread parallel_bcsr_AA_mul_B (csr.h:323-355) performs the same additions in
 * the same order (its single ytmp replica is summed onto 0.0). */
FSO_API void fso_bcsr_aa_mul(double *y, int nrow, int ncol, const int *row_ptr,
                             const int *cols, const double *x)
{
  memset(y, 0, sizeof(double) * (size_t)ncol);
  for (int r = 0; r < nrow; r++) {
    double s = 0;
    for (int64_t i = row_ptr[r]; i < row_ptr[r + 1]; i++) s += x[cols[i]];
    for (int64_t i = row_ptr[r]; i < row_ptr[r + 1]; i++) y[cols[i]] += s;
  }
}

/* ------------------------------------------------------------------ */
/* COO kernels (serial in the reference)                               */
/* ------------------------------------------------------------------ */

/* y = A x on COO.  vals == NULL: A_mul_B (sparse.h:58-65); else sdm_A_mul_B
 * (dsparse.h:43-51).  y zeroed first, duplicates accumulate. */
FSO_API void fso_coo_mul(double *y, int nrow, int64_t nnz, const int *rows,
                         const int *cols, const double *vals, const double *x)
{
  memset(y, 0, sizeof(double) * (size_t)nrow);
  if (vals) for (int64_t j = 0; j < nnz; j++) y[rows[j]] += x[cols[j]] * vals[j];
  else      for (int64_t j = 0; j < nnz; j++) y[rows[j]] += x[cols[j]];
}

/* y = A' x on COO.  vals == NULL: At_mul_B (sparse.h:68-75); else
 * sdm_At_mul_B (dsparse.h:54-62). */
FSO_API void fso_coo_tmul(double *y, int ncol, int64_t nnz, const int *rows,
                          const int *cols, const double *vals, const double *x)
{
  memset(y, 0, sizeof(double) * (size_t)ncol);
  if (vals) for (int64_t j = 0; j < nnz; j++) y[cols[j]] += x[rows[j]] * vals[j];
  else      for (int64_t j = 0; j < nnz; j++) y[cols[j]] += x[rows[j]];
}

/* Y = B X on a row-blocked COO with k row-major right-hand sides.
 * vals == NULL: bsbm_A_mul_B/_B2/_B4/_Bn (sparse.h:259-336); vals != NULL and
 * k == 1: bsdm_A_mul_B (dsparse.h:176-191).  Each block zeroes its own slice
 * of Y and scatter-adds its entries in storage order; blocks touch disjoint
 * rows, so the parallel-for over blocks (sparse.h:260) is bit-neutral. */
FSO_API void fso_blocked_mul_n(double *Y, int nblocks, const int *start_row,
                               const int *blk_nnz, const int64_t *blk_off,
                               const int *rows, const int *cols, const double *vals,
                               const double *X, int k)
{
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < nblocks; b++) {
    memset(Y + (int64_t)k * start_row[b], 0,
           sizeof(double) * (size_t)k * (size_t)(start_row[b + 1] - start_row[b]));
    int64_t o = blk_off[b];
    for (int64_t j = o; j < o + blk_nnz[b]; j++) {
      double *yr = Y + (int64_t)rows[j] * k;
      const double *xr = X + (int64_t)cols[j] * k;
      if (vals) { double v = vals[j]; for (int c = 0; c < k; c++) yr[c] += xr[c] * v; }
      else      {                      for (int c = 0; c < k; c++) yr[c] += xr[c]; }
    }
  }
}

/* ------------------------------------------------------------------ */
/* Column-blocked binary CSR                                           */
/* ------------------------------------------------------------------ */

/* y = A x on ColBinaryCSR (cbcsr_A_mul_B, cbcsr.h:76-106), in the order a
 * single thread executes it: ytmp[row] accumulates the cell sums block by
 * block (cell sum first, then ytmp[row] += cell sum), and y[row] = 0 + ytmp[row].
 * With more threads the reference's cross-thread order is schedule dependent
 * (SURVEY 8a row a19); this is its deterministic representative. */
FSO_API void fso_cbcsr_mul(double *y, int nrow, int nblocks, const int *row_ptr,
                           const int *cols, const double *x)
{
#pragma omp parallel for schedule(static)
  for (int r = 0; r < nrow; r++) {
    double tot = 0;
    for (int b = 0; b < nblocks; b++) {
      int64_t cell = (int64_t)b * nrow + r;
      double s = 0;
      for (int64_t i = row_ptr[cell]; i < row_ptr[cell + 1]; i++) s += x[cols[i]];
      tot += s;
    }
    y[r] = 0.0 + tot;
  }
}

/* ------------------------------------------------------------------ */
/* Error bound helper used by the fp64 tolerance tests:    */
/* scale[r] = sum_i |a_ri| |x_ci| for a CSR (vals NULL -> |a| = 1).    */
/* ------------------------------------------------------------------ */
FSO_API void fso_csr_abs_scale(double *s, int nrow, const int *row_ptr, const int *cols,
                               const double *vals, const double *x)
{
#pragma omp parallel for schedule(static)
  for (int r = 0; r < nrow; r++) {
    double acc = 0;
    for (int64_t i = row_ptr[r]; i < row_ptr[r + 1]; i++)
      acc += fabs(x[cols[i]]) * (vals ? fabs(vals[i]) : 1.0);
    s[r] = acc;
  }
}

FSO_API int fso_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
