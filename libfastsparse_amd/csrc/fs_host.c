/*
 * fs_host.c -- host-side containers and format builders behind the reference's constructor
 * names (include/sparse.h, dsparse.h, csr.h, cbcsr.h).  Plain C, no GPU: these run once per
 * matrix; the products they feed run on the device (fs_dropin.hip -> fs_kernels.hip).
 *
 * Written from the behaviour documented in SURVEY.md 8(a) (rows a2, a6, a11, a14, a16, a18,
 * a19), not from the reference's code: all counters are 64-bit, builders share one bucket
 * routine, and a failed allocation is reported instead of dereferenced.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

#include "cbcsr.h"
#include "csr.h"
#include "dsparse.h"
#include "sparse.h"

#define FS_EXPORT __attribute__((visibility("default")))

/* side table of device copies (fs_dropin.hip): a struct whose arrays go away or change roles drops its copy */
void fs_invalidate(const void *host_struct);
/* format construction on the device (fs_format.hip, include/fastsparse_hip.h) */
int fs_bucket_coo(int kind, int param, int nrow, int ncol, int64_t nbuckets, int64_t nnz, const int *rows, const int *cols,
                  const double *vals, int *offsets, int *rows_out, int *cols_out, double *vals_out);
int fs_device_build_wanted(int64_t nnz);
const char *fs_last_error(void);

static void device_build_failed(const char *who)
{
  fprintf(stderr, "libfastsparse_hip: %s: building the format on the device failed: %s\n", who, fs_last_error());
  exit(1);
}

static void *xmalloc(size_t bytes)
{
  void *p = malloc(bytes ? bytes : 1);
  if (!p) {
    fprintf(stderr, "libfastsparse_hip: out of host memory (%zu bytes)\n", bytes);
    exit(1);
  }
  return p;
}

static int blocks_for(int n, int block_size) { return (int)ceil(n / (double)block_size); }

/* Stable bucketing shared by every builder: entry k goes to bucket key(k); on return
 * offsets[b] is the first slot of bucket b (nbuckets + 1 values) and slot_of[k] the
 * slot of entry k.  Entries of one bucket keep their input order. */
static int64_t *bucket_slots(int64_t nnz, int64_t nbuckets, const int64_t *keys, int64_t *offsets)
{
  int64_t *slot_of = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)nnz);
  memset(offsets, 0, sizeof(int64_t) * ((size_t)nbuckets + 1));
  for (int64_t k = 0; k < nnz; k++) offsets[keys[k] + 1]++;
  for (int64_t b = 0; b < nbuckets; b++) offsets[b + 1] += offsets[b];
  int64_t *next = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)(nbuckets > 0 ? nbuckets : 1));
  memcpy(next, offsets, sizeof(int64_t) * (size_t)nbuckets);
  for (int64_t k = 0; k < nnz; k++) slot_of[k] = next[keys[k]]++;
  free(next);
  return slot_of;
}

/* ---- COO containers (sparse.h:21-55, dsparse.h:22-40 of the reference) ------------------- */
FS_EXPORT struct SparseBinaryMatrix *new_sbm(long nrow, long ncol, long nnz, int *rows, int *cols)
{
  struct SparseBinaryMatrix *A = (struct SparseBinaryMatrix *)xmalloc(sizeof *A);
  A->nrow = (int)nrow; A->ncol = (int)ncol; A->nnz = nnz; A->rows = rows; A->cols = cols;
  return A;
}

FS_EXPORT void free_sbm(struct SparseBinaryMatrix *sbm)
{
  fs_invalidate(sbm);   /* both device copies: the A_mul_B handle and the At_mul_B handle */
  free(sbm->rows);
  free(sbm->cols);
}

FS_EXPORT struct SparseBinaryMatrix *new_transpose(struct SparseBinaryMatrix *A)
{
  return new_sbm(A->ncol, A->nrow, A->nnz, A->cols, A->rows); /* aliases A's arrays */
}

FS_EXPORT void transpose(struct SparseBinaryMatrix *A)
{
  fs_invalidate(A);
  int *r = A->rows; A->rows = A->cols; A->cols = r;
  int n = A->nrow; A->nrow = A->ncol; A->ncol = n;
}

FS_EXPORT struct SparseDoubleMatrix *new_sdm(long nrow, long ncol, long nnz, int *rows, int *cols, double *vals)
{
  struct SparseDoubleMatrix *A = (struct SparseDoubleMatrix *)xmalloc(sizeof *A);
  A->nrow = (int)nrow; A->ncol = (int)ncol; A->nnz = nnz; A->rows = rows; A->cols = cols; A->vals = vals;
  return A;
}

FS_EXPORT void sdm_transpose(struct SparseDoubleMatrix *A)
{
  fs_invalidate(A);
  int *r = A->rows; A->rows = A->cols; A->cols = r;
  int n = A->nrow; A->nrow = A->ncol; A->ncol = n;
}

/* ---- small helpers of sparse.h / timing.h / omp_util.h (host, off the product path) ----------
 * The samplers draw from drand48 exactly as the reference's do (sparse.h:77-110), so a caller that
 * seeds with srand48 sees the same sequence. */
FS_EXPORT double exprand(void) { return log1p(1.0 - drand48()); }
FS_EXPORT double randexp(void) { return -log(1.0 - drand48()); }

FS_EXPORT long randsubseq(long N, long max_samples, double p, long *samples)
{
  /* geometric gaps between kept indices: gap = ceil(Exp(1) * scale), scale = -1/log(1-p) */
  const double scale = -1.0 / log1p(-p);
  long last = -1, count = 0;
  for (;;) {
    double gap = randexp() * scale;
    if (gap + last >= N - 1) break;
    last += (long)ceil(gap);
    samples[count++] = last;
    if (count >= max_samples) break;
  }
  return count;
}

FS_EXPORT void timing(double *wcTime, double *cpuTime)
{
  struct timeval now;
  struct rusage use;
  gettimeofday(&now, NULL);
  getrusage(RUSAGE_SELF, &use);
  *wcTime = now.tv_sec + now.tv_usec * 1e-6;
  *cpuTime = use.ru_utime.tv_sec + use.ru_utime.tv_usec * 1e-6;
}

/* the products run on the GPU: the calling thread is the only host thread the library uses */
FS_EXPORT int thread_num(void) { return 0; }
FS_EXPORT int nthreads(void) { return 1; }
FS_EXPORT int thread_limit(void) { return 1; }
FS_EXPORT void threads_init(void) { }

/* ---- fixture files: 3 x int64 header, int32 rows, int32 cols, [float64 vals], 1-based ------ */
FS_EXPORT long read_long(FILE *fh)
{
  long v;
  if (fread(&v, sizeof v, 1, fh) != 1) {
    fprintf(stderr, "File reading error for a long. File is corrupt.\n");
    exit(1);
  }
  return v;
}

static void read_coo(const char *filename, long *nrow, long *ncol, long *nnz, int **rows, int **cols, double **vals)
{
  FILE *fh = fopen(filename, "rb");
  if (!fh) {
    fprintf(stderr, "File error: %s\n", filename);
    exit(1);
  }
  *nrow = read_long(fh); *ncol = read_long(fh); *nnz = read_long(fh);
  size_t n = (size_t)*nnz;
  *rows = (int *)xmalloc(sizeof(int) * n);
  *cols = (int *)xmalloc(sizeof(int) * n);
  int ok = fread(*rows, sizeof(int), n, fh) == n && fread(*cols, sizeof(int), n, fh) == n;
  if (ok && vals) {
    *vals = (double *)xmalloc(sizeof(double) * n);
    ok = fread(*vals, sizeof(double), n, fh) == n;
  }
  fclose(fh);
  if (!ok) {
    fprintf(stderr, "File read error: %s\n", filename);
    exit(1);
  }
  for (size_t i = 0; i < n; i++) { (*rows)[i]--; (*cols)[i]--; }
}

FS_EXPORT struct SparseBinaryMatrix *read_sbm(const char *filename)
{
  long nrow, ncol, nnz; int *rows, *cols;
  read_coo(filename, &nrow, &ncol, &nnz, &rows, &cols, NULL);
  return new_sbm(nrow, ncol, nnz, rows, cols);
}

FS_EXPORT struct SparseDoubleMatrix *read_sdm(const char *filename)
{
  long nrow, ncol, nnz; int *rows, *cols; double *vals;
  read_coo(filename, &nrow, &ncol, &nnz, &rows, &cols, &vals);
  return new_sdm(nrow, ncol, nnz, rows, cols, vals);
}

/* ---- CSR builders (csr.h:30-74, 375-422) ------------------------------------------------- */
static void build_csr(int64_t nnz, int nrow, int ncol, const int *rows, const int *cols, const double *vals, int **row_ptr,
                      int **out_cols, double **out_vals)
{
  if (fs_device_build_wanted(nnz)) {   /* large matrices: upload once, stable device sort, download (fs_bucket_coo) */
    *row_ptr = (int *)xmalloc(sizeof(int) * ((size_t)nrow + 1));
    *out_cols = (int *)xmalloc(sizeof(int) * (size_t)nnz);
    if (vals) *out_vals = (double *)xmalloc(sizeof(double) * (size_t)nnz);
    if (fs_bucket_coo(0, 1, nrow, ncol, nrow, nnz, rows, cols, vals, *row_ptr, NULL, *out_cols, vals ? *out_vals : NULL))
      device_build_failed("new_csr / new_bcsr");
    return;
  }
  int64_t *keys = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)nnz);
  int64_t *off = (int64_t *)xmalloc(sizeof(int64_t) * ((size_t)nrow + 1));
  for (int64_t k = 0; k < nnz; k++) keys[k] = rows[k];
  int64_t *slot = bucket_slots(nnz, nrow, keys, off);
  *row_ptr = (int *)xmalloc(sizeof(int) * ((size_t)nrow + 1));
  *out_cols = (int *)xmalloc(sizeof(int) * (size_t)nnz);
  if (vals) *out_vals = (double *)xmalloc(sizeof(double) * (size_t)nnz);
  for (int r = 0; r <= nrow; r++) (*row_ptr)[r] = (int)off[r];
  for (int64_t k = 0; k < nnz; k++) {
    (*out_cols)[slot[k]] = cols[k];
    if (vals) (*out_vals)[slot[k]] = vals[k];
  }
  free(keys); free(off); free(slot);
}

FS_EXPORT void new_bcsr(struct BinaryCSR *A, long nnz, int nrow, int ncol, int *rows, int *cols)
{
  A->nnz = nnz; A->nrow = nrow; A->ncol = ncol;
  build_csr(nnz, nrow, ncol, rows, cols, NULL, &A->row_ptr, &A->cols, NULL);
}

FS_EXPORT void bcsr_from_sbm(struct BinaryCSR *A, struct SparseBinaryMatrix *sbm)
{
  new_bcsr(A, sbm->nnz, sbm->nrow, sbm->ncol, sbm->rows, sbm->cols);
}

FS_EXPORT void new_csr(struct CSR *A, long nnz, int nrow, int ncol, int *rows, int *cols, double *vals)
{
  A->nnz = nnz; A->nrow = nrow; A->ncol = ncol;
  build_csr(nnz, nrow, ncol, rows, cols, vals, &A->row_ptr, &A->cols, &A->vals);
}

/* ---- column-blocked binary CSR (cbcsr.h:16-73) ------------------------------------------- */
FS_EXPORT void new_cbcsr(struct ColBinaryCSR *A, int colblocksize, long nnz, int nrow, int ncol, int *rows, int *cols)
{
  A->nnz = (int)nnz; A->nrow = nrow; A->ncol = ncol;
  A->nblocks = blocks_for(ncol, colblocksize);
  A->colblocksize = colblocksize;
  int64_t ncell = (int64_t)A->nblocks * nrow;
  if (fs_device_build_wanted(nnz) && ncell < ((int64_t)1 << 31) - 1) {
    A->row_ptr = (int *)xmalloc(sizeof(int) * ((size_t)ncell + 1));
    A->cols = (int *)xmalloc(sizeof(int) * (size_t)nnz);
    if (fs_bucket_coo(1, colblocksize, nrow, ncol, ncell, nnz, rows, cols, NULL, A->row_ptr, NULL, A->cols, NULL))
      device_build_failed("new_cbcsr");
    return;
  }
  int64_t *keys = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)nnz);
  int64_t *off = (int64_t *)xmalloc(sizeof(int64_t) * ((size_t)ncell + 1));
  for (int64_t k = 0; k < nnz; k++) keys[k] = (int64_t)(cols[k] / colblocksize) * nrow + rows[k];
  int64_t *slot = bucket_slots(nnz, ncell, keys, off);
  A->row_ptr = (int *)xmalloc(sizeof(int) * ((size_t)ncell + 1));
  A->cols = (int *)xmalloc(sizeof(int) * (size_t)nnz);
  for (int64_t c = 0; c <= ncell; c++) A->row_ptr[c] = (int)off[c];
  for (int64_t k = 0; k < nnz; k++) A->cols[slot[k]] = cols[k];
  free(keys); free(off); free(slot);
}

FS_EXPORT void cbcsr_from_sbm(struct ColBinaryCSR *A, struct SparseBinaryMatrix *sbm, int colblocksize)
{
  new_cbcsr(A, colblocksize, sbm->nnz, sbm->nrow, sbm->ncol, sbm->rows, sbm->cols);
}

/* ---- row-blocked COO (sparse.h:175-213, dsparse.h:132-173) --------------------------------- */
static void build_blocks(int64_t nnz, int nrow, int ncol, int block_size, const int *rows, const int *cols, const double *vals,
                         int *nblocks, int **start_row, int **blk_nnz, int ***brows, int ***bcols, double ***bvals)
{
  int nb = blocks_for(nrow, block_size);
  *nblocks = nb;
  *start_row = (int *)xmalloc(sizeof(int) * ((size_t)nb + 1));
  *blk_nnz = (int *)xmalloc(sizeof(int) * (size_t)nb);
  *brows = (int **)xmalloc(sizeof(int *) * (size_t)nb);
  *bcols = (int **)xmalloc(sizeof(int *) * (size_t)nb);
  if (bvals) *bvals = (double **)xmalloc(sizeof(double *) * (size_t)nb);
  if (fs_device_build_wanted(nnz)) {
    /* entries in block order from the device, then cut into the per-block arrays the struct wants */
    int *off32 = (int *)xmalloc(sizeof(int) * ((size_t)nb + 1));
    int *sr = (int *)xmalloc(sizeof(int) * (size_t)nnz), *sc = (int *)xmalloc(sizeof(int) * (size_t)nnz);
    double *sv = vals ? (double *)xmalloc(sizeof(double) * (size_t)nnz) : NULL;
    if (fs_bucket_coo(2, block_size, nrow, ncol, nb, nnz, rows, cols, vals, off32, sr, sc, sv)) device_build_failed("new_bsbm / new_bsdm");
    for (int b = 0; b < nb; b++) {
      size_t n = (size_t)(off32[b + 1] - off32[b]);
      (*start_row)[b] = b * block_size;
      (*blk_nnz)[b] = (int)n;
      (*brows)[b] = (int *)xmalloc(sizeof(int) * n);
      (*bcols)[b] = (int *)xmalloc(sizeof(int) * n);
      memcpy((*brows)[b], sr + off32[b], sizeof(int) * n);
      memcpy((*bcols)[b], sc + off32[b], sizeof(int) * n);
      if (bvals) {
        (*bvals)[b] = (double *)xmalloc(sizeof(double) * n);
        memcpy((*bvals)[b], sv + off32[b], sizeof(double) * n);
      }
    }
    (*start_row)[nb] = nrow;
    free(off32); free(sr); free(sc); free(sv);
    return;
  }
  int64_t *keys = (int64_t *)xmalloc(sizeof(int64_t) * (size_t)nnz);
  int64_t *off = (int64_t *)xmalloc(sizeof(int64_t) * ((size_t)nb + 1));
  for (int64_t k = 0; k < nnz; k++) keys[k] = rows[k] / block_size;
  int64_t *slot = bucket_slots(nnz, nb, keys, off);
  for (int b = 0; b < nb; b++) {
    int n = (int)(off[b + 1] - off[b]);
    (*start_row)[b] = b * block_size;
    (*blk_nnz)[b] = n;
    (*brows)[b] = (int *)xmalloc(sizeof(int) * (size_t)n);
    (*bcols)[b] = (int *)xmalloc(sizeof(int) * (size_t)n);
    if (bvals) (*bvals)[b] = (double *)xmalloc(sizeof(double) * (size_t)n);
  }
  (*start_row)[nb] = nrow;
  for (int64_t k = 0; k < nnz; k++) {
    int b = (int)keys[k];
    int64_t i = slot[k] - off[b];
    (*brows)[b][i] = rows[k];
    (*bcols)[b][i] = cols[k];
    if (bvals) (*bvals)[b][i] = vals[k];
  }
  free(keys); free(off); free(slot);
}

FS_EXPORT struct BlockedSBM *new_bsbm(struct SparseBinaryMatrix *A, int block_size)
{
  struct BlockedSBM *B = (struct BlockedSBM *)xmalloc(sizeof *B);
  B->nrow = A->nrow; B->ncol = A->ncol;
  build_blocks(A->nnz, A->nrow, A->ncol, block_size, A->rows, A->cols, NULL, &B->nblocks, &B->start_row, &B->nnz, &B->rows,
               &B->cols, NULL);
  return B;
}

FS_EXPORT struct BlockedSDM *new_bsdm(struct SparseDoubleMatrix *A, int block_size)
{
  struct BlockedSDM *B = (struct BlockedSDM *)xmalloc(sizeof *B);
  B->nrow = A->nrow; B->ncol = A->ncol;
  build_blocks(A->nnz, A->nrow, A->ncol, block_size, A->rows, A->cols, A->vals, &B->nblocks, &B->start_row, &B->nnz, &B->rows,
               &B->cols, &B->vals);
  return B;
}

/* ---- linalg.h helpers (host; linalg.h:6-88 of the reference) -------------------------------------- */
#include "linalg.h"

FS_EXPORT double dist(double *x, double *y, int n)
{
  double d = 0;
  for (int i = 0; i < n; i++) { double t = x[i] - y[i]; d += t * t; }
  return sqrt(d);
}

FS_EXPORT double pnormsq(double *x, int n)
{
  double s = 0;
  for (int i = 0; i < n; i++) s += x[i] * x[i];
  return s;
}

FS_EXPORT double pdot(double *x, double *y, int n)
{
  double s = 0;
  for (int i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}

FS_EXPORT void pdot2sym(double *D, double *X, double *Y, int n)
{
  double aa = 0, bb = 0, ab = 0;
  for (int i = 0; i < n; i++) { aa += X[2 * i] * Y[2 * i]; bb += X[2 * i + 1] * Y[2 * i + 1]; ab += X[2 * i] * Y[2 * i + 1]; }
  D[0] = aa; D[1] = bb; D[2] = ab;
}

FS_EXPORT void pnormsq2(double *normsq, double *X, int n)
{
  double d[3];
  pdot2sym(d, X, X, n);
  normsq[0] = d[0]; normsq[1] = d[1];
}

FS_EXPORT void pouter2(double *outer, double *X, int n) { pdot2sym(outer, X, X, n); }

FS_EXPORT void solve2sym(double *X, double *A, double *RHS)
{
  double dinv = 1.0 / (A[0] * A[1] - A[2] * A[2]);
  double i0 = dinv * A[1], i1 = dinv * A[0], i2 = -dinv * A[2];
  X[0] = i0 * RHS[0] + i2 * RHS[1];
  X[1] = i2 * RHS[0] + i1 * RHS[1];
  X[2] = i0 * RHS[2] + i2 * RHS[3];
  X[3] = i2 * RHS[2] + i1 * RHS[3];
}

/* ---- BinaryCSR on-disk form (csr.h:83-146 of the reference) ----------------------
 * text-tagged raw dump: header line, "struct BinaryCSR\n" + the 32-byte struct as it sits in memory (its two
 * pointers are meaningless on disk), "int[nrow+1]\n" + row_ptr, "int[nnz]\n" + cols.  Files written by the
 * reference load here and vice versa. */
#define FS_BCSR_HEADER "BINARY_CSR: struct BinaryCSR, int[nrow], int[nnz]\n"

static void expect_line(FILE *f, const char *want, const char *what)
{
  char buf[256];
  if (!fgets(buf, sizeof buf, f) || strncmp(buf, want, sizeof buf)) {
    printf("ERROR: could not read data from file, %s\n  expected: \"%s\"\n      read: \"%s\"\n", what, want, buf);
    exit(-1);
  }
}

FS_EXPORT void serialize_to_file(const struct BinaryCSR *bcsr, const char *filename)
{
  FILE *f = fopen(filename, "w+");
  if (!f) { fprintf(stderr, "File error: %s\n", filename); exit(1); }
  fputs(FS_BCSR_HEADER, f);
  fputs("struct BinaryCSR\n", f);
  fwrite(bcsr, sizeof *bcsr, 1, f);
  fprintf(f, "int[%d]\n", bcsr->nrow + 1);
  fwrite(bcsr->row_ptr, sizeof(int), (size_t)bcsr->nrow + 1, f);
  fprintf(f, "int[%ld]\n", bcsr->nnz);
  fwrite(bcsr->cols, sizeof(int), (size_t)bcsr->nnz, f);
  fclose(f);
}

FS_EXPORT void deserialize_from_file(struct BinaryCSR *bcsr, const char *filename)
{
  char tag[32];
  FILE *f = fopen(filename, "r");
  if (!f) { fprintf(stderr, "File error: %s\n", filename); exit(1); }
  expect_line(f, FS_BCSR_HEADER, "Invalid file format or version");
  expect_line(f, "struct BinaryCSR\n", "struct data corrupted");
  if (fread(bcsr, sizeof *bcsr, 1, f) != 1) { printf("ERROR: struct data truncated\n"); exit(-1); }
  bcsr->row_ptr = (int *)xmalloc(sizeof(int) * ((size_t)bcsr->nrow + 1));
  bcsr->cols = (int *)xmalloc(sizeof(int) * (size_t)bcsr->nnz);
  snprintf(tag, sizeof tag, "int[%d]\n", bcsr->nrow + 1);
  expect_line(f, tag, "nrow data corrupted");
  if (fread(bcsr->row_ptr, sizeof(int), (size_t)bcsr->nrow + 1, f) != (size_t)bcsr->nrow + 1) { printf("ERROR: row_ptr truncated\n"); exit(-1); }
  snprintf(tag, sizeof tag, "int[%ld]\n", bcsr->nnz);
  expect_line(f, tag, "cols data corrupted");
  if (fread(bcsr->cols, sizeof(int), (size_t)bcsr->nnz, f) != (size_t)bcsr->nnz) { printf("ERROR: cols truncated\n"); exit(-1); }
  fclose(f);
}
