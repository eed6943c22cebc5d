/*
 * bench_a_mul_b.c -- driver for the A_mul_B family on the MI355X build, output-compatible with the
 * reference's bench_a_mul_b.c (same options, same section labels, same line format
 * "[label]\tWall: %0.5e\tcpu: %0.5e").  Written against include/sparse.h and
 * include/csr.h exactly as a user of the reference would write it; links with -lfastsparse_hip.
 *
 *   bench_a_mul_b -f <matrix_file> [-b block_size] [-t] [-r] [-c] [-d]
 *     -t  transpose the matrix            -r  also run the CSR kernels
 *     -c  also solve (A'A + 0.5 I) X = B for two right-hand sides with bsbm_cg2 (device-resident block CG)
 *     -d  keep x / y in HBM (device pointers): times the kernels without the PCIe copies
 *
 * Differences: an extra "[csr-f64]" section runs read_sdm -> new_csr -> csr_A_mul_B / csr_At_mul_B when the
 * file carries values (BASELINE config 1); -c reports the CG solve as one line instead of the reference's
 * residual trace.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "cg.h"
#include "csr.h"
#include "dsparse.h"
#include "fastsparse_hip.h"
#include "sparse.h"

static void now(double *wall, double *cpu)
{
  struct timeval tv;
  struct rusage ru;
  gettimeofday(&tv, NULL);
  getrusage(RUSAGE_SELF, &ru);
  *wall = tv.tv_sec + 1e-6 * tv.tv_usec;
  *cpu = ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec;
}

/* One timed section exactly as the reference times it -- `reps` calls between two clock readings, mean per call, warm-up
 * calls only where the reference has them (bench_a_mul_b.c:207,216,225,254,264,275,285,363 and the check call :189) -- plus,
 * on a second line, the first call and the mean of the others: on this build the first call on a matrix (or with a new k)
 * carries its one-time device work, which the reference's loops do not have. */
static double w0, c0, first_call;
static int calls;
static void tic(void) { now(&w0, &c0); calls = 0; first_call = 0.0; }
static void lap(void)
{
  if (calls++ == 0) { double w1, c1; now(&w1, &c1); first_call = w1 - w0; }
}
static void toc(const char *label, int reps)
{
  double w1, c1;
  now(&w1, &c1);
  printf("[%s]\tWall: %0.5e\tcpu: %0.5e\n", label, (w1 - w0) / reps, (c1 - c0) / reps);
  if (calls > 1) printf("  first call %0.5e, others %0.5e\n", first_call, (w1 - w0 - first_call) / (calls - 1));
}
#define TIMED(label, reps, body) do { tic(); for (int i_ = 0; i_ < (reps); i_++) { body; lap(); } toc(label, reps); } while (0)

static int on_device = 0;

/* a dense vector of n doubles filled by f(i): host malloc, or HBM with -d */
static double *vec(long n, double (*f)(long))
{
  double *h = (double *)malloc(sizeof(double) * (n ? n : 1));
  for (long i = 0; i < n; i++) h[i] = f ? f(i) : 0.0;
  if (!on_device) return h;
  double *d = (double *)fs_device_alloc(sizeof(double) * n);
  if (!d || fs_copy_to_device(d, h, sizeof(double) * n)) { fprintf(stderr, "device vector: %s\n", fs_last_error()); exit(1); }
  free(h);
  return d;
}

static double fx(long i) { return sin(7.0 * i + 0.3); }
static double fx2(long i) { return (i & 1) ? sin(11.0 * (i / 2) - 0.2) : sin(7.0 * (i / 2) + 0.3); }
static double fx4(long i) { return sin(7.0 * (i / 4) + 17.0 * (i % 4) + 0.3); }
static double fx8(long i) { return sin(7.0 * (i / 8) + 17.0 * (i % 8) + 0.3); }

struct mul2_job { double *Y, *X; struct BlockedSBM *B, *Bt; int reps; };
static void *mul2_thread(void *arg)
{
  struct mul2_job *j = (struct mul2_job *)arg;
  for (int i = 0; i < j->reps; i++) { bsbm_A_mul_B2(j->Y, j->B, j->X); bsbm_A_mul_B2(j->X, j->Bt, j->Y); }
  return NULL;
}

int main(int argc, char **argv)
{
  int block_size = 1024, tflag = 0, csrflag = 0, cgflag = 0, c;
  const char *filename = NULL;
  while ((c = getopt(argc, argv, "b:cf:rtd")) != -1) {
    switch (c) {
      case 'b': block_size = atoi(optarg); break;
      case 'c': cgflag = 1; break;
      case 'f': filename = optarg; break;
      case 'r': csrflag = 1; break;
      case 't': tflag = 1; break;
      case 'd': on_device = 1; break;
      default: fprintf(stderr, "usage: %s -f <matrix_file> [-b block_size] [-t] [-c] [-r] [-d]\n", argv[0]); return 1;
    }
  }
  if (!filename) { fprintf(stderr, "usage: %s -f <matrix_file> [-b block_size] [-t] [-c] [-r] [-d]\n", argv[0]); return 1; }
  const int nrepeats = 10, cgrepeats = 20;

  struct SparseBinaryMatrix *A = read_sbm(filename);
  if (tflag) transpose(A);
  printf("%s (MI355X build, vectors %s)\nSize of A is %d x %d.\nNumber of nnz = %ld\nBlock size = %d\n", fs_version(),
         on_device ? "in HBM" : "on the host", A->nrow, A->ncol, A->nnz, block_size);

  double *y = vec(A->nrow, NULL), *x = vec(A->ncol, fx);
  double *Y2 = vec(2L * A->nrow, NULL), *X2 = vec(2L * A->ncol, fx2);
  double *Y4 = vec(4L * A->nrow, NULL), *X4 = vec(4L * A->ncol, fx4);
  double *Y8 = vec(8L * A->nrow, NULL), *X8 = vec(8L * A->ncol, fx8);

  TIMED("unsorted", nrepeats, A_mul_B(y, A, x));             /* cold, like the reference (:159-165) */

  sort_sbm(A); /* Hilbert order, in place; drops the cached device copy */
  A_mul_B(y, A, x);                                             /* the reference's check call (:189) */
  TIMED("sort", nrepeats, A_mul_B(y, A, x));

  struct BlockedSBM *B = new_bsbm(A, block_size);
  struct SparseBinaryMatrix *At = new_transpose(A);
  struct BlockedSBM *Bt = new_bsbm(At, block_size);
  bsbm_A_mul_B(y, B, x);                                        /* :207 */
  TIMED("block", nrepeats, bsbm_A_mul_B(y, B, x));
  bsbm_A_mul_B2(Y2, B, X2);                                     /* :216 */
  TIMED("2xblock", nrepeats, bsbm_A_mul_B2(Y2, B, X2));
  bsbm_A_mul_B2(Y2, B, X2);                                     /* :225 */
  TIMED("2xblock*", nrepeats, bsbm_A_mul_Bn(Y2, B, X2, 2));
  TIMED("cg", cgrepeats, { bsbm_A_mul_B(y, B, x); bsbm_A_mul_B(x, Bt, y); });       /* Bt cold (:234-238) */
  TIMED("cg2", cgrepeats, { bsbm_A_mul_B2(Y2, B, X2); bsbm_A_mul_B2(X2, Bt, Y2); });

  if (csrflag) {
    struct BinaryCSR csr, csrt;
    bcsr_from_sbm(&csr, A);
    bcsr_A_mul_B(y, &csr, x);                                   /* :254 */
    TIMED("csr", nrepeats, bcsr_A_mul_B(y, &csr, x));
    bcsr_A_mul_B2(Y2, &csr, X2);                                /* :264 */
    TIMED("csr2", nrepeats, bcsr_A_mul_B2(Y2, &csr, X2));
    bcsr_from_sbm(&csrt, At);
    bcsr_A_mul_B2(X2, &csrt, Y2);                               /* :275 */
    TIMED("cg2-csr", cgrepeats, { bcsr_A_mul_B2(Y2, &csr, X2); bcsr_A_mul_B2(X2, &csrt, Y2); });
    bcsr_A_mul_B4(X4, &csrt, Y4);                               /* :285 -- csr with _B4 stays cold, as in the reference */
    TIMED("cg4-csr", cgrepeats, { bcsr_A_mul_B4(Y4, &csr, X4); bcsr_A_mul_B4(X4, &csrt, Y4); });
    TIMED("cg8-csr", cgrepeats, { bcsr_A_mul_B8(Y8, &csr, X8); bcsr_A_mul_B8(X8, &csrt, Y8); });
    TIMED("cg8a-csr", cgrepeats, { bcsr_A_mul_B8_auto(Y8, &csr, X8); bcsr_A_mul_B8_auto(X8, &csrt, Y8); });
    TIMED("cg8*-csr", cgrepeats, { bcsr_A_mul_Bn(Y8, &csr, X8, 8); bcsr_A_mul_Bn(X8, &csrt, Y8, 8); });
    TIMED("cg8**-csr", cgrepeats, { bcsr_A_mul_B32n(Y8, &csr, X8, 8); bcsr_A_mul_B32n(X8, &csrt, Y8, 8); });
    free_bcsr(&csr);
    free_bcsr(&csrt);
  }
  if (cgflag) { /* bench_a_mul_b.c:332-360: two right-hand sides, (A'A + 0.5 I) X = B */
    int iters = 0;
    double *Bh = (double *)malloc(sizeof(double) * 2 * A->ncol), *Xh = (double *)malloc(sizeof(double) * 2 * A->ncol);
    for (long i = 0; i < 2L * A->ncol; i++) Bh[i] = fx2(i);
    tic();
    bsbm_cg2(Xh, B, Bt, Bh, 0.5, 1e-6, &iters);
    toc("cg solver", 1);
    printf("  bsbm_cg2: %d iterations\n", iters);
    free(Bh); free(Xh);
  }

  bsbm_A_mul_B4(Y4, B, X4);                                     /* :363 */
  TIMED("4xblock", nrepeats, bsbm_A_mul_B4(Y4, B, X4));
  sort_bsbm(B);                                                 /* no warm-up after either sort (:383-398) */
  TIMED("sort+block", nrepeats, bsbm_A_mul_B(y, B, x));
  sort_bsbm_byrow(B);
  TIMED("rowsort+block", nrepeats, bsbm_A_mul_B(y, B, x));

  { /* two host threads multiplying at once on shared matrices (reference: nested OpenMP, bench_a_mul_b.c:401-421) */
    double *Y2b = vec(2L * A->nrow, NULL), *X2b = vec(2L * A->ncol, fx2);
    struct mul2_job j1 = {Y2, X2, B, Bt, cgrepeats / 2}, j2 = {Y2b, X2b, B, Bt, cgrepeats / 2};   /* :414-416 */
    pthread_t t1, t2;
    tic();
    pthread_create(&t1, NULL, mul2_thread, &j1);
    pthread_create(&t2, NULL, mul2_thread, &j2);
    pthread_join(t1, NULL);
    pthread_join(t2, NULL);
    toc("2x cg2", cgrepeats / 2);
  }

  /* BASELINE config 1: fp64 CSR on the same file when it carries values (24 + 16*nnz bytes) */
  struct stat st;
  if (stat(filename, &st) == 0 && st.st_size >= 24 + 16 * A->nnz) {
    struct SparseDoubleMatrix *D = read_sdm(filename);
    if (tflag) sdm_transpose(D);
    struct CSR csr;
    new_csr(&csr, D->nnz, D->nrow, D->ncol, D->rows, D->cols, D->vals);
    double *xr = vec(D->nrow, fx);
    csr_A_mul_B(y, &csr, x);
    TIMED("csr-f64", nrepeats, csr_A_mul_B(y, &csr, x));
    csr_At_mul_B(x, &csr, xr);
    TIMED("csr-f64 At", nrepeats, csr_At_mul_B(x, &csr, xr));
    free_csr(&csr);
  }
  fs_release_all();
  return 0;
}
