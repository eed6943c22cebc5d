/*
 * sparse.h -- drop-in replacement for libfastsparse's sparse.h (COO pattern matrices and
 * row-blocked COO), MI355X build.  Same struct layouts, same function names, same
 * signatures; the bodies live in libfastsparse_hip.so and run on the GPU.
 *
 * Callers written against the reference header recompile unchanged:
 *     #include "sparse.h"   ...   A_mul_B(y, A, x);      cc ... -lfastsparse_hip
 * and objects compiled against the ORIGINAL header whose calls were not inlined (C99
 * inline leaves them as undefined references) link against the same library.
 *
 * x and y may be host pointers (copied through, the call returns when y is complete) or
 * device pointers (used in place).  The matrix is uploaded on first use and cached per host
 * struct; after changing a matrix's arrays in place call fs_invalidate(A) (fastsparse_hip.h).
 */
#ifndef SPARSE_H
#define SPARSE_H

#include <stdio.h>
#include <stdlib.h>

#include "utils.h"
#include "hilbert.h"
#include "quickSort.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: sparse.h:11-18 (32 bytes, LP64) */
struct SparseBinaryMatrix
{
  int nrow;
  int ncol;
  long nnz;
  int* rows;
  int* cols;
};

/* reference: sparse.h:163-172 (48 bytes) -- per-block arrays, global row ids */
struct BlockedSBM {
  int nrow;
  int ncol;
  int nblocks;
  int* start_row;
  int* nnz;
  int** rows;
  int** cols;
};

/* containers -- host side, same ownership rules as the reference */
struct SparseBinaryMatrix* new_sbm(long nrow, long ncol, long nnz, int* rows, int* cols); /* adopts rows/cols, sparse.h:21 */
void free_sbm(struct SparseBinaryMatrix* sbm);                                            /* frees arrays only, sparse.h:31 */
struct SparseBinaryMatrix* new_transpose(struct SparseBinaryMatrix* A);                   /* aliases A's arrays, sparse.h:38 */
void transpose(struct SparseBinaryMatrix* A);                                             /* swaps in place, sparse.h:48 */
struct SparseBinaryMatrix* read_sbm(const char* filename);                                /* sparse.h:112 */
struct BlockedSBM* new_bsbm(struct SparseBinaryMatrix* A, int block_size);                /* sparse.h:175 */

/* samplers over drand48 (host; same draws as the reference for the same srand48 seed) */
double exprand(void);                                                                     /* sparse.h:77 */
double randexp(void);                                                                     /* sparse.h:83 */
long randsubseq(long N, long max_samples, double p, long* samples);                       /* sparse.h:92 */

/* locality re-orderings (host; they permute the entries in place and drop the cached device copy) */
void sort_sbm(struct SparseBinaryMatrix* A);                                              /* Hilbert order, sparse.h:142 */
void sort_bsbm(struct BlockedSBM* B);                                                     /* per-block Hilbert order, sparse.h:215 */
void sort_bsbm_byrow(struct BlockedSBM* B);                                               /* per-block (row, col) order, sparse.h:238 */

/* products (GPU) */
void A_mul_B(double* y, struct SparseBinaryMatrix* A, double* x);      /* y[nrow] = A x,  sparse.h:58 */
void At_mul_B(double* y, struct SparseBinaryMatrix* A, double* x);     /* y[ncol] = A' x, sparse.h:68 */
void bsbm_A_mul_B(double* y, struct BlockedSBM* B, double* x);         /* sparse.h:259 */
void bsbm_A_mul_B2(double* y, struct BlockedSBM* B, double* x);        /* 2 row-major columns, sparse.h:276 */
void bsbm_A_mul_B4(double* y, struct BlockedSBM* B, double* x);        /* 4 row-major columns, sparse.h:296 */
void bsbm_A_mul_Bn(double* y, struct BlockedSBM* B, double* x, int ncol); /* sparse.h:318 */

#ifdef __cplusplus
}
#endif
#endif /* SPARSE_H */
