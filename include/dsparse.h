/*
 * dsparse.h -- drop-in replacement for libfastsparse's dsparse.h (valued COO and row-blocked
 * valued COO), MI355X build.  See sparse.h in this directory for the rules.
 */
#ifndef DSPARSE_H
#define DSPARSE_H

#include <stdio.h>
#include <stdlib.h>

#include "hilbert.h"
#include "quickSortD.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: dsparse.h:11-19 (40 bytes) */
struct SparseDoubleMatrix
{
  int nrow;
  int ncol;
  long nnz;
  int* rows;
  int* cols;
  double* vals;
};

/* reference: dsparse.h:119-129 (56 bytes) */
struct BlockedSDM {
  int nrow;
  int ncol;
  int nblocks;
  int* start_row;
  int* nnz;
  int** rows;
  int** cols;
  double** vals;
};

struct SparseDoubleMatrix* new_sdm(long nrow, long ncol, long nnz, int* rows, int* cols, double* vals); /* adopts, dsparse.h:22 */
void sdm_transpose(struct SparseDoubleMatrix* A);                              /* dsparse.h:33 */
struct SparseDoubleMatrix* read_sdm(const char* filename);                     /* dsparse.h:64 */
struct BlockedSDM* new_bsdm(struct SparseDoubleMatrix* A, int block_size);     /* dsparse.h:132 */

void sort_sdm(struct SparseDoubleMatrix* A);                                   /* Hilbert order (host), dsparse.h:96 */
void sort_bsdm(struct BlockedSDM* B);                                          /* per-block Hilbert order, dsparse.h:193 */

void sdm_A_mul_B(double* y, struct SparseDoubleMatrix* A, double* x);          /* y[nrow] = A x,  dsparse.h:43 */
void sdm_At_mul_B(double* y, struct SparseDoubleMatrix* A, double* x);         /* y[ncol] = A' x, dsparse.h:54 */
void bsdm_A_mul_B(double* y, struct BlockedSDM* B, double* x);                 /* dsparse.h:176 */

#ifdef __cplusplus
}
#endif
#endif /* DSPARSE_H */
