/*
 * cbcsr.h -- drop-in replacement for libfastsparse's cbcsr.h (column-blocked binary CSR),
 * MI355X build.  Include after csr.h, as with the reference.
 */
#ifndef CBCSR_H
#define CBCSR_H

#include "csr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: cbcsr.h:5-14 (40 bytes; note nnz is int here) */
struct ColBinaryCSR
{
  int nrow;
  int ncol;
  int nblocks;
  int colblocksize;
  int nnz;
  int* row_ptr; /* nblocks * nrow + 1 entries, cell = block * nrow + row */
  int* cols;
};

void new_cbcsr(struct ColBinaryCSR* A, int colblocksize, long nnz, int nrow, int ncol, int* rows, int* cols); /* cbcsr.h:16 */
void cbcsr_from_sbm(struct ColBinaryCSR* A, struct SparseBinaryMatrix* sbm, int colblocksize);                /* cbcsr.h:67 */
void cbcsr_A_mul_B(double* y, struct ColBinaryCSR* A, double* x);                                             /* cbcsr.h:76 */

#ifdef __cplusplus
}
#endif
#endif /* CBCSR_H */
