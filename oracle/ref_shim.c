/*
 * ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Translation unit that compiles the REAL reference headers where they lie
 * (/root/reference, passed with -I by oracle/Makefile) into
 * oracle/_ref/libfsref.so.  No reference source is copied: this file only
 * #includes the reference's own headers and, for the three constructors the
 * reference declares `static inline` (csr.h:30, csr.h:375, cbcsr.h:16, plus
 * the two *_from_sbm helpers), adds a one-line exported wrapper so that
 * ctypes can reach them.
 *
 * Built with -std=gnu99 -fgnu89-inline so that every (non-static) `inline`
 * definition in the reference headers also emits an external symbol
 * (A_mul_B, csr_A_mul_B, ...), and WITHOUT -fopenmp: omp_util.h:7 pulls in
 * <cblas.h> under _OPENMP, a header this image does not have, and no stand-in
 * is written for it.  Without OpenMP the reference's pragmas are ignored and
 * every kernel runs its loops serially in the same per-row order, which is
 * exactly what a parity oracle wants.
 */
#include <stdio.h>
#include "sparse.h"
#include "dsparse.h"
#include "csr.h"
#include "cbcsr.h"
#include "cg.h"   /* bsbm_cg / bsbm_cg2: plain (non-inline) definitions, exported as they are */

void ref_new_bcsr(struct BinaryCSR *A, long nnz, int nrow, int ncol, int *rows, int *cols)
{ new_bcsr(A, nnz, nrow, ncol, rows, cols); }

void ref_new_csr(struct CSR *A, long nnz, int nrow, int ncol, int *rows, int *cols, double *vals)
{ new_csr(A, nnz, nrow, ncol, rows, cols, vals); }

void ref_new_cbcsr(struct ColBinaryCSR *A, int colblocksize, long nnz, int nrow, int ncol,
                   int *rows, int *cols)
{ new_cbcsr(A, colblocksize, nnz, nrow, ncol, rows, cols); }

void ref_bcsr_from_sbm(struct BinaryCSR *A, struct SparseBinaryMatrix *sbm)
{ bcsr_from_sbm(A, sbm); }

void ref_cbcsr_from_sbm(struct ColBinaryCSR *A, struct SparseBinaryMatrix *sbm, int colblocksize)
{ cbcsr_from_sbm(A, sbm, colblocksize); }

/* static in csr.h:97,117 */
void ref_serialize_to_file(const struct BinaryCSR *b, const char *fn) { serialize_to_file(b, fn); }
void ref_deserialize_from_file(struct BinaryCSR *b, const char *fn) { deserialize_from_file(b, fn); }

/* static in sparse.h:142,215,238 */
void ref_sort_sbm(struct SparseBinaryMatrix *A) { sort_sbm(A); }
void ref_sort_bsbm(struct BlockedSBM *B) { sort_bsbm(B); }
void ref_sort_bsbm_byrow(struct BlockedSBM *B) { sort_bsbm_byrow(B); }
