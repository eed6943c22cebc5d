/*
 * csr.h -- drop-in replacement for libfastsparse's csr.h (BinaryCSR and CSR), MI355X build.
 * See sparse.h in this directory for the rules.
 */
#ifndef CSR_H
#define CSR_H

#include "sparse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference: csr.h:15-22 (32 bytes) */
struct BinaryCSR
{
  int nrow;
  int ncol;
  long nnz;
  int* row_ptr; /* nrow + 1 row starts */
  int* cols;
};

/* reference: csr.h:358-366 (40 bytes) */
struct CSR
{
  int nrow;
  int ncol;
  long nnz;
  int* row_ptr;
  int* cols;
  double* vals;
};

/* host-side builders: stable in input order per row, columns neither sorted nor merged */
void new_bcsr(struct BinaryCSR* A, long nnz, int nrow, int ncol, int* rows, int* cols);            /* csr.h:30 */
void bcsr_from_sbm(struct BinaryCSR* A, struct SparseBinaryMatrix* sbm);                            /* csr.h:69 */
void free_bcsr(struct BinaryCSR* bcsr);                                                             /* csr.h:24 */
void new_csr(struct CSR* A, long nnz, int nrow, int ncol, int* rows, int* cols, double* vals);     /* csr.h:375 */
void free_csr(struct CSR* csr);                                                                     /* csr.h:368 */

/* on-disk form of a BinaryCSR (host I/O; same bytes as the reference writes, csr.h:97-146) */
void serialize_to_file(const struct BinaryCSR* bcsr, const char* filename);
void deserialize_from_file(struct BinaryCSR* bcsr, const char* filename);

/* binary CSR products (GPU); X and Y are row-major ("row-ordered") for the multi-column forms */
void bcsr_A_mul_B(double* y, struct BinaryCSR* A, double* x);                  /* csr.h:149 */
void bcsr_A_mul_B2(double* Y, struct BinaryCSR* A, double* X);                 /* csr.h:164 */
void bcsr_A_mul_B4(double* Y, struct BinaryCSR* A, double* X);                 /* csr.h:184 */
void bcsr_A_mul_B8(double* Y, struct BinaryCSR* A, double* X);                 /* csr.h:205 */
void bcsr_A_mul_B8_auto(double* Y, struct BinaryCSR* A, double* X);            /* csr.h:225 */
void bcsr_A_mul_Bn(double* Y, struct BinaryCSR* A, double* X, const int ncol); /* csr.h:257 */
void bcsr_A_mul_B32n(double* Y, struct BinaryCSR* A, double* X, const int ncol); /* ncol <= 32, csr.h:283 */
void bcsr_AA_mul_B(double* y, struct BinaryCSR* A, double* x);                 /* y = A'A x, csr.h:305 */
void parallel_bcsr_AA_mul_B(double* y, struct BinaryCSR* A, double* x, double* ytmp); /* ytmp unused on the GPU, csr.h:323 */

/* fp64 CSR products (GPU) */
void csr_A_mul_B(double* y, struct CSR* A, double* x);                         /* csr.h:425 */
void csr_A_mul_Bn(double* Y, struct CSR* A, double* X, const int ncol);        /* csr.h:441 */

/* Additions (the reference has no CSR transposed product, SURVEY.md note N3): y[ncol] = A' x
 * with sdm_At_mul_B semantics (y overwritten), through a cached device CSR of A'. */
void csr_At_mul_B(double* y, struct CSR* A, double* x);
void bcsr_At_mul_B(double* y, struct BinaryCSR* A, double* x);

#ifdef __cplusplus
}
#endif
#endif /* CSR_H */
