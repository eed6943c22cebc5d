/*
 * cg.h -- drop-in replacement for libfastsparse's cg.h: conjugate gradients on (A'A + lambda I) over row-blocked
 * pattern matrices, MI355X build.  Same signatures; the solves run device resident (vectors stay in HBM for the
 * whole solve, products on the GPU kernels of the A_mul_B path).  x / b may be host or device pointers.
 */
#ifndef CG_H
#define CG_H

#include "linalg.h"
#include "sparse.h"

#ifdef __cplusplus
extern "C" {
#endif

/* y = A'A x + lambda x; tmp: caller scratch of A->nrow doubles (cg.h:9) */
void bsbm_AtA(double* y, struct BlockedSBM* A, struct BlockedSBM* At, double* x, double* tmp, double lambda);
/* solves (A'A + lambda I) x = b (cg.h:25) */
void bsbm_cg(double* x, struct BlockedSBM* A, struct BlockedSBM* At, double* b, double lambda, double tol, int* out_iter);
/* the same for two right-hand sides, X and B row-major with 2 columns (cg.h:85) */
void bsbm_cg2(double* X, struct BlockedSBM* A, struct BlockedSBM* At, double* B, double lambda, double tol, int* out_iter);

#ifdef __cplusplus
}
#endif
#endif /* CG_H */
