/*
 * linalg.h -- drop-in replacement for libfastsparse's linalg.h: the small dense helpers the CG drivers and
 * their callers use.  Host functions (they work on a handful of doubles or on host vectors); the device-side
 * reductions of the GPU solvers live in libfastsparse_hip.so (fs_cg.hip).
 */
#ifndef LINALG_H
#define LINALG_H

#ifdef __cplusplus
extern "C" {
#endif

double dist(double* x, double* y, int n);                        /* ||x - y||_2,              linalg.h:6  */
double pnormsq(double* x, int n);                                /* x'x,                      linalg.h:15 */
void   pnormsq2(double* normsq, double* X, int n);               /* {a'a, b'b} of X = [a b],  linalg.h:24 */
void   pouter2(double* outer, double* X, int n);                 /* {a'a, b'b, a'b},          linalg.h:37 */
double pdot(double* x, double* y, int n);                        /* x'y,                      linalg.h:51 */
void   pdot2sym(double* D, double* X, double* Y, int n);         /* {xa'ya, xb'yb, xa'yb},    linalg.h:61 */
void   solve2sym(double* X, double* A, double* RHS);             /* 2x2 symmetric solve,      linalg.h:77 */

#ifdef __cplusplus
}
#endif
#endif /* LINALG_H */
