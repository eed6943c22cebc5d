/*
 * hilbert.h -- drop-in replacement for libfastsparse's hilbert.h: Hilbert-curve index helpers used by the
 * locality sorters (host functions in libfastsparse_hip.so).
 */
#ifndef HILBERT_H
#define HILBERT_H

#ifdef __cplusplus
extern "C" {
#endif

int  ceilPower2(int x);                               /* smallest power of two >= x,           hilbert.h:11 */
long xy2d(int n, int x, int y);                       /* (x, y) -> curve position, n x n grid, hilbert.h:16 */
void d2xy(int n, long d, int* x, int* y);             /* curve position -> (x, y),             hilbert.h:30 */
void rot(int n, int* x, int* y, int rx, int ry);      /* quadrant rotate / flip,               hilbert.h:45 */
long row_xy2d(int n, int x, int y);                   /* curve over an n-row block,            hilbert.h:60 */
void row_d2xy(int n, long d, int* x, int* y);         /*                                       hilbert.h:68 */

#ifdef __cplusplus
}
#endif
#endif /* HILBERT_H */
