/* quickSortD.h -- drop-in replacement for libfastsparse's quickSortD.h: sort a[l..r] ascending, moving v along. */
#ifndef QUICKSORTD_H
#define QUICKSORTD_H
#ifdef __cplusplus
extern "C" {
#endif
void quickSortD(long a[], long l, long r, double* v); /* quickSortD.h:12 */
#ifdef __cplusplus
}
#endif
#endif /* QUICKSORTD_H */
