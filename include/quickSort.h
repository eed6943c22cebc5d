/* quickSort.h -- drop-in replacement for libfastsparse's quickSort.h: ascending sort of a[l..r] (inclusive). */
#ifndef QUICKSORT_H
#define QUICKSORT_H
#ifdef __cplusplus
extern "C" {
#endif
void quickSort(long a[], long l, long r);             /* quickSort.h:10 */
#ifdef __cplusplus
}
#endif
#endif /* QUICKSORT_H */
