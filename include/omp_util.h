/*
 * omp_util.h -- drop-in for libfastsparse's omp_util.h.  The products run on the GPU and the
 * library starts no host threads, so these report the reference's serial answers (its
 * non-OpenMP branch, omp_util.h:30-33): a caller that sizes per-thread buffers with them
 * gets one buffer.
 */
#ifndef OMP_UTIL_H
#define OMP_UTIL_H

#ifdef __cplusplus
extern "C" {
#endif

int thread_num(void);     /* 0 */
int nthreads(void);       /* 1 */
int thread_limit(void);   /* 1 */
void threads_init(void);  /* nothing to start */

#ifdef __cplusplus
}
#endif

#endif /* OMP_UTIL_H */
