/*
 * timing.h -- drop-in for libfastsparse's timing.h (wall clock + user CPU time of the process).
 * The reference defines the function in the header; here it is declared and the body lives in
 * libfastsparse_hip.so, so including the header from several translation units links.
 */
#ifndef FS_TIMING_H
#define FS_TIMING_H

#ifdef __cplusplus
extern "C" {
#endif

void timing(double* wcTime, double* cpuTime);   /* seconds; timing.h:9 */

#ifdef __cplusplus
}
#endif

#endif /* FS_TIMING_H */
