/*
 * utils.h -- drop-in for libfastsparse's utils.h: the fixture-file word reader.  The body
 * lives in libfastsparse_hip.so (fs_host.c).
 */
#ifndef UTILS_H
#define UTILS_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

long read_long(FILE* fh);   /* one native long; prints an error and exits on a short read, utils.h:4 */

#ifdef __cplusplus
}
#endif

#endif /* UTILS_H */
