// fs_kernel_util.h -- small device helpers shared by the kernel translation units (fs_kernels.hip, fs_kernels_tiled.hip,
// fs_kernels_twopass.hip).  Not installed.
#pragma once

#include "fs_common.h"

namespace fs {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <bool NT, typename T>
__device__ __forceinline__ void stream_store(T v, T *p)
{
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

}  // namespace fs
