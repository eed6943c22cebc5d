// probe_ldsatomic.hip -- tuning probe (not product): rate of ds_add_f64 (no return) on random / strided LDS addresses,
// against plain ds_write_b64, with no memory traffic at all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>   // 0: atomic add random, 1: atomic add consecutive (lane i -> slot base+i), 2: plain store random, 3: atomic random, 32-bit float
__global__ __launch_bounds__(1024, 8) void k(double* out, int iters, unsigned mul) {
  __shared__ double yt[8192];
  const int t = threadIdx.x;
  for (int i = t; i < 8192; i += 1024) yt[i] = 0.0;
  __syncthreads();
  unsigned h = t * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < iters; ++i) {
    h = h * mul + 1013904223u;
    unsigned idx = (MODE == 1) ? ((h >> 19) & ~63u & 8191u) + (t & 63) : (h >> 19);   // 13 bits
    if (MODE == 1) idx &= 8191u;
    if (MODE == 2) yt[idx] = (double)i;
    else if (MODE == 3) __hip_atomic_fetch_add((float*)yt + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(yt + idx, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (t < 64) out[blockIdx.x * 64 + t] = yt[t * 7];
}

template <int MODE>
void run(const char* name, double* out) {
  const int blocks = 512, iters = 1250;    // 512*1024*1250 = 655 M operations = 2.56 M per CU
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<MODE><<<blocks, 1024>>>(out, iters, 1664525u); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k<MODE><<<blocks, 1024>>>(out, iters, 1664525u);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double ops = (double)blocks * 1024 * iters;
  printf("{\"probe\": \"%s\", \"ms\": %.4f, \"Gops\": %.1f, \"ops_per_clk_per_cu_at_2.4GHz\": %.2f}\n", name, ms, ops / ms * 1e-6,
         ops / 256 / (ms * 1e-3 * 2.4e9));
}

int main() {
  double* out; CK(hipMalloc(&out, 512 * 64 * 8));
  run<0>("ds_add_f64 random", out);
  run<1>("ds_add_f64 consecutive", out);
  run<2>("ds_write_b64 random", out);
  run<3>("ds_add_f32 random", out);
  return 0;
}
