// probe_mall.hip -- tuning probe (not product): does a write-then-read working set that fits the 256 MB Infinity Cache run
// faster than one that does not?  Kernel W writes a buffer of S bytes (nt or plain stores), kernel R reads it back; the
// pair is repeated, and the combined rate reported per size.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(1024) void wr(double* p, long n2) {   // n2 = number of 16-byte elements
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n2; i += (long)gridDim.x * 1024) {
    v2d v = {(double)i, 1.0};
    if (NT) __builtin_nontemporal_store(v, (v2d*)p + i); else ((v2d*)p)[i] = v;
  }
}
template <bool NT>
__global__ __launch_bounds__(1024) void rd(const double* p, long n2, double* out) {
  double acc = 0;
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n2; i += (long)gridDim.x * 1024) {
    v2d v = NT ? __builtin_nontemporal_load((const v2d*)p + i) : ((const v2d*)p)[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;
}

template <bool NT>
void run(double* buf, double* out, long bytes) {
  const long n2 = bytes / 16;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int reps = (int)(8e9 / bytes) + 2;
  for (int w = 0; w < 2; ++w) { wr<NT><<<1024, 1024>>>(buf, n2); rd<NT><<<1024, 1024>>>(buf, n2, out); }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) { wr<NT><<<1024, 1024>>>(buf, n2); rd<NT><<<1024, 1024>>>(buf, n2, out); }
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("{\"probe\": \"write+read %s\", \"MB\": %ld, \"pairs\": %d, \"us_per_pair\": %.1f, \"GBps_combined\": %.0f}\n", NT ? "nt" : "plain", bytes >> 20, reps,
         ms * 1e3 / reps, 2.0 * bytes * reps / ms * 1e-6);
  fflush(stdout);
}

int main() {
  double *buf, *out; CK(hipMalloc(&buf, 2048L << 20)); CK(hipMalloc(&out, 64));
  for (long mb : {16L, 64L, 128L, 192L, 384L, 1024L, 2048L}) { run<true>(buf, out, mb << 20); run<false>(buf, out, mb << 20); }
  return 0;
}
