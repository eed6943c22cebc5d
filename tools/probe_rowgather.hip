// probe_rowgather.hip -- tuning probe (not product): how fast does gfx950 gather whole ROWS of R bytes from a table far larger
// than the caches?  This is the access pattern of the row SpMM kernel (Y = A X with k row-major right-hand sides: every entry
// reads the 8k bytes of one X row): R = 32 / 64 / 128 / 256 bytes <-> k = 4 / 8 / 16 / 32.  A group of R / W lanes reads one row
// with W-byte loads (W = 8: global_load_dwordx2, what spmm_kernel does; W = 16: dwordx4), U rows in flight per group.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_rowgather tools/probe_rowgather.hip && tools/probe_rowgather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

__global__ void make_idx(int* idx, long n, long rows) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (int)(((unsigned __int128)splitmix64((uint64_t)i * 0x9E3779B97F4A7C15ull + 777) * (uint64_t)rows) >> 64);
}

template <int R, int W, int U>
__global__ __launch_bounds__(256) void rowgather(const int* __restrict__ idx, const char* __restrict__ table, double* __restrict__ out) {
  constexpr int G = R / W;                                      // lanes per row
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long g = t / G;
  const int j = (int)(t % G);
  int ix[U];
#pragma unroll
  for (int u = 0; u < U; u += 4) {
    int4 v = *reinterpret_cast<const int4*>(idx + g * U + u);
    ix[u] = v.x; ix[u + 1] = v.y; ix[u + 2] = v.z; ix[u + 3] = v.w;
  }
  double acc = 0;
  if (W == 8) {
    double w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const double*>(table + (long)ix[u] * R + j * 8);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += w[u];
  } else {
    double2 w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const double2*>(table + (long)ix[u] * R + j * 16);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += w[u].x + w[u].y;
  }
  out[t] = acc;
}

template <int R, int W, int U>
void run(const int* idx, const char* table, double* out, long n, long rows) {
  constexpr int G = R / W;
  long blocks = n / U * G / 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((rowgather<R, W, U>), dim3(blocks), dim3(256), 0, 0, idx, table, out);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < 5; i++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((rowgather<R, W, U>), dim3(blocks), dim3(256), 0, 0, idx, table, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  float ms = ts[2];
  printf("{\"probe\":\"rowgather\",\"row_bytes\":%d,\"load_bytes\":%d,\"in_flight\":%d,\"table_MB\":%.0f,\"rows_gathered\":%ld,\"ms\":%.4f,"
         "\"Grows_s\":%.1f,\"row_TBs\":%.3f,\"TBs_if_64B_granules\":%.3f,\"TBs_if_128B_granules\":%.3f}\n",
         R, W, U, rows * (double)R / 1e6, n, ms, n / ms / 1e6, n * (double)R / ms / 1e9, n * (double)std::max(R, 64) / ms / 1e9,
         n * (double)std::max(R, 128) / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long n = 160L * 1000 * 1000 / 8192 * 8192;      // rows gathered per launch (config 4's entries)
  const long rows = argc > 1 ? atol(argv[1]) : 10L * 1000 * 1000;   // table rows (config 4's columns)
  int* idx; char* table; double* out;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&table, rows * 256)); CK(hipMalloc(&out, n / 8 * 32 * 8));
  CK(hipMemset(table, 0, rows * 256));
  hipLaunchKernelGGL(make_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, rows);
  CK(hipDeviceSynchronize());
  run<16, 8, 8>(idx, table, out, n, rows);   run<16, 16, 8>(idx, table, out, n, rows);
  run<32, 8, 8>(idx, table, out, n, rows);   run<32, 16, 8>(idx, table, out, n, rows);  run<32, 16, 16>(idx, table, out, n, rows);
  run<64, 8, 8>(idx, table, out, n, rows);   run<64, 16, 8>(idx, table, out, n, rows);  run<64, 16, 16>(idx, table, out, n, rows);
  run<128, 8, 8>(idx, table, out, n, rows);  run<128, 16, 8>(idx, table, out, n, rows); run<128, 16, 16>(idx, table, out, n, rows);
  run<256, 8, 8>(idx, table, out, n, rows);  run<256, 16, 8>(idx, table, out, n, rows); run<256, 16, 16>(idx, table, out, n, rows);
  return 0;
}
