// probe_mix.hip -- tuning probe (not product): L2-resident 8-byte gathers with and without a concurrent
// HBM stream of 12 bytes per gather (what the tiled SpMV kernel does), in several load shapes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void make_idx(int* idx, long n, long table_elems) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (int)(((unsigned __int128)splitmix64((uint64_t)i * 77 + 5) * (uint64_t)table_elems) >> 64);
}

// MODE 0: gathers only (idx stream 4 B/gather).  MODE 1: + vals stream 8 B/gather, scalar-width loads (dword / dwordx2 per lane,
// element i of the block = q*BLOCK + t).  MODE 2: + vals stream, 16-byte loads (4 consecutive elements per lane).
// NT: stream loads non-temporal.  BLOCK threads, U elements per thread, blocks loop over ITER chunks (persistent-ish) to mimic a sweep.
template <int BLOCK, int U, int MODE, bool NT>
__global__ __launch_bounds__(BLOCK) void mix(const int* __restrict__ idx, const double* __restrict__ vals, const double* __restrict__ table,
                                             double* __restrict__ out, long chunks_per_block) {
  const int t = threadIdx.x;
  double acc = 0;
  for (long c = 0; c < chunks_per_block; ++c) {
    const long base = ((long)blockIdx.x * chunks_per_block + c) * (long)(BLOCK * U);
    int ix[U]; double v[U];
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < U; u += 4) {
        const long e = base + (long)(u / 4) * BLOCK * 4 + t * 4;
        v4i a = NT ? __builtin_nontemporal_load((const v4i*)(idx + e)) : *(const v4i*)(idx + e);
        ix[u] = a.x; ix[u+1] = a.y; ix[u+2] = a.z; ix[u+3] = a.w;
        v2d p0 = NT ? __builtin_nontemporal_load((const v2d*)(vals + e)) : *(const v2d*)(vals + e);
        v2d p1 = NT ? __builtin_nontemporal_load((const v2d*)(vals + e + 2)) : *(const v2d*)(vals + e + 2);
        v[u] = p0.x; v[u+1] = p0.y; v[u+2] = p1.x; v[u+3] = p1.y;
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long e = base + (long)u * BLOCK + t;
        ix[u] = NT ? __builtin_nontemporal_load(idx + e) : idx[e];
        v[u] = (MODE == 1) ? (NT ? __builtin_nontemporal_load(vals + e) : vals[e]) : 1.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += table[ix[u]] * v[u];
  }
  out[(long)blockIdx.x * BLOCK + t] = acc;
}

template <int BLOCK, int U, int MODE, bool NT>
float run(const int* idx, const double* vals, const double* table, double* out, long n, long blocks) {
  long cpb = n / (blocks * BLOCK * U);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((mix<BLOCK, U, MODE, NT>), dim3(blocks), dim3(BLOCK), 0, 0, idx, vals, table, out, cpb);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < 5; i++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((mix<BLOCK, U, MODE, NT>), dim3(blocks), dim3(BLOCK), 0, 0, idx, vals, table, out, cpb);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  printf("{\"probe\":\"mix\",\"block\":%d,\"U\":%d,\"mode\":%d,\"nt\":%d,\"blocks\":%ld,\"ms\":%.4f,\"Ggather_s\":%.1f}\n", BLOCK, U, MODE, (int)NT, blocks, ts[2],
         (double)(cpb * blocks * BLOCK * U) / ts[2] / 1e6);
  fflush(stdout);
  return ts[2];
}

int main() {
  const long n = 160L * 1000 * 1000 / (1 << 20) * (1 << 20);
  const long table_elems = (2L << 20) / 8;   // 2 MiB table: resident in every L2
  int* idx; double *vals, *table, *out;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&vals, n * 8)); CK(hipMalloc(&table, table_elems * 8)); CK(hipMalloc(&out, (n / 4) * 8));   /* largest launch writes n/8 outputs */
  CK(hipMemset(table, 0, table_elems * 8)); CK(hipMemset(vals, 0, n * 8));
  hipLaunchKernelGGL(make_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, table_elems);
  CK(hipDeviceSynchronize());
  // many small blocks (occupancy-limited by registers only)
  run<256, 8, 0, false>(idx, vals, table, out, n, n / (256 * 8));
  run<256, 8, 1, false>(idx, vals, table, out, n, n / (256 * 8));
  run<256, 8, 1, true>(idx, vals, table, out, n, n / (256 * 8));
  run<256, 8, 2, false>(idx, vals, table, out, n, n / (256 * 8));
  run<256, 8, 2, true>(idx, vals, table, out, n, n / (256 * 8));
  // one 512-thread block per CU, looping (like the tiled kernel), 8 and 16 elements per thread per chunk
  run<512, 8, 0, false>(idx, vals, table, out, n, 256);
  run<512, 8, 1, true>(idx, vals, table, out, n, 256);
  run<512, 8, 2, true>(idx, vals, table, out, n, 256);
  run<512, 16, 2, true>(idx, vals, table, out, n, 256);
  run<512, 8, 2, true>(idx, vals, table, out, n, 512);
  run<512, 8, 2, true>(idx, vals, table, out, n, 1024);
  run<1024, 8, 2, true>(idx, vals, table, out, n, 256);
  run<1024, 8, 2, true>(idx, vals, table, out, n, 512);
  run<256, 8, 2, true>(idx, vals, table, out, n, 2048);
  return 0;
}
