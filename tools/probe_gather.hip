// probe_gather.hip -- tuning probe (not product): how fast can gfx950 gather 8-byte words?
//   out[b*256+t] = sum_{u<U} table[idx[...]]   with U independent gathers in flight per lane.
// Patterns: uniform over a table of T bytes; "tiled": the blocks that run together on one XCD
// (blockIdx % 8 equal) draw from the same window of W bytes for K consecutive blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

// idx for element i of block b: uniform in [base(b), base(b)+W)
__global__ void make_idx(int* idx, long n, long table_elems, long window_elems, int K, int per_block, int sorted16) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long b = i / per_block;
  long nwin = table_elems / window_elems;
  long win = ((b % 8) + 8 * ((b / 8) / K)) % nwin;
  uint64_t h = splitmix64((uint64_t)i * 0x9E3779B97F4A7C15ull + 12345);
  long v = win * window_elems + (long)(((unsigned __int128)h * (uint64_t)window_elems) >> 64);
  idx[i] = (int)v;
}

template <int U, int KIND>
__global__ __launch_bounds__(256) void gather(const int* __restrict__ idx, const double* __restrict__ table, double* __restrict__ out) {
  const long base = ((long)blockIdx.x * 256 + threadIdx.x) * U;   // U consecutive idx per lane (16B loads when U%4==0)
  int ix[U];
#pragma unroll
  for (int u = 0; u < U; u += 4) {
    int4 v = *reinterpret_cast<const int4*>(idx + base + u);
    ix[u] = v.x; ix[u+1] = v.y; ix[u+2] = v.z; ix[u+3] = v.w;
  }
  double acc = 0;
  double w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (KIND == 0) w[u] = table[ix[u]];
    else if (KIND == 1) w[u] = __builtin_nontemporal_load(table + ix[u]);
    else if (KIND == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(w[u]) : "v"(table + ix[u]) : "memory");
    else if (KIND == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(w[u]) : "v"(table + ix[u]) : "memory");
  }
  if (KIND >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < U; ++u) acc += w[u];
  out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int U, int KIND>
float run(const int* idx, const double* table, double* out, long n, int iters) {
  long blocks = n / (256L * U);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL((gather<U, KIND>), dim3(blocks), dim3(256), 0, 0, idx, table, out);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < iters; i++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<U, KIND>), dim3(blocks), dim3(256), 0, 0, idx, table, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const long n = 160L * 1000 * 1000 / 4096 * 4096;   // gathers per launch
  const long max_table = 1024L * 1024 * 1024 / 8;
  int* idx; double *table, *out;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&table, max_table * 8)); CK(hipMalloc(&out, n / 4 * 8));
  CK(hipMemset(table, 0, max_table * 8));
  const char* only = argc > 1 ? argv[1] : "all";
  bool pmc = !strcmp(only, "pmc");
  auto gen = [&](long table_bytes, long window_bytes, int K, int U) {
    hipLaunchKernelGGL(make_idx, dim3((n + 255) / 256), dim3(256), 0, 0, idx, n, table_bytes / 8, window_bytes / 8, K, 256 * U, 0);
    CK(hipDeviceSynchronize());
  };
  auto rep = [&](const char* name, long tb, long wb, int K, int U, const char* kind, float ms) {
    printf("{\"probe\":\"%s\",\"table_MB\":%.1f,\"window_MB\":%.2f,\"K\":%d,\"U\":%d,\"kind\":\"%s\",\"ms\":%.4f,\"Ggather_s\":%.1f}\n",
           name, tb / 1e6, wb / 1e6, K, U, kind, ms, n / ms / 1e6);
    fflush(stdout);
  };
  if (pmc) {   // few launches, for rocprofv3 --pmc passes
    gen(80L << 20, 80L << 20, 1, 8);  float t = run<8, 0>(idx, table, out, n, 3); rep("uniform", 80L << 20, 80L << 20, 1, 8, "plain", t);
    gen(2L << 20, 2L << 20, 1, 8);    t = run<8, 0>(idx, table, out, n, 3);       rep("uniform", 2L << 20, 2L << 20, 1, 8, "plain", t);
    gen(640L << 20, 640L << 20, 1, 8); t = run<8, 0>(idx, table, out, n, 3);      rep("uniform", 640L << 20, 640L << 20, 1, 8, "plain", t);
    return 0;
  }
  long sizes[] = {1, 2, 3, 4, 6, 8, 16, 32, 80, 256, 1024};
  for (long mb : sizes) {
    long tb = mb << 20;
    gen(tb, tb, 1, 8);
    rep("uniform", tb, tb, 1, 8, "plain", run<8, 0>(idx, table, out, n, 5));
    if (mb == 2 || mb == 80 || mb == 1024) {
      rep("uniform", tb, tb, 1, 8, "nt", run<8, 1>(idx, table, out, n, 5));
      rep("uniform", tb, tb, 1, 8, "sc1", run<8, 2>(idx, table, out, n, 5));
      rep("uniform", tb, tb, 1, 8, "sc0sc1", run<8, 3>(idx, table, out, n, 5));
      gen(tb, tb, 1, 4);  rep("uniform", tb, tb, 1, 4, "plain", run<4, 0>(idx, table, out, n, 5));
      gen(tb, tb, 1, 16); rep("uniform", tb, tb, 1, 16, "plain", run<16, 0>(idx, table, out, n, 5));
    }
  }
  // tiled: 80 MB table, per-XCD windows
  long wins[] = {256L << 10, 512L << 10, 1L << 20, 2L << 20, 3L << 20, 4L << 20, 8L << 20};
  int Ks[] = {64, 1024};
  for (long w : wins) for (int K : Ks) {
    gen(80L << 20, w, K, 8);
    rep("tiled", 80L << 20, w, K, 8, "plain", run<8, 0>(idx, table, out, n, 5));
  }
  return 0;
}
