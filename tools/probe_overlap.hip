// probe_overlap.hip -- tuning probe (not product): do L2-hit gathers and an HBM read+write stream overlap when they run in ONE
// launch (workgroups of both roles resident on every CU)?  Role G: 8-byte gathers from a 2 MiB table (resident in every L2) driven
// by a 4-byte index stream + an 8-byte value stream (what a gather SpMV kernel does per entry).  Role S: per "entry" 10 bytes read
// and 8 bytes written (what pass 1 of the two-pass SpMV moves).  Times: G alone, S alone, both in one grid (roles alternate in groups of 8 blocks).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned short v4h __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void make_idx(int* idx, long n, long table_elems) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (int)(((unsigned __int128)splitmix64((uint64_t)i * 77 + 5) * (uint64_t)table_elems) >> 64);
}

constexpr int BLOCK = 256;
// mode bit 0: gather role present, bit 1: stream role present.  Blocks with (blockIdx & 1) == 0 gather, the others stream (when
// only one role is present all blocks take it).  Each role's blocks share its n entries equally, 4 per thread and step.
template <bool SNT>
__global__ __launch_bounds__(BLOCK) void overlap(int mode, const int* __restrict__ idx, const double* __restrict__ vals, const double* __restrict__ table,
                                                 double* __restrict__ gout, long ng, const unsigned short* __restrict__ scol, const double* __restrict__ sval,
                                                 double* __restrict__ sout, long ns) {
  const int t = threadIdx.x;
  const bool both = mode == 3;
  // consecutive blocks go round the 8 XCDs: roles alternate in groups of 8 blocks, so that each role runs on every XCD
  const bool gather = both ? ((blockIdx.x >> 3) & 1) == 0 : mode == 1;
  const long nb = both ? gridDim.x / 2 : gridDim.x, b = both ? ((blockIdx.x >> 4) << 3) + (blockIdx.x & 7) : blockIdx.x;
  if (gather) {
    const long per = ng / nb / (BLOCK * 4) * (BLOCK * 4);
    double acc = 0;
    for (long e0 = b * per; e0 < (b + 1) * per; e0 += BLOCK * 8) {
      v4i a[2]; v2d p[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long e = e0 + u * BLOCK * 4 + t * 4;
        a[u] = __builtin_nontemporal_load((const v4i*)(idx + e));
        p[u][0] = __builtin_nontemporal_load((const v2d*)(vals + e));
        p[u][1] = __builtin_nontemporal_load((const v2d*)(vals + e + 2));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        acc += table[a[u].x] * p[u][0].x + table[a[u].y] * p[u][0].y + table[a[u].z] * p[u][1].x + table[a[u].w] * p[u][1].y;
    }
    gout[(long)b * BLOCK + t] = acc;
  } else {
    const long per = ns / nb / (BLOCK * 4) * (BLOCK * 4);
    for (long e0 = b * per; e0 < (b + 1) * per; e0 += BLOCK * 8) {
      v4h c[2]; v2d p[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long e = e0 + u * BLOCK * 4 + t * 4;
        if (SNT) {
          c[u] = __builtin_nontemporal_load((const v4h*)(scol + e));
          p[u][0] = __builtin_nontemporal_load((const v2d*)(sval + e));
          p[u][1] = __builtin_nontemporal_load((const v2d*)(sval + e + 2));
        } else {
          c[u] = *(const v4h*)(scol + e);
          p[u][0] = *(const v2d*)(sval + e);
          p[u][1] = *(const v2d*)(sval + e + 2);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long e = e0 + u * BLOCK * 4 + t * 4;
        v2d q0 = {p[u][0].x * (double)c[u].x, p[u][0].y * (double)c[u].y}, q1 = {p[u][1].x * (double)c[u].z, p[u][1].y * (double)c[u].w};
        __builtin_nontemporal_store(q0, (v2d*)(sout + e));
        __builtin_nontemporal_store(q1, (v2d*)(sout + e + 2));
      }
    }
  }
}

template <bool SNT>
static float run(int mode, long blocks, const int* idx, const double* vals, const double* table, double* gout, long ng, const unsigned short* scol,
                 const double* sval, double* sout, long ns) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; i++) hipLaunchKernelGGL(overlap<SNT>, dim3(blocks), dim3(BLOCK), 0, 0, mode, idx, vals, table, gout, ng, scol, sval, sout, ns);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < 5; i++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(overlap<SNT>, dim3(blocks), dim3(BLOCK), 0, 0, mode, idx, vals, table, gout, ng, scol, sval, sout, ns);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[2];
}

int main(int argc, char** argv) {
  const long N = 160L << 20;
  const long table_elems = (2L << 20) / 8;
  int* idx; double *vals, *table, *gout, *sval, *sout; unsigned short* scol;
  CK(hipMalloc(&idx, N * 4)); CK(hipMalloc(&vals, N * 8)); CK(hipMalloc(&table, table_elems * 8)); CK(hipMalloc(&gout, (1 << 24) * 8));
  CK(hipMalloc(&scol, N * 2)); CK(hipMalloc(&sval, N * 8)); CK(hipMalloc(&sout, N * 8));
  CK(hipMemset(table, 0, table_elems * 8)); CK(hipMemset(vals, 0, N * 8)); CK(hipMemset(scol, 0, N * 2)); CK(hipMemset(sval, 0, N * 8));
  hipLaunchKernelGGL(make_idx, dim3((N + 255) / 256), dim3(256), 0, 0, idx, N, table_elems);
  CK(hipDeviceSynchronize());
  for (long blocks : {2048L, 32768L}) {
    for (int gshare = 2; gshare <= 8; gshare += 2) {        // tenths of the 160 M entries that take the gather role
      const long ng = N / 10 * gshare, ns = N - ng;
      const float tg = run<false>(1, blocks / 2, idx, vals, table, gout, ng, scol, sval, sout, ns);
      const float tsm = run<false>(2, blocks / 2, idx, vals, table, gout, ng, scol, sval, sout, ns);
      const float tb = run<false>(3, blocks, idx, vals, table, gout, ng, scol, sval, sout, ns);
      const float tsn = run<true>(2, blocks / 2, idx, vals, table, gout, ng, scol, sval, sout, ns);
      const float tbn = run<true>(3, blocks, idx, vals, table, gout, ng, scol, sval, sout, ns);
      printf("{\"probe\":\"overlap\",\"blocks\":%ld,\"gather_share\":%.1f,\"gather_M\":%.0f,\"stream_M\":%.0f,\"ms_gather_alone\":%.4f,\"ms_stream_alone\":%.4f,"
             "\"ms_both_one_launch\":%.4f,\"ms_stream_alone_nt_loads\":%.4f,\"ms_both_one_launch_nt_loads\":%.4f}\n", blocks, gshare / 10.0, ng / 1e6, ns / 1e6, tg,
             tsm, tb, tsn, tbn);
      fflush(stdout);
    }
  }
  return 0;
}
