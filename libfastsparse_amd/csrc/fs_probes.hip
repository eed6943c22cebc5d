// fs_probes.hip -- two hardware probes behind DESIGN.md section 4 ("why config 2 sits at 0.30 of peak"), inside the library so
// that bench.py can RUN them next to the product instead of quoting numbers from earlier rounds.  Diagnostics:
// not declared in include/fastsparse_hip.h, not used by any product path.  The stand-alone forms with every variant are
// tools/probe_gather.hip and tools/probe_mix.hip; these are their headline shapes.
//
//   fs_debug_probe_gather  n independent 8-byte gathers, 8 in flight per lane, from a table of `table_bytes`; the workgroups that
//                          run together on one XCD (blockIdx % 8 equal) draw from one window of `window_bytes` for K consecutive
//                          workgroups -- window = table: uniformly random columns over all of x (one fabric request per gather);
//                          window <= 2 MiB: every gather an L2 hit (the best any row-panel x column-band tiling can arrange)
//   fs_debug_probe_mix     the same L2-resident gathers NEXT TO the 12-byte entry stream of a CSR (4-byte index + 8-byte value per
//                          gather, 16-byte loads) in one kernel and no y at all: what a gather kernel's inner loop costs at least
#include <algorithm>
#include <vector>

#include "fs_common.h"

namespace {

__host__ __device__ inline uint64_t probe_mix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void probe_idx_kernel(int *idx, int64_t n, int64_t table_elems, int64_t window_elems, int K, int per_block)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / per_block;
  const int64_t nwin = table_elems / window_elems;
  const int64_t win = ((b % 8) + 8 * ((b / 8) / K)) % nwin;
  const uint64_t h = probe_mix64((uint64_t)i * 0x9E3779B97F4A7C15ull + 12345);
  idx[i] = (int)(win * window_elems + (int64_t)(((unsigned __int128)h * (uint64_t)window_elems) >> 64));
}

constexpr int kU = 8;

__global__ __launch_bounds__(256) void probe_gather_kernel(const int *__restrict__ idx, const double *__restrict__ table, double *__restrict__ out)
{
  const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kU;
  int ix[kU];
#pragma unroll
  for (int u = 0; u < kU; u += 4) {
    const int4 v = *reinterpret_cast<const int4 *>(idx + base + u);
    ix[u] = v.x; ix[u + 1] = v.y; ix[u + 2] = v.z; ix[u + 3] = v.w;
  }
  double w[kU], acc = 0;
#pragma unroll
  for (int u = 0; u < kU; ++u) w[u] = table[ix[u]];
#pragma unroll
  for (int u = 0; u < kU; ++u) acc += w[u];
  out[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef int probe_v4i __attribute__((ext_vector_type(4)));
typedef double probe_v2d __attribute__((ext_vector_type(2)));

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void probe_mix_kernel(const int *__restrict__ idx, const double *__restrict__ vals, const double *__restrict__ table,
                                                          double *__restrict__ out, int64_t chunks_per_block)
{
  const int t = threadIdx.x;
  double acc = 0;
  for (int64_t c = 0; c < chunks_per_block; ++c) {
    const int64_t base = ((int64_t)blockIdx.x * chunks_per_block + c) * (int64_t)(BLOCK * kU);
    int ix[kU];
    double v[kU];
#pragma unroll
    for (int u = 0; u < kU; u += 4) {
      const int64_t e = base + (int64_t)(u / 4) * BLOCK * 4 + t * 4;
      const probe_v4i a = *(const probe_v4i *)(idx + e);
      ix[u] = a.x; ix[u + 1] = a.y; ix[u + 2] = a.z; ix[u + 3] = a.w;
      const probe_v2d p0 = *(const probe_v2d *)(vals + e), p1 = *(const probe_v2d *)(vals + e + 2);
      v[u] = p0.x; v[u + 1] = p0.y; v[u + 2] = p1.x; v[u + 3] = p1.y;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) acc += table[ix[u]] * v[u];
  }
  out[(int64_t)blockIdx.x * BLOCK + t] = acc;
}

struct ProbeBuffers {
  int *idx = nullptr;
  double *vals = nullptr, *table = nullptr, *out = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  ~ProbeBuffers()
  {
    for (void *p : {(void *)idx, (void *)vals, (void *)table, (void *)out})
      if (p) (void)hipFree(p);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
};

template <typename Launch>
int median_ms(Launch launch, int iters, ProbeBuffers &B, float *out_ms)
{
  for (int i = 0; i < 2; ++i) launch();
  FS_HIP(hipGetLastError());
  FS_HIP(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) {
    FS_HIP(hipEventRecord(B.a, nullptr));
    launch();
    FS_HIP(hipEventRecord(B.b, nullptr));
    FS_HIP(hipEventSynchronize(B.b));
    float ms = 0.f;
    FS_HIP(hipEventElapsedTime(&ms, B.a, B.b));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  *out_ms = ts[ts.size() / 2];
  return FS_OK;
}

}  // namespace

extern "C" {

// median ms of `iters` launches of n gathers (n is rounded down to a multiple of 4096)
int fs_debug_probe_gather(int64_t n, int64_t table_bytes, int64_t window_bytes, int K, int iters, float *ms)
{
  if (n < 4096 || table_bytes < 4096 || window_bytes < 4096 || window_bytes > table_bytes || K < 1 || iters < 1 || !ms) {
    fs::set_error("fs_debug_probe_gather: bad argument");
    return FS_ERR_ARG;
  }
  n = n / 4096 * 4096;
  ProbeBuffers B;
  FS_HIP(hipMalloc(&B.idx, sizeof(int) * (size_t)n));
  FS_HIP(hipMalloc(&B.table, (size_t)table_bytes));
  FS_HIP(hipMalloc(&B.out, sizeof(double) * (size_t)(n / kU)));
  FS_HIP(hipMemset(B.table, 0, (size_t)table_bytes));
  FS_HIP(hipEventCreate(&B.a));
  FS_HIP(hipEventCreate(&B.b));
  hipLaunchKernelGGL(probe_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, B.idx, n, table_bytes / 8, window_bytes / 8, K, 256 * kU);
  FS_HIP(hipGetLastError());
  const unsigned blocks = (unsigned)(n / (256 * kU));
  return median_ms([&] { hipLaunchKernelGGL(probe_gather_kernel, dim3(blocks), dim3(256), 0, nullptr, B.idx, B.table, B.out); }, iters, B, ms);
}

// median ms of n gathers from an L2-resident table of `table_bytes` next to their 12-byte entry stream; block = 256 (many small
// workgroups, blocks = 0) or 512 / 1024 with `blocks` persistent workgroups (the shape an LDS-resident y slice forces)
int fs_debug_probe_mix(int64_t n, int64_t table_bytes, int block, int blocks, int iters, float *ms)
{
  if (n < (1 << 20) || table_bytes < 4096 || iters < 1 || !ms || (block != 256 && block != 512 && block != 1024)) {
    fs::set_error("fs_debug_probe_mix: bad argument");
    return FS_ERR_ARG;
  }
  n = n / (1 << 20) * (1 << 20);
  if (blocks <= 0) blocks = (int)(n / ((int64_t)block * kU));
  const int64_t cpb = n / ((int64_t)blocks * block * kU);
  if (cpb < 1) { fs::set_error("fs_debug_probe_mix: too many workgroups for n"); return FS_ERR_ARG; }
  ProbeBuffers B;
  FS_HIP(hipMalloc(&B.idx, sizeof(int) * (size_t)n));
  FS_HIP(hipMalloc(&B.vals, sizeof(double) * (size_t)n));
  FS_HIP(hipMalloc(&B.table, (size_t)table_bytes));
  FS_HIP(hipMalloc(&B.out, sizeof(double) * (size_t)blocks * (size_t)block));
  FS_HIP(hipMemset(B.table, 0, (size_t)table_bytes));
  FS_HIP(hipMemset(B.vals, 0, sizeof(double) * (size_t)n));
  FS_HIP(hipEventCreate(&B.a));
  FS_HIP(hipEventCreate(&B.b));
  hipLaunchKernelGGL(probe_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, B.idx, n, table_bytes / 8, table_bytes / 8, 1, 256 * kU);
  FS_HIP(hipGetLastError());
  auto launch = [&] {
    if (block == 256) hipLaunchKernelGGL(probe_mix_kernel<256>, dim3(blocks), dim3(256), 0, nullptr, B.idx, B.vals, B.table, B.out, cpb);
    else if (block == 512) hipLaunchKernelGGL(probe_mix_kernel<512>, dim3(blocks), dim3(512), 0, nullptr, B.idx, B.vals, B.table, B.out, cpb);
    else hipLaunchKernelGGL(probe_mix_kernel<1024>, dim3(blocks), dim3(1024), 0, nullptr, B.idx, B.vals, B.table, B.out, cpb);
  };
  return median_ms(launch, iters, B, ms);
}

}  // extern "C"
