// probe_ldsdma.hip -- tuning probe (not product): how fast does a CU pull an L2-resident slice into LDS?
//   mode 0  global_load_lds_dwordx4 (LDS DMA: 16 bytes per lane land in LDS, no registers)
//   mode 1  global_load_dwordx4 into registers, then ds_write_b128
//   mode 2  global_load_dwordx4 into registers only (summed)
// One 1024-thread workgroup per CU; every wave moves 1 KiB per step from an 8 MB table (config 3's x) with DEPTH steps in flight;
// waves per workgroup that take part: 16, 4, 1.   hipcc --offload-arch=gfx950 -O3 -o tools/probe_ldsdma tools/probe_ldsdma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void pull(const double* __restrict__ table, long table_doubles, int steps, int active_waves, double* __restrict__ out)
{
  __shared__ __attribute__((aligned(16))) double buf[DEPTH][2048];      // DEPTH slices of 16 KiB
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  if (wave >= active_waves) return;
  double acc = 0.0;
  // wave w of block b walks the table in 16 KiB slices; its 1 KiB piece of slice s: doubles [2048 s + 128 w, +128)
  long s0 = ((long)blockIdx.x * 7919) % (table_doubles / 2048);
  v2d r[DEPTH];
  for (int i = 0; i < steps; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long s = (s0 + i + d) % (table_doubles / 2048);
      const double* src = table + s * 2048 + 128 * wave + 2 * lane;
      if (MODE == 0)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)(&buf[d][128 * wave]), 16, 0, 0);
      else
        r[d] = *reinterpret_cast<const v2d*>(src);
    }
    if (MODE == 0) {
      __builtin_amdgcn_s_waitcnt(0 | (0x7 << 4) | (0xF << 8));        // vmcnt(0)
      acc += buf[i % DEPTH][128 * wave + lane];
    } else {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        if (MODE == 1) *reinterpret_cast<v2d*>(&buf[d][128 * wave + 2 * lane]) = r[d];
        else acc += r[d].x + r[d].y;
      }
      if (MODE == 1) acc += buf[i % DEPTH][128 * wave + ((lane * 7) & 127)];
    }
  }
  out[(long)blockIdx.x * 1024 + t] = acc;
}

template <int MODE, int DEPTH>
void run(const double* table, long td, double* out, int active)
{
  const int steps = 4096, blocks = 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((pull<MODE, DEPTH>), dim3(blocks), dim3(1024), 0, 0, table, td, steps, active, out);
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < 5; i++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((pull<MODE, DEPTH>), dim3(blocks), dim3(1024), 0, 0, table, td, steps, active, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  const double bytes = (double)blocks * active * 1024.0 * steps;
  printf("{\"probe\":\"ldsdma\",\"mode\":\"%s\",\"steps_in_flight\":%d,\"waves\":%d,\"ms\":%.4f,\"TBs\":%.3f,\"bytes_per_clk_per_CU_at_2.1GHz\":%.1f}\n",
         MODE == 0 ? "global_load_lds_dwordx4" : (MODE == 1 ? "global_load_dwordx4 + ds_write_b128" : "global_load_dwordx4 only"), DEPTH, active,
         ts[2], bytes / ts[2] / 1e9, bytes / blocks / (ts[2] * 1e-3 * 2.1e9));
  fflush(stdout);
}

int main()
{
  const long td = 1L << 20;        // 8 MB of doubles
  double *table, *out;
  CK(hipMalloc(&table, td * 8)); CK(hipMalloc(&out, 256L * 1024 * 8));
  CK(hipMemset(table, 0, td * 8));
  for (int active : {16, 4, 1}) {
    run<0, 3>(table, td, out, active);
    run<1, 3>(table, td, out, active);
    run<2, 3>(table, td, out, active);
  }
  run<0, 6>(table, td, out, 16);
  run<1, 6>(table, td, out, 16);
  run<2, 6>(table, td, out, 16);
  return 0;
}
