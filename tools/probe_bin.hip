// probe_bin.hip -- tuning probe (not product): the traffic pattern of a two-pass ("expand, then reduce") SpMV.
//   pass 1 (expand): a workgroup holds one band of x in LDS (XB doubles), streams its entries' 16-bit local column ids,
//                    gathers from LDS and writes the 8-byte products to HBM in runs (scattered by groups of GROUP entries);
//   pass 2 (reduce): a workgroup owns a row panel (R rows, y tile in LDS), streams products + 16-bit local row ids and adds
//                    them into the tile with ds_add_f64, then writes the tile.
// Indices are synthetic and uniform; the question is only how fast the two streaming passes run.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void make_u16(uint16_t* idx, long n, unsigned range, uint64_t salt) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (uint16_t)(((unsigned __int128)splitmix64((uint64_t)i * 77 + salt) * (uint64_t)range) >> 64);
}
__global__ void fill(double* p, long n, double v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v + (double)(i & 7);
}

constexpr int GROUP = 256;   // entries written contiguously before jumping elsewhere

// pass 1.  BLOCK threads; segment of `seg` entries per workgroup (multiple of BLOCK*8).  Lane handles 8 consecutive entries
// (one 16-byte id load), gathers 8 doubles from LDS, stores 4 x 16 bytes.
template <int BLOCK, int XB>
__global__ __launch_bounds__(BLOCK) void expand(const uint16_t* __restrict__ lcol, const double* __restrict__ x, double* __restrict__ prod,
                                                long seg, long ngroups, long nbands) {
  extern __shared__ double xs[];
  const int t = threadIdx.x;
  const long band = blockIdx.x % nbands;
  for (int i = t * 2; i < XB; i += BLOCK * 2) *(v2d*)(xs + i) = __builtin_nontemporal_load((const v2d*)(x + band * XB + i));
  __syncthreads();
  const long base = (long)blockIdx.x * seg;
  for (long o = 0; o < seg; o += (long)BLOCK * 8) {
    const long e = base + o + (long)t * 8;
    v4u a = __builtin_nontemporal_load((const v4u*)(lcol + e));
    double p[8];
    p[0] = xs[a.x & 0xffff]; p[1] = xs[a.x >> 16]; p[2] = xs[a.y & 0xffff]; p[3] = xs[a.y >> 16];
    p[4] = xs[a.z & 0xffff]; p[5] = xs[a.z >> 16]; p[6] = xs[a.w & 0xffff]; p[7] = xs[a.w >> 16];
    const long g = e / GROUP;
    const long gd = (g * 1000003L) % ngroups;       // scattered destination of this group
    double* d = prod + gd * GROUP + (e % GROUP);
#pragma unroll
    for (int k = 0; k < 8; k += 2) { v2d w = {p[k], p[k + 1]}; __builtin_nontemporal_store(w, (v2d*)(d + k)); }
  }
}

// pass 1, variant B: lane handles 2 consecutive entries per step (4-byte id load, one fully coalesced 16-byte store), 4 steps in flight
template <int BLOCK, int XB>
__global__ __launch_bounds__(BLOCK) void expand2(const uint16_t* __restrict__ lcol, const double* __restrict__ x, double* __restrict__ prod,
                                                 long seg, long ngroups, long nbands) {
  extern __shared__ double xs[];
  const int t = threadIdx.x;
  const long band = blockIdx.x % nbands;
  for (int i = t * 2; i < XB; i += BLOCK * 2) *(v2d*)(xs + i) = __builtin_nontemporal_load((const v2d*)(x + band * XB + i));
  __syncthreads();
  const long base = (long)blockIdx.x * seg;
  for (long o = 0; o < seg; o += (long)BLOCK * 8) {
    unsigned a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = __builtin_nontemporal_load((const unsigned*)(lcol + base + o + (long)k * BLOCK * 2 + t * 2));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long e = base + o + (long)k * BLOCK * 2 + t * 2;
      const long g = e / GROUP;
      const long gd = (g * 1000003L) % ngroups;
      v2d w = {xs[a[k] & 0xffff], xs[a[k] >> 16]};
      __builtin_nontemporal_store(w, (v2d*)(prod + gd * GROUP + (e % GROUP)));
    }
  }
}

// pass 2.  One workgroup per panel of R rows; `per_panel` entries each (multiple of BLOCK*8).
template <int BLOCK, int R>
__global__ __launch_bounds__(BLOCK) void reduce(const uint16_t* __restrict__ lrow, const double* __restrict__ prod, double* __restrict__ y, long per_panel) {
  extern __shared__ double yt[];
  const int t = threadIdx.x;
  for (int i = t; i < R; i += BLOCK) yt[i] = 0.0;
  __syncthreads();
  const long base = (long)blockIdx.x * per_panel;
  for (long o = 0; o < per_panel; o += (long)BLOCK * 8) {
    const long e = base + o + (long)t * 8;
    if (e >= base + per_panel) break;
    v4u a = __builtin_nontemporal_load((const v4u*)(lrow + e));
    v2d p0 = __builtin_nontemporal_load((const v2d*)(prod + e));
    v2d p1 = __builtin_nontemporal_load((const v2d*)(prod + e + 2));
    v2d p2 = __builtin_nontemporal_load((const v2d*)(prod + e + 4));
    v2d p3 = __builtin_nontemporal_load((const v2d*)(prod + e + 6));
    __hip_atomic_fetch_add(yt + (a.x & 0xffff), p0.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(yt + (a.x >> 16), p0.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(yt + (a.y & 0xffff), p1.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(yt + (a.y >> 16), p1.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(yt + (a.z & 0xffff), p2.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(yt + (a.z >> 16), p2.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(yt + (a.w & 0xffff), p3.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(yt + (a.w >> 16), p3.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  for (int i = t; i < R; i += BLOCK) y[((long)blockIdx.x * R + i) % (10L << 20)] = yt[i];   // y holds 10 Mi rows
}

// lrow ascending inside runs of `runlen` entries (what a (band, panel) run looks like)
__global__ void make_lrow_runs(uint16_t* idx, long n, int runlen, int rows) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long k = i % runlen;
  const unsigned step = rows / runlen;
  idx[i] = (uint16_t)((k * step + splitmix64((uint64_t)i * 31 + 7) % step) % rows);
}
// gdst: runs of `rungroups` groups go to scattered places
__global__ void make_gdst(unsigned* gd, long ngroups, int rungroups) {
  long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  const long nruns = ngroups / rungroups;
  const long r = g / rungroups;
  if (r >= nruns) { gd[g] = (unsigned)g; return; }
  const long rd = (r * 1000003L) % nruns;
  gd[g] = (unsigned)(rd * rungroups + g % rungroups);
}

// pass 1 as the product does it: persistent workgroups, gdst table, 2 entries per lane and step
template <int BLOCK, int XB>
__global__ __launch_bounds__(BLOCK) void expand3(const uint16_t* __restrict__ lcol, const unsigned* __restrict__ gdst, const double* __restrict__ x,
                                                 double* __restrict__ prod, long n, long per_band, long nbands) {
  extern __shared__ double xs[];
  const int t = threadIdx.x;
  const long e_lo = n / gridDim.x * blockIdx.x, e_hi = n / gridDim.x * (blockIdx.x + 1);
  for (long e0 = e_lo; e0 < e_hi;) {
    const long band = e0 / per_band;
    long e1 = (band + 1) * per_band; if (e1 > e_hi) e1 = e_hi;
    __syncthreads();
    for (int i = t; i < XB; i += BLOCK) xs[i] = __builtin_nontemporal_load(x + (band % nbands) * XB + i);
    __syncthreads();
    for (long o = e0 + 2 * t; o < e1; o += (long)BLOCK * 8) {
      unsigned a[4], d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long e = o + (long)k * BLOCK * 2;
        const bool ok = e < e1;
        a[k] = ok ? __builtin_nontemporal_load((const unsigned*)(lcol + e)) : 0u;
        d[k] = ok ? __builtin_nontemporal_load(gdst + (e >> 3)) : 0u;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long e = o + (long)k * BLOCK * 2;
        v2d w = {xs[a[k] & 0xffff], xs[a[k] >> 16]};
        if (e < e1) __builtin_nontemporal_store(w, (v2d*)(prod + (long)d[k] * 8 + (e & 7)));
      }
    }
    e0 = e1;
  }
}

template <typename F>
float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms / reps;
}

template <int BLOCK, int XB, bool V2>
void run_expand(const uint16_t* lcol, const double* x, double* prod, long n, long seg, long ncol) {
  const long blocks = n / seg, ngroups = n / GROUP, nbands = ncol / XB;
  CK(hipFuncSetAttribute((const void*)(V2 ? expand2<BLOCK, XB> : expand<BLOCK, XB>), hipFuncAttributeMaxDynamicSharedMemorySize, XB * 8));
  float ms = timeit([&] {
    if (V2) expand2<BLOCK, XB><<<blocks, BLOCK, XB * 8>>>(lcol, x, prod, seg, ngroups, nbands);
    else expand<BLOCK, XB><<<blocks, BLOCK, XB * 8>>>(lcol, x, prod, seg, ngroups, nbands);
  }, 5);
  const double bytes = n * 10.0 + (double)blocks * XB * 8;
  printf("{\"pass\": \"expand%s\", \"block\": %d, \"xband\": %d, \"seg\": %ld, \"wgs\": %ld, \"ms\": %.4f, \"GBps\": %.0f}\n", V2 ? "2" : "", BLOCK, XB, seg,
         blocks, ms, bytes / ms * 1e-6);
  fflush(stdout);
}

template <int BLOCK, int R>
void run_reduce(const uint16_t* lrow, const double* prod, double* y, long n, long nrow) {
  const long panels = nrow / R;
  long per = n / panels; per -= per % (BLOCK * 8);
  CK(hipFuncSetAttribute((const void*)reduce<BLOCK, R>, hipFuncAttributeMaxDynamicSharedMemorySize, R * 8));
  float ms = timeit([&] { reduce<BLOCK, R><<<panels, BLOCK, R * 8>>>(lrow, prod, y, per); }, 5);
  const double bytes = (double)per * panels * 10.0 + (double)panels * R * 8;
  printf("{\"pass\": \"reduce\", \"block\": %d, \"rows\": %d, \"panels\": %ld, \"per_panel\": %ld, \"ms\": %.4f, \"GBps\": %.0f}\n", BLOCK, R, panels, per, ms,
         bytes / ms * 1e-6);
  fflush(stdout);
}

template <int BLOCK, int R>
void run_reduce_p(const char* tag, const uint16_t* lrow, const double* prod, double* y, long panels, long per) {
  CK(hipFuncSetAttribute((const void*)reduce<BLOCK, R>, hipFuncAttributeMaxDynamicSharedMemorySize, R * 8));
  float ms = timeit([&] { reduce<BLOCK, R><<<panels, BLOCK, R * 8>>>(lrow, prod, y, per); }, 5);
  const double bytes = (double)per * panels * 10.0 + (double)panels * R * 8;
  printf("{\"pass\": \"reduce %s\", \"block\": %d, \"rows\": %d, \"panels\": %ld, \"per_panel\": %ld, \"ms\": %.4f, \"GBps\": %.0f}\n", tag, BLOCK, R, panels, per,
         ms, bytes / ms * 1e-6);
  fflush(stdout);
}

template <int BLOCK, int XB>
void run_expand3(const char* tag, const uint16_t* lcol, const unsigned* gd, const double* x, double* prod, long n, long per_band, long ncol, int wgs) {
  CK(hipFuncSetAttribute((const void*)expand3<BLOCK, XB>, hipFuncAttributeMaxDynamicSharedMemorySize, XB * 8));
  float ms = timeit([&] { expand3<BLOCK, XB><<<wgs, BLOCK, XB * 8>>>(lcol, gd, x, prod, n, per_band, ncol / XB); }, 5);
  const double bytes = n * 10.5 + (double)(n / per_band + wgs) * XB * 8;
  printf("{\"pass\": \"expand3 %s\", \"wgs\": %d, \"per_band\": %ld, \"ms\": %.4f, \"GBps\": %.0f}\n", tag, wgs, per_band, ms, bytes / ms * 1e-6);
  fflush(stdout);
}

int main() {
  const long n = 160L << 20;         // 167.8M entries
  const long ncol = 10L << 20, nrow = 10L << 20;
  uint16_t *lcol, *lrow; double *x, *y, *prod;
  CK(hipMalloc(&lcol, n * 2 + 64)); CK(hipMalloc(&lrow, n * 2 + 64));
  CK(hipMalloc(&x, ncol * 8)); CK(hipMalloc(&y, nrow * 8)); CK(hipMalloc(&prod, n * 8 + 64));
  fill<<<(ncol + 255) / 256, 256>>>(x, ncol, 1.0);
  make_u16<<<(n + 255) / 256, 256>>>(lcol, n, 8192, 5);     // valid for every band size tried (>= 8192)
  make_u16<<<(n + 255) / 256, 256>>>(lrow, n, 8192, 9);     // valid for every panel size tried (>= 8192)
  CK(hipDeviceSynchronize());

  run_expand<1024, 16384, false>(lcol, x, prod, n, 65536, ncol);
  run_expand<1024, 16384, false>(lcol, x, prod, n, 131072, ncol);
  run_expand<1024, 16384, true>(lcol, x, prod, n, 131072, ncol);
  run_expand<512, 8192, false>(lcol, x, prod, n, 65536, ncol);
  run_expand<512, 8192, true>(lcol, x, prod, n, 65536, ncol);
  run_expand<256, 8192, false>(lcol, x, prod, n, 32768, ncol);
  run_expand<1024, 8192, false>(lcol, x, prod, n, 65536, ncol);

  run_reduce<1024, 8192>(lrow, prod, y, n, nrow);
  run_reduce<512, 8192>(lrow, prod, y, n, nrow);
  run_reduce<1024, 16384>(lrow, prod, y, n, nrow);
  run_reduce<256, 8192>(lrow, prod, y, n, nrow);
  // closer to the product
  unsigned* gd; CK(hipMalloc(&gd, n / 8 * 4));
  for (int rg : {21, 22, 32, 42, 64}) {
    make_gdst<<<(n / 8 + 255) / 256, 256>>>(gd, n / 8, rg);
    char tag[64]; snprintf(tag, sizeof tag, "runs%d persistent", rg * 8);
    run_expand3<1024, 16384>(tag, lcol, gd, x, prod, n, 262144, ncol, 256);
  }
  make_gdst<<<(n / 8 + 255) / 256, 256>>>(gd, n / 8, 42);
  run_expand3<1024, 16384>("runs336 wgs512", lcol, gd, x, prod, n, 262144, ncol, 512);
  // per_panel * panels <= n = 167772160 and every e + 8 <= n; y index wraps inside the kernel
  run_reduce_p<1024, 8192>("p1536", lrow, prod, y, 1536, 106496);
  run_reduce_p<1024, 8192>("p1536 ragged", lrow, prod, y, 1536, 109000 / 8 * 8);
  run_reduce_p<1024, 8192>("p2560", lrow, prod, y, 2560, 65536);
  run_reduce_p<1024, 16384>("p768", lrow, prod, y, 768, 212992);
  run_reduce_p<1024, 16384>("p512", lrow, prod, y, 512, 327680);
  run_reduce_p<1024, 16384>("p1024", lrow, prod, y, 1024, 163840);
  make_lrow_runs<<<(n + 255) / 256, 256>>>(lrow, n, 170, 6500);
  run_reduce_p<1024, 8192>("p1536 ascending runs", lrow, prod, y, 1536, 106496);
  run_reduce_p<1024, 8192>("p1280 ascending runs", lrow, prod, y, 1280, 131072);
  make_lrow_runs<<<(n + 255) / 256, 256>>>(lrow, n, 340, 13000);
  run_reduce_p<1024, 16384>("p768 ascending runs", lrow, prod, y, 768, 212992);
  double h[4]; CK(hipMemcpy(h, y, 32, hipMemcpyDeviceToHost));
  printf("{\"check\": %.1f}\n", h[0] + h[1]);
  return 0;
}
