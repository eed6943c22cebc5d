// fs_kernels_twopass.hip -- the kernels whose every random access is an LDS access: the two-pass SpMV (expand by column band,
// reduce by row panel, a sequential stream of products between them; the fixed-order pass 2), the long-row side path, the
// k-column sweeps for k = 2, 4, and their launchers.  Split from fs_kernels.hip in round 4.
#include <stdlib.h>

#include "fs_kernel_util.h"

namespace fs {

// ------------------------------------------------------------------------------------------
// y = A x in two streaming passes (BinnedCsr in fs_common.h).  Same callers as the kernels above.
//
// Why: with 16 entries per row and a vector x of tens of MB, a gather kernel is bound by the rate at which
// L2 answers 8-byte requests (measured 240 G/s when every request hits, 53 G/s when every one goes to HBM),
// not by bytes.  Here every random access is an LDS access: pass 1 gathers x from a 128 KiB band held in LDS,
// pass 2 scatters into a 128 KiB slice of y held in LDS, and what travels between them is a sequential stream
// of products laid out so that each pass reads and writes whole lines.  20.5 bytes per entry at stream speed
// beat 4 bytes per entry at gather speed.
//
// Sum order: a row's terms are added band by band, and inside a band by LDS atomics in no fixed order --
// the result is exact for pattern matrices with integer-valued x and within the usual rounding bound
// otherwise; strict_order keeps the chunk-streaming kernel.
// ------------------------------------------------------------------------------------------

// pass 1: persistent workgroups, one per CU; workgroup w streams the w-th equal share of the (band, panel)-ordered
// groups and reloads its x band when the share crosses into the next band (every band is loaded once, plus once
// per share boundary: cutting bands into many small workgroups instead re-reads x several times over)
// xband: BC + 8 doubles of LDS; slot BC is the zero the padding entries point at.  The share is groups [g0, g1).
template <bool VALUED, int U, bool NTLD, bool NTST, int BC>
__device__ __forceinline__ void expand_share(double *__restrict__ xband, int ncol, int B, const unsigned *__restrict__ band_ptr,
                                             const uint16_t *__restrict__ lcol, const double *__restrict__ vals,
                                             const unsigned *__restrict__ gdst, const double *__restrict__ x, int xs,
                                             double *__restrict__ prod, unsigned g0, unsigned g1)
{
  const int t = threadIdx.x;
  if (g0 >= g1) return;
  // band of the first group: last b with band_ptr[b] <= g0
  int b;
  {
    int lo = 0, hi = B - 1;
    while (lo < hi) {
      const int mid = lo + ((hi - lo + 1) >> 1);
      if (band_ptr[mid] <= g0) lo = mid; else hi = mid - 1;
    }
    b = lo;
  }
  for (unsigned g = g0; g < g1; ++b) {
    const unsigned gb = band_ptr[b + 1] < g1 ? band_ptr[b + 1] : g1;   // end of this band's part of the share
    if (gb <= g) continue;                                               // empty band
    const int c0 = b * BC;
    const int w = (ncol - c0 < BC) ? ncol - c0 : BC;
    __syncthreads();                                                     // everyone is done with the previous band
    {
      // 16 loads per thread in flight together (clamped addresses, masking afterwards: a select next to the load
      // would make every one of them wait for itself); slots BC .. BC+7 are the zero padding points at
      double r[BC / kBinBlock];
#pragma unroll
      for (int j = 0; j < BC / kBinBlock; ++j) {
        const int i = j * kBinBlock + t;
        r[j] = __builtin_nontemporal_load(x + (int64_t)(c0 + (i < w ? i : w - 1)) * xs);
      }
#pragma unroll
      for (int j = 0; j < BC / kBinBlock; ++j) {
        const int i = j * kBinBlock + t;
        xband[i] = (i < w) ? r[j] : 0.0;
      }
      if (t < 8) xband[BC + t] = 0.0;
    }
    __syncthreads();
    const int64_t e0 = (int64_t)g * kBinGroup, e1 = (int64_t)gb * kBinGroup;
    // a lane takes 2 consecutive entries per step, so a wave's stores are 1 KiB of consecutive products; U steps in
    // flight.  Whole rounds (every step of every lane inside the segment) are straight-line code: all loads, a
    // scheduling barrier, then gathers and stores -- with a guard anywhere in it the compiler sinks the loads of a
    // step behind that step's guard and the steps run one after the other.  The last, partial round is guarded.
    constexpr int64_t kRound = 2 * U * kBinBlock;
    int64_t o = e0 + 2 * t;
    for (; o - 2 * t + kRound <= e1; o += kRound) {
      unsigned a[U], d[U];
      v2d v[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t e = o + (int64_t)k * 2 * kBinBlock;
        a[k] = stream_load<NTLD>((const unsigned *)(lcol + e));
        d[k] = stream_load<NTLD>(gdst + (e >> kBinGroupLog));
        if (VALUED) v[k] = stream_load<NTLD>((const v2d *)(vals + e));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t e = o + (int64_t)k * 2 * kBinBlock;
        v2d p = {xband[a[k] & 0xffffu], xband[a[k] >> 16]};
        if (VALUED) { p.x *= v[k].x; p.y *= v[k].y; }
        stream_store<NTST>(p, (v2d *)(prod + (int64_t)d[k] * kBinGroup + (e & (kBinGroup - 1))));
      }
    }
    for (; o < e1; o += 2 * kBinBlock) {
      const unsigned a = stream_load<NTLD>((const unsigned *)(lcol + o));
      const unsigned d = stream_load<NTLD>(gdst + (o >> kBinGroupLog));
      v2d p = {xband[a & 0xffffu], xband[a >> 16]};
      if (VALUED) {
        const v2d v = stream_load<NTLD>((const v2d *)(vals + o));
        p.x *= v.x; p.y *= v.y;
      }
      stream_store<NTST>(p, (v2d *)(prod + (int64_t)d * kBinGroup + (o & (kBinGroup - 1))));
    }
    g = gb;
  }
}

template <bool VALUED, int U, bool NTLD, bool NTST, int BC = kBinCols>
__global__ __launch_bounds__(kBinBlock) void spmv_expand_kernel(
    int ncol, int B, const unsigned *__restrict__ band_ptr, const uint16_t *__restrict__ lcol,
    const double *__restrict__ vals, const unsigned *__restrict__ gdst, const double *__restrict__ x, int xs,
    double *__restrict__ prod, unsigned gbeg, unsigned gend)
{
  __shared__ double xband[BC + 8];
  // this launch covers the groups gbeg .. gend (everything, or the bands whose part of x has arrived: fs_spmv_host)
  const uint64_t groups = gend - gbeg;
  const unsigned g0 = gbeg + (unsigned)(groups * blockIdx.x / gridDim.x), g1 = gbeg + (unsigned)(groups * (blockIdx.x + 1) / gridDim.x);
  expand_share<VALUED, U, NTLD, NTST, BC>(xband, ncol, B, band_ptr, lcol, vals, gdst, x, xs, prod, g0, g1);
}

// the local rows of the 8 entries e .. e + 7 (e a multiple of 8) as four pairs of 16-bit ids, in two steps so that the loads of
// several steps can be in flight before the first is used: row_ids_load issues the loads, row_ids decodes what they brought.
// L8 (BinnedCsr::lrow8): one byte per entry, the step from the slot before it; gbase[g] = the row in front of group g's first slot.
// A lane sums its 8 steps; the lane with the second half of a group (e = 8 mod 16: an ODD lane, since a panel's entries start on a
// group boundary and lane t takes e0 + 8 t + whole rounds) adds the first half's total, fetched from the lane below it (DPP
// row_shr:1 -- every lane of the wave executes the decode, and an odd lane inside the range has its even neighbour inside too).
template <bool NTLD, bool L8>
__device__ __forceinline__ v4u row_ids_load(const uint16_t *__restrict__ lrow, const uint8_t *__restrict__ lrow8,
                                            const uint16_t *__restrict__ gbase, int64_t e)
{
  if (!L8) return stream_load<NTLD>((const v4u *)(lrow + e));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const v2u d = stream_load<NTLD>((const v2u *)(lrow8 + e));
  return v4u{d.x, d.y, (unsigned)gbase[e >> kBinGroupLog], (unsigned)(e >> 3) & 1u};
}

template <bool L8>
__device__ __forceinline__ v4u row_ids(v4u raw)
{
  if (!L8) return raw;
  const unsigned dx = raw.x, dy = raw.y;
  const unsigned s0 = dx & 0xffu, s1 = s0 + ((dx >> 8) & 0xffu), s2 = s1 + ((dx >> 16) & 0xffu), s3 = s2 + (dx >> 24);
  const unsigned s4 = s3 + (dy & 0xffu), s5 = s4 + ((dy >> 8) & 0xffu), s6 = s5 + ((dy >> 16) & 0xffu), s7 = s6 + (dy >> 24);
  const unsigned below = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s7, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  const unsigned off = raw.z + (raw.w ? below : 0u);
  return v4u{(off + s0) | ((off + s1) << 16), (off + s2) | ((off + s3) << 16), (off + s4) | ((off + s5) << 16), (off + s6) | ((off + s7) << 16)};
}

// pass 2: workgroup = one row panel; its products are contiguous.  ytile: the panel's slice of y in LDS.
template <bool NTLD, bool L8 = false>
__device__ __forceinline__ void reduce_panel(double *__restrict__ ytile, int panel, const unsigned *__restrict__ bin_ptr,
                                             const int *__restrict__ panel_row, const uint16_t *__restrict__ lrow,
                                             const double *__restrict__ prod, double *__restrict__ y, int ys,
                                             const uint8_t *__restrict__ lrow8 = nullptr, const uint16_t *__restrict__ gbase = nullptr)
{
  const int t = threadIdx.x;
  const int r0 = panel_row[panel], nr = panel_row[panel + 1] - r0;
  for (int i = t; i < nr; i += kBinBlock) ytile[i] = 0.0;
  __syncthreads();
  const int64_t e0 = (int64_t)bin_ptr[panel] * kBinGroup, e1 = (int64_t)bin_ptr[panel + 1] * kBinGroup;
  // 8 entries (64 bytes of products) per lane and step, two steps in flight in whole rounds (straight-line code:
  // see expand_share); the last, partial round is guarded
#define FS_ADD(idx, val) __hip_atomic_fetch_add(&ytile[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FS_ADD8(A, P)                                                  \
  FS_ADD(A.x & 0xffffu, P[0].x); FS_ADD(A.x >> 16, P[0].y);            \
  FS_ADD(A.y & 0xffffu, P[1].x); FS_ADD(A.y >> 16, P[1].y);            \
  FS_ADD(A.z & 0xffffu, P[2].x); FS_ADD(A.z >> 16, P[2].y);            \
  FS_ADD(A.w & 0xffffu, P[3].x); FS_ADD(A.w >> 16, P[3].y);
  constexpr int64_t kRound = 16 * kBinBlock;
  int64_t e = e0 + 8 * t;
  for (; e - 8 * t + kRound <= e1; e += kRound) {
    v4u a[2];
    v2d p[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t ek = e + (int64_t)k * 8 * kBinBlock;
      a[k] = row_ids_load<NTLD, L8>(lrow, lrow8, gbase, ek);
#pragma unroll
      for (int j = 0; j < 4; ++j) p[k][j] = stream_load<NTLD>((const v2d *)(prod + ek + 2 * j));
    }
    __builtin_amdgcn_sched_barrier(0);
    a[0] = row_ids<L8>(a[0]);
    a[1] = row_ids<L8>(a[1]);
    FS_ADD8(a[0], p[0])
    FS_ADD8(a[1], p[1])
  }
  // the last, partial round: whole waves stay together (the one-byte row ids shift a value between neighbouring lanes), the lanes
  // past the end read the panel's last eight entries and skip their adds
  for (; e - 8 * (t & 63) < e1; e += 8 * kBinBlock) {
    const int64_t ec = e < e1 ? e : e1 - 8;
    const v4u raw = row_ids_load<NTLD, L8>(lrow, lrow8, gbase, ec);
    v2d p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = stream_load<NTLD>((const v2d *)(prod + ec + 2 * j));
    const v4u a = row_ids<L8>(raw);
    if (e < e1) { FS_ADD8(a, p) }
  }
#undef FS_ADD8
#undef FS_ADD
  __syncthreads();
  for (int i = t; i < nr; i += kBinBlock) y[(int64_t)(r0 + i) * ys] = ytile[i];
}

template <bool NTLD, int RM = kBinRowsMax, bool L8 = false>
__global__ __launch_bounds__(kBinBlock) void spmv_reduce_kernel(
    const unsigned *__restrict__ bin_ptr, const int *__restrict__ panel_row, const uint16_t *__restrict__ lrow,
    const double *__restrict__ prod, double *__restrict__ y, int ys, int pbase, const uint8_t *__restrict__ lrow8 = nullptr,
    const uint16_t *__restrict__ gbase = nullptr)
{
  __shared__ double ytile[RM];
  reduce_panel<NTLD, L8>(ytile, pbase + blockIdx.x, bin_ptr, panel_row, lrow, prod, y, ys, lrow8, gbase);
}

// pass 2 with a FIXED order of additions: ONE wave per panel walks the panel's products in stream order, 512 entries per step
// (eight per lane, as above).  A wave's LDS instructions execute in program order, so every y slot receives its addends in the
// order of the stream (band by band, inside a band in CSR order; lanes of one instruction that hit the same slot are
// serialised by the LDS in a fixed order): the result is bit-identical run to run, which the sixteen-wave kernel above -- whose
// waves add into the same slots concurrently -- is not.  One wave can do it because the pass is a stream: DEPTH steps of loads
// (80 bytes per lane each) stay in flight in registers, and eight ds_add_f64 per 512 entries are far below what one wave may
// issue.
#ifndef FS_ORDERED_DEPTH
#define FS_ORDERED_DEPTH 16   // steps of loads in flight (config 2: 4 / 6 / 8 / 12 / 16 / 20 -> +13 / +7 / +6 / +4.3 / +3.5 / +3 % over the 16-wave pass)
#endif
template <bool NTLD, int RM, int DEPTH, bool L8 = false>
__global__ __launch_bounds__(64) void spmv_reduce_ordered_kernel(
    const unsigned *__restrict__ bin_ptr, const int *__restrict__ panel_row, const uint16_t *__restrict__ lrow,
    const double *__restrict__ prod, double *__restrict__ y, int ys, int pbase, const uint8_t *__restrict__ lrow8 = nullptr,
    const uint16_t *__restrict__ gbase = nullptr)
{
  __shared__ __attribute__((aligned(16))) double ytile[RM];
  const int t = threadIdx.x;
  const int panel = pbase + blockIdx.x;
  const int r0 = panel_row[panel], nr = panel_row[panel + 1] - r0;
  for (int i = 2 * t; i < nr; i += 128) *reinterpret_cast<v2d *>(&ytile[i]) = v2d{0.0, 0.0};   // (RM is even: a pair past nr stays inside)
  const int64_t e0 = (int64_t)bin_ptr[panel] * kBinGroup, e1 = (int64_t)bin_ptr[panel + 1] * kBinGroup;
#define FS_ADD(idx, val) __hip_atomic_fetch_add(&ytile[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define FS_ADD8(A, P)                                                  \
  FS_ADD(A.x & 0xffffu, P[0].x); FS_ADD(A.x >> 16, P[0].y);            \
  FS_ADD(A.y & 0xffffu, P[1].x); FS_ADD(A.y >> 16, P[1].y);            \
  FS_ADD(A.z & 0xffffu, P[2].x); FS_ADD(A.z >> 16, P[2].y);            \
  FS_ADD(A.w & 0xffffu, P[3].x); FS_ADD(A.w >> 16, P[3].y);
  constexpr int64_t kStep = 8 * 64;
  v4u a[DEPTH];
  v2d p[DEPTH][4];
  // steps past the end re-read the segment's last eight entries (their adds are skipped): every load is unconditional
  auto fetch = [&](int k, int64_t e) {
    const int64_t ec = (e + 8 <= e1) ? e : (e1 - e0 >= 8 ? e1 - 8 : e0);
    a[k] = row_ids_load<NTLD, L8>(lrow, lrow8, gbase, ec);      // (decoded where it is used: the loads of DEPTH steps stay in flight)
#pragma unroll
    for (int j = 0; j < 4; ++j) p[k][j] = stream_load<NTLD>((const v2d *)(prod + ec + 2 * j));
  };
  if (e1 > e0) {
    int64_t e = e0 + 8 * t;
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) fetch(k, e + (int64_t)k * kStep);
    for (; e - 8 * t < e1; e += (int64_t)DEPTH * kStep) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        const int64_t ek = e + (int64_t)k * kStep;
        const v4u ak = row_ids<L8>(a[k]);
        v2d pk[4] = {p[k][0], p[k][1], p[k][2], p[k][3]};
        fetch(k, ek + (int64_t)DEPTH * kStep);
        if (ek + 8 <= e1) { FS_ADD8(ak, pk) }
      }
    }
  }
#undef FS_ADD8
#undef FS_ADD
  __syncthreads();
  for (int i = t; i < nr; i += 64) y[(int64_t)(r0 + i) * ys] = ytile[i];
}

// ------------------------------------------------------------------------------------------
// The longest rows of a heavy-tailed matrix in ONE pass (LongRows, fs_common.h): persistent workgroups stream equal shares of
// the (band, long row)-ordered entries; the band of x (128 KiB) AND one accumulator per long row (<= 24 KiB) sit in LDS.
// A lane takes two consecutive entries; entries are sorted by row inside a band, so a wave adds up the products of equal rows
// with a segmented scan over its lanes (a row of 10^6 entries has ~160 of them per band: 64 lanes hammering one LDS
// address would serialise) and only the last lane of every run adds to the accumulator (ds_add_f64).  At the end of its
// share a workgroup adds its accumulators to ylong in HBM (one atomic per row it touched).  10 bytes per entry where the
// two-pass pair moves 28.
// ORDERED (fixed-order sums: option "reproducible", the solvers): every long row belongs to ONE wave of the workgroup and the
// builder keeps an owner's entries of a band in one contiguous segment (seg_ptr); wave w walks ITS segments, so an accumulator
// only ever sees the LDS instructions of one wave, which execute in program order -- the same sum, bit for bit, every run.
// The workgroups' sums go to ypart and are added up in workgroup order by longrows_combine_kernel instead of with atomics.
// ------------------------------------------------------------------------------------------
template <bool VALUED, int BC, int NACC, bool ORDERED>
__global__ __launch_bounds__(kBinBlock) void spmv_longrows_kernel(
    int ncol, int B, int nlong, const int64_t *__restrict__ band_ptr, const unsigned *__restrict__ seg_ptr,
    const uint16_t *__restrict__ lcol, const uint16_t *__restrict__ lrow, const double *__restrict__ vals,
    const double *__restrict__ x, int xs, double *__restrict__ ylong, double *__restrict__ ypart)
{
  __shared__ double xband[BC + 8];
  __shared__ double acc[NACC];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < nlong; i += kBinBlock) acc[i] = 0.0;
  const int64_t pairs = band_ptr[B] >> 1;                         // every segment holds an even number of entries
  const int64_t e_beg = 2 * (pairs * blockIdx.x / gridDim.x), e_end = 2 * (pairs * (blockIdx.x + 1) / gridDim.x);
  if (e_beg < e_end) {
    int b;
    {
      int lo = 0, hi = B - 1;
      while (lo < hi) {
        const int mid = lo + ((hi - lo + 1) >> 1);
        if (band_ptr[mid] <= e_beg) lo = mid; else hi = mid - 1;
      }
      b = lo;
    }
    // a band holds few entries here (config-5 shard: 16 K per 64 KiB of x), so the band of x is requested one band AHEAD into
    // registers and only copied to LDS at the band switch: its latency hides under the entries of the band before
    double rn[BC / kBinBlock];
    int have = -1;                                                   // band whose x is in rn
    auto request = [&](int bb) {
      const int cb = bb * BC;
      const int wb = (ncol - cb < BC) ? ncol - cb : BC;
#pragma unroll
      for (int j = 0; j < BC / kBinBlock; ++j) {
        const int i = j * kBinBlock + t;
        rn[j] = __builtin_nontemporal_load(x + (int64_t)(cb + (i < wb ? i : wb - 1)) * xs);
      }
      have = bb;
    };
    // 128 entries of one wave: products, run sums by a segmented scan over the lanes, one LDS add per run
    auto wave_step = [&](int64_t o, bool live, int64_t oc) {
      const unsigned a = *reinterpret_cast<const unsigned *>(lcol + oc);
      const unsigned rr = *reinterpret_cast<const unsigned *>(lrow + oc);
      v2d p = {xband[a & 0xffffu], xband[a >> 16]};
      if (VALUED) { const v2d v = *reinterpret_cast<const v2d *>(vals + oc); p.x *= v.x; p.y *= v.y; }
      unsigned key = 0xffffffffu;
      double sum = 0.0;
      if (live) {
        const unsigned r0 = rr & 0xffffu, r1 = rr >> 16;
        if (r0 == r1) { key = r0; sum = p.x + p.y; }
        else { __hip_atomic_fetch_add(&acc[r0], p.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); key = r1; sum = p.y; }
      }
      // segmented inclusive scan over the wave: keys are sorted, so an equal key d lanes down means one run
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned ku = __shfl_up(key, d);
        const double su = __shfl_up(sum, d);
        if (lane >= d && ku == key) sum += su;
      }
      const unsigned kn = __shfl_down(key, 1);
      if (live && (lane == 63 || kn != key))
        __hip_atomic_fetch_add(&acc[key], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)o;
    };
    for (int64_t e = e_beg; e < e_end; ++b) {
      const int64_t eb = band_ptr[b + 1] < e_end ? band_ptr[b + 1] : e_end;
      if (eb <= e) continue;
      const int c0 = b * BC;
      const int w = (ncol - c0 < BC) ? ncol - c0 : BC;
      if (have != b) request(b);                                     // the first band of the share (or after empty bands)
      __syncthreads();                                               // everyone is done with the previous band
#pragma unroll
      for (int j = 0; j < BC / kBinBlock; ++j) {
        const int i = j * kBinBlock + t;
        xband[i] = (i < w) ? rn[j] : 0.0;
      }
      if (t < 8) xband[BC + t] = 0.0;
      __syncthreads();
      if (b + 1 < B && eb < e_end) request(b + 1);
      if (ORDERED) {
        // this wave's segment of the band, clipped to the share: nobody else touches the accumulators of its rows
        const int64_t s0 = band_ptr[b] + seg_ptr[b * (kLongOwners + 1) + wave], s1 = band_ptr[b] + seg_ptr[b * (kLongOwners + 1) + wave + 1];
        const int64_t lo = s0 > e ? s0 : e, hi = s1 < eb ? s1 : eb;
        for (int64_t o0 = lo; o0 < hi; o0 += 128) {
          const int64_t o = o0 + 2 * lane;
          const bool live = o < hi;
          wave_step(o, live, live ? o : lo);
        }
      } else {
        // whole rounds of the workgroup (2048 entries); lanes past the end of the segment carry a sentinel key and add nothing
        for (int64_t o0 = e; o0 < eb; o0 += 2 * kBinBlock) {
          const int64_t o = o0 + 2 * t;
          const bool live = o < eb;
          wave_step(o, live, live ? o : e);                          // clamped address, masked inside
        }
      }
      e = eb;
    }
  }
  __syncthreads();
  if (ORDERED) {
    double *__restrict__ mine = ypart + (int64_t)blockIdx.x * nlong;
    for (int i = t; i < nlong; i += kBinBlock) mine[i] = acc[i];
  } else {
    for (int i = t; i < nlong; i += kBinBlock) {
      const double v = acc[i];
      if (v != 0.0) unsafeAtomicAdd(ylong + i, v);
    }
  }
}

// ylong[i] = the workgroups' sums of long row i, added in workgroup order
__global__ __launch_bounds__(kBlock) void longrows_combine_kernel(int nlong, int nwg, const double *__restrict__ ypart, double *__restrict__ ylong)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nlong) return;
  double acc = 0.0;
  for (int w = 0; w < nwg; ++w) acc += ypart[(int64_t)w * nlong + i];
  ylong[i] = acc;
}

// y[row[i]] = ylong[i] for the long rows inside [row0, row1): the two-pass pair wrote 0 there (their entries are not in it)
__global__ __launch_bounds__(kBlock) void longrows_scatter_kernel(int nlong, const int *__restrict__ row, const double *__restrict__ ylong,
                                                                 double *__restrict__ y, int ys, int row0, int row1)
{
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nlong) return;
  const int r = row[i];
  if (r >= row0 && r < row1) y[(int64_t)r * ys] = 0.0 + ylong[i];
}

static int launch_longrows(const DeviceCsr &A, const LongRows &L, const double *x, int xs, hipStream_t s)
{
  const bool ordered = reproducible_now();
  if (!ordered || L.n == 0 || L.nwg == 0) FS_HIP(hipMemsetAsync(L.ylong, 0, sizeof(double) * (size_t)L.nlong, s));
  if (L.n == 0 || L.nwg == 0) return FS_OK;
#define FS_LONG(V, BC, NA, ORD)                                                                                              \
  hipLaunchKernelGGL((spmv_longrows_kernel<V, BC, NA, ORD>), dim3(L.nwg), dim3(kBinBlock), 0, s, A.ncol, L.B, L.nlong, L.band_ptr, \
                     L.seg_ptr, L.lcol, L.lrow, L.vals, x, xs, L.ylong, L.ypart)
#define FS_LONG2(V, BC, NA) do { if (ordered) FS_LONG(V, BC, NA, true); else FS_LONG(V, BC, NA, false); } while (0)
  if (L.bcols == kLongBandB) { if (A.has_vals()) FS_LONG2(true, kLongBandB, kLongRowsB); else FS_LONG2(false, kLongBandB, kLongRowsB); }
  else                       { if (A.has_vals()) FS_LONG2(true, kLongBandA, kLongRowsA); else FS_LONG2(false, kLongBandA, kLongRowsA); }
#undef FS_LONG2
#undef FS_LONG
  FS_HIP(hipGetLastError());
  if (ordered) {
    hipLaunchKernelGGL(longrows_combine_kernel, dim3((unsigned)((L.nlong + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, L.nlong, L.nwg, L.ypart,
                       L.ylong);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

// (Both passes in ONE persistent launch behind a device-wide arrival counter were built and measured in round 3 and withdrawn:
// on config 2 the single launch took 1.13 ms against 0.857 ms for the pair -- every wave's agent-scope release is a
// buffer_wbl2 over the XCD's whole L2 -- and the kernel trace shows there is nothing to win: pass 2 starts 0.0 us after pass 1
// ends, the next product 9 us after that (profiles/r03_gap_probe.jsonl, r03_kernel_gaps.txt).)

// ------------------------------------------------------------------------------------------
// Y = A X for K = 2 or 4 row-major right-hand sides in ONE sweep of a two-pass copy built with bands of
// kBinCols / K columns (BinnedCsr::kw == K): the north_star's "LDS-tiled dense B panel".
//   replaces bcsr_A_mul_B2 / _B4 (csr.h:164-202), bsbm_A_mul_B2 / _B4 (sparse.h:276-315), csr_A_mul_Bn /
//   bcsr_A_mul_Bn / bsbm_A_mul_Bn with ncol = 2, 4 (csr.h:441-465, 257-280, sparse.h:318-336) and the two
//   products of every bsbm_cg2 iteration (cg.h:134-135).
// Pass 1 keeps a band of X -- kBinCols / K rows of K doubles, 128 KiB -- in LDS; an entry is read once (2-byte
// local column, value) and gives K products, written as one 128-byte line per group of kBinGroup / K entries to
// the place of its run in (panel, band) order.  Pass 2 keeps the K-column Y slice of a panel (<= kBinRowsMax / K
// rows) in LDS and adds the products up.  Per entry: 2 + 8 + 8K written + 8K read + 2 bytes, against K times
// 28.25 for K sweeps of the single-vector pair.  Sum order as the single-vector pair (band-major, LDS atomics).
// ------------------------------------------------------------------------------------------
template <bool VALUED, int K, int U>
__global__ __launch_bounds__(kBinBlock) void spmm_expand_kernel(
    int ncol, int B, const unsigned *__restrict__ band_ptr, const uint16_t *__restrict__ lcol,
    const double *__restrict__ vals, const unsigned *__restrict__ gdst, const double *__restrict__ X, int xs,
    double *__restrict__ prod)
{
  constexpr int BC = kBinCols / K;       // columns per band
  constexpr int GE = kBinGroup / K;      // entries per group
  constexpr int LP = K / 2;              // lanes per entry: every lane owns two neighbouring products (one 16-byte store)
  constexpr int EPS = kBinBlock / LP;    // entries per step of the workgroup
  __shared__ __attribute__((aligned(16))) double xband[kBinCols + 8];  // [BC][K]; row BC is the zero row the padding entries point at
  const int t = threadIdx.x;
  const int le = t / LP, h = t % LP;
  const uint64_t groups = band_ptr[B];
  const unsigned g0 = (unsigned)(groups * blockIdx.x / gridDim.x), g1 = (unsigned)(groups * (blockIdx.x + 1) / gridDim.x);
  if (g0 >= g1) return;
  int b;
  {
    int lo = 0, hi = B - 1;
    while (lo < hi) {
      const int mid = lo + ((hi - lo + 1) >> 1);
      if (band_ptr[mid] <= g0) lo = mid; else hi = mid - 1;
    }
    b = lo;
  }
  for (unsigned g = g0; g < g1; ++b) {
    const unsigned gb = band_ptr[b + 1] < g1 ? band_ptr[b + 1] : g1;
    if (gb <= g) continue;
    const int c0 = b * BC;
    const int w = (ncol - c0 < BC) ? ncol - c0 : BC;
    __syncthreads();
    {
      // the band: rows c0 .. c0+w of X, K doubles each (contiguous when xs == K); 16 loads per thread in flight
      double r[kBinCols / kBinBlock];
#pragma unroll
      for (int j = 0; j < kBinCols / kBinBlock; ++j) {
        const int f = j * kBinBlock + t, i = f / K, q = f % K;
        r[j] = __builtin_nontemporal_load(X + (int64_t)(c0 + (i < w ? i : w - 1)) * xs + q);
      }
#pragma unroll
      for (int j = 0; j < kBinCols / kBinBlock; ++j) {
        const int f = j * kBinBlock + t;
        xband[f] = (f / K < w) ? r[j] : 0.0;
      }
      if (t < 8) xband[kBinCols + t] = 0.0;
    }
    __syncthreads();
    const int64_t e0 = (int64_t)g * GE, e1 = (int64_t)gb * GE;
    constexpr int64_t kRound = (int64_t)U * EPS;
    int64_t o = e0 + le;
    for (; o - le + kRound <= e1; o += kRound) {
      unsigned a[U], d[U];
      double v[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t e = o + (int64_t)k * EPS;
        a[k] = lcol[e];
        d[k] = gdst[e / GE];
        if (VALUED) v[k] = vals[e];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t e = o + (int64_t)k * EPS;
        v2d p = *reinterpret_cast<const v2d *>(&xband[a[k] * K + 2 * h]);
        if (VALUED) { p.x *= v[k]; p.y *= v[k]; }
        __builtin_nontemporal_store(p, (v2d *)(prod + ((int64_t)d[k] * GE + (e % GE)) * K + 2 * h));
      }
    }
    for (; o < e1; o += EPS) {
      const unsigned a = lcol[o];
      const unsigned d = gdst[o / GE];
      v2d p = *reinterpret_cast<const v2d *>(&xband[a * K + 2 * h]);
      if (VALUED) { const double v = vals[o]; p.x *= v; p.y *= v; }
      __builtin_nontemporal_store(p, (v2d *)(prod + ((int64_t)d * GE + (o % GE)) * K + 2 * h));
    }
    g = gb;
  }
}

template <int K>
__global__ __launch_bounds__(kBinBlock) void spmm_reduce_kernel(
    const unsigned *__restrict__ bin_ptr_all, const int *__restrict__ panel_row_all, const uint16_t *__restrict__ lrow,
    const double *__restrict__ prod, double *__restrict__ Y, int ys, int pbase)
{
  const unsigned *__restrict__ bin_ptr = bin_ptr_all + pbase;          // this launch covers the panels pbase .. pbase + gridDim.x
  const int *__restrict__ panel_row = panel_row_all + pbase;
  constexpr int GE = kBinGroup / K;
  constexpr int EPL = 8 / K;             // entries per lane and step: 64 bytes of products
  __shared__ double ytile[kBinRowsMax];  // [rows of the panel][K]
  const int t = threadIdx.x;
  const int r0 = panel_row[blockIdx.x], nr = panel_row[blockIdx.x + 1] - r0;
  for (int i = t; i < nr * K; i += kBinBlock) ytile[i] = 0.0;
  __syncthreads();
  const int64_t e0 = (int64_t)bin_ptr[blockIdx.x] * GE, e1 = (int64_t)bin_ptr[blockIdx.x + 1] * GE;
  typedef uint16_t rows_t __attribute__((ext_vector_type(EPL)));
#define FS_ADD(idx, val) __hip_atomic_fetch_add(&ytile[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  auto add8 = [&](const rows_t a, const v2d (&p)[4]) {
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
      const int base = (int)a[q] * K;
#pragma unroll
      for (int j = 0; j < K; j += 2) {
        FS_ADD(base + j, p[(q * K + j) / 2].x);
        FS_ADD(base + j + 1, p[(q * K + j) / 2].y);
      }
    }
  };
  constexpr int64_t kStep = (int64_t)EPL * kBinBlock;
  constexpr int64_t kRound = 2 * kStep;
  int64_t e = e0 + (int64_t)EPL * t;
  for (; e - EPL * t + kRound <= e1; e += kRound) {
    rows_t a[2];
    v2d p[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t ek = e + k * kStep;
      a[k] = *reinterpret_cast<const rows_t *>(lrow + ek);
#pragma unroll
      for (int j = 0; j < 4; ++j) p[k][j] = *reinterpret_cast<const v2d *>(prod + ek * K + 2 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
    add8(a[0], p[0]);
    add8(a[1], p[1]);
  }
  for (; e < e1; e += kStep) {
    const rows_t a = *reinterpret_cast<const rows_t *>(lrow + e);
    v2d p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = *reinterpret_cast<const v2d *>(prod + e * K + 2 * j);
    add8(a, p);
  }
#undef FS_ADD
  __syncthreads();
  for (int i = t; i < nr * K; i += kBinBlock) Y[(int64_t)(r0 + i / K) * ys + (i % K)] = ytile[i];
}

// the same pass with a fixed order of additions: one wave per panel, stream order (see spmv_reduce_ordered_kernel)
template <int K, int DEPTH>
__global__ __launch_bounds__(64) void spmm_reduce_ordered_kernel(
    const unsigned *__restrict__ bin_ptr_all, const int *__restrict__ panel_row_all, const uint16_t *__restrict__ lrow,
    const double *__restrict__ prod, double *__restrict__ Y, int ys, int pbase)
{
  const unsigned *__restrict__ bin_ptr = bin_ptr_all + pbase;
  const int *__restrict__ panel_row = panel_row_all + pbase;
  constexpr int GE = kBinGroup / K;
  constexpr int EPL = 8 / K;             // entries per lane and step: 64 bytes of products
  __shared__ double ytile[kBinRowsMax];  // [rows of the panel][K]
  const int t = threadIdx.x;
  const int r0 = panel_row[blockIdx.x], nr = panel_row[blockIdx.x + 1] - r0;
  for (int i = t; i < nr * K; i += 64) ytile[i] = 0.0;
  const int64_t e0 = (int64_t)bin_ptr[blockIdx.x] * GE, e1 = (int64_t)bin_ptr[blockIdx.x + 1] * GE;
  typedef uint16_t rows_t __attribute__((ext_vector_type(EPL)));
#define FS_ADD(idx, val) __hip_atomic_fetch_add(&ytile[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  auto add8 = [&](const rows_t a, const v2d (&p)[4]) {
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
      const int base = (int)a[q] * K;
#pragma unroll
      for (int j = 0; j < K; j += 2) {
        FS_ADD(base + j, p[(q * K + j) / 2].x);
        FS_ADD(base + j + 1, p[(q * K + j) / 2].y);
      }
    }
  };
  constexpr int64_t kStep = (int64_t)EPL * 64;
  rows_t a[DEPTH];
  v2d p[DEPTH][4];
  auto fetch = [&](int k, int64_t e) {   // steps past the end re-read the segment's last entries (their adds are skipped)
    const int64_t ec = (e + EPL <= e1) ? e : (e1 - e0 >= EPL ? e1 - EPL : e0);
    a[k] = *reinterpret_cast<const rows_t *>(lrow + ec);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[k][j] = *reinterpret_cast<const v2d *>(prod + ec * K + 2 * j);
  };
  if (e1 > e0) {
    int64_t e = e0 + (int64_t)EPL * t;
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) fetch(k, e + (int64_t)k * kStep);
    for (; e - EPL * t < e1; e += (int64_t)DEPTH * kStep) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        const int64_t ek = e + (int64_t)k * kStep;
        const rows_t ak = a[k];
        const v2d pk[4] = {p[k][0], p[k][1], p[k][2], p[k][3]};
        fetch(k, ek + (int64_t)DEPTH * kStep);
        if (ek + EPL <= e1) add8(ak, pk);
      }
    }
  }
#undef FS_ADD
  __syncthreads();
  for (int i = t; i < nr * K; i += 64) Y[(int64_t)(r0 + i / K) * ys + (i % K)] = ytile[i];
}

// Y[r, 0:K] = sum of the virtual rows of row r, in storage order (yv holds K doubles per virtual row)
template <int K>
__global__ __launch_bounds__(kBlock) void tiled_combine_k_kernel(int nrow, const int *__restrict__ vfirst,
                                                                const double *__restrict__ yv, double *__restrict__ Y, int ys,
                                                                int row0 = 0)
{
  const int64_t i = (int64_t)row0 * K + (int64_t)blockIdx.x * kBlock + threadIdx.x;   // rows row0 .. nrow of this launch
  const int64_t r = i / K;
  const int j = (int)(i % K);
  if (r >= nrow) return;
  const int a = vfirst[r], b = vfirst[r + 1];
  double acc = yv[(int64_t)a * K + j];
  for (int v = a + 1; v < b; ++v) acc += yv[(int64_t)v * K + j];
  Y[r * ys + j] = acc;
}

// p0 .. p1: the panels pass 2 covers in this launch (p1 < 0: all of them); pass 1 runs when p0 == 0.  With cut rows the
// combine pass covers rows row0 .. row1 (the rows whose last piece lies in a panel below p1: see spmv_part_bounds).
// The one-wave pass 2 of fixed-order sums is bound by what ONE wave can issue per step, so the decode of one-byte row ids costs it
// what it saves the sixteen-wave pass (config 2: + 11.5 % instead of + 3.6 % over the arrival-order product).  The first fixed-order
// product on such a copy therefore writes the two-byte ids out once (0.1 ms for 160 M entries + one allocation of 2 bytes per entry)
// and the ordered kernel keeps reading those; the default product never touches them.  If the allocation fails the ordered kernel
// decodes on the fly (spmv_reduce_ordered_kernel<..., true>).
__global__ void rows8_decode_kernel(int64_t groups, const uint8_t *__restrict__ lrow8, const uint16_t *__restrict__ gbase,
                                    uint16_t *__restrict__ lrow)
{
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const v4u d = *(const v4u *)(lrow8 + g * kBinGroup);
  unsigned r = gbase[g];
  unsigned out[kBinGroup / 2];
#pragma unroll
  for (int j = 0; j < kBinGroup; j += 2) {
    const unsigned w = j < 4 ? d.x : j < 8 ? d.y : j < 12 ? d.z : d.w;
    const unsigned a = r + ((w >> (8 * (j & 3))) & 0xffu), b = a + ((w >> (8 * ((j + 1) & 3))) & 0xffu);
    out[j / 2] = a | (b << 16);
    r = b;
  }
#pragma unroll
  for (int j = 0; j < kBinGroup / 8; ++j)
    *(v4u *)(lrow + g * kBinGroup + 8 * j) = v4u{out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]};
}

static void ensure_two_byte_ids(BinnedCsr &N, hipStream_t s)
{
  if (N.lrow || !N.lrow8 || N.lrow_tried) return;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return; }
  if (cap != hipStreamCaptureStatusNone) return;      // a product being captured into a graph: no allocation, no wait -- decode on the fly
  N.lrow_tried = true;
  uint16_t *p = nullptr;
  if (hipMalloc(&p, sizeof(uint16_t) * (size_t)N.n) != hipSuccess) { (void)hipGetLastError(); return; }
  const int64_t groups = N.n >> kBinGroupLog;
  if (groups > 0) hipLaunchKernelGGL(rows8_decode_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, groups, N.lrow8, N.gbase, p);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return; }
  N.lrow = p;       // (behind the synchronize: a later product on another stream finds the ids complete)
}

int launch_spmv_binned(const DeviceCsr &A, double *y, const double *x, hipStream_t s, int xs, int ys, int p0, int p1, int row0,
                       int row1)
{
  const BinnedCsr &N = *A.binned;
  double *out = N.split ? N.yv : y;              // cut rows: virtual sums first, combined below
  const int os = N.split ? 1 : ys;
  const bool part = p1 >= 0;
  if (!part) { p0 = 0; p1 = N.P; row0 = 0; row1 = A.nrow; }
  const int nwg1 = (options().bin_wgs > 0 && options().bin_wgs < N.nwg1) ? options().bin_wgs : N.nwg1;
  if (nwg1 > 0 && p0 == 0) {
    // tuning switches (A/B runs): bits 0-1 pass-1 unroll (1: 8, 2: 2 steps; default 4), bit 2 pass-2 loads
    // non-temporal, bit 3 pass-1 stores plain, bit 4 pass-1 loads non-temporal.  Defaults, measured on config 2
    // (valued / pattern-only, ms per product): plain loads in both passes and non-temporal stores 0.95 / 0.69;
    // non-temporal loads in pass 1 0.98 / 0.73, in pass 2 as well 1.09 / 0.84; plain stores 1.01 / 0.75
    const int flags = options().bin_flags;
#define FS_EXPAND4(V, U)                                                                                          \
  do {                                                                                                            \
    if (flags & 16) { if (flags & 8) FS_EXPAND(V, U, true, false);  else FS_EXPAND(V, U, true, true); }           \
    else            { if (flags & 8) FS_EXPAND(V, U, false, false); else FS_EXPAND(V, U, false, true); }          \
  } while (0)
#define FS_EXPAND(V, U, NL, NS)                                                                                    \
  hipLaunchKernelGGL((spmv_expand_kernel<V, U, NL, NS>), dim3(nwg1), dim3(kBinBlock), 0, s, A.ncol, N.B, N.band_ptr, \
                     N.lcol, N.vals, N.gdst, x, xs, N.prod, 0u, (unsigned)(N.n >> kBinGroupLog))
    if (N.bcols == kBinColsBig) {                // the large-band copy: default switches only
      if (A.has_vals())
        hipLaunchKernelGGL((spmv_expand_kernel<true, 4, false, true, kBinColsBig>), dim3(nwg1), dim3(kBinBlock), 0, s, A.ncol, N.B,
                           N.band_ptr, N.lcol, N.vals, N.gdst, x, xs, N.prod, 0u, (unsigned)(N.n >> kBinGroupLog));
      else
        hipLaunchKernelGGL((spmv_expand_kernel<false, 4, false, true, kBinColsBig>), dim3(nwg1), dim3(kBinBlock), 0, s, A.ncol, N.B,
                           N.band_ptr, N.lcol, N.vals, N.gdst, x, xs, N.prod, 0u, (unsigned)(N.n >> kBinGroupLog));
    } else
    if (A.has_vals()) { if ((flags & 3) == 1) FS_EXPAND4(true, 8); else if ((flags & 3) == 2) FS_EXPAND4(true, 2); else FS_EXPAND4(true, 4); }
    else        { if ((flags & 3) == 1) FS_EXPAND4(false, 8); else if ((flags & 3) == 2) FS_EXPAND4(false, 2); else FS_EXPAND4(false, 4); }
#undef FS_EXPAND4
#undef FS_EXPAND
    FS_HIP(hipGetLastError());
  }
  if (N.lr && p0 == 0)          // behind pass 1, in front of pass 2: HBM-bound like both
    if (int rc = launch_longrows(A, *N.lr, x, xs, s)) return rc;
  if (p1 > p0) {
    // "reproducible" (or bit 5 of bin_flags): one wave per panel, additions in stream order, bit-identical run to run
    const bool ordered = reproducible_now() || (options().bin_flags & 32);
    if (ordered && N.lrow8 && !N.lrow) ensure_two_byte_ids(*A.binned, s);
    if (N.bcols == kBinColsBig && ordered)
      hipLaunchKernelGGL((spmv_reduce_ordered_kernel<false, kBinRowsBig, FS_ORDERED_DEPTH>), dim3(p1 - p0), dim3(64), 0, s, N.bin_ptr,
                         N.panel_row, N.lrow, N.prod, out, os, p0);
    else if (N.bcols == kBinColsBig)
      hipLaunchKernelGGL((spmv_reduce_kernel<false, kBinRowsBig>), dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row,
                         N.lrow, N.prod, out, os, p0);
    else if (ordered && N.lrow8 && !N.lrow)
      hipLaunchKernelGGL((spmv_reduce_ordered_kernel<false, kBinRowsMax, FS_ORDERED_DEPTH, true>), dim3(p1 - p0), dim3(64), 0, s, N.bin_ptr,
                         N.panel_row, N.lrow, N.prod, out, os, p0, N.lrow8, N.gbase);
    else if (ordered)
      hipLaunchKernelGGL((spmv_reduce_ordered_kernel<false, kBinRowsMax, FS_ORDERED_DEPTH>), dim3(p1 - p0), dim3(64), 0, s, N.bin_ptr,
                         N.panel_row, N.lrow, N.prod, out, os, p0);
    else if (N.lrow8)       // one byte per row id (BinnedCsr::lrow8): default loads only
      hipLaunchKernelGGL((spmv_reduce_kernel<false, kBinRowsMax, true>), dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow,
                         N.prod, out, os, p0, N.lrow8, N.gbase);
    else if (options().bin_flags & 4)
      hipLaunchKernelGGL(spmv_reduce_kernel<true>, dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod,
                         out, os, p0);
    else
      hipLaunchKernelGGL(spmv_reduce_kernel<false>, dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod,
                         out, os, p0);
    FS_HIP(hipGetLastError());
  }
  if (N.split && row1 > row0) {
    if (int rc = launch_tiled_combine(row1, N.vfirst, N.yv, y, ys, row0, s)) return rc;
  }
  if (N.lr && row1 > row0) {   // the long rows of this range: their sums were left in ylong by the launch behind pass 1
    hipLaunchKernelGGL(longrows_scatter_kernel, dim3((unsigned)((N.lr->nlong + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, N.lr->nlong,
                       N.lr->row, N.lr->ylong, y, ys, row0, row1);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

// pass 1 for the groups g0 .. g1 only, pass 2 for the panels p0 .. p1 only, default switches: the pieces fs_spmv_host overlaps with
// its PCIe copies (the band range whose part of x has arrived; the panel range whose rows go down next)
int launch_expand_groups(const DeviceCsr &A, const double *x, unsigned g0, unsigned g1, int wgs, hipStream_t s)
{
  const BinnedCsr &N = *A.binned;
  if (g1 <= g0 || wgs <= 0) return FS_OK;
#define FS_XH(V, BC)                                                                                                    \
  hipLaunchKernelGGL((spmv_expand_kernel<V, 4, false, true, BC>), dim3(wgs), dim3(kBinBlock), 0, s, A.ncol, N.B, N.band_ptr, \
                     N.lcol, N.vals, N.gdst, x, 1, N.prod, g0, g1)
  if (N.bcols == kBinColsBig) { if (A.has_vals()) FS_XH(true, kBinColsBig); else FS_XH(false, kBinColsBig); }
  else                        { if (A.has_vals()) FS_XH(true, kBinCols); else FS_XH(false, kBinCols); }
#undef FS_XH
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int launch_reduce_panels(const DeviceCsr &A, double *y, int p0, int p1, hipStream_t s)
{
  const BinnedCsr &N = *A.binned;
  if (p1 <= p0) return FS_OK;
  if (N.bcols == kBinColsBig)
    hipLaunchKernelGGL((spmv_reduce_kernel<false, kBinRowsBig>), dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod,
                       y, 1, p0);
  else if (N.lrow8)
    hipLaunchKernelGGL((spmv_reduce_kernel<false, kBinRowsMax, true>), dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod,
                       y, 1, p0, N.lrow8, N.gbase);
  else
    hipLaunchKernelGGL(spmv_reduce_kernel<false>, dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod, y, 1, p0);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// one sweep of a k-column two-pass copy: Y[:, 0:kw] = A X[:, 0:kw]; X / Y rows are xs / ys doubles apart
// p0 .. p1 / row0 .. row1: as launch_spmv_binned (p1 < 0: the whole sweep)
int launch_spmm_binned(const DeviceCsr &A, const BinnedCsr &N, double *Y, const double *X, hipStream_t s, int xs, int ys, int p0,
                       int p1, int row0, int row1)
{
  const int K = N.kw;
  double *out = N.split ? N.yv : Y;
  const int os = N.split ? K : ys;
  if (p1 < 0) { p0 = 0; p1 = N.P; row0 = 0; row1 = A.nrow; }
  const int nwg1 = (p0 != 0) ? 0 : ((options().bin_wgs > 0 && options().bin_wgs < N.nwg1) ? options().bin_wgs : N.nwg1);
#define FS_XP(V, KK)                                                                                                  \
  hipLaunchKernelGGL((spmm_expand_kernel<V, KK, 4>), dim3(nwg1), dim3(kBinBlock), 0, s, A.ncol, N.B, N.band_ptr, N.lcol, \
                     N.vals, N.gdst, X, xs, N.prod)
  if (nwg1 > 0) {
    if (K == 2) { if (A.has_vals()) FS_XP(true, 2); else FS_XP(false, 2); }
    else        { if (A.has_vals()) FS_XP(true, 4); else FS_XP(false, 4); }
    FS_HIP(hipGetLastError());
  }
#undef FS_XP
  if (p1 > p0) {
    const bool ordered = reproducible_now() || (options().bin_flags & 32);   // one wave per panel, stream order
    if (K == 2 && ordered)
      hipLaunchKernelGGL((spmm_reduce_ordered_kernel<2, FS_ORDERED_DEPTH>), dim3(p1 - p0), dim3(64), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod, out, os, p0);
    else if (ordered)
      hipLaunchKernelGGL((spmm_reduce_ordered_kernel<4, FS_ORDERED_DEPTH>), dim3(p1 - p0), dim3(64), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod, out, os, p0);
    else if (K == 2)
      hipLaunchKernelGGL(spmm_reduce_kernel<2>, dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod, out, os, p0);
    else
      hipLaunchKernelGGL(spmm_reduce_kernel<4>, dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod, out, os, p0);
    FS_HIP(hipGetLastError());
  }
  if (N.split && row1 > row0) {
    const unsigned grid = (unsigned)(((int64_t)(row1 - row0) * K + kBlock - 1) / kBlock);
    if (K == 2)
      hipLaunchKernelGGL(tiled_combine_k_kernel<2>, dim3(grid), dim3(kBlock), 0, s, row1, N.vfirst, N.yv, Y, ys, row0);
    else
      hipLaunchKernelGGL(tiled_combine_k_kernel<4>, dim3(grid), dim3(kBlock), 0, s, row1, N.vfirst, N.yv, Y, ys, row0);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

}  // namespace fs
