// fs_kernels.hip -- hand-written gfx950 kernels of the A_mul_B / At_mul_B path.
//
// Compiled with -ffp-contract=off: a valued term is one rounded multiply followed by one
// rounded add, exactly what the strict-IEEE reference loop `tmp += x[cols[i]] * vals[i]`
// (csr.h:434, dsparse.h:49) computes, so that every kernel that adds a row's terms in
// storage order is bit-identical to the CPU order.  All kernels are HBM/gather bound
// (0.17 flop/B), the spare multiply issue slot costs nothing.
//
// Where the kernels live
//   fs_kernels_twopass.hip  spmv_expand_kernel / spmv_reduce_kernel (+ the fixed-order pass 2), spmv_longrows_kernel,
//                           spmm_expand_kernel / spmm_reduce_kernel: y = A x in two streaming passes on the two-pass copy (column
//                           bands x row panels), the default for large matrices; neither pass gathers from L2 or HBM
//   fs_kernels_tiled.hip    spmv_ldsx_dma_kernel / spmv_ldsx_pipe_kernel (the band's slice of x staged in LDS: dense tiles,
//                           config 3, cbcsr), spmv_tiled_kernel (x gathered from an L2-resident band), ata_ldsx_kernel
//   here                    spmv_stream_kernel + fix-ups (one workgroup per 2048-entry chunk, storage-order sums: small x,
//                           strict_order), spmv_vector_kernel (lanes per row, A/B alternative), spmm_kernel / spmm_wide_kernel /
//                           spmm_mfma_kernel (k right-hand sides, one lane per output column), cbcsr_kernel, and everything that
//                           decides: spmv_choice, spmm_plan, products in parts, host-vector products
//   (which copy a matrix keeps is the format builder's measured choice: fs_format.hip, choose_copy)
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "fs_kernel_util.h"

namespace fs {

// LDS slot of the i-th product of a chunk: one pad slot per 16 products makes the stride
// between consecutive 16-entry rows odd (17), so a thread-per-row sweep is conflict free.
__device__ __forceinline__ int lds_slot(int i) { return i + (i >> 4); }
constexpr int kLdsDoubles = kChunk + (kChunk >> 4) + 8;

// ------------------------------------------------------------------------------------------
// y = A x, chunk-streaming kernel.
//   replaces csr_A_mul_B (csr.h:425-438), bcsr_A_mul_B (csr.h:149-161) and, through the
//   device CSR built at upload, A_mul_B / At_mul_B (sparse.h:58-75), sdm_A_mul_B /
//   sdm_At_mul_B (dsparse.h:43-62), bsbm_A_mul_B (sparse.h:259-273), bsdm_A_mul_B
//   (dsparse.h:176-191).
// Chunk c owns non-zeros [c*kChunk, min((c+1)*kChunk, nnz)) and finishes the rows whose first
// non-zero is in that range, rows [first_row[c], first_row[c+1]).  Non-zeros in front of
// row_ptr[first_row[c]] belong to a row that began in an earlier chunk: their sum goes to
// head[c].  If the last row runs past the chunk its partial sum goes to tail[c]; the fix-up
// kernel combines tail[c] + head[c+1] + ... left to right.
// Row sums: with G == 1 (chosen when the chunk holds >= 64 rows, or always under
// strict_order) one thread adds a row's products in storage order.  Chunks with few, long rows
// use G = 2..64 lanes per row and a butterfly reduction.
// ------------------------------------------------------------------------------------------
// phase 1 of the streaming kernel for one thread: 8 non-zeros as two groups of 4 consecutive
// entries.  FULL (every chunk but possibly the last) is straight-line code: six 16-byte
// streaming loads, then eight independent 8-byte gathers of x, all in flight together.
template <bool VALUED, bool NT, bool FULL>
__device__ __forceinline__ void stream_products(double *__restrict__ prod, const int *__restrict__ cols,
                                                const double *__restrict__ vals, const double *__restrict__ x,
                                                int64_t s, int64_t e, int t)
{
  int ci[kPerThread];
  double vv[kPerThread];
#pragma unroll
  for (int j = 0; j < kPerThread / 4; ++j) {
    const int l = 4 * (j * kBlock + t);
    const int64_t g = s + l;
    if (FULL) {
      const v4i cc = stream_load<NT>(reinterpret_cast<const v4i *>(cols + g));
      ci[4 * j + 0] = cc.x; ci[4 * j + 1] = cc.y; ci[4 * j + 2] = cc.z; ci[4 * j + 3] = cc.w;
      if (VALUED) {
        const v2d a = stream_load<NT>(reinterpret_cast<const v2d *>(vals + g));
        const v2d b = stream_load<NT>(reinterpret_cast<const v2d *>(vals + g + 2));
        vv[4 * j + 0] = a.x; vv[4 * j + 1] = a.y; vv[4 * j + 2] = b.x; vv[4 * j + 3] = b.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool in = g + q < e;
        ci[4 * j + q] = in ? cols[g + q] : -1;
        if (VALUED) vv[4 * j + q] = in ? vals[g + q] : 0.0;
      }
    }
  }
  double p[kPerThread];
#pragma unroll
  for (int i = 0; i < kPerThread; ++i) p[i] = (FULL || ci[i] >= 0) ? x[ci[i]] : 0.0;
#pragma unroll
  for (int j = 0; j < kPerThread / 4; ++j) {
    const int l = 4 * (j * kBlock + t);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * j + q;
      prod[lds_slot(l + q)] = VALUED ? p[i] * vv[i] : p[i];
    }
  }
}

template <bool VALUED, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(
    int nrow, int64_t nnz, const int *__restrict__ row_ptr, const int *__restrict__ cols,
    const double *__restrict__ vals, const int *__restrict__ first_row, const double *__restrict__ x,
    double *__restrict__ y, double *__restrict__ head, double *__restrict__ tail, int strict)
{
  __shared__ double prod[kLdsDoubles];
  const int t = threadIdx.x;
  const int c = blockIdx.x;
  const int64_t s = (int64_t)c * kChunk;
  const bool full = s + kChunk <= nnz;
  const int64_t e = full ? s + kChunk : nnz;

  // row bookkeeping first: these loads do not depend on phase 1 and overlap with it.
  // Virtual row vr: vr == 0 is the head (non-zeros [s, row_ptr[r_begin]) of a row that began
  // earlier), vr >= 1 is row r_begin + vr - 1.  Its non-zeros are [A(vr), B(vr)) with
  // A(0) = s, A(vr) = row_ptr[r_begin + vr - 1], B(vr) = row_ptr[r_begin + vr].
  const int r_begin = first_row[c];
  const int r_end = first_row[c + 1];
  const int nv = r_end - r_begin + 1;
  int lg = 0;  // log2(lanes per row)
  if (!strict && nv < 64) {
    lg = 31 - __clz(kBlock / nv);  // floor(log2(256 / nv)) in [2, 8]
    if (lg > 6) lg = 6;
  }
  const int G = 1 << lg;
  const int groups = kBlock >> lg;
  const int gid = t >> lg;
  const int gl = t & (G - 1);
  const int s32 = (int)s, e32 = (int)e;  // nnz <= INT_MAX because row_ptr is int
  // unconditional (index-clamped) loads: no branch, so no wait is forced in front of phase 1
  const int vg = gid < nv ? gid : nv - 1;
  const int la = row_ptr[r_begin + (vg > 0 ? vg - 1 : 0)];
  int pb = row_ptr[r_begin + vg];
  int pa = gid == 0 ? s32 : la;

  // ---- phase 1: stream cols/vals, gather x, park the products in LDS ------------------------
  if (full) stream_products<VALUED, NT, true>(prod, cols, vals, x, s, e, t);
  else      stream_products<VALUED, NT, false>(prod, cols, vals, x, s, e, t);
  __syncthreads();

  // ---- phase 2: per-row sums out of LDS -----------------------------------------------------
  for (int vr = gid; vr < nv; vr += groups) {
    if (vr != gid) {
      pa = row_ptr[r_begin + vr - 1];
      pb = row_ptr[r_begin + vr];
    }
    const bool cont = pb > e32;
    const int lo = pa - s32;
    const int hi = (cont ? e32 : pb) - s32;
    double acc = 0.0;
    int i = lo + gl;
    for (; i + 7 * G < hi; i += 8 * G) {
      double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = prod[lds_slot(i + u * G)];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += w[u];
    }
    for (; i < hi; i += G) acc += prod[lds_slot(i)];
    for (int m = G >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (gl == 0) {
      if (vr == 0) head[c] = acc;
      else if (cont) tail[c] = acc;
      else y[r_begin + vr - 1] = acc;
    }
  }
}

// rows that cross chunk boundaries: y[r] = tail[c] + head[c+1] + ... + head[last chunk of r]
__global__ void spmv_fixup_kernel(int nchunks, int64_t nnz, const int *__restrict__ row_ptr,
                                  const int *__restrict__ first_row, const double *__restrict__ head,
                                  const double *__restrict__ tail, double *__restrict__ y)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int r0 = first_row[c], r1 = first_row[c + 1];
  if (r1 <= r0) return;
  const int r = r1 - 1;
  const int64_t b = row_ptr[r + 1];
  int64_t e = (int64_t)(c + 1) * kChunk;
  if (e > nnz) e = nnz;
  if (b <= e) return;
  double acc = tail[c];
  const int last = (int)((b - 1) / kChunk);
  for (int cc = c + 1; cc <= last; ++cc) acc += head[cc];
  y[r] = acc;
}

// strict_order variant: a row that crosses chunks must still be ONE left-to-right sum (csr.h:430-437), so the running sum tail[c]
// is continued with the remaining terms themselves instead of with the later chunks' partial sums.  Storage order forbids
// re-associating the ADDS, not prefetching the TERMS: a crossing row belongs to one WAVE; its 64 lanes load 64 consecutive
// cols / vals and gather x[col] -- independent loads, kFixU steps of 64 in flight, the entries two super-steps and the gathers
// one super-step ahead of the sum -- and the products are then added in lane order through v_readlane (every lane carries the
// same running sum).  One thread walking the row with dependent loads took 593 ms on a config-5 shard (a 10^6-entry row).
// Lanes past the end of the row contribute +0.0: a sum that started from +0.0 is never -0.0, so x + (+0.0) == x bit for bit.
constexpr int kFixU = 4;

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <bool VALUED>
__global__ __launch_bounds__(kBlock) void spmv_fixup_strict_kernel(int nchunks, int64_t nnz, const int *__restrict__ row_ptr,
                                                                   const int *__restrict__ first_row, const int *__restrict__ cols,
                                                                   const double *__restrict__ vals, const double *__restrict__ x,
                                                                   const double *__restrict__ tail, double *__restrict__ y)
{
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);      // one wave per chunk: everything below is wave-uniform
  if (c >= nchunks) return;
  const int r0 = first_row[c], r1 = first_row[c + 1];
  if (r1 <= r0) return;
  const int r = r1 - 1;
  const int64_t b = row_ptr[r + 1];
  int64_t e = (int64_t)(c + 1) * kChunk;
  if (e > nnz) e = nnz;
  if (b <= e) return;
  double acc = tail[c];
  int ci[kFixU];
  double vv[kFixU], p[kFixU], pn[kFixU];
  auto load_entries = [&](int64_t at) {
#pragma unroll
    for (int u = 0; u < kFixU; ++u) {
      const int64_t i = at + u * 64 + lane;
      const bool in = i < b;
      ci[u] = in ? cols[i] : -1;
      if (VALUED) vv[u] = in ? vals[i] : 0.0;
    }
  };
  auto gather = [&](double (&out)[kFixU]) {
#pragma unroll
    for (int u = 0; u < kFixU; ++u) {
      const double xv = ci[u] >= 0 ? x[ci[u]] : 0.0;
      out[u] = ci[u] >= 0 ? (VALUED ? xv * vv[u] : xv) : 0.0;
    }
  };
  load_entries(e);
  gather(p);
  load_entries(e + 64 * kFixU);
  for (int64_t base = e; base < b; base += 64 * kFixU) {
    gather(pn);                                   // the next super-step's x: in flight under the sum below
    load_entries(base + 2 * 64 * kFixU);          // and the entries of the one after
#pragma unroll
    for (int u = 0; u < kFixU; ++u)
#pragma unroll
      for (int j = 0; j < 64; ++j) acc += readlane_f64(p[u], j);
#pragma unroll
    for (int u = 0; u < kFixU; ++u) p[u] = pn[u];
  }
  if (lane == 0) y[r] = acc;
}

// ------------------------------------------------------------------------------------------
// G-lanes-per-row CSR kernel (A/B alternative; also the simplest correct baseline).
// ------------------------------------------------------------------------------------------
template <bool VALUED>
__global__ __launch_bounds__(kBlock) void spmv_vector_kernel(int nrow, int lg, const int *__restrict__ row_ptr,
                                                            const int *__restrict__ cols,
                                                            const double *__restrict__ vals,
                                                            const double *__restrict__ x, double *__restrict__ y)
{
  const int G = 1 << lg;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> lg;
  const int gl = threadIdx.x & (G - 1);
  if (row >= nrow) return;
  const int a = row_ptr[row], b = row_ptr[row + 1];
  double acc = 0.0;
  for (int64_t i = (int64_t)a + gl; i < b; i += G) {
    const double xv = x[cols[i]];
    acc += VALUED ? xv * vals[i] : xv;
  }
  for (int m = G >> 1; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
  if (gl == 0) y[row] = acc;
}

// ------------------------------------------------------------------------------------------
// Y = A X with k row-major right-hand sides.
//   replaces csr_A_mul_Bn (csr.h:441-465), bcsr_A_mul_B2/_B4/_B8/_B8_auto/_Bn/_B32n
//   (csr.h:164-302), bsbm_A_mul_B2/_B4/_Bn (sparse.h:276-336).
// A group of KP = 2^lg lanes (KP >= min(k,64)) owns one row; lane j owns output column j and
// adds the row's terms in storage order (bit-identical to the CPU loops).  The group fetches
// KP (col,val) pairs with one coalesced load and broadcasts them with shuffles; each term is a
// contiguous 8k-byte read of X row cols[i] -- the whole group reads one X row per step.
// ------------------------------------------------------------------------------------------
template <bool VALUED, int LG>
__global__ __launch_bounds__(kBlock) void spmm_kernel(int nrow, int k, const int *__restrict__ row_ptr,
                                                     const int *__restrict__ cols,
                                                     const double *__restrict__ vals,
                                                     const double *__restrict__ X, double *__restrict__ Y)
{
  constexpr int KP = 1 << LG;                    // lanes per row = output columns handled together
  constexpr int EPL = LG < 3 ? (8 >> LG) : 1;    // entries per lane and step: a step always covers >= 8 entries
  constexpr int EB = KP * EPL;                   // entries per step
  constexpr int gpb = kBlock >> LG;
  const int j = threadIdx.x & (KP - 1);
  const int64_t row = (int64_t)blockIdx.x * gpb + (threadIdx.x >> LG);
  if (row >= nrow) return;
  const int a = row_ptr[row], b = row_ptr[row + 1];
  for (int j0 = 0; j0 < k; j0 += KP) {
    const int col = j0 + j;
    const bool act = col < k;
    const int colc = act ? col : k - 1;          // lanes past the last column read a valid address and store nothing
    double acc = 0.0;
    for (int64_t base = a; base < b; base += EB) {   // 64-bit: a row may end within EB of INT_MAX
      // the step's (column, value) pairs, spread over the row's lanes; positions past the row's end re-read its last
      // entry so that every load below is unconditional (no load sits behind a branch: see spmv_tiled_kernel)
      int myc[EPL];
      double myv[EPL];
#pragma unroll
      for (int q = 0; q < EPL; ++q) {
        const int64_t e = base + q * KP + j;
        const int64_t ec = e < b ? e : b - 1;
        myc[q] = cols[ec];
        if (VALUED) myv[q] = vals[ec];
      }
      const int n = (b - base < EB) ? (int)(b - base) : EB;
#pragma unroll
      for (int i0 = 0; i0 < EB; i0 += 8) {
        if (i0 < n) {
          double xv[8], wv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + u;                // entry i of the step lives in lane i % KP, register i / KP
            const int cc = __shfl(myc[(i / KP) % EPL], i & (KP - 1), KP);
            if (VALUED) wv[u] = __shfl(myv[(i / KP) % EPL], i & (KP - 1), KP);
            xv[u] = X[(int64_t)cc * k + colc];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (i0 + u < n) acc += VALUED ? xv[u] * wv[u] : xv[u];
        }
      }
    }
    if (act) Y[row * k + col] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// The same product with 16-byte loads: a lane owns TWO neighbouring output columns, a row is served by 2^LG >= k/2 lanes, up
// to sixteen X rows are in flight per lane.  tools/probe_rowgather (profiles/r03_probe_rowgather.jsonl) is why: from a table
// far larger than the caches HBM hands out 128-byte lines at 53 G lines/s whatever part of the line is used (160 M rows of 16,
// 32 or 64 bytes: 3.0 ms each), and for rows of 128 / 256 bytes dwordx4 loads with 16 rows in flight reach 6.2 / 5.6 TB/s where
// dwordx2 loads with 8 in flight reach 5.4 / 4.9.  Every column still adds its terms in storage order (bit-identical to
// spmm_kernel).  Needs an even k and 16-byte aligned X and Y.
// ------------------------------------------------------------------------------------------
template <bool VALUED, int LG>
__global__ __launch_bounds__(kBlock) void spmm_wide_kernel(int nrow, int k, const int *__restrict__ row_ptr,
                                                          const int *__restrict__ cols,
                                                          const double *__restrict__ vals,
                                                          const double *__restrict__ X, double *__restrict__ Y)
{
  constexpr int KP = 1 << LG;                    // lanes per row = column PAIRS handled together
  constexpr int EPL = LG < 4 ? (16 >> LG) : 1;   // entries per lane and step: a step always covers >= 16 entries
  constexpr int EB = KP * EPL;
  constexpr int gpb = kBlock >> LG;
  const int j = threadIdx.x & (KP - 1);
  const int64_t row = (int64_t)blockIdx.x * gpb + (threadIdx.x >> LG);
  if (row >= nrow) return;
  const int a = row_ptr[row], b = row_ptr[row + 1];
  const int kh = k >> 1;
  const double2 *__restrict__ X2 = reinterpret_cast<const double2 *>(X);
  double2 *__restrict__ Y2 = reinterpret_cast<double2 *>(Y);
  for (int j0 = 0; j0 < kh; j0 += KP) {
    const int pr = j0 + j;
    const bool act = pr < kh;
    const int prc = act ? pr : kh - 1;           // lanes past the last pair read a valid address and store nothing
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t base = a; base < b; base += EB) {
      int myc[EPL];
      double myv[EPL];
#pragma unroll
      for (int q = 0; q < EPL; ++q) {            // unconditional loads from clamped positions, as in spmm_kernel
        const int64_t e = base + q * KP + j;
        const int64_t ec = e < b ? e : b - 1;
        myc[q] = cols[ec];
        if (VALUED) myv[q] = vals[ec];
      }
      const int n = (b - base < EB) ? (int)(b - base) : EB;
#define FS_WIDE_GROUP(U)                                                                           \
  {                                                                                                \
    double2 xv[U];                                                                                 \
    double wv[U];                                                                                  \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
      const int i = i0 + u;                                                                        \
      const int cc = __shfl(myc[(i / KP) % EPL], i & (KP - 1), KP);                                \
      if (VALUED) wv[u] = __shfl(myv[(i / KP) % EPL], i & (KP - 1), KP);                           \
      xv[u] = X2[(int64_t)cc * kh + prc];                                                          \
    }                                                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                                  \
      if (i0 + u < n) {                                                                            \
        acc0 += VALUED ? xv[u].x * wv[u] : xv[u].x;                                                \
        acc1 += VALUED ? xv[u].y * wv[u] : xv[u].y;                                                \
      }                                                                                            \
  }
#pragma unroll
      for (int i0 = 0; i0 < EB; i0 += 16) {
        if (i0 < n) {
          if (n - i0 > 8) FS_WIDE_GROUP(16)
          else FS_WIDE_GROUP(8)
        }
      }
#undef FS_WIDE_GROUP
    }
    if (act) Y2[row * kh + pr] = make_double2(acc0, acc1);
  }
}

// ------------------------------------------------------------------------------------------
// Y = A X on the matrix cores: the experiment BASELINE.json's north_star asks for ("MFMA only to the dense panel
// accumulate"), kept as an opt-in kernel (option spmm_kernel = 4) so that its counters can be put next to the row
// kernel's.  One wave per CSR row.  v_mfma_f64_16x16x4_f64 computes D(16x16) += A(16x4) B(4x16) with ONE f64 of A and
// of B per lane (lane l: A[l & 15][l >> 4], B[l >> 4][l & 15]; D[(l >> 4) + 4 reg][l & 15]).  A row of Y is a sum of
// scaled rows of X, so B = four gathered rows of X (16 columns of them) and A carries the row's four values in
// its row 0 only: the columns of different CSR rows are unrelated, nothing else can share B -- 1/16 of the
// multiply-adds of an instruction are useful, which is the structural answer to "how much of this SpMM is a GEMM".
// The multiply-adds inside the instruction are fused (one rounding), so results agree with the strict CPU order
// to rounding (1e-12 bar), bit for bit only for pattern matrices with integer-valued X.
// ------------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));

template <bool VALUED>
__global__ __launch_bounds__(kBlock) void spmm_mfma_kernel(int nrow, int k, const int *__restrict__ row_ptr,
                                                          const int *__restrict__ cols, const double *__restrict__ vals,
                                                          const double *__restrict__ X, double *__restrict__ Y)
{
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= nrow) return;                       // whole waves leave together
  const int a = row_ptr[row], b = row_ptr[row + 1];
  const int kk = lane >> 4, jj = lane & 15;
  for (int j0 = 0; j0 < k; j0 += 32) {
    const int c0 = j0 + jj, c1 = j0 + 16 + jj;
    const int c0c = c0 < k ? c0 : k - 1, c1c = c1 < k ? c1 : k - 1;   // clamped: loads stay unconditional
    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int64_t base = a; base < b; base += 16) {
      // 16 entries per round: four MFMA steps, eight X loads per lane in flight
      double av[4], b0[4], b1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t e = base + 4 * u + kk;
        const int64_t ec = e < b ? e : b - 1;
        const int c = cols[ec];
        const double v = VALUED ? vals[ec] : 1.0;
        av[u] = (jj == 0 && e < b) ? v : 0.0;     // row 0 of A; entries past the row's end contribute 0
        b0[u] = X[(int64_t)c * k + c0c];
        b1[u] = X[(int64_t)c * k + c1c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (base + 4 * u < b) {                   // wave-uniform
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b0[u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], b1[u], acc1, 0, 0, 0);
        }
      }
    }
    if (kk == 0) {                                // D row 0 lives in register 0 of lanes 0-15
      if (c0 < k) Y[row * k + c0] = acc0.x;
      if (c1 < k) Y[row * k + c1] = acc1.x;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Column-blocked binary CSR (cbcsr_A_mul_B, cbcsr.h:76-106): cell = block*nrow + row.
// One thread per row; the workgroup walks the column blocks in order, so every row adds its
// cell sums block by block (the order one CPU thread produces).  With STAGE the x tile of the
// current column block is first copied to LDS (coalesced) and gathers hit LDS instead of L2.
// ------------------------------------------------------------------------------------------
constexpr int kCbTile = 8192;  // doubles of x staged per column block (64 KiB)

template <bool STAGE>
__global__ __launch_bounds__(kBlock) void cbcsr_kernel(int nrow, int ncol, int nblocks, int cbs,
                                                      const int *__restrict__ row_ptr,
                                                      const int *__restrict__ cols,
                                                      const double *__restrict__ x, double *__restrict__ y)
{
  __shared__ double xt[STAGE ? kCbTile : 1];
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  double tot = 0.0;
  for (int b = 0; b < nblocks; ++b) {
    const int c0 = b * cbs;
    if (STAGE) {
      const int w = (ncol - c0 < cbs) ? ncol - c0 : cbs;
      __syncthreads();
      for (int i = threadIdx.x; i < w; i += kBlock) xt[i] = x[c0 + i];
      __syncthreads();
    }
    if (r < nrow) {
      const int64_t cell = (int64_t)b * nrow + r;
      const int lo = row_ptr[cell], hi = row_ptr[cell + 1];
      double s = 0.0;
      for (int i = lo; i < hi; ++i) s += STAGE ? xt[cols[i] - c0] : x[cols[i]];
      tot += s;
    }
  }
  if (r < nrow) y[r] = 0.0 + tot;
}

// ------------------------------------------------------------------------------------------
// launchers (those of the tiled kernels: fs_kernels_tiled.hip; of the two-pass kernels: fs_kernels_twopass.hip)
// ------------------------------------------------------------------------------------------
static int ceil_log2(int v)
{
  int lg = 0;
  while ((1 << lg) < v) ++lg;
  return lg;
}

// ------------------------------------------------------------------------------------------
// y = A x in parts, for callers that ship finished rows while the rest still computes: the all-gather of the y shards of
// a row-sharded product overlapped with the product itself ("chunk rows, launch all-gather of finished
// chunks while later chunks compute").  The product is cut where its kernel finishes rows anyway:
//   two-pass copy      pass 1 as a whole with part 0, pass 2 (one workgroup per row panel) in ranges of panels; with cut
//                      rows the combine pass follows in ranges of rows (a row is final once its last piece's panel is done)
//   L2-tiled / LDS-staged copy whose workgroups own their rows: ranges of whole generations of resident workgroups
//   anything else      (chunk-streaming kernel, chunks sharing panels, strict_order): everything with part 0
// spmv_part_bounds gives the row cuts rows[0 .. nparts] (rows [rows[p], rows[p+1]) are final after part p) and the unit
// cuts (panels / workgroups) for a given nparts; cached per handle.
// ------------------------------------------------------------------------------------------
// cuts of a two-pass copy (single-vector or k-column): pass 2 by whole generations of resident workgroups, rows by the panels'
// first rows (with cut rows: the rows whose every piece lies below the cut).  *cut = false: too few panels to cut.
static int binned_part_cuts(const DeviceCsr &A, BinnedCsr &N, int nparts, std::vector<int> &rows, std::vector<int> &units, bool *cut)
{
  *cut = false;
  const int slots = N.slots > 0 ? N.slots : 256;
  if (N.nwg1 <= 0 || nparts <= 1 || N.P <= slots) return FS_OK;
  if (!N.h_panel_row) {
    int *hp = (int *)malloc(sizeof(int) * ((size_t)N.P + 1));
    if (!hp) { set_error("out of host memory"); return FS_ERR_HIP; }
    const hipError_t e = hipMemcpy(hp, N.panel_row, sizeof(int) * ((size_t)N.P + 1), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(hp); return hip_fail(e, "hipMemcpy(panel_row)", __FILE__, __LINE__); }
    N.h_panel_row = hp;
  }
  if (N.split && !N.h_vfirst) {
    int *hv = (int *)malloc(sizeof(int) * ((size_t)A.nrow + 1));
    if (!hv) { set_error("out of host memory"); return FS_ERR_HIP; }
    const hipError_t e = hipMemcpy(hv, N.vfirst, sizeof(int) * ((size_t)A.nrow + 1), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(hv); return hip_fail(e, "hipMemcpy(vfirst)", __FILE__, __LINE__); }
    N.h_vfirst = hv;
  }
  const int *vfirst = N.h_vfirst;
  // one pass-2 workgroup per CU (its y slice fills LDS) and launches of one stream run one after the other: a part is a
  // whole number of generations of resident workgroups -- 4 parts of 192 panels on 256 CUs would take 4 generations where
  // the undivided pass takes 3 (config 2: 768 panels)
  const int gens = (N.P + slots - 1) / slots;
  const int c = nparts < gens ? nparts : gens;
  for (int p = 0; p <= nparts; ++p) {
    const int64_t w = p >= c ? N.P : (int64_t)slots * ((int64_t)gens * p / c);
    units[(size_t)p] = (int)(w < N.P ? w : N.P);
    const int vcut = N.h_panel_row[units[(size_t)p]];
    rows[(size_t)p] = !N.split ? vcut : p == nparts ? A.nrow :
                      (int)(std::upper_bound(vfirst, vfirst + A.nrow + 1, vcut) - vfirst) - 1;
  }
  *cut = true;
  return FS_OK;
}

int spmv_part_bounds(DeviceCsr &A, int nparts, const int **rows_out, const int **units_out, int *kind_out, bool *cut_out)
{
  const Options &o = options();
  const int kind = spmv_choice(A, o);
  for (const DeviceCsr::PartCuts &C : A.part_plans)
    if (C.n == nparts && C.kind == kind) {
      *rows_out = C.rows.data(); if (units_out) *units_out = C.units.data();
      if (kind_out) *kind_out = kind;
      if (cut_out) *cut_out = C.cut;
      return FS_OK;
    }
  std::vector<int> rows((size_t)nparts + 1, A.nrow), units((size_t)nparts + 1, 0);
  rows[0] = 0;
  bool cut = false;
  if (kind == 7) {
    if (int rc = binned_part_cuts(A, *A.binned, nparts, rows, units, &cut)) return rc;
  } else if ((kind == 8 || kind == 6) && nparts > 1) {
    TiledCsr &M = kind == 8 ? *A.tiledx : *A.tiled;
    if (!M.split && !(M.ldsx && M.shared)) {
      if (!M.h_panel_row) {
        int *hp = (int *)malloc(sizeof(int) * ((size_t)M.P + 1));
        int *hc = (int *)malloc(sizeof(int) * (size_t)(M.nchunks > 0 ? M.nchunks : 1));
        if (!hp || !hc) { free(hp); free(hc); set_error("out of host memory"); return FS_ERR_HIP; }
        hipError_t e = hipMemcpy(hp, M.panel_row, sizeof(int) * ((size_t)M.P + 1), hipMemcpyDeviceToHost);
        if (e == hipSuccess && M.ldsx && M.nchunks > 0)
          e = hipMemcpy(hc, M.chunk_panel, sizeof(int) * (size_t)M.nchunks, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(hp); free(hc); return hip_fail(e, "hipMemcpy(panel tables)", __FILE__, __LINE__); }
        M.h_panel_row = hp; M.h_chunk_panel = hc;
      }
      const int nwg = M.P;
      bool own = !M.ldsx || M.nchunks == M.P;      // workgroup w owns panel w
      for (int w = 0; own && M.ldsx && w < nwg; ++w) own = M.h_chunk_panel[w] == w;
      const int slots = M.slots > 0 ? M.slots : 256;
      const int gens = (nwg + slots - 1) / slots;
      if (own && gens >= 2) {
        // one workgroup per CU and launches of one stream run one after the other: a part is a whole number of generations
        const int c = nparts < gens ? nparts : gens;
        for (int p = 0; p <= nparts; ++p) {
          const int64_t w = p >= c ? nwg : (int64_t)slots * ((int64_t)gens * p / c);
          units[(size_t)p] = (int)(w < nwg ? w : nwg);
          rows[(size_t)p] = M.h_panel_row[units[(size_t)p]];
        }
        cut = true;
      }
    }
  }
  if (!cut) {                                        // everything with part 0: rows = {0, nrow, nrow, ...}
    units.assign((size_t)nparts + 1, 0);
    rows.assign((size_t)nparts + 1, A.nrow);
    rows[0] = 0;
  }
  if (A.part_plans.size() >= 8) A.part_plans.erase(A.part_plans.begin());
  A.part_plans.emplace_back();
  DeviceCsr::PartCuts &C = A.part_plans.back();
  C.n = nparts; C.kind = kind; C.cut = cut;
  C.rows.swap(rows); C.units.swap(units);
  *rows_out = C.rows.data(); if (units_out) *units_out = C.units.data();
  if (kind_out) *kind_out = kind;
  if (cut_out) *cut_out = cut;
  return FS_OK;
}

int launch_spmv_part(DeviceCsr &A, double *y, const double *x, int part, int nparts, hipStream_t s)
{
  if (A.nrow == 0) return FS_OK;
  const int *rows = nullptr, *units = nullptr;
  int kind = 0;
  bool cut = false;
  if (int rc = spmv_part_bounds(A, nparts, &rows, &units, &kind, &cut)) return rc;
  if (!cut) return part == 0 ? launch_spmv(A, y, x, s) : FS_OK;
  if (kind == 7) return launch_spmv_binned(A, y, x, s, 1, 1, units[part], units[part + 1], rows[part], rows[part + 1]);
  if (units[part + 1] <= units[part]) return FS_OK;
  return launch_spmv_tiled(A, kind == 8 ? *A.tiledx : *A.tiled, y, x, s, 1, 1, units[part], units[part + 1]);
}

// which kernel a single-vector product on A runs under the options o: 7 two-pass, 8 LDS-staged, 6 L2-tiled, 2 lanes per
// row, 1 chunk-streaming.  The copy the format builder kept unless the caller asked for storage-order (strict_order) or
// run-to-run identical (reproducible) sums, which the kernels that add in arrival order cannot give (LDS-staged; the k-column
// sweeps; the long-row path).
int spmv_choice(const DeviceCsr &A, const Options &o)
{
  // (fixed-order sums -- "reproducible", the solvers -- keep the builder's choice: pass 2 of the two-pass pair then runs one wave per
  // panel in stream order, the long-row path gives every row to one wave, the LDS-staged kernel waits for a phase's adds before
  // its barrier; only an LDS-staged copy whose items could not be arranged row-per-wave, TiledCsr::orderable, drops out)
  // Option "reproducible" is a REQUIREMENT: such a copy then drops out and the product falls to a kernel that can give fixed-order
  // sums.  A solver's scope (cg_fixed_order, FixedOrderScope) is a WISH: the kept copy keeps running -- in arrival order where it is
  // not orderable -- instead of silently moving every product of a solve to the chunk-streaming kernel;
  // fs_debug_fixed_order_honoured tells which it is.
  const bool must = o.reproducible != 0;
  if (A.binned && A.binned->built && !o.strict_order && (o.spmv_kernel == 0 || o.spmv_kernel == 7)) return 7;
  if (A.tiledx && A.tiledx->built && !o.strict_order && (!must || A.tiledx->orderable) && (o.spmv_kernel == 0 || o.spmv_kernel == 8)) return 8;
  if (A.tiled && A.tiled->built && !o.strict_order && (o.spmv_kernel == 0 || o.spmv_kernel == 6)) return 6;
  return o.spmv_kernel == 2 ? 2 : 1;
}

int launch_spmv(const DeviceCsr &A, double *y, const double *x, hipStream_t s, bool force_stream)
{
  if (A.nrow == 0) return FS_OK;
  Options o = options();
  if (force_stream) o.spmv_kernel = 1;
  const bool valued = A.vals != nullptr;
  switch (spmv_choice(A, o)) {
    case 7: return launch_spmv_binned(A, y, x, s);
    case 8: return launch_spmv_tiled(A, *A.tiledx, y, x, s);
    case 6: return launch_spmv_tiled(A, *A.tiled, y, x, s);
    default: break;
  }
  if (int rc = need_plain_csr(A, "this product (strict_order, spmv_kernel 1-3, or \"reproducible\" on a copy that cannot give fixed-order sums)")) return rc;
  // strict_order (storage-order sums: only the chunk-streaming kernel gives them), or "reproducible" on a matrix whose kept
  // LDS-staged copy could not be arranged row-per-wave: the product runs on the chunk-streaming kernel (correct, fixed order, but
  // slow on large matrices).  Said once under FS_TRACE_BUILD.
  if ((o.reproducible || o.strict_order) && o.spmv_kernel == 0 && ((A.binned && A.binned->built) || (A.tiledx && A.tiledx->built))) {
    static const bool trace = getenv("FS_TRACE_BUILD") != nullptr;
    static bool said = false;
    if (trace && !said) {
      said = true;
      fprintf(stderr, "[fastsparse] %d x %d: option %s: the kept %s copy cannot give such sums%s; products run on the chunk-streaming "
              "kernel\n", A.nrow, A.ncol, o.strict_order ? "strict_order" : "reproducible", A.binned && A.binned->built ? "two-pass" : "LDS-staged",
              o.strict_order ? "" : " (some work item holds more of one row than one wave can take: TiledCsr::orderable)");
    }
  }
  if (o.spmv_kernel == 2) {
    const double avg = A.nrow ? (double)A.nnz / A.nrow : 0.0;
    int lg = o.strict_order ? 0 : ceil_log2((int)(avg < 1 ? 1 : (avg > 64 ? 64 : avg)));
    if (lg > 6) lg = 6;
    const int64_t threads = (int64_t)A.nrow << lg;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    if (valued)
      hipLaunchKernelGGL(spmv_vector_kernel<true>, dim3(grid), dim3(kBlock), 0, s, A.nrow, lg, A.row_ptr, A.cols,
                         A.vals, x, y);
    else
      hipLaunchKernelGGL(spmv_vector_kernel<false>, dim3(grid), dim3(kBlock), 0, s, A.nrow, lg, A.row_ptr, A.cols,
                         A.vals, x, y);
    FS_HIP(hipGetLastError());
    return FS_OK;
  }
  const bool nt = o.spmv_kernel != 3;  // 3 = streaming kernel with plain (cached) loads, for A/B runs
  const dim3 grid(A.nchunks), block(kBlock);
#define FS_LAUNCH_STREAM(V, N)                                                                              \
  hipLaunchKernelGGL((spmv_stream_kernel<V, N>), grid, block, 0, s, A.nrow, A.nnz, A.row_ptr, A.cols, A.vals, \
                     A.first_row, x, y, A.head, A.tail, o.strict_order)
  if (valued) { if (nt) FS_LAUNCH_STREAM(true, true); else FS_LAUNCH_STREAM(true, false); }
  else        { if (nt) FS_LAUNCH_STREAM(false, true); else FS_LAUNCH_STREAM(false, false); }
#undef FS_LAUNCH_STREAM
  FS_HIP(hipGetLastError());
  if (A.spanning > 0 && o.strict_order) {
    const dim3 fg((A.nchunks + kBlock / 64 - 1) / (kBlock / 64)), fb(kBlock);    // one wave per chunk
    if (valued)
      hipLaunchKernelGGL(spmv_fixup_strict_kernel<true>, fg, fb, 0, s, A.nchunks, A.nnz, A.row_ptr, A.first_row,
                         A.cols, A.vals, x, A.tail, y);
    else
      hipLaunchKernelGGL(spmv_fixup_strict_kernel<false>, fg, fb, 0, s, A.nchunks, A.nnz, A.row_ptr, A.first_row,
                         A.cols, A.vals, x, A.tail, y);
    FS_HIP(hipGetLastError());
  } else if (A.spanning > 0) {
    hipLaunchKernelGGL(spmv_fixup_kernel, dim3((A.nchunks + kBlock - 1) / kBlock), dim3(kBlock), 0, s, A.nchunks,
                       A.nnz, A.row_ptr, A.first_row, A.head, A.tail, y);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

// ------------------------------------------------------------------------------------------
// y_host = A x_host: what the reference's callers hand over (csr_A_mul_B(y, A, x) with malloc'ed vectors).  The two
// 8-byte-per-element copies over PCIe cost more than the product (config 2: 1.43 ms each against 0.9 ms), and y needs
// all of x -- but the two-pass kernels do not: pass 1 needs the band of x a group belongs to, pass 2 finishes y panel
// by panel.  So x goes up in C ranges of bands and the pass-1 groups of a range are launched (on the handle's own
// non-blocking stream) as soon as its part of x has landed; pass 2 is launched in C ranges of panels and every range
// of y goes down as soon as its event fires, while the later ranges still run.  Only the last pass-1 range and the
// first pass-2 range are left outside the copies.  Other kept copies (LDS-staged, tiled, stream), cut rows and
// strict_order / reproducible: copy, product, copy.
// ------------------------------------------------------------------------------------------
void free_host_pipe(HostPipe &H)
{
  if (H.sx) (void)hipFree(H.sx);
  if (H.sy) (void)hipFree(H.sy);
  for (int i = 0; i < H.nev; ++i) (void)hipEventDestroy(H.ev[i]);
  if (H.stream) (void)hipStreamDestroy(H.stream);
  H = HostPipe();
}

static int host_pipe_ready(HostPipe &H, size_t nx, size_t ny, int nev)
{
  if (H.cx < nx) {
    if (H.sx) (void)hipFree(H.sx);
    H.sx = nullptr; H.cx = 0;
    FS_HIP(hipMalloc(&H.sx, sizeof(double) * nx));
    H.cx = nx;
  }
  if (H.cy < ny) {
    if (H.sy) (void)hipFree(H.sy);
    H.sy = nullptr; H.cy = 0;
    FS_HIP(hipMalloc(&H.sy, sizeof(double) * ny));
    H.cy = ny;
  }
  if (!H.stream) FS_HIP(hipStreamCreateWithFlags(&H.stream, hipStreamNonBlocking));
  for (; H.nev < nev; ++H.nev) FS_HIP(hipEventCreateWithFlags(&H.ev[H.nev], hipEventDisableTiming));
  return FS_OK;
}

// which way the last product with host vectors went (diagnostics / tests): 0 copy + product + copy, 1 two-pass in ranges
// of bands and panels, 2 a panel kernel in ranges of workgroups
static int g_last_host_path = 0;
int last_host_path() { return g_last_host_path; }

int spmv_host_vectors(const DeviceCsr &A, HostPipe &H, double *y_host, const double *x_host)
{
  g_last_host_path = 0;
  if (A.nrow == 0) return FS_OK;
  static const int want_chunks = [] {
    const char *e = getenv("FS_HOST_CHUNKS");
    const int v = e ? atoi(e) : 8;
    return v < 1 ? 1 : (v > HostPipe::kMaxChunks ? HostPipe::kMaxChunks : v);
  }();
  const Options &o = options();
  const size_t nx = A.ncol > 0 ? (size_t)A.ncol : 1, ny = (size_t)A.nrow;
  if (int rc = host_pipe_ready(H, nx, ny, want_chunks)) return rc;
  const bool two_pass = A.binned && A.binned->built && !A.binned->split && !A.binned->lr && A.binned->nwg1 > 0 && !o.strict_order &&
                        !reproducible_now() && (o.spmv_kernel == 0 || o.spmv_kernel == 7) && o.bin_flags == 0 && want_chunks > 1;
  if (!two_pass) {
    // the panel kernels (LDS-staged, L2-tiled) need all of x, but a workgroup that owns its rows finishes them: launched in
    // ranges of workgroups, y comes down range by range under the later ranges (a tall matrix: config 3, y 80 MB, x 8 MB)
    const TiledCsr *T = nullptr;
    if (!(A.binned && A.binned->built) && !o.strict_order && want_chunks > 1) {
      if (A.tiledx && A.tiledx->built && !reproducible_now() && (o.spmv_kernel == 0 || o.spmv_kernel == 8)) T = A.tiledx;
      else if (A.tiled && A.tiled->built && (o.spmv_kernel == 0 || o.spmv_kernel == 6)) T = A.tiled;
    }
    if (T && T->ldsx && T->shared && T->nchunks >= 2 * want_chunks && A.ncol >= (1 << 20)) {
      // Few, long rows (config 3 transposed: x 80 MB, y 8 MB): several chunks per panel, launched stretch of bands by
      // stretch of bands (fs_format.hip "Launch order"), so the chunks at the front of the order only read the front of x.
      // need[w] = columns the chunks 0 .. w read; x goes up in ranges and the chunks a range completes are launched behind it.
      TiledCsr &M = const_cast<TiledCsr &>(*T);
      if (!M.h_chunk_need) {
        std::vector<int> ci(2 * (size_t)M.nchunks);
        std::vector<int4> it((size_t)M.nitems > 0 ? (size_t)M.nitems : 1);
        int *need = (int *)malloc(sizeof(int) * (size_t)M.nchunks);
        if (!need) { set_error("out of host memory"); return FS_ERR_HIP; }
        hipError_t e = hipMemcpy(ci.data(), M.chunk_item, sizeof(int) * ci.size(), hipMemcpyDeviceToHost);
        if (e == hipSuccess && M.nitems > 0) e = hipMemcpy(it.data(), M.items, sizeof(int4) * (size_t)M.nitems, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(need); return hip_fail(e, "hipMemcpy(chunk tables)", __FILE__, __LINE__); }
        int64_t most = 0;
        for (int w = 0; w < M.nchunks; ++w) {
          for (int i = ci[2 * (size_t)w]; i < ci[2 * (size_t)w + 1]; ++i) {
            const int64_t end = ((int64_t)it[(size_t)i].z + 1) * M.W;
            if (end > most) most = end;
          }
          need[w] = (int)(most < A.ncol ? most : A.ncol);
        }
        int *hc = (int *)malloc(sizeof(int) * (size_t)M.nchunks);
        if (!hc) { free(need); set_error("out of host memory"); return FS_ERR_HIP; }
        e = hipMemcpy(hc, M.chunk_panel, sizeof(int) * (size_t)M.nchunks, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(need); free(hc); return hip_fail(e, "hipMemcpy(chunk tables)", __FILE__, __LINE__); }
        free(M.h_chunk_panel);
        M.h_chunk_panel = hc;
        M.h_chunk_need = need;
      }
      g_last_host_path = 3;
      FS_HIP(hipMemsetAsync(M.yv, 0, sizeof(double) * (size_t)A.nrow, H.stream));
      // a range = one or more whole stretch groups of the launch order (the panel id falls back where the next group starts),
      // at least one generation of resident workgroups long
      const int slots = M.slots > 0 ? M.slots : 256;
      int64_t have = 0;
      int w0 = 0;
      static const bool trace = getenv("FS_HOST_TRACE") != nullptr;   // the timeline of the call on stderr
      const auto tt0 = std::chrono::steady_clock::now();
      auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count(); };
      while (w0 < M.nchunks) {
        int w1 = w0 + 1;
        while (w1 < M.nchunks &&
               ((M.h_chunk_panel[w1] & 0x7fffffff) >= (M.h_chunk_panel[w1 - 1] & 0x7fffffff) || w1 - w0 < slots)) ++w1;
        // no short tail, and nothing is cut once all of x is needed (every launch ends in a partly filled generation)
        if (M.nchunks - w1 < slots || M.h_chunk_need[w1 - 1] >= A.ncol) w1 = M.nchunks;
        const int64_t upto = M.h_chunk_need[w1 - 1];
        if (upto > have) {
          FS_HIP(hipMemcpy(H.sx + have, x_host + have, sizeof(double) * (size_t)(upto - have), hipMemcpyHostToDevice));
          have = upto;
        }
        if (trace) fprintf(stderr, "[host] %.3f ms: x up to %lld, chunks %d..%d\n", now_ms(), (long long)have, w0, w1);
        if (int rc = launch_spmv_tiled(A, M, H.sy, H.sx, H.stream, 1, 1, w0, w1)) return rc;
        w0 = w1;
      }
      if (int rc = launch_strided_copy(A.nrow, M.yv, H.sy, 1, H.stream)) return rc;
      FS_HIP(hipStreamSynchronize(H.stream));
      if (trace) fprintf(stderr, "[host] %.3f ms: kernels done\n", now_ms());
      FS_HIP(hipMemcpy(y_host, H.sy, sizeof(double) * ny, hipMemcpyDeviceToHost));
      if (trace) fprintf(stderr, "[host] %.3f ms: y down\n", now_ms());
      return FS_OK;
    }
    if (A.ncol > 0) FS_HIP(hipMemcpy(H.sx, x_host, sizeof(double) * (size_t)A.ncol, hipMemcpyHostToDevice));
    if (T && !T->split && !(T->ldsx && T->shared) && T->P >= 2 * want_chunks) {
      TiledCsr &M = const_cast<TiledCsr &>(*T);
      if (!M.h_panel_row) {
        int *hp = (int *)malloc(sizeof(int) * ((size_t)M.P + 1));
        int *hc = (int *)malloc(sizeof(int) * (size_t)(M.nchunks > 0 ? M.nchunks : 1));
        if (!hp || !hc) { free(hp); free(hc); set_error("out of host memory"); return FS_ERR_HIP; }
        hipError_t e = hipMemcpy(hp, M.panel_row, sizeof(int) * ((size_t)M.P + 1), hipMemcpyDeviceToHost);
        if (e == hipSuccess && M.ldsx && M.nchunks > 0)
          e = hipMemcpy(hc, M.chunk_panel, sizeof(int) * (size_t)M.nchunks, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(hp); free(hc); return hip_fail(e, "hipMemcpy(panel tables)", __FILE__, __LINE__); }
        M.h_panel_row = hp; M.h_chunk_panel = hc;
      }
      // workgroup w owns panel w (LDS-staged: one chunk per panel when no panel is shared)
      const int nwg = M.P;
      bool own = !M.ldsx || M.nchunks == M.P;
      for (int w = 0; own && M.ldsx && w < nwg; ++w) own = M.h_chunk_panel[w] == w;
      // One workgroup per CU and launches of one stream run one after the other, so a range is a whole number of
      // generations of resident workgroups (config 3: 698 panels = 256 + 256 + 186; eight ranges of 87 were measured to
      // take eight generations, 2.42 ms per call against 1.87): y starts to come down after the first generation
      const int slots = M.slots > 0 ? M.slots : 256;
      const int gens = (nwg + slots - 1) / slots;
      const int c = want_chunks < gens ? want_chunks : gens;
      if (own && c >= 2) {
        g_last_host_path = 2;
        auto cut = [&](int j) { const int64_t w = (int64_t)slots * ((int64_t)gens * j / c); return (int)(w < nwg ? w : nwg); };
        for (int j = 0; j < c; ++j) {
          const int w0 = cut(j), w1 = j == c - 1 ? nwg : cut(j + 1);
          if (int rc = launch_spmv_tiled(A, M, H.sy, H.sx, H.stream, 1, 1, w0, w1)) return rc;
          FS_HIP(hipEventRecord(H.ev[j], H.stream));
        }
        for (int j = 0; j < c; ++j) {
          const int w0 = cut(j), w1 = j == c - 1 ? nwg : cut(j + 1);
          const int64_t r0 = M.h_panel_row[w0], r1 = M.h_panel_row[w1];
          FS_HIP(hipEventSynchronize(H.ev[j]));
          if (r1 > r0) FS_HIP(hipMemcpy(y_host + r0, H.sy + r0, sizeof(double) * (size_t)(r1 - r0), hipMemcpyDeviceToHost));
        }
        return FS_OK;
      }
    }
    if (int rc = launch_spmv(A, H.sy, H.sx, H.stream)) return rc;
    FS_HIP(hipStreamSynchronize(H.stream));
    FS_HIP(hipMemcpy(y_host, H.sy, sizeof(double) * ny, hipMemcpyDeviceToHost));
    return FS_OK;
  }
  g_last_host_path = 1;
  BinnedCsr &N = *A.binned;
  if (!N.h_band_ptr) {
    unsigned *hb = (unsigned *)malloc(sizeof(unsigned) * ((size_t)N.B + 1));
    int *hp = (int *)malloc(sizeof(int) * ((size_t)N.P + 1));
    if (!hb || !hp) { free(hb); free(hp); set_error("out of host memory"); return FS_ERR_HIP; }
    hipError_t e = hipMemcpy(hb, N.band_ptr, sizeof(unsigned) * ((size_t)N.B + 1), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(hp, N.panel_row, sizeof(int) * ((size_t)N.P + 1), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { free(hb); free(hp); return hip_fail(e, "hipMemcpy(two-pass tables)", __FILE__, __LINE__); }
    free(N.h_panel_row);                         // products in parts may have fetched it already
    N.h_band_ptr = hb; N.h_panel_row = hp;
  }
  const int nwg1 = (o.bin_wgs > 0 && o.bin_wgs < N.nwg1) ? o.bin_wgs : N.nwg1;
  // pass 1, band range by band range
  const int c1 = want_chunks < N.B ? want_chunks : N.B;
  for (int j = 0; j < c1; ++j) {
    const int b0 = (int)((int64_t)N.B * j / c1), b1 = (int)((int64_t)N.B * (j + 1) / c1);
    const int64_t x0 = (int64_t)b0 * N.bcols;
    int64_t x1 = (int64_t)b1 * N.bcols;
    if (x1 > A.ncol || j == c1 - 1) x1 = A.ncol;
    if (x1 > x0) FS_HIP(hipMemcpy(H.sx + x0, x_host + x0, sizeof(double) * (size_t)(x1 - x0), hipMemcpyHostToDevice));
    const unsigned g0 = N.h_band_ptr[b0], g1 = N.h_band_ptr[b1];
    if (g1 <= g0) continue;
    // a range holds 1 / c1 of the groups: fewer persistent workgroups than CUs only when a share would fall under the minimum
    int wgs = nwg1;
    const int64_t cap = ((int64_t)(g1 - g0) * kBinGroup + kBinShareMin - 1) / kBinShareMin;
    if (cap < wgs) wgs = (int)(cap < 1 ? 1 : cap);
    if (int rc = launch_expand_groups(A, H.sx, g0, g1, wgs, H.stream)) return rc;
  }
  // pass 2, panel range by panel range, an event behind each
  const int c2 = want_chunks < N.P ? want_chunks : N.P;
  for (int j = 0; j < c2; ++j) {
    const int p0 = (int)((int64_t)N.P * j / c2), p1 = (int)((int64_t)N.P * (j + 1) / c2);
    if (int rc = launch_reduce_panels(A, H.sy, p0, p1, H.stream)) return rc;
    FS_HIP(hipEventRecord(H.ev[j], H.stream));
  }
  for (int j = 0; j < c2; ++j) {
    const int p0 = (int)((int64_t)N.P * j / c2), p1 = (int)((int64_t)N.P * (j + 1) / c2);
    const int64_t r0 = N.h_panel_row[p0], r1 = N.h_panel_row[p1];
    FS_HIP(hipEventSynchronize(H.ev[j]));
    if (r1 > r0) FS_HIP(hipMemcpy(y_host + r0, H.sy + r0, sizeof(double) * (size_t)(r1 - r0), hipMemcpyDeviceToHost));
  }
  return FS_OK;
}

// dst[dst_off[i] + j] = src[src_off[i] + j], j < count[i], for nseg segments: blockIdx.y = segment.  Unpacks the padded
// receive buffer of an all-gather of unequal shards (or parts of shards) into y with ONE launch.
__global__ __launch_bounds__(kBlock) void copy_segments_kernel(int nseg, const int64_t *__restrict__ tab, const double *__restrict__ src,
                                                               double *__restrict__ dst)
{
  const int sgm = blockIdx.y;
  const int64_t d0 = tab[sgm], s0 = tab[nseg + sgm], n = tab[2 * nseg + sgm];
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) dst[d0 + j] = src[s0 + j];
}

int launch_copy_segments(int nseg, const int64_t *tab_dev, int64_t max_count, const double *src, double *dst, hipStream_t s)
{
  if (nseg <= 0 || max_count <= 0) return FS_OK;
  int64_t gx = (max_count + kBlock - 1) / kBlock;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)gx, (unsigned)nseg), dim3(kBlock), 0, s, nseg, tab_dev, src, dst);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// row-major n x k  <->  k columns of `ld` doubles each (column-major)
__global__ __launch_bounds__(kBlock) void rows_to_columns_kernel(int64_t n, int k, int64_t ld, const double *__restrict__ rm,
                                                                 double *__restrict__ cm)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  for (int j = 0; j < k; ++j) cm[(int64_t)j * ld + i] = rm[i * k + j];
}

__global__ __launch_bounds__(kBlock) void columns_to_rows_kernel(int64_t n, int k, int64_t ld, const double *__restrict__ cm,
                                                                 double *__restrict__ rm)
{
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  for (int j = 0; j < k; ++j) rm[i * k + j] = cm[(int64_t)j * ld + i];
}

// ------------------------------------------------------------------------------------------
// Multi-column products Y = A X (X, Y row-major, k columns).  Which kernel serves a given (matrix, k) is ONE decision
// (spmm_plan) over the copies the handle holds and the options; fs_spmm never builds a copy, never allocates a k-column
// copy and never waits for the device: everything that costs more than a launch happens in prepare_spmm (C-ABI:
// fs_matrix_prepare; the drop-in layer calls it on the first product with a new k, or for the ks listed in
// FS_PREPARE_K when the device copy of a matrix is made).
//
//   plan 2  k = 2..4 on a matrix that keeps the two-pass copy: ONE sweep with a k-column band of X in LDS (k = 3: a
//           2-column sweep and a single-vector sweep) -- needs the k-column copy, which prepare builds
//   plan 3  one single-vector two-pass sweep per column (strided gathers and stores): k <= 3 without a k-column copy
//   plan 5  a matrix on the LDS-staged copy (dense tiles: config 3's class): one unit-stride sweep per column on
//           COLUMN-major copies of X and Y (two transposes; the handle's scratch, allocated by prepare ONLY: before it the strided
//           sweeps, k = 2, or the row kernel serve);
//           for k >= 3 prepare times this against the row kernel on the matrix and keeps the faster (neither wins
//           everywhere: config 3's shape 0.77 ms per column against a row kernel bound by its X-row gathers at 51 G/s,
//           12.5 ms for any k from 4 to 16; 10 M rows x 16 with X of 131 K rows, k = 8: 3.5 ms in sweeps, 2.05 ms on
//           the row kernel, X in L2)
//   plan 6 / 7  k <= 2 strided sweeps of the LDS-staged / L2-tiled kernel
//   plan 4  the v_mfma_f64_16x16x4_f64 experiment (option spmm_kernel = 4)
//   plan 1  the row kernel: k >= 5 on two-pass matrices, k > 16, strict_order / reproducible, small matrices
// ------------------------------------------------------------------------------------------
constexpr int kLdsxSweepMaxK = 16;
enum { kPlanRow = 1, kPlanBinnedK = 2, kPlanBinnedCols = 3, kPlanMfma = 4, kPlanLdsxColumns = 5, kPlanLdsxStrided = 6,
       kPlanTiledStrided = 7 };

// doubles of column-major scratch a k-column product on the LDS-staged copy needs (X and Y, 16-byte aligned columns)
static size_t spmm_scratch_need(const DeviceCsr &A, int k, int64_t *ldx_out, int64_t *ldy_out)
{
  const int64_t ldx = ((int64_t)A.ncol + 1) & ~(int64_t)1, ldy = ((int64_t)A.nrow + 1) & ~(int64_t)1;
  *ldx_out = ldx; *ldy_out = ldy;
  return (size_t)k * (size_t)(ldx + ldy);
}

static bool spmm_scratch_ready(const DeviceCsr &A, int k)
{
  int64_t ldx, ldy;
  return A.spmm_scratch && A.spmm_scratch_doubles >= spmm_scratch_need(A, k, &ldx, &ldy);
}

// *needs_prepare: what prepare_spmm would still do for this k -- bit 0 build the k-column two-pass copy, bit 1 measure
// column sweeps against the row kernel, bit 2 allocate the column-major scratch (0: the plan is final)
int spmm_plan(const DeviceCsr &A, int k, int *needs_prepare)
{
  const Options &o = options();
  if (needs_prepare) *needs_prepare = 0;
  const bool free_order = !o.strict_order && (!o.reproducible || (A.tiledx && A.tiledx->orderable));   // fixed-order sums on the LDS-staged copy: see spmv_choice
  // the two-pass kernels: under "reproducible" their pass 2 runs one wave per panel in stream order -- not the long-row side path
  const bool bin_order = !o.strict_order;
  const int want = o.spmm_kernel;
  const bool hb = A.binned && A.binned->built, hx = A.tiledx && A.tiledx->built, ht = A.tiled && A.tiled->built;
  const bool bin_ok = o.spmv_kernel == 0 || o.spmv_kernel == 7, ldsx_ok = o.spmv_kernel == 0 || o.spmv_kernel == 8;
  // (the k-column copy only where the format builder kept the two-pass copy for the single-vector product: that is the
  // class of matrices -- large x, thin tiles -- on which streaming products beats gathering; config 3's dense tiles stay
  // on the LDS-staged kernel, two sweeps of 0.9 ms against 36 bytes per entry here)
  if (bin_order && k >= 2 && k <= 4 && (want == 0 || want == 2) && o.binning != 0 && bin_ok && (hb || o.binning == 2 || want == 2)) {
    const BinnedCsr *slot = k == 4 ? A.binned4 : A.binned2;
    const bool tried = k == 4 ? A.tried4 : A.tried2;
    if (slot && slot->built && (k != 3 || hb)) return kPlanBinnedK;
    if (!slot && !tried && needs_prepare) *needs_prepare |= 1;
  }
  // column by column on the single-vector pair where that beats the row kernel, whose every X-row gather misses L2 (three
  // sweeps: 3.0 ms on config 2 against 3.8 ms for the row kernel; four: 4.0 against 3.5)
  if (want != 1 && (k <= 3 || want == 3) && hb && bin_order && bin_ok) return kPlanBinnedCols;
  if (want != 1 && want != 4 && k >= 2 && k <= kLdsxSweepMaxK && hx && !hb && free_order && ldsx_ok) {
    // the sweeps run on column-major copies of X and Y in the handle's scratch, which only prepare_spmm allocates (a product
    // never allocates: hipMalloc synchronises the device and fails under stream capture -- ADVICE r3): until then the strided
    // sweeps (k = 2) or the row kernel serve
    if (!spmm_scratch_ready(A, k)) {
      if (needs_prepare) *needs_prepare |= 4 | ((want == 0 && k > 2 && A.spmm_choice[k] == 0) ? 2 : 0);
      return k <= 2 ? kPlanLdsxStrided : kPlanRow;
    }
    if (want != 0 || k == 2) return kPlanLdsxColumns;          // k = 2: the sweeps won every measurement
    if (A.spmm_choice[k] == 0 && needs_prepare) *needs_prepare |= 2;
    return A.spmm_choice[k] == 2 ? kPlanRow : kPlanLdsxColumns;
  }
  if (want != 1 && k <= 2 && hx && free_order && ldsx_ok) return kPlanLdsxStrided;
  if (want != 1 && k <= 2 && ht && !o.strict_order && (o.spmv_kernel == 0 || o.spmv_kernel == 6)) return kPlanTiledStrided;
  return want == 4 ? kPlanMfma : kPlanRow;
}

static int spmm_scratch_alloc(DeviceCsr &A, int k)      // prepare_spmm only
{
  int64_t ldx, ldy;
  const size_t need = spmm_scratch_need(A, k, &ldx, &ldy);
  if (A.spmm_scratch_doubles < need) {
    if (A.spmm_scratch) FS_HIP(hipFree(A.spmm_scratch));
    A.spmm_scratch = nullptr; A.spmm_scratch_doubles = 0;
    FS_HIP(hipMalloc(&A.spmm_scratch, sizeof(double) * need));
    A.spmm_scratch_doubles = need;
  }
  return FS_OK;
}

static int launch_spmm_row(const DeviceCsr &A, double *Y, const double *X, int k, hipStream_t s)
{
  if (int rc = need_plain_csr(A, "the row kernel of multi-column products")) return rc;
  // two columns per lane and 16-byte loads where the layout allows it and it measured faster (config 4's matrix, k = 4 / 6 / 8 /
  // 12: 3.39 / 3.88 / 3.32 / 5.00 ms -> 3.30 / 3.76 / 3.19 / 4.83; k = 16 / 32 / 64: 3.65 / 7.94 / 16.15 -> 3.71 / 7.93 / 16.20:
  // there both sit at the HBM rate for the lines they pull, profiles/r03_spmm_wide_ab.jsonl); option spmm_wide: 1 wherever
  // legal, -1 never
  const int wide = options().spmm_wide;
  if (wide >= 0 && (k & 1) == 0 && (wide > 0 || (k >= 4 && k <= 14)) && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0) {
    const int kh = k >> 1;
    const int lg = ceil_log2(kh > 64 ? 64 : kh);
    const int gpb = kBlock >> lg;
    const unsigned grid = (unsigned)(((int64_t)A.nrow + gpb - 1) / gpb);
#define FS_SPMMW(V, L) \
  hipLaunchKernelGGL((spmm_wide_kernel<V, L>), dim3(grid), dim3(kBlock), 0, s, A.nrow, k, A.row_ptr, A.cols, A.vals, X, Y)
#define FS_SPMMW_LG(V)                                                                                  \
  switch (lg) {                                                                                         \
    case 0: FS_SPMMW(V, 0); break; case 1: FS_SPMMW(V, 1); break; case 2: FS_SPMMW(V, 2); break;        \
    case 3: FS_SPMMW(V, 3); break; case 4: FS_SPMMW(V, 4); break; case 5: FS_SPMMW(V, 5); break;        \
    default: FS_SPMMW(V, 6); break;                                                                     \
  }
    if (A.vals) { FS_SPMMW_LG(true) } else { FS_SPMMW_LG(false) }
#undef FS_SPMMW_LG
#undef FS_SPMMW
    FS_HIP(hipGetLastError());
    return FS_OK;
  }
  const int lg = ceil_log2(k > 64 ? 64 : k);
  const int gpb = kBlock >> lg;
  const unsigned grid = (unsigned)(((int64_t)A.nrow + gpb - 1) / gpb);
#define FS_SPMM(V, L) \
  hipLaunchKernelGGL((spmm_kernel<V, L>), dim3(grid), dim3(kBlock), 0, s, A.nrow, k, A.row_ptr, A.cols, A.vals, X, Y)
#define FS_SPMM_LG(V)                                                                                   \
  switch (lg) {                                                                                         \
    case 0: FS_SPMM(V, 0); break; case 1: FS_SPMM(V, 1); break; case 2: FS_SPMM(V, 2); break;           \
    case 3: FS_SPMM(V, 3); break; case 4: FS_SPMM(V, 4); break; case 5: FS_SPMM(V, 5); break;           \
    default: FS_SPMM(V, 6); break;                                                                      \
  }
  if (A.vals) { FS_SPMM_LG(true) } else { FS_SPMM_LG(false) }
#undef FS_SPMM_LG
#undef FS_SPMM
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// one unit-stride sweep of the LDS-staged kernel per column, on column-major copies of X and Y in the handle's scratch
static int launch_spmm_ldsx_columns(DeviceCsr &A, double *Y, const double *X, int k, hipStream_t s)
{
  int64_t ldx = 0, ldy = 0;
  if (A.spmm_scratch_doubles < spmm_scratch_need(A, k, &ldx, &ldy)) { set_error("launch_spmm: no column-major scratch (fs_matrix_prepare allocates it)"); return FS_ERR_ARG; }
  double *xt = A.spmm_scratch, *yt = A.spmm_scratch + (size_t)k * (size_t)ldx;
  if (A.ncol > 0)
    hipLaunchKernelGGL(rows_to_columns_kernel, dim3((unsigned)(((int64_t)A.ncol + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       (int64_t)A.ncol, k, ldx, X, xt);
  FS_HIP(hipGetLastError());
  for (int j = 0; j < k; ++j)
    if (int rc = launch_spmv_tiled(A, *A.tiledx, yt + (int64_t)j * ldy, xt + (int64_t)j * ldx, s, 1, 1)) return rc;
  hipLaunchKernelGGL(columns_to_rows_kernel, dim3((unsigned)(((int64_t)A.nrow + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                     (int64_t)A.nrow, k, ldy, yt, Y);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

static int launch_spmm_plan(DeviceCsr &A, int plan, double *Y, const double *X, int k, hipStream_t s)
{
  switch (plan) {
    case kPlanBinnedK: {
      const BinnedCsr &N = k == 4 ? *A.binned4 : *A.binned2;
      if (int rc = launch_spmm_binned(A, N, Y, X, s, k, k)) return rc;
      if (k == 3) return launch_spmv_binned(A, Y + 2, X + 2, s, k, k);
      return FS_OK;
    }
    case kPlanBinnedCols:
      for (int j = 0; j < k; ++j)
        if (int rc = launch_spmv_binned(A, Y + j, X + j, s, k, k)) return rc;
      return FS_OK;
    case kPlanLdsxColumns:
      return launch_spmm_ldsx_columns(A, Y, X, k, s);
    case kPlanLdsxStrided:
    case kPlanTiledStrided: {
      const TiledCsr &T = plan == kPlanLdsxStrided ? *A.tiledx : *A.tiled;
      for (int j = 0; j < k; ++j)
        if (int rc = launch_spmv_tiled(A, T, Y + j, X + j, s, k, k)) return rc;
      return FS_OK;
    }
    case kPlanMfma: {   // the matrix-core experiment (see spmm_mfma_kernel)
      if (int rc = need_plain_csr(A, "the matrix-core SpMM experiment")) return rc;
      const unsigned g4 = (unsigned)(((int64_t)A.nrow + kBlock / 64 - 1) / (kBlock / 64));
      if (A.vals) hipLaunchKernelGGL(spmm_mfma_kernel<true>, dim3(g4), dim3(kBlock), 0, s, A.nrow, k, A.row_ptr, A.cols, A.vals, X, Y);
      else        hipLaunchKernelGGL(spmm_mfma_kernel<false>, dim3(g4), dim3(kBlock), 0, s, A.nrow, k, A.row_ptr, A.cols, A.vals, X, Y);
      FS_HIP(hipGetLastError());
      return FS_OK;
    }
    default:
      return launch_spmm_row(A, Y, X, k, s);
  }
}

int launch_spmm(DeviceCsr &A, double *Y, const double *X, int k, hipStream_t s)
{
  if (A.nrow == 0) return FS_OK;
  return launch_spmm_plan(A, spmm_plan(A, k, nullptr), Y, X, k, s);
}

// The k-column product in parts: only the one-sweep plan (k = 2, 4 on the k-column two-pass copy) is cut, like the single-vector
// pair; every other plan does everything with part 0.  The cuts are cached per handle and k.
int spmm_part_bounds(DeviceCsr &A, int k, int nparts, const int **rows_out, const int **units_out, int *plan_out)
{
  const int plan = spmm_plan(A, k, nullptr);
  if (plan_out) *plan_out = plan;
  DeviceCsr::PartCuts &C = A.partk[k == 4 ? 1 : 0];
  const bool sweep = plan == kPlanBinnedK && (k == 2 || k == 4);
  if (C.n == nparts && C.kind == (sweep ? k : -k) && !C.rows.empty()) {
    *rows_out = C.rows.data(); if (units_out) *units_out = C.units.data();
    return FS_OK;
  }
  std::vector<int> rows((size_t)nparts + 1, A.nrow), units((size_t)nparts + 1, 0);
  rows[0] = 0;
  bool cut = false;
  if (sweep)
    if (int rc = binned_part_cuts(A, k == 4 ? *A.binned4 : *A.binned2, nparts, rows, units, &cut)) return rc;
  if (!cut) {
    units.assign((size_t)nparts + 1, 0);
    rows.assign((size_t)nparts + 1, A.nrow);
    rows[0] = 0;
  }
  C.n = nparts; C.kind = sweep ? k : -k; C.cut = cut;
  C.rows.swap(rows); C.units.swap(units);
  *rows_out = C.rows.data(); if (units_out) *units_out = C.units.data();
  return FS_OK;
}

int launch_spmm_part(DeviceCsr &A, double *Y, const double *X, int k, int part, int nparts, hipStream_t s)
{
  if (A.nrow == 0) return FS_OK;
  if (k != 2 && k != 4) return part == 0 ? launch_spmm(A, Y, X, k, s) : FS_OK;    // (k = 3 is two sweeps: not cut)
  const int *rows = nullptr, *units = nullptr;
  int plan = 0;
  if (int rc = spmm_part_bounds(A, k, nparts, &rows, &units, &plan)) return rc;
  if (!A.partk[k == 4 ? 1 : 0].cut) return part == 0 ? launch_spmm_plan(A, plan, Y, X, k, s) : FS_OK;
  return launch_spmm_binned(A, k == 4 ? *A.binned4 : *A.binned2, Y, X, s, k, k, units[part], units[part + 1], rows[part], rows[part + 1]);
}

namespace {
struct EventPair {   // destroyed however the function leaves
  hipEvent_t a = nullptr, b = nullptr;
  int create() { FS_HIP(hipEventCreate(&a)); FS_HIP(hipEventCreate(&b)); return FS_OK; }
  ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
struct Scratch2 {    // zero operands of a timed run
  double *x = nullptr, *y = nullptr;
  int alloc(size_t nx, size_t ny)
  {
    FS_HIP(hipMalloc(&x, sizeof(double) * (nx ? nx : 1)));
    FS_HIP(hipMalloc(&y, sizeof(double) * (ny ? ny : 1)));
    return FS_OK;
  }
  ~Scratch2() { if (x) (void)hipFree(x); if (y) (void)hipFree(y); }
};
}  // namespace

// Everything a product with k columns on this matrix may need beyond a launch: the k-column two-pass copy (k = 2..4),
// the column-major scratch and the measured choice between column sweeps and the row kernel (LDS-staged copy, k = 3..16).
// Synchronous (builds and timed runs wait for the device); idempotent and cheap once done.
int prepare_spmm(DeviceCsr &A, int k, hipStream_t s)
{
  if (A.nrow == 0 || A.nnz == 0 || k < 2) return FS_OK;
  int needs = 0;
  int plan = spmm_plan(A, k, &needs);
  if (needs & 1) {
    if (int rc = need_plain_csr(A, "fs_matrix_prepare (the k-column copy is built from the plain arrays)")) return rc;
    if (int rc = build_binned_k(A, k == 4 ? 4 : 2, s)) return rc;     // declines (and says so in tried2 / tried4) where it would not pay
    plan = spmm_plan(A, k, &needs);
  }
  if (needs & 4) {                                // scratch first: its hipMalloc may stall and must not be inside a timed run
    if (spmm_scratch_alloc(A, k) != FS_OK) {
      (void)hipGetLastError();                     // no room for the column-major copies: this k stays on the strided sweeps / the row kernel
      return FS_OK;
    }
    plan = spmm_plan(A, k, &needs);
  }
  (void)plan;
  if (!(needs & 2)) return FS_OK;
  // LDS-staged copy, k = 3..16: time one run of each candidate on zero operands (same addresses and traffic as any X)
  // after an untimed run of each (code objects loaded, TLB warm)
  Scratch2 xy;
  if (xy.alloc((size_t)A.ncol * k, (size_t)A.nrow * k) != FS_OK) { (void)hipGetLastError(); return FS_OK; }
  FS_HIP(hipMemsetAsync(xy.x, 0, sizeof(double) * (size_t)A.ncol * k, s));
  EventPair e0, e1;
  if (e0.create() != FS_OK || e1.create() != FS_OK) { (void)hipGetLastError(); return FS_OK; }
  float t_sweeps = 0.f, t_row = 0.f;
  for (int cand = 0; cand < 2; ++cand) {
    const int pl = cand == 0 ? kPlanLdsxColumns : kPlanRow;
    if (int rc = launch_spmm_plan(A, pl, xy.y, xy.x, k, s)) return rc;            // warm-up
    FS_HIP(hipEventRecord(cand == 0 ? e0.a : e1.a, s));
    if (int rc = launch_spmm_plan(A, pl, xy.y, xy.x, k, s)) return rc;
    FS_HIP(hipEventRecord(cand == 0 ? e0.b : e1.b, s));
  }
  FS_HIP(hipEventSynchronize(e1.b));
  FS_HIP(hipEventElapsedTime(&t_sweeps, e0.a, e0.b));
  FS_HIP(hipEventElapsedTime(&t_row, e1.a, e1.b));
  A.spmm_choice[k] = t_sweeps <= t_row ? 1 : 2;
  static const bool trace = getenv("FS_TRACE_BUILD") != nullptr;
  if (trace)
    fprintf(stderr, "[fastsparse] %d x %d, k = %d: one sweep per column %.2f ms, row kernel %.2f ms\n", A.nrow, A.ncol, k,
            t_sweeps, t_row);
  return FS_OK;
}

// y[r] = 0 + sum over column blocks, in block order, of the cell sums (cbcsr.h:96-103 with one thread)
__global__ __launch_bounds__(kBlock) void cbcsr_combine_kernel(int nrow, int nblocks, const double *__restrict__ cell,
                                                              double *__restrict__ y)
{
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= nrow) return;
  double tot = 0.0;
  for (int b = 0; b < nblocks; ++b) tot += cell[(int64_t)b * nrow + r];
  y[r] = 0.0 + tot;
}

int launch_cbcsr(const fs_cbcsr_s &A, double *y, const double *x, hipStream_t s)
{
  if (A.nrow == 0) return FS_OK;
  // largest matrices: the plain CSR of the same entries on the general SpMV path (spmv_kernel 4 / 5 / 9 force the
  // column-block kernels below)
  // (not under strict_order: the reference adds cell sums, not one flat sum per row)
  if (A.use_rows && !options().strict_order && options().spmv_kernel != 4 && options().spmv_kernel != 5 &&
      options().spmv_kernel != 9)
    return launch_spmv(A.rows, y, x, s);
  // large matrices: cell sums by the chunk-streaming kernel over the (block, row) cells, then one pass that
  // adds each row's cells block by block.  spmv_kernel 4 / 5 force the one-thread-per-row kernels below.
  if (A.use_cells && options().spmv_kernel != 4 && options().spmv_kernel != 5) {
    if (int rc = launch_spmv(A.cells, A.cell_sums, x, s, /*force_stream=*/true)) return rc;
    hipLaunchKernelGGL(cbcsr_combine_kernel, dim3((unsigned)(((int64_t)A.nrow + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       A.nrow, A.nblocks, A.cell_sums, y);
    FS_HIP(hipGetLastError());
    return FS_OK;
  }
  const unsigned grid = (unsigned)(((int64_t)A.nrow + kBlock - 1) / kBlock);
  // stage the x tile in LDS when it fits and a workgroup's cells of one block hold enough
  // entries to amortise the copy (one tile read per workgroup per block)
  const double per_wg_block = A.nblocks ? (double)A.nnz / A.nblocks / grid : 0.0;
  const int forced = options().spmv_kernel;  // 4 = always stage (if it fits), 5 = never
  bool stage = A.colblocksize <= kCbTile && per_wg_block * 4 >= A.colblocksize;
  if (forced == 4) stage = A.colblocksize <= kCbTile;
  if (forced == 5) stage = false;
  if (stage)
    hipLaunchKernelGGL(cbcsr_kernel<true>, dim3(grid), dim3(kBlock), 0, s, A.nrow, A.ncol, A.nblocks, A.colblocksize,
                       A.row_ptr, A.cols, x, y);
  else
    hipLaunchKernelGGL(cbcsr_kernel<false>, dim3(grid), dim3(kBlock), 0, s, A.nrow, A.ncol, A.nblocks, A.colblocksize,
                       A.row_ptr, A.cols, x, y);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

}  // namespace fs
