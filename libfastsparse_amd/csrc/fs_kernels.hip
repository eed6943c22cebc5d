// fs_kernels.hip -- hand-written gfx950 kernels of the A_mul_B / At_mul_B path.
//
// Compiled with -ffp-contract=off: a valued term is one rounded multiply followed by one
// rounded add, exactly what the strict-IEEE reference loop `tmp += x[cols[i]] * vals[i]`
// (csr.h:434, dsparse.h:49) computes, so that every kernel that adds a row's terms in
// storage order is bit-identical to the CPU order.  All kernels are HBM/gather bound
// (0.17 flop/B), the spare multiply issue slot costs nothing.
//
// Kernels
//   spmv_expand_kernel / spmv_reduce_kernel
//                        y = A x in two streaming passes on the two-pass copy (column bands x row panels): the
//                        default for large matrices.  Pass 1 keeps a band of x in LDS and writes the products,
//                        regrouped by row panel, to HBM; pass 2 keeps a panel of y in LDS and adds them up.
//                        Neither pass gathers from L2 or HBM.
//   spmv_ldsx_dma_kernel / spmv_ldsx_pipe_kernel
//                        y = A x on a tiled copy with bands of <= 2048 columns: the band's slice of x is staged in LDS
//                        (by LDS DMA when x is contiguous and aligned), so gathers and adds are LDS operations.  For dense
//                        tiles (config 3, cbcsr).  ata_ldsx_kernel: the fused A'A x on the same copy (opt-in).
//   spmv_tiled_kernel    y = A x on the L2-tiled copy (row panels x column bands): x gathered from L2 inside
//                        the current band.  One 1024-thread workgroup per CU; producer waves stream the
//                        entries and gather, consumer waves reduce the staged products into the panel's y
//                        slice in LDS.  The fixed-order (run-to-run reproducible) kernel for large matrices.
//   (which of the three runs on a matrix is the format builder's measured choice: fs_format.hip, choose_copy)
//   spmv_stream_kernel   y = A x, CSR or pattern-only CSR.  One 256-thread workgroup streams a
//                        fixed 2048-non-zero chunk of cols/vals with 16-byte loads, gathers x,
//                        parks the products in LDS and reduces them per row.  Work per
//                        workgroup is independent of the row-length distribution.
//   spmv_fixup_kernel    adds up the partial sums of rows that cross chunk boundaries
//   spmv_vector_kernel   classic G-lanes-per-row CSR kernel (A/B alternative, option spmv_kernel=2)
//   spmm_kernel          Y = A X, k row-major right-hand sides, one lane per output column
//   cbcsr_kernel         column-blocked binary CSR, x tile staged in LDS per column block
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "fs_common.h"

namespace fs {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// LDS slot of the i-th product of a chunk: one pad slot per 16 products makes the stride
// between consecutive 16-entry rows odd (17), so a thread-per-row sweep is conflict free.
__device__ __forceinline__ int lds_slot(int i) { return i + (i >> 4); }
constexpr int kLdsDoubles = kChunk + (kChunk >> 4) + 8;

template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <bool NT, typename T>
__device__ __forceinline__ void stream_store(T v, T *p)
{
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

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

// strict_order variant: a row that crosses chunks must still be ONE left-to-right sum, so the
// thread continues the running sum tail[c] with the remaining terms themselves (recomputed from
// global memory) instead of adding the later chunks' partial sums.
template <bool VALUED>
__global__ void spmv_fixup_strict_kernel(int nchunks, int64_t nnz, const int *__restrict__ row_ptr,
                                         const int *__restrict__ first_row, const int *__restrict__ cols,
                                         const double *__restrict__ vals, const double *__restrict__ x,
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
  for (int64_t i = e; i < b; ++i) {
    const double xv = x[cols[i]];
    acc += VALUED ? xv * vals[i] : xv;
  }
  y[r] = acc;
}

// ------------------------------------------------------------------------------------------
// y = A x on the L2-tiled copy (TiledCsr, fs_common.h).  Same products and the same per-row
// terms as csr_A_mul_B (csr.h:425-438); the order in which a row's terms are added is
// band-major (deterministic, run-to-run reproducible), so arbitrary x agrees with the CPU order
// to rounding (1e-12 bar) and integer-valued x bit for bit.
//
// Why: with x far larger than the 4 MiB L2 of an XCD every gather of x[col] misses and pulls a
// whole line across the fabric; measured 53-56 G gathers/s however the kernel is shaped, against
// ~240 G/s when the gathered range is L2-resident (tools/probe_gather, profiles/).  Here one
// workgroup owns a panel of R rows (its y slice lives in LDS) and sweeps the column bands left
// to right; the workgroups resident together start together and advance at the same pace, so at
// any moment an XCD gathers from one or two bands (<= 2 MiB each) that stay in its L2.
//
// Per work item (<= 2048 consecutive entries of one tile): coalesced loads of the packed
// (head, row, col) words and the values, gathers of x inside the band, products parked in LDS,
// barrier, then every entry that starts a row-run adds the run's sum into its y slot (rows of
// different runs are distinct inside an item, so plain LDS read-add-write is race free).
// The next item's entries are loaded while the current one is being reduced.
// ------------------------------------------------------------------------------------------
constexpr int kTiledPer = kTiledItem / kTiledProd;  // 4 entries per producer (and per consumer) thread

// ---- producer side (waves 0-7): entries of one item for producer thread tp are positions q*512 + tp.
// Whole 512-entry slabs past the item's end are skipped (wave-uniform test); inside the last slab the
// position is clamped to the last entry, so the loads themselves are unconditional and the clamped lanes
// re-read one cached word.
// Pattern-only: every load is unconditional and nothing touches its result before the phase that needs it -- a load
// under a branch (or a select on its result) makes the compiler lose count of what is in flight and wait for more
// than it has to (LDS-staged kernel below: 1.63 ms with skipped slabs, 1.08 ms with straight-line phases; here 0.85
// -> 0.82 ms).  Valued: the kernel sits at the 128-register limit and the 512-entry slabs past an item's end are
// still skipped (unconditional: 1.23 ms, skipped: 1.05 ms on config 2).
template <bool VALUED, bool NT>
__device__ __forceinline__ void tiled_load(const int4 d, int tp, const unsigned *__restrict__ pk,
                                           const double *__restrict__ vals, unsigned (&w)[kTiledPer],
                                           double (&v)[kTiledPer])
{
  const int last = d.y > 0 ? d.y - 1 : 0;
#pragma unroll
  for (int q = 0; q < kTiledPer; ++q) {
    if (!VALUED || q * kTiledProd < d.y || q == 0) {   // see above: slabs past the item's end are skipped when valued
      const int pos = q * kTiledProd + tp;
      const int64_t e = (int64_t)d.x + (pos < last ? pos : last);
      w[q] = stream_load<NT>(pk + e);
      if (VALUED) v[q] = stream_load<NT>(vals + e);
    }
  }
}

// x may be one column of a row-major k-column X: element c lives at x[c * xs] (xs = 1 for a plain vector)
template <bool VALUED>
__device__ __forceinline__ void tiled_gather(const int4 d, int W, unsigned cmask, const double *__restrict__ x, int xs,
                                             const unsigned (&w)[kTiledPer], double (&xv)[kTiledPer])
{
  const double *xb = x + (int64_t)d.z * W * xs;
#pragma unroll
  for (int q = 0; q < kTiledPer; ++q)
    if (!VALUED || q * kTiledProd < d.y || q == 0) xv[q] = xb[(int64_t)(w[q] & cmask) * xs];
}

// products and packed words of one item into a stage buffer (entry i at spk[i + 1]; spk[0], spk[n + 1] guards)
template <bool VALUED>
__device__ __forceinline__ void tiled_stage(double *__restrict__ sprod, unsigned *__restrict__ spk, int tp, int n,
                                            const unsigned (&w)[kTiledPer], const double (&v)[kTiledPer],
                                            const double (&xv)[kTiledPer])
{
#pragma unroll
  for (int q = 0; q < kTiledPer; ++q) {
    const int pos = q * kTiledProd + tp;
    if (pos < n) {
      sprod[pos] = VALUED ? xv[q] * v[q] : xv[q];
      spk[pos + 1] = w[q];
    }
  }
  if (tp == 0) { spk[0] = 0xFFFFFFFFu; spk[n + 1] = 0xFFFFFFFFu; }  // row id no entry has: runs stop at both ends
}

// ---- consumer side (waves 8-15): every entry that starts a row-run adds the run's sum to its row of the
// y slice.  The runs of one item are distinct rows: one add per address, so the LDS atomic (fire and forget,
// no read-add-write chain in the wave) gives the same bits as a plain update, in a fixed order.
__device__ __forceinline__ void tiled_reduce(double *__restrict__ ytile, const double *__restrict__ sprod,
                                             const unsigned *__restrict__ spk, int tc, int n, int lcol_bits)
{
#pragma unroll
  for (int q = 0; q < kTiledPer; ++q) {
    const int pos = q * kTiledProd + tc;
    if (pos < n) {
      const unsigned lr = spk[pos + 1] >> lcol_bits;
      if ((spk[pos] >> lcol_bits) != lr) {  // previous entry is another row (or the item's start): run head
        double sum = sprod[pos];
        int k = pos + 1;
        while ((spk[k + 1] >> lcol_bits) == lr) { sum += sprod[k]; ++k; }
        __hip_atomic_fetch_add(&ytile[lr], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}

// ONE 1024-thread workgroup per CU and row panel.  Waves 0-7 are producers: they stream the panel's
// entries, gather x inside the current column band and park products in one of two LDS stage buffers.
// Waves 8-15 are consumers: they reduce the other stage buffer into the y slice.  One barrier per item
// separates the roles' phases, so the LDS reduction of item k-1 overlaps the memory work of items k..k+3.
// Producer software pipeline: four register sets rotate by name (a register copy would force in-flight
// loads to complete); while item k is staged, the gathers of k+1 and k+2 and the entry loads of k+3 are in
// flight.  Inside a phase the loads of k+3 are issued before the gathers of k+2: vmcnt retires in order
// and the loads are needed one phase earlier than the gathers issued with them.
template <bool VALUED, bool NT, bool DEBUG = false>
__global__ __launch_bounds__(kTiledBlock, 4) void spmv_tiled_kernel(
    const int *__restrict__ panel_row, int W, int lcol_bits, const int4 *__restrict__ items,
    const int *__restrict__ item_ptr, const unsigned *__restrict__ pk, const double *__restrict__ vals,
    const double *__restrict__ x, double *__restrict__ y, int xs, int ys, long long *__restrict__ dbg_time = nullptr,
    int *__restrict__ dbg_xcc = nullptr)
{
  __shared__ double ytile[kTiledRowsMax];
  __shared__ double sprod[2][kTiledItem];
  __shared__ unsigned spk[2][kTiledItem + 2];
  const int t = threadIdx.x;
  const bool producer = t < kTiledProd;      // wave-uniform: waves 0-7
  const int tr = producer ? t : t - kTiledProd;  // index inside the role
  const int p = blockIdx.x;
  const int row0 = panel_row[p];
  const int nr = panel_row[p + 1] - row0;
  for (int i = t; i < nr; i += kTiledBlock) ytile[i] = 0.0;
  const unsigned cmask = (1u << lcol_bits) - 1u;
  const int it0 = item_ptr[p], it1 = item_ptr[p + 1];
  const int4 none = make_int4(0, 0, 0, 0);
  // descriptor reads outside the panel are clamped to its items (an empty panel reads the item in front of it; the
  // array always holds at least one) ...
  const int itl = it1 > it0 ? it1 - 1 : (it0 > 0 ? it0 - 1 : 0);
  const int itf = it1 > it0 ? it0 : itl;
  auto item_at = [&](int i) {
    int4 d = items[i < itf ? itf : (i < itl ? i : itl)];
    if (i < it0 || i >= it1) d.y = 0;          // ... and emptied: one entry is loaded and gathered, nothing is staged
    return d;
  };
#define FS_ITEM(i) item_at(i)
  if (DEBUG && t == 0) {  // diagnostic build only: which XCD runs this panel, and when each item starts
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg_xcc[p] = (int)(xcc & 0xf);
  }
  // There is no separate prologue: the sweep starts three phases early on empty items with zeroed register sets
  // (local column 0 of band 0 is a valid address), so the pipeline fills through the same code that keeps it full
  // and the compiler sees one steady state of loads in flight at the loop's back edge.
  int4 dA = none, dB = none, dC = none, dD = none;
  unsigned wA[kTiledPer] = {}, wB[kTiledPer] = {}, wC[kTiledPer] = {}, wD[kTiledPer] = {};
  double vA[kTiledPer] = {}, vB[kTiledPer] = {}, vC[kTiledPer] = {}, vD[kTiledPer] = {};
  double xA[kTiledPer] = {}, xB[kTiledPer] = {}, xC[kTiledPer] = {}, xD[kTiledPer] = {};
  __syncthreads();  // ytile zeroed
  // phase IT: producers stage item IT (register set 0) into buffer IT&1, then issue the loads of IT+3
  // (set 3) and the gathers of IT+2 (set 2); consumers reduce item IT-1 from the other buffer.
#define FS_PHASE(IT, D0, W0, V0, X0, D2, W2, X2, D3, W3, V3)                                   \
  if (producer) {                                                                              \
    if (DEBUG && t == 0 && (IT) >= it0 && (IT) < it1) dbg_time[(IT)] = (long long)wall_clock64(); \
    if ((IT) >= it0 && (IT) < it1) tiled_stage<VALUED>(sprod[(IT) & 1], spk[(IT) & 1], tr, D0.y, W0, V0, X0); \
    D3 = FS_ITEM((IT) + 3);                                                                    \
    tiled_load<VALUED, NT>(D3, tr, pk, vals, W3, V3);                                          \
    tiled_gather<VALUED>(D2, W, cmask, x, xs, W2, X2);                                         \
  } else if ((IT) > it0 && (IT) <= it1) {                                                      \
    tiled_reduce(ytile, sprod[((IT) - 1) & 1], spk[((IT) - 1) & 1], tr, items[(IT) - 1].y, lcol_bits); \
  }                                                                                            \
  __syncthreads();
  // whole rounds of four phases (no early exit: a loop body with one way through is what lets the compiler count
  // the loads in flight); phases past the last item stage and reduce nothing
  for (int it = it0 - 3; it <= it1; it += 4) {
    FS_PHASE(it, dA, wA, vA, xA, dC, wC, xC, dD, wD, vD)
    FS_PHASE(it + 1, dB, wB, vB, xB, dD, wD, xD, dA, wA, vA)
    FS_PHASE(it + 2, dC, wC, vC, xC, dA, wA, xA, dB, wB, vB)
    FS_PHASE(it + 3, dD, wD, vD, xD, dB, wB, xB, dC, wC, vC)
  }
#undef FS_ITEM
#undef FS_PHASE
  for (int i = t; i < nr; i += kTiledBlock) y[(int64_t)(row0 + i) * ys] = ytile[i];
}

// ------------------------------------------------------------------------------------------
// y = A x on a tiled copy whose bands are narrow enough for the band's slice of x to live in LDS
// (W <= kLdsxCols): the north_star's "LDS staging of the dense x tile".  For matrices whose tiles are
// dense enough (config 3: 10 M x 1 M, 64 per row -> 1 700 entries per 13 021 x 2 048 tile) loading the
// slice costs less than gathering from L2 entry by entry: a slice is 128 full lines from L2, the tile's
// gathers would be 1 700 separate requests.
//
// ONE 1024-thread workgroup per CU and row panel, y slice (<= 120 KiB) and two x slices in LDS, all 16
// waves in the same role.  Phase IT = work item IT (<= 2048 entries of one tile, 2 per thread): gather x from
// the LDS slice of the item's band, multiply, ds_add_f64 into the y slice.  Memory runs three phases ahead
// in registers (four register sets rotating by name): in phase IT the entries and the x slice of item IT+3
// are requested, the slice of item IT+1 is copied from registers to the other LDS buffer, and one barrier
// ends the phase.  Sum order: band-major, inside an item by LDS atomics in arrival order (see the two-pass
// kernels above for what that means).  Because the order inside an item is free, the format builder arranges every
// item so that the 32 lanes of a half-wave add into 32 different LDS bank pairs (local row mod 32, round-robin over
// the residue classes): ds_add_f64 on random rows runs at 2.97 lanes per clock, conflict-free at 6.9, and the adds
// are the largest share of the LDS time (config 3: 1.07 -> 0.87 ms with perfectly conflict-free rows).
// ------------------------------------------------------------------------------------------
// s_waitcnt immediate of gfx9: vmcnt in bits 3:0 and 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8
#define FS_WAIT_IMM(VM, LGKM) (((VM) & 0xF) | (0x7 << 4) | (((LGKM) & 0xF) << 8) | (((VM) >> 4) << 14))

// The panel's slice of y leaves LDS.  A workgroup that owns its rows stores them.  Chunks that share a panel add theirs into the
// (zeroed) output with HBM atomics -- in arrival order, or, for fixed-order sums (`ordered`), one chunk after the other in the
// order of their ordinals inside the panel: *ticket says whose turn it is.  Chunks of one panel are launched in ascending
// ordinal order and workgroups are dispatched in index order, so the chunk waited for is running or done; the wait is bounded
// all the same (a chunk that gives up adds out of turn: a wrong ORDER, never a hang).
__device__ __forceinline__ void ldsx_store_slice(const double *__restrict__ ytile, int nr, int row0, double *__restrict__ y, int ys, bool shared,
                                                 bool ordered, int *__restrict__ ticket, int ord)
{
  const int t = threadIdx.x;
  if (!shared) {
    for (int i = t; i < nr; i += kTiledBlock) y[(int64_t)(row0 + i) * ys] = ytile[i];
    return;
  }
  // (the adds are device-scope atomics, performed at the memory side, and the ticket is read and written there too: relaxed
  // accesses and a wait for this chunk's atomics to be acknowledged order the chunks' adds -- no cache flush is involved, which an
  // acquire / release pair would cost on every chunk)
  if (ordered) {
    if (t == 0) {
      int spins = 0;
      while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ord && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
  }
  for (int i = t; i < nr; i += kTiledBlock) unsafeAtomicAdd(&y[(int64_t)(row0 + i) * ys], ytile[i]);
  if (ordered) {
    __builtin_amdgcn_s_waitcnt(FS_WAIT_IMM(0, 0));   // this thread's adds are acknowledged ...
    __syncthreads();                                 // ... and everybody's: the next chunk may start its own
    if (t == 0) __hip_atomic_store(ticket, ord + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

constexpr int kLdsxSets = 4;                          // register sets = items in flight (8 measured no faster: 0.92 vs
                                                      // 0.89 ms on config 3, 1.87 vs 1.84 ms on its transpose)
constexpr int kLdsxPer = kTiledItem / kTiledBlock;    // entries per thread and item (2)
constexpr int kLdsxXPer = kLdsxCols / kTiledBlock;    // x values per thread and slice (2)

// Every load of a phase is unconditional and its result is not touched before the phase that needs it (addresses
// are clamped here, lanes outside the slice are masked when the slice is published): a load under a branch, or a
// select on its result, makes the compiler wait for it on the spot and the pipeline collapses.
template <bool VALUED, bool NT>
__device__ __forceinline__ void ldsx_load(const int4 d, int t, int W, int ncol, const unsigned *__restrict__ pk,
                                          const double *__restrict__ vals, const double *__restrict__ x, int xs,
                                          unsigned (&w)[kLdsxPer], double (&v)[kLdsxPer], double (&xr)[kLdsxXPer])
{
  // the slice first: it is needed one phase before the entries and vmcnt retires in order
  const int c0 = d.z * W;
#pragma unroll
  for (int q = 0; q < kLdsxXPer; ++q) {
    const int lc = q * kTiledBlock + t;
    const int c = c0 + lc;
    xr[q] = x[(int64_t)(c < ncol ? c : ncol - 1) * xs];
  }
  const int last = d.y > 0 ? d.y - 1 : 0;
#pragma unroll
  for (int q = 0; q < kLdsxPer; ++q) {
    const int pos = q * kTiledBlock + t;
    const int64_t e = (int64_t)d.x + (pos < last ? pos : last);
    w[q] = stream_load<NT>(pk + e);
    if (VALUED) v[q] = stream_load<NT>(vals + e);
  }
}

// The fused y = A'A x of bcsr_AA_mul_B (csr.h:305-319) on the LDS-staged copy (fs_ata_mul, option ata_kernel = 2; opt-in: measured
// slower than the two products it replaces).  A workgroup takes one chunk = one whole panel (the launcher checks).  Sweep 1 is the
// LDS-staged product in its first, simplest form -- phase IT gathers x from the slice of item IT and adds into the y slice, memory
// NSETS - 1 phases ahead in registers -- and leaves t = (A x) of the panel's rows in LDS; sweep 2 walks the same work items again
// and scatters t back through the tiles, y[col] += t[row], accumulating a band's slice of y in LDS (the buffer the x slices
// used) and adding it to y in HBM with atomics whenever the sweep moves to another band.  One pass over A's copy per sweep, no
// copy of A'.  Needs a zeroed y.
template <bool VALUED, bool NT, int NSETS>
__global__ __launch_bounds__(kTiledBlock) void ata_ldsx_kernel(
    const int *__restrict__ panel_row, int W, int lcol_bits, int ncol, const int4 *__restrict__ items,
    const int *__restrict__ chunk_panel, const int *__restrict__ chunk_item, const unsigned *__restrict__ pk,
    const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y, int xs, int ys)
{
  __shared__ double ytile[kLdsxRows];
  __shared__ double xsl[2][kLdsxCols];
  const int t = threadIdx.x;
  const int p = chunk_panel[blockIdx.x] & 0x7fffffff;
  const int row0 = panel_row[p];
  const int nr = panel_row[p + 1] - row0;
  for (int i = t; i < nr; i += kTiledBlock) ytile[i] = 0.0;
  const unsigned cmask = (1u << lcol_bits) - 1u;
  const int it0 = chunk_item[2 * blockIdx.x], it1 = chunk_item[2 * blockIdx.x + 1];   // [first, last) work item of the chunk
  // descriptor reads outside the chunk are clamped to its items (an empty chunk reads the item in front of it; the
  // array always holds at least one) ...
  const int itl = it1 > it0 ? it1 - 1 : (it0 > 0 ? it0 - 1 : 0);
  const int itf = it1 > it0 ? it0 : itl;
  auto item = [&](int i) {
    int4 d = items[i < itf ? itf : (i < itl ? i : itl)];
    if (i < it0 || i >= it1) d.y = 0;           // ... and emptied: loads one entry and one slice, contributes nothing
    return d;
  };
  // NSETS register sets hold the items in flight (set k % NSETS belongs to item k; all indices below are constants
  // after unrolling, so the sets are registers).  No separate prologue: the sweep starts NSETS-1 phases early on empty
  // items (see spmv_tiled_kernel).
  int4 dset[NSETS];
  unsigned w[NSETS][kLdsxPer];
  double v[NSETS][kLdsxPer];
  double xr[NSETS][kLdsxXPer];
#pragma unroll
  for (int k = 0; k < NSETS; ++k) {
    dset[k] = item(it0 - 1);                     // an empty descriptor with valid addresses
#pragma unroll
    for (int q = 0; q < kLdsxPer; ++q) { w[k][q] = 0; v[k][q] = 0.0; }
#pragma unroll
    for (int q = 0; q < kLdsxXPer; ++q) xr[k][q] = 0.0;
  }
  const int first = it0 - (NSETS - 1);
  int4 dN = item(first + NSETS - 1);              // descriptor of the item the first phase requests
  __syncthreads();   // ytile zeroed
  // phase IT (IT = it + ph, ph constant): request item IT+NSETS-1 into the set item IT-1 has just left (its
  // descriptor was fetched a phase ago) and fetch the descriptor after it; consume item IT from slice buffer IT&1;
  // publish the slice of item IT+1 in the other buffer.  Whole rounds of NSETS phases, no early exit (phases past the
  // last item add nothing: their items are empty).  Buffer parity: `first` may be odd, so it is carried explicitly.
  for (int it = first; it < it1; it += NSETS) {
#pragma unroll
    for (int ph = 0; ph < NSETS; ++ph) {
      const int IT = it + ph;
      const int s0 = ph, s1 = (ph + 1) % NSETS, sl = (ph + NSETS - 1) % NSETS;
      dset[sl] = dN;
      dN = item(IT + NSETS);
      ldsx_load<VALUED, NT>(dset[sl], t, W, ncol, pk, vals, x, xs, w[sl], v[sl], xr[sl]);
      const int buf = IT & 1;
#pragma unroll
      for (int q = 0; q < kLdsxPer; ++q) {
        const int pos = q * kTiledBlock + t;
        if (pos < dset[s0].y) {
          double pr = xsl[buf][w[s0][q] & cmask];
          if (VALUED) pr *= v[s0][q];
          __hip_atomic_fetch_add(&ytile[w[s0][q] >> lcol_bits], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
#pragma unroll
      for (int q = 0; q < kLdsxXPer; ++q) {
        const int lc = q * kTiledBlock + t;
        xsl[buf ^ 1][lc] = (lc < W && dset[s1].z * W + lc < ncol) ? xr[s1][q] : 0.0;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  double *yb = xsl[0];                              // the band's slice of y (W <= kLdsxCols doubles)
  for (int i = t; i < kLdsxCols; i += kTiledBlock) yb[i] = 0.0;
  int band = it0 < it1 ? items[it0].z : 0;
  __syncthreads();
  for (int it = it0; it < it1; ++it) {
    const int4 d = items[it];
    if (d.z != band) {                              // wave-uniform: the sweep leaves the band, its slice goes to HBM
      __syncthreads();
      for (int lc = t; lc < W; lc += kTiledBlock) {
        const double v = yb[lc];
        if (v != 0.0) unsafeAtomicAdd(&y[(int64_t)band * W + lc], v);
        yb[lc] = 0.0;
      }
      band = d.z;
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < kLdsxPer; ++q) {
      const int pos = q * kTiledBlock + t;
      if (pos < d.y) {
        const unsigned wq = pk[(int64_t)d.x + pos];
        double pr = ytile[wq >> lcol_bits];
        if (VALUED) pr *= vals[(int64_t)d.x + pos];
        __hip_atomic_fetch_add(&yb[wq & cmask], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  for (int lc = t; lc < W; lc += kTiledBlock) {
    const double v = yb[lc];
    if (v != 0.0) unsafeAtomicAdd(&y[(int64_t)band * W + lc], v);
  }
}

// ------------------------------------------------------------------------------------------
// y = A'A x, fused, on the plain CSR: one wave per row adds up xv = sum x[cols] and scatters it back,
// y[cols] += xv, with HBM atomics (the loop nest of bcsr_AA_mul_B, csr.h:305-319, rows in parallel like
// parallel_bcsr_AA_mul_B csr.h:323-355, whose per-thread replicas of y become atomics).  The general form of the
// fused product: any matrix, no copy at all; y must be zeroed first.
// ------------------------------------------------------------------------------------------
template <bool VALUED>
__global__ __launch_bounds__(kBlock) void ata_csr_kernel(int nrow, const int *__restrict__ row_ptr, const int *__restrict__ cols,
                                                        const double *__restrict__ vals, const double *__restrict__ x,
                                                        double *__restrict__ y)
{
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= nrow) return;
  const int a = row_ptr[row], b = row_ptr[row + 1];
  double acc = 0.0;
  for (int64_t i = (int64_t)a + lane; i < b; i += 64) {
    const double xv = x[cols[i]];
    acc += VALUED ? xv * vals[i] : xv;
  }
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
  for (int64_t i = (int64_t)a + lane; i < b; i += 64) unsafeAtomicAdd(&y[cols[i]], VALUED ? acc * vals[i] : acc);
}

// The same sweep with (1) the LDS work of consecutive items overlapped and (2) half as many vector-memory instructions.
//
// (1) In the first version (what ata_ldsx_kernel's first sweep still is) a phase is a dependency chain per wave -- gather, wait, add, gather, wait, add, publish, drain,
// barrier -- that all 16 waves walk in step.  Here item k's slice is published in phase k-2, its x values are gathered in
// phase k-1 and added in phase k: nothing inside a phase waits for anything issued in it.  Costs a third slice buffer
// (panels of <= 14336 rows) and two more register sets.  Worth 2 % (A) to 8 % (A') on config 3.
// (2) What the kernel was really bound by (ablation, gpurun_out/r2n: without its global loads 0.52 ms, with them 0.88 ms,
// while dropping the atomics, the gathers or the slice writes gained 4-5 % each and deeper pipelines nothing): the number
// of vector-memory INSTRUCTIONS.  A wave64 memory instruction occupies the CU's address unit for ~16 cycles whatever its
// width, and a phase issued 64 of them (two 4-byte entry loads and two 8-byte slice loads per thread): ~1000 of the
// phase's ~1400 cycles.  Now a thread loads its two entries -- adjacent ones, the builder arranges the items for that
// (ldsx_reorder_kernel) -- with ONE 8-byte load and its two slice values with ONE 16-byte load.
typedef unsigned v2u_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));

template <bool XS1>
__device__ __forceinline__ void ldsx_load2_slice(const int4 d, int t, int W, int ncol, const double *__restrict__ x, int xs,
                                                 double (&xr)[2])
{
  const int c0 = d.z * W + 2 * t;
  // the pair clamped into the vector: past the end it is (ncol-2, ncol-1), so the LAST column arrives in the second value
  // when c0 == ncol - 1; which value is which is sorted out when the slice is published -- no select next to the load
  const int cc = c0 + 1 < ncol ? c0 : (ncol >= 2 ? ncol - 2 : 0);
  if (XS1) {
    const v2d_a8 p = *reinterpret_cast<const v2d_a8 *>(x + cc);
    xr[0] = p.x; xr[1] = p.y;
  } else {
    xr[0] = x[(int64_t)cc * xs];
    xr[1] = x[(int64_t)(cc + 1 < ncol ? cc + 1 : cc) * xs];
  }
}

template <bool VALUED, bool NT>
__device__ __forceinline__ void ldsx_load2_entries(const int4 d, int t, const unsigned *__restrict__ pk,
                                                   const double *__restrict__ vals, unsigned (&w)[2], double (&v)[2])
{
  // entries 2t, 2t+1 of the item; threads wholly past its end re-read its first pair (masked at the add), the thread
  // on an odd end reads one entry of the next item (or of the slack behind the array), masked too
  const int64_t e = (int64_t)d.x + (2 * t < d.y ? 2 * t : 0);
  // (the pair types carry the alignment of ONE element: spelled out here, a template would deduce the plain vector type)
  const v2u_a4 *pp = reinterpret_cast<const v2u_a4 *>(pk + e);
  const v2u_a4 pw = NT ? __builtin_nontemporal_load(pp) : *pp;
  w[0] = pw.x; w[1] = pw.y;
  if (VALUED) {
    const v2d_a8 *vp = reinterpret_cast<const v2d_a8 *>(vals + e);
    const v2d_a8 pv = NT ? __builtin_nontemporal_load(vp) : *vp;
    v[0] = pv.x; v[1] = pv.y;
  }
}

#ifndef FS_PIPE_SETS
#define FS_PIPE_SETS 6
#endif
constexpr int kLdsxPipeSets = FS_PIPE_SETS;       // item k requested in phase k-5, its slice needed in phase k-2

template <bool VALUED, bool NT, bool XS1, int NSETS>
__global__ __launch_bounds__(kTiledBlock) void spmv_ldsx_pipe_kernel(
    const int *__restrict__ panel_row, int W, int lcol_bits, int ncol, const int4 *__restrict__ items,
    const int *__restrict__ chunk_panel, const int *__restrict__ chunk_item, const unsigned *__restrict__ pk,
    const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y, int xs, int ys, int ordered,
    const int *__restrict__ chunk_ord, int *__restrict__ ticket)
{
  static_assert(kLdsxPer == 2 && kLdsxXPer == 2, "pair loads assume two entries and two slice values per thread");
  __shared__ double ytile[kLdsxRows];
  __shared__ __attribute__((aligned(16))) double xsl[3][kLdsxCols];
  const int t = threadIdx.x;
  const int cp = chunk_panel[blockIdx.x];
  const int p = cp & 0x7fffffff;
  const bool shared = cp < 0;
  const int row0 = panel_row[p];
  const int nr = panel_row[p + 1] - row0;
  for (int i = t; i < nr; i += kTiledBlock) ytile[i] = 0.0;
  const unsigned cmask = (1u << lcol_bits) - 1u;
  const int it0 = chunk_item[2 * blockIdx.x], it1 = chunk_item[2 * blockIdx.x + 1];
  const int itl = it1 > it0 ? it1 - 1 : (it0 > 0 ? it0 - 1 : 0);
  const int itf = it1 > it0 ? it0 : itl;
  auto item = [&](int i) {
    int4 d = items[i < itf ? itf : (i < itl ? i : itl)];
    if (i < it0 || i >= it1) d.y = 0;
    return d;
  };
  int4 dset[NSETS];
  unsigned w[NSETS][2];
  double v[NSETS][2];
  double xr[NSETS][2];
#pragma unroll
  for (int k = 0; k < NSETS; ++k) {
    dset[k] = item(it0 - 1);
    w[k][0] = w[k][1] = 0;
    v[k][0] = v[k][1] = 0.0;
    xr[k][0] = xr[k][1] = 0.0;
  }
  double gcur[2] = {0.0, 0.0};
  const int first = it0 - (NSETS - 1);
  int4 dN = item(first + NSETS - 1);
  int bi = 0;                                     // slice buffer of item IT: (IT - first) % 3
  __syncthreads();
  // phase IT: request item IT+NSETS-1; gather item IT+1 (slice published a phase ago); add item IT (values gathered a
  // phase ago); publish the slice of item IT+2; barrier
  for (int it = first; it < it1; it += NSETS) {
#pragma unroll
    for (int ph = 0; ph < NSETS; ++ph) {
      const int IT = it + ph;
      const int s0 = ph, s1 = (ph + 1) % NSETS, s2 = (ph + 2) % NSETS, sl = (ph + NSETS - 1) % NSETS;
      dset[sl] = dN;
      dN = item(IT + NSETS);
      // the phase's memory instructions are spread between its LDS work (slice first: needed a phase before the
      // entries, and vmcnt retires in order): issued in one burst at the top, the 32 wave-instructions of a phase queue up
      // in front of the address unit and the waves stall at issue while the LDS array idles
      const int b1 = bi == 2 ? 0 : bi + 1, b2 = b1 == 2 ? 0 : b1 + 1;
      double gnew[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) gnew[q] = xsl[b1][w[s1][q] & cmask];
      ldsx_load2_slice<XS1>(dset[sl], t, W, ncol, x, xs, xr[sl]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (2 * t + q < dset[s0].y) {
          double pr = gcur[q];
          if (VALUED) pr *= v[s0][q];
          __hip_atomic_fetch_add(&ytile[w[s0][q] >> lcol_bits], pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      ldsx_load2_entries<VALUED, NT>(dset[sl], t, pk, vals, w[sl], v[sl]);
      {
        const int lc = 2 * t, c = dset[s2].z * W + lc;
        v2d sv;
        sv.x = (lc < W && c < ncol) ? ((c + 1 == ncol && ncol >= 2) ? xr[s2][1] : xr[s2][0]) : 0.0;   // see ldsx_load2
        sv.y = (lc + 1 < W && c + 1 < ncol) ? xr[s2][1] : 0.0;
        *reinterpret_cast<v2d *>(&xsl[b2][lc]) = sv;
      }
      __syncthreads();
      gcur[0] = gnew[0]; gcur[1] = gnew[1];
      bi = b1;
    }
  }
  __syncthreads();
  // (every phase ends in __syncthreads, which waits for the phase's adds: with the builder's row-per-wave items the sums of this
  // kernel are in a fixed order as they are; `ordered` only matters for chunks that share a panel)
  ldsx_store_slice(ytile, nr, row0, y, ys, shared, ordered != 0, ticket + p, shared ? chunk_ord[blockIdx.x] : 0);
}

// y[r * ys] = v[r] (output of a product that went through a contiguous scratch vector)
__global__ __launch_bounds__(kBlock) void strided_copy_kernel(int n, const double *__restrict__ v, double *__restrict__ y, int ys)
{
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < n) y[r * ys] = v[r];
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

// The pipelined sweep with the x slices sent STRAIGHT from global memory into LDS (global_load_lds_dwordx4: 16 bytes per
// lane land at M0 + 16 * lane, so a wave fills 1 KiB of a slice with one instruction): no pass through the registers, no
// ds_write of the slice, four VGPRs fewer per register set.  The default of the LDS-staged copy when x has unit stride, is
// 16-byte aligned and has an even number of columns (a pair load at the end of an odd vector would read past it);
// spmv_ldsx_pipe_kernel otherwise and with tiled_flags bit 2.  Config 3: A 0.745 -> 0.70 ms, A' 0.826 -> 0.775 ms.
//   phase IT:  gather item IT+1 (its slice landed a phase ago) | start the DMA of item IT+3's slice into the buffer item IT
//   used (free since the barrier) | wait for the gathers | add item IT | request the entries of item IT+NSETS-1 | wait until
//   this wave's DMA of item IT+2 has landed | barrier.
// The slice buffers are three separate LDS objects: the compiler tracks an LDS DMA per object and would otherwise make
// every gather wait for the DMA still under way into another buffer.  Even so it makes each LDS instruction it knows about
// wait for every LDS DMA under way (vmcnt(0) in front of every ds_add_f64, and again before the barrier), which would end
// the prefetch of the entries as well -- vmcnt retires in order.  So inside the loop every LDS access is inline assembly,
// the barrier is the bare s_barrier, and the waits are placed by hand:
//   after the gathers and the DMA   lgkmcnt(0): the gathered values are in their registers (an empty asm that takes them
//                        as in/out operands keeps the compiler from giving those registers to anything else before this
//                        point -- the hardware writes them some time after the ds_read was issued) and their buffer may
//                        be refilled once every wave is past the barrier.  The adds go out AFTER this wait and are not
//                        waited for: nothing but the end of the kernel reads the y slice.
//   before the barrier   vmcnt(n): this wave's DMA of item IT+2 has landed; the n operations issued after it stay in
//                        flight (the entries requested behind it a phase ago, this phase's DMA and entries: n = 2 * loads
//                        per entry pair + 1), so the order "DMA, then entries" inside a phase is pinned by scheduling
//                        barriers.  tests/test_isa_guards.py checks n against the compiled code.
#ifndef FS_DMA_SETS
#define FS_DMA_SETS 6
#endif
constexpr int kLdsxDmaSets = FS_DMA_SETS;
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

#ifdef FS_LAB   // tools/build_variants.py only: the instrumented copy of this kernel (ablations, phase clocks) takes its place
#include "experiments/ldsx_dma_lab.inc"
#else
// ORDERED (fixed-order sums): the builder has put all entries of a row inside one work item with ONE wave (TiledCsr::orderable),
// whose LDS adds execute in program order; here the adds of a phase are additionally waited for before the phase's barrier, so
// that adds of different phases -- which may come from different waves -- reach a y slot in phase order.
template <bool VALUED, bool NT, int NSETS, bool ORDERED>
__global__ __launch_bounds__(kTiledBlock) void spmv_ldsx_dma_kernel(
    const int *__restrict__ panel_row, int W, int lcol_bits, int ncol, const int4 *__restrict__ items,
    const int *__restrict__ chunk_panel, const int *__restrict__ chunk_item, const unsigned *__restrict__ pk,
    const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y, int ys,
    const int *__restrict__ chunk_ord, int *__restrict__ ticket)
{
  static_assert(NSETS % 3 == 0, "the slice buffer of a phase is a compile-time constant");
  static_assert(NSETS >= 4, "the sweep starts NSETS - 1 phases early and the DMA of an item is sent three phases before it");
  __shared__ double ytile[kLdsxRows];
  __shared__ __attribute__((aligned(16))) double xs0[kLdsxCols];
  __shared__ __attribute__((aligned(16))) double xs1[kLdsxCols];
  __shared__ __attribute__((aligned(16))) double xs2[kLdsxCols];
  const int t = threadIdx.x;
  const int cp = chunk_panel[blockIdx.x];
  const int p = cp & 0x7fffffff;
  const bool shared = cp < 0;
  const int row0 = panel_row[p];
  const int nr = panel_row[p + 1] - row0;
  for (int i = t; i < nr; i += kTiledBlock) ytile[i] = 0.0;
  const unsigned cmask = (1u << lcol_bits) - 1u;
  const int it0 = chunk_item[2 * blockIdx.x], it1 = chunk_item[2 * blockIdx.x + 1];
  const int itl = it1 > it0 ? it1 - 1 : (it0 > 0 ? it0 - 1 : 0);
  const int itf = it1 > it0 ? it0 : itl;
  auto item = [&](int i) {
    int4 d = items[i < itf ? itf : (i < itl ? i : itl)];
    if (i < it0 || i >= it1) d.y = 0;
    return d;
  };
  int4 dset[NSETS];
  unsigned w[NSETS][2];
  double v[NSETS][2];
#pragma unroll
  for (int k = 0; k < NSETS; ++k) {
    dset[k] = item(it0 - 1);
    w[k][0] = w[k][1] = 0;
    v[k][0] = v[k][1] = 0.0;
  }
  double gcur[2] = {0.0, 0.0};
  const int first = it0 - (NSETS - 1);
  int4 dN = item(first + NSETS - 1);
  int4 dS = item(first + 3);                      // descriptor of the item whose slice this phase sends for
  const int wave_cols = 2 * (t & ~63);            // first column (inside the slice) of this wave's 1 KiB
  const unsigned ybase = lds_addr(ytile);
  const unsigned xbase[3] = {lds_addr(xs0), lds_addr(xs1), lds_addr(xs2)};
  __syncthreads();
  for (int it = first; it < it1; it += NSETS) {
#pragma unroll
    for (int ph = 0; ph < NSETS; ++ph) {
      const int IT = it + ph;
      const int s0 = ph, s1 = (ph + 1) % NSETS, sl = (ph + NSETS - 1) % NSETS;
      dset[sl] = dN;
      dN = item(IT + NSETS);
      // buffers: item j lives in buffer (j - first) % 3; NSETS is a multiple of 3, so these are constants per unrolled phase
      double *const bfree = ph % 3 == 0 ? xs0 : (ph % 3 == 1 ? xs1 : xs2);               // item IT's: refilled for IT+3
      double gnew[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned a = xbase[(ph + 1) % 3] + ((w[s1][q] & cmask) << 3);                // item IT+1: gathered now
        asm volatile("ds_read_b64 %0, %1" : "=v"(gnew[q]) : "v"(a) : "memory");
      }
      {
        const int c0 = dS.z * W + 2 * t;
        const int cc = c0 + 1 < ncol ? c0 : ncol - 2;            // ncol is even and >= 2 here
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(x + cc),
                                         (void __attribute__((address_space(3))) *)(bfree + wave_cols), 16, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      dS = item(IT + 4);
      // the gathers (and the descriptor loads) have returned: their buffer may be refilled once every wave is past the
      // barrier.  The adds go out AFTER this wait and (unless ORDERED) are not waited for: nothing but the end of the kernel
      // reads the y slice, so they drain under the barrier and the next phase instead of holding it up
      __builtin_amdgcn_s_waitcnt(FS_WAIT_IMM(63, 0));
      // the gathered values are written by the hardware some time after the ds_read was issued: tell the compiler they are
      // live up to here, whatever uses them later, so that it can never hand their registers to something else in between
      asm volatile("" : "+v"(gnew[0]), "+v"(gnew[1]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (2 * t + q < dset[s0].y) {
          double pr = gcur[q];
          if (VALUED) pr *= v[s0][q];
          const unsigned a = ybase + ((w[s0][q] >> lcol_bits) << 3);
          asm volatile("ds_add_f64 %0, %1" : : "v"(a), "v"(pr) : "memory");
        }
      }
      // (the entries after the adds: requested right behind the DMA, 0.70 -> 0.74 ms -- memory instructions issued in a
      // burst queue up in front of the address unit)
      ldsx_load2_entries<VALUED, NT>(dset[sl], t, pk, vals, w[sl], v[sl]);
      __builtin_amdgcn_sched_barrier(0);
      // this wave's DMA of item IT+2 has landed (the operations issued behind it stay in flight); ORDERED: and its adds are done
      __builtin_amdgcn_s_waitcnt(FS_WAIT_IMM(VALUED ? 5 : 3, ORDERED ? 0 : 15));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      gcur[0] = gnew[0]; gcur[1] = gnew[1];
    }
  }
  __builtin_amdgcn_s_waitcnt(FS_WAIT_IMM(0, 0));
  __syncthreads();
  ldsx_store_slice(ytile, nr, row0, y, ys, shared, ORDERED, ticket + p, shared ? chunk_ord[blockIdx.x] : 0);
}
#endif   // FS_LAB

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
typedef unsigned v4u __attribute__((ext_vector_type(4)));

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

// pass 2: workgroup = one row panel; its products are contiguous.  ytile: the panel's slice of y in LDS.
template <bool NTLD>
__device__ __forceinline__ void reduce_panel(double *__restrict__ ytile, int panel, const unsigned *__restrict__ bin_ptr,
                                             const int *__restrict__ panel_row, const uint16_t *__restrict__ lrow,
                                             const double *__restrict__ prod, double *__restrict__ y, int ys)
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
      a[k] = stream_load<NTLD>((const v4u *)(lrow + ek));
#pragma unroll
      for (int j = 0; j < 4; ++j) p[k][j] = stream_load<NTLD>((const v2d *)(prod + ek + 2 * j));
    }
    __builtin_amdgcn_sched_barrier(0);
    FS_ADD8(a[0], p[0])
    FS_ADD8(a[1], p[1])
  }
  for (; e < e1; e += 8 * kBinBlock) {
    const v4u a = stream_load<NTLD>((const v4u *)(lrow + e));
    v2d p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = stream_load<NTLD>((const v2d *)(prod + e + 2 * j));
    FS_ADD8(a, p)
  }
#undef FS_ADD8
#undef FS_ADD
  __syncthreads();
  for (int i = t; i < nr; i += kBinBlock) y[(int64_t)(r0 + i) * ys] = ytile[i];
}

template <bool NTLD, int RM = kBinRowsMax>
__global__ __launch_bounds__(kBinBlock) void spmv_reduce_kernel(
    const unsigned *__restrict__ bin_ptr, const int *__restrict__ panel_row, const uint16_t *__restrict__ lrow,
    const double *__restrict__ prod, double *__restrict__ y, int ys, int pbase)
{
  __shared__ double ytile[RM];
  reduce_panel<NTLD>(ytile, pbase + blockIdx.x, bin_ptr, panel_row, lrow, prod, y, ys);
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
template <bool NTLD, int RM, int DEPTH>
__global__ __launch_bounds__(64) void spmv_reduce_ordered_kernel(
    const unsigned *__restrict__ bin_ptr, const int *__restrict__ panel_row, const uint16_t *__restrict__ lrow,
    const double *__restrict__ prod, double *__restrict__ y, int ys, int pbase)
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
    a[k] = stream_load<NTLD>((const v4u *)(lrow + ec));
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
        const v4u ak = a[k];
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
  if (L.bcols == kLongBandB) { if (A.vals) FS_LONG2(true, kLongBandB, kLongRowsB); else FS_LONG2(false, kLongBandB, kLongRowsB); }
  else                       { if (A.vals) FS_LONG2(true, kLongBandA, kLongRowsA); else FS_LONG2(false, kLongBandA, kLongRowsA); }
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

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static int ceil_log2(int v)
{
  int lg = 0;
  while ((1 << lg) < v) ++lg;
  return lg;
}

// y[r] = sum of the virtual rows of row r (rows that were not cut: a copy).  Up to 32 pieces: one thread, in storage
// order.  Longer rows (a power-law matrix has rows of 10^5..10^6 entries = thousands of pieces) are summed by the whole
// wave, 64 pieces per step and a butterfly at the end -- a fixed order, so still reproducible run to run; one thread
// walking 3 900 dependent loads made this pass 0.49 ms of a 2.9 ms product on a config-5 shard
// (profiles/r02_c5_pmc_summary.csv).
__global__ __launch_bounds__(kBlock) void tiled_combine_kernel(int nrow, const int *__restrict__ vfirst,
                                                              const double *__restrict__ yv, double *__restrict__ y, int ys,
                                                              int row0 = 0)
{
  const int64_t r = (int64_t)row0 + (int64_t)blockIdx.x * kBlock + threadIdx.x;   // rows row0 .. nrow of this launch
  const int lane = threadIdx.x & 63;
  int a = 0, b = 0;
  if (r < nrow) { a = vfirst[r]; b = vfirst[r + 1]; }
  const bool long_row = b - a > 32;
  if (r < nrow && !long_row) {
    double acc = yv[a];
    for (int v = a + 1; v < b; ++v) acc += yv[v];
    y[r * ys] = acc;
  }
  unsigned long long todo = __ballot(long_row);
  while (todo) {                                   // wave-uniform
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int ra = __shfl(a, src), rb = __shfl(b, src);
    double acc = 0.0;
    for (int v = ra + lane; v < rb; v += 64) acc += yv[v];
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == src) y[r * ys] = acc;
  }
}

// c0 .. c1: the workgroups of this launch (chunks of the LDS-staged kernel, panels of the L2-tiled one); c1 < 0 = all.
// A part launch needs every workgroup to own its rows (no chunks sharing a panel, no cut rows): spmv_host_vectors
int launch_spmv_tiled(const DeviceCsr &A, const TiledCsr &T, double *y, const double *x, hipStream_t s, int xs, int ys,
                      int c0, int c1)
{
  const bool nt = !(options().tiled_flags & 1);  // bit 0: cached (not nt) entry loads
  double *out = T.split ? T.yv : y;              // cut rows: virtual sums first, combined below
  const int os = T.split ? 1 : ys;
  const bool part = c1 >= 0;
  if (part && T.split) { set_error("launch_spmv_tiled: a copy with cut rows cannot be launched in parts"); return FS_ERR_ARG; }
  if (!part) { c0 = 0; c1 = T.ldsx ? T.nchunks : T.P; }
  if (c1 <= c0) return FS_OK;
  if (T.ldsx) {
    // chunks of one panel add into the same rows: the output then goes through the zeroed scratch vector (a part launch
    // adds into T.yv as it is and leaves the copy to y to its caller, who zeroed T.yv before the first part)
    // fixed-order sums: the chunks of a panel take turns (a ticket per panel, zeroed here) -- see ldsx_store_slice
    const bool ordered = reproducible_now() && T.orderable;
    if (T.shared) {
      if (!part) FS_HIP(hipMemsetAsync(T.yv, 0, sizeof(double) * (size_t)A.nrow, s));
      if (!part && ordered) FS_HIP(hipMemsetAsync(T.ticket, 0, sizeof(int) * (size_t)T.P, s));
      out = T.yv;
    }
    const int ost = T.shared ? 1 : ys;
    if (T.nchunks > 0) {
#define FS_LDSXP(V, N, X1)                                                                                         \
  hipLaunchKernelGGL((spmv_ldsx_pipe_kernel<V, N, X1, kLdsxPipeSets>), dim3(c1 - c0), dim3(kTiledBlock), 0, s,       \
                     T.panel_row, T.W, T.lcol_bits, A.ncol, T.items, T.chunk_panel + c0, T.chunk_item + 2 * c0, T.pk, T.vals, x, \
                     out, xs, ost, (int)ordered, T.chunk_ord + c0, T.ticket)
      // slices by LDS DMA: unit-stride x, 16-byte aligned, an even number of columns (bit 2 of tiled_flags turns it off)
      if (!(options().tiled_flags & 4) && xs == 1 && A.ncol >= 2 && (A.ncol & 1) == 0 &&
          (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
#define FS_LDSXD(V, N, O)                                                                                           \
  hipLaunchKernelGGL((spmv_ldsx_dma_kernel<V, N, kLdsxDmaSets, O>), dim3(c1 - c0), dim3(kTiledBlock), 0, s, T.panel_row, T.W, \
                     T.lcol_bits, A.ncol, T.items, T.chunk_panel + c0, T.chunk_item + 2 * c0, T.pk, T.vals, x, out, ost,    \
                     T.chunk_ord + c0, T.ticket)
#define FS_LDSXD2(V, N) do { if (ordered) FS_LDSXD(V, N, true); else FS_LDSXD(V, N, false); } while (0)
        if (A.vals) { if (nt) FS_LDSXD2(true, true); else FS_LDSXD2(true, false); }
        else        { if (nt) FS_LDSXD2(false, true); else FS_LDSXD2(false, false); }
#undef FS_LDSXD2
#undef FS_LDSXD
      } else if (xs == 1 && A.ncol >= 2) {
        if (A.vals) { if (nt) FS_LDSXP(true, true, true); else FS_LDSXP(true, false, true); }
        else        { if (nt) FS_LDSXP(false, true, true); else FS_LDSXP(false, false, true); }
      } else {                                     // one column of a row-major X: strided slice loads
        if (A.vals) FS_LDSXP(true, true, false); else FS_LDSXP(false, true, false);
      }
#undef FS_LDSXP
      FS_HIP(hipGetLastError());
    }
    if (T.shared && !part) {
      hipLaunchKernelGGL(strided_copy_kernel, dim3((unsigned)(((int64_t)A.nrow + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                         A.nrow, T.yv, y, ys);
      FS_HIP(hipGetLastError());
    }
    return FS_OK;   // rows are never cut for this kernel: no combine pass
  } else {
#define FS_TILED(V, N)                                                                                         \
  hipLaunchKernelGGL((spmv_tiled_kernel<V, N>), dim3(c1 - c0), dim3(kTiledBlock), 0, s, T.panel_row + c0, T.W, T.lcol_bits, \
                     T.items, T.item_ptr + c0, T.pk, T.vals, x, out, xs, os)
    if (A.vals) { if (nt) FS_TILED(true, true); else FS_TILED(true, false); }
    else        { if (nt) FS_TILED(false, true); else FS_TILED(false, false); }
#undef FS_TILED
  }
  FS_HIP(hipGetLastError());
  if (T.split) {
    hipLaunchKernelGGL(tiled_combine_kernel, dim3((unsigned)(((int64_t)A.nrow + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       A.nrow, T.vfirst, T.yv, y, ys);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
}

// p0 .. p1: the panels pass 2 covers in this launch (p1 < 0: all of them); pass 1 runs when p0 == 0.  With cut rows the
// combine pass covers rows row0 .. row1 (the rows whose last piece lies in a panel below p1: see spmv_part_bounds).
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
      if (A.vals)
        hipLaunchKernelGGL((spmv_expand_kernel<true, 4, false, true, kBinColsBig>), dim3(nwg1), dim3(kBinBlock), 0, s, A.ncol, N.B,
                           N.band_ptr, N.lcol, N.vals, N.gdst, x, xs, N.prod, 0u, (unsigned)(N.n >> kBinGroupLog));
      else
        hipLaunchKernelGGL((spmv_expand_kernel<false, 4, false, true, kBinColsBig>), dim3(nwg1), dim3(kBinBlock), 0, s, A.ncol, N.B,
                           N.band_ptr, N.lcol, N.vals, N.gdst, x, xs, N.prod, 0u, (unsigned)(N.n >> kBinGroupLog));
    } else
    if (A.vals) { if ((flags & 3) == 1) FS_EXPAND4(true, 8); else if ((flags & 3) == 2) FS_EXPAND4(true, 2); else FS_EXPAND4(true, 4); }
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
    if (N.bcols == kBinColsBig && ordered)
      hipLaunchKernelGGL((spmv_reduce_ordered_kernel<false, kBinRowsBig, FS_ORDERED_DEPTH>), dim3(p1 - p0), dim3(64), 0, s, N.bin_ptr,
                         N.panel_row, N.lrow, N.prod, out, os, p0);
    else if (N.bcols == kBinColsBig)
      hipLaunchKernelGGL((spmv_reduce_kernel<false, kBinRowsBig>), dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row,
                         N.lrow, N.prod, out, os, p0);
    else if (ordered)
      hipLaunchKernelGGL((spmv_reduce_ordered_kernel<false, kBinRowsMax, FS_ORDERED_DEPTH>), dim3(p1 - p0), dim3(64), 0, s, N.bin_ptr,
                         N.panel_row, N.lrow, N.prod, out, os, p0);
    else if (options().bin_flags & 4)
      hipLaunchKernelGGL(spmv_reduce_kernel<true>, dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod,
                         out, os, p0);
    else
      hipLaunchKernelGGL(spmv_reduce_kernel<false>, dim3(p1 - p0), dim3(kBinBlock), 0, s, N.bin_ptr, N.panel_row, N.lrow, N.prod,
                         out, os, p0);
    FS_HIP(hipGetLastError());
  }
  if (N.split && row1 > row0) {
    hipLaunchKernelGGL(tiled_combine_kernel, dim3((unsigned)(((int64_t)(row1 - row0) + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       row1, N.vfirst, N.yv, y, ys, row0);
    FS_HIP(hipGetLastError());
  }
  if (N.lr && row1 > row0) {   // the long rows of this range: their sums were left in ylong by the launch behind pass 1
    hipLaunchKernelGGL(longrows_scatter_kernel, dim3((unsigned)((N.lr->nlong + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, N.lr->nlong,
                       N.lr->row, N.lr->ylong, y, ys, row0, row1);
    FS_HIP(hipGetLastError());
  }
  return FS_OK;
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
    if (K == 2) { if (A.vals) FS_XP(true, 2); else FS_XP(false, 2); }
    else        { if (A.vals) FS_XP(true, 4); else FS_XP(false, 4); }
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

// y = A'A x in one kernel (fs_ata_mul, option ata_kernel = 2): on the LDS-staged copy when the matrix has one with
// one chunk per panel, else on the plain CSR.  y is zeroed here.
int launch_ata_fused(const DeviceCsr &A, double *y, const double *x, hipStream_t s)
{
  if (A.ncol == 0) return FS_OK;
  FS_HIP(hipMemsetAsync(y, 0, sizeof(double) * (size_t)A.ncol, s));
  if (A.nrow == 0 || A.nnz == 0) return FS_OK;
  const TiledCsr *T = A.tiledx;
  if (T && T->built && !T->shared && T->nchunks > 0) {
#define FS_ATA(V)                                                                                                        \
  hipLaunchKernelGGL((ata_ldsx_kernel<V, true, kLdsxSets>), dim3(T->nchunks), dim3(kTiledBlock), 0, s, T->panel_row, \
                     T->W, T->lcol_bits, A.ncol, T->items, T->chunk_panel, T->chunk_item, T->pk, T->vals, x, y, 1, 1)
    if (A.vals) FS_ATA(true); else FS_ATA(false);
#undef FS_ATA
  } else {
    const unsigned grid = (unsigned)(((int64_t)A.nrow + kBlock / 64 - 1) / (kBlock / 64));
    if (A.vals) hipLaunchKernelGGL(ata_csr_kernel<true>, dim3(grid), dim3(kBlock), 0, s, A.nrow, A.row_ptr, A.cols, A.vals, x, y);
    else        hipLaunchKernelGGL(ata_csr_kernel<false>, dim3(grid), dim3(kBlock), 0, s, A.nrow, A.row_ptr, A.cols, A.vals, x, y);
  }
  FS_HIP(hipGetLastError());
  return FS_OK;
}

// diagnostic: one launch of the tiled kernel that also records, per work item, the start time
// (100 MHz wall clock) and, per panel, the XCD that ran it
int launch_spmv_tiled_trace(const DeviceCsr &A, double *y, const double *x, long long *times_dev, int *xcc_dev,
                            hipStream_t s)
{
  const TiledCsr &T = *A.tiled;
  double *out = T.split ? T.yv : y;
  if (A.vals)
    hipLaunchKernelGGL((spmv_tiled_kernel<true, true, true>), dim3(T.P), dim3(kTiledBlock), 0, s, T.panel_row, T.W,
                       T.lcol_bits, T.items, T.item_ptr, T.pk, T.vals, x, out, 1, 1, times_dev, xcc_dev);
  else
    hipLaunchKernelGGL((spmv_tiled_kernel<false, true, true>), dim3(T.P), dim3(kTiledBlock), 0, s, T.panel_row, T.W,
                       T.lcol_bits, T.items, T.item_ptr, T.pk, T.vals, x, out, 1, 1, times_dev, xcc_dev);
  FS_HIP(hipGetLastError());
  return FS_OK;
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
  const bool repro = o.reproducible != 0 || tl_fixed_order > 0;
  if (A.binned && A.binned->built && !o.strict_order && (o.spmv_kernel == 0 || o.spmv_kernel == 7)) return 7;
  if (A.tiledx && A.tiledx->built && !o.strict_order && (!repro || A.tiledx->orderable) && (o.spmv_kernel == 0 || o.spmv_kernel == 8)) return 8;
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
  // "reproducible" / "strict_order" set AFTER the matrix was created leave its kept copy unusable: the product then runs on
  // the chunk-streaming kernel (correct, but slow on large matrices).  Said once under FS_TRACE_BUILD.
  if ((reproducible_now() || o.strict_order) && o.spmv_kernel == 0 && ((A.binned && A.binned->built) || (A.tiledx && A.tiledx->built))) {
    static const bool trace = getenv("FS_TRACE_BUILD") != nullptr;
    static bool said = false;
    if (trace && !said) {
      said = true;
      fprintf(stderr, "[fastsparse] %d x %d: option %s was set after this matrix was created with a %s copy; its products run on "
              "the chunk-streaming kernel (set the option before creating the matrix to get a fixed-order copy)\n", A.nrow, A.ncol,
              o.strict_order ? "strict_order" : "reproducible", A.binned && A.binned->built ? "two-pass" : "LDS-staged");
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
    const dim3 fg((A.nchunks + kBlock - 1) / kBlock), fb(kBlock);
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
      hipLaunchKernelGGL(strided_copy_kernel, dim3((unsigned)(((int64_t)A.nrow + kBlock - 1) / kBlock)), dim3(kBlock), 0, H.stream,
                         A.nrow, M.yv, H.sy, 1);
      FS_HIP(hipGetLastError());
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
#define FS_XH(V, BC)                                                                                                    \
  hipLaunchKernelGGL((spmv_expand_kernel<V, 4, false, true, BC>), dim3(wgs), dim3(kBinBlock), 0, H.stream, A.ncol, N.B, \
                     N.band_ptr, N.lcol, N.vals, N.gdst, H.sx, 1, N.prod, g0, g1)
    if (N.bcols == kBinColsBig) { if (A.vals) FS_XH(true, kBinColsBig); else FS_XH(false, kBinColsBig); }
    else                        { if (A.vals) FS_XH(true, kBinCols); else FS_XH(false, kBinCols); }
#undef FS_XH
    FS_HIP(hipGetLastError());
  }
  // pass 2, panel range by panel range, an event behind each
  const int c2 = want_chunks < N.P ? want_chunks : N.P;
  for (int j = 0; j < c2; ++j) {
    const int p0 = (int)((int64_t)N.P * j / c2), p1 = (int)((int64_t)N.P * (j + 1) / c2);
    if (N.bcols == kBinColsBig)
      hipLaunchKernelGGL((spmv_reduce_kernel<false, kBinRowsBig>), dim3(p1 - p0), dim3(kBinBlock), 0, H.stream, N.bin_ptr,
                         N.panel_row, N.lrow, N.prod, H.sy, 1, p0);
    else
      hipLaunchKernelGGL(spmv_reduce_kernel<false>, dim3(p1 - p0), dim3(kBinBlock), 0, H.stream, N.bin_ptr, N.panel_row, N.lrow,
                         N.prod, H.sy, 1, p0);
    FS_HIP(hipGetLastError());
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
//           COLUMN-major copies of X and Y (two transposes; the handle's scratch, allocated by prepare or on first use);
//           for k >= 3 prepare times this against the row kernel on the matrix and keeps the faster (neither wins
//           everywhere: config 3's shape 0.77 ms per column against a row kernel bound by its X-row gathers at 51 G/s,
//           12.5 ms for any k from 4 to 16; 10 M rows x 16 with X of 131 K rows, k = 8: 3.5 ms in sweeps, 2.05 ms on
//           the row kernel, X in L2).  Unprepared: the sweeps (the kernel the builder kept this copy for)
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
  const bool repro = reproducible_now();
  const bool free_order = !o.strict_order && (!repro || (A.tiledx && A.tiledx->orderable));   // fixed-order sums on the LDS-staged copy: see spmv_choice
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
