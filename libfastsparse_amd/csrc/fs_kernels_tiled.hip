// fs_kernels_tiled.hip -- the kernels that keep a row panel's slice of y in LDS and sweep column bands: the L2-tiled kernel (x
// gathered from an L2-resident band) and the LDS-staged kernels (the band's slice of x in LDS as well: by LDS DMA, or through
// the registers), the fused A'A x on the LDS-staged copy, and their launchers.  Split from fs_kernels.hip in round 4.
#include <stdlib.h>

#include "fs_kernel_util.h"

namespace fs {

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
// ordinal order and workgroups are dispatched in index order (an ASSUMPTION about the hardware's dispatcher), so
// the chunk waited for is running or done; the wait is bounded all the same: a chunk that gives up adds out of turn -- a wrong
// ORDER, never a hang -- and says so in *giveups, which fs_debug_ldsx_ticket_giveups reads and the fixed-order tests assert is 0.
__device__ __forceinline__ void ldsx_store_slice(const double *__restrict__ ytile, int nr, int row0, double *__restrict__ y, int ys, bool shared,
                                                 bool ordered, int *__restrict__ ticket, int ord, int *__restrict__ giveups)
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
      if (spins >= (1 << 24)) atomicAdd(giveups, 1);
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
  ldsx_store_slice(ytile, nr, row0, y, ys, shared, ordered != 0, ticket + p, shared ? chunk_ord[blockIdx.x] : 0, ticket - 1);
}

// y[r * ys] = v[r] (output of a product that went through a contiguous scratch vector)
__global__ __launch_bounds__(kBlock) void strided_copy_kernel(int n, const double *__restrict__ v, double *__restrict__ y, int ys)
{
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < n) y[r * ys] = v[r];
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
  ldsx_store_slice(ytile, nr, row0, y, ys, shared, ORDERED, ticket + p, shared ? chunk_ord[blockIdx.x] : 0, ticket - 1);
}
#endif   // FS_LAB

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

int launch_tiled_combine(int row_end, const int *vfirst, const double *yv, double *y, int ys, int row0, hipStream_t s)
{
  if (row_end <= row0) return FS_OK;
  hipLaunchKernelGGL(tiled_combine_kernel, dim3((unsigned)(((int64_t)(row_end - row0) + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, row_end, vfirst,
                     yv, y, ys, row0);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int launch_strided_copy(int n, const double *v, double *y, int ys, hipStream_t s)
{
  if (n <= 0) return FS_OK;
  hipLaunchKernelGGL(strided_copy_kernel, dim3((unsigned)(((int64_t)n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, n, v, y, ys);
  FS_HIP(hipGetLastError());
  return FS_OK;
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
        if (A.has_vals()) { if (nt) FS_LDSXD2(true, true); else FS_LDSXD2(true, false); }
        else        { if (nt) FS_LDSXD2(false, true); else FS_LDSXD2(false, false); }
#undef FS_LDSXD2
#undef FS_LDSXD
      } else if (xs == 1 && A.ncol >= 2) {
        if (A.has_vals()) { if (nt) FS_LDSXP(true, true, true); else FS_LDSXP(true, false, true); }
        else        { if (nt) FS_LDSXP(false, true, true); else FS_LDSXP(false, false, true); }
      } else {                                     // one column of a row-major X: strided slice loads
        if (A.has_vals()) FS_LDSXP(true, true, false); else FS_LDSXP(false, true, false);
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
    if (A.has_vals()) { if (nt) FS_TILED(true, true); else FS_TILED(true, false); }
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
    if (A.has_vals()) FS_ATA(true); else FS_ATA(false);
#undef FS_ATA
  } else {
    if (int rc = need_plain_csr(A, "the fused A'A kernel on the plain CSR")) return rc;
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

}  // namespace fs
