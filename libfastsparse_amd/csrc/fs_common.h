// fs_common.h -- internal declarations shared by the HIP translation units of
// libfastsparse_hip.so.  Not installed; the public surface is include/*.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <vector>

#include "fastsparse_hip.h"

namespace fs {

// ---- kernel geometry (see DESIGN.md "Kernels") ---------------------------------------
constexpr int kBlock = 256;            // threads per workgroup = 4 wave64
constexpr int kChunk = 2048;           // non-zeros one workgroup streams through LDS
constexpr int kPerThread = kChunk / kBlock;

struct TiledCsr;
struct BinnedCsr;

// One CSR in HBM plus the chunk schedule of the streaming SpMV kernel.
struct DeviceCsr {
  int nrow = 0, ncol = 0;
  int64_t nnz = 0;
  int *row_ptr = nullptr;      // nrow + 1
  int *cols = nullptr;         // nnz
  double *vals = nullptr;      // nnz, or nullptr for a pattern-only matrix
  bool owns = true;            // false: arrays borrowed from the caller
  // fs_matrix_release_csr: row_ptr / cols / vals and the chunk schedule were given back (only the kept re-ordered copy is left);
  // everything that reads the plain arrays then fails with FS_ERR_RELEASED until fs_matrix_restore_csr
  bool released = false, released_valued = false;
  bool has_vals() const { return vals != nullptr || (released && released_valued); }
  // schedule: chunk c streams non-zeros [c*kChunk, (c+1)*kChunk) and finishes the rows
  // whose first non-zero lies in that range: rows [first_row[c], first_row[c+1]).
  int nchunks = 0;
  int *first_row = nullptr;    // nchunks + 1
  double *head = nullptr;      // nchunks: sum of the chunk's leading non-zeros that belong to an earlier row
  double *tail = nullptr;      // nchunks: sum of the chunk's trailing non-zeros of a row that ends later
  int spanning = 0;            // number of rows that cross a chunk boundary (0 => no fix-up launch)
  TiledCsr *tiled = nullptr;  // optional L2-tiled copy (owned)
  TiledCsr *tiledx = nullptr; // optional copy in the same layout with the LDS-staged kernel's geometry (owned)
  BinnedCsr *binned = nullptr;  // optional two-pass copy (owned)
  // two-pass copies for k = 2 / k = 4 right-hand sides in ONE sweep (a k-column band of X in LDS, k products per entry):
  // built on the first multi-column product that can use them (launch_spmm); `tried` = the builder declined once
  BinnedCsr *binned2 = nullptr, *binned4 = nullptr;
  bool tried2 = false, tried4 = false;
  // multi-column products of a matrix on the LDS-staged copy: X and Y column-major in here, one unit-stride sweep per column
  double *spmm_scratch = nullptr;
  size_t spmm_scratch_doubles = 0;
  signed char spmm_choice[17] = {};   // per k: 0 not measured (column sweeps run), 1 one sweep per column, 2 the row kernel (prepare_spmm measures)
  // what the format builder measured when it chose (ms per product, median of 5; 0 = candidate not built / not timed):
  // [0] chunk-streaming, [1] L2-tiled, [2] LDS-staged tiled, [3] two-pass
  float candidate_ms[4] = {0.f, 0.f, 0.f, 0.f};
  mutable int max_row_len = -1;      // longest row, once a builder has asked (-1: not yet); the arrays of a handle never change
  bool two_pass_clear_win = false;   // the builder did not build the L2-tiled copy: the two-pass pair already ran faster than that kernel ever has
  // where the one-time work of this matrix went, in ms of host wall time (fs_matrix_build_ms): [0] the arrays into HBM (upload or
  // device copy) + validation, [1] ordering (COO -> CSR, or the transpose), [2] chunk schedule, [3] two-pass copy built, [4] L2-tiled
  // copy built, [5] LDS-staged copy built, [6] the candidates timed, [7] the losers freed + scratch trimmed
  float build_ms[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // products in parts (spmv_part_bounds): the cuts per (number of parts, kernel), computed on first use (a synchronous download
  // of the panel tables: call fs_spmv_part_rows before timing) and kept
  struct PartCuts { int n = 0, kind = 0; bool cut = false; std::vector<int> rows, units; };
  std::vector<PartCuts> part_plans;
  PartCuts partk[2];           // the same for the k-column sweeps: [0] k = 2, [1] k = 4
};

// L2-tiled copy of a CSR for the column-band kernel (see DESIGN.md "spmv_tiled_kernel").
// Rows are cut into panels of R rows (one workgroup each, y slice in LDS), columns into bands of W
// columns (x slice L2-resident).  Entries are ordered by (panel, band), then row, then CSR order.
// An entry is one 32-bit word  local row:(32-lcol_bits) | local col:lcol_bits  plus its value.
// Work items are runs of at most kTiledItem consecutive entries of one (panel, band) tile.
constexpr int kTiledBlock = 1024;      // threads per workgroup of the tiled kernel: ONE workgroup per CU (two
                                       // co-resident workgroups were measured to run at different speeds -- the
                                       // older one wins issue arbitration, 401 vs 450 us per panel -- which pulls
                                       // the band sweep of an XCD apart and out of its L2)
constexpr int kTiledProd = 512;        // waves 0-7 produce (stream + gather), waves 8-15 consume (LDS reduction)
constexpr int kTiledItem = 2048;       // entries per work item (4 per producer thread)
constexpr int kTiledRowsMax = 13056;   // R <= this: 102 KiB of y per workgroup
constexpr int kTiledColBits = 18;      // W <= 262144 columns (2 MiB of x); the other 14 bits are the local row

constexpr int kLdsxRows = 14336;       // LDS-staged kernel: rows per panel (112 KiB of y in LDS, + 3 x 16 KiB slices = 160 KiB)
constexpr int kLdsxCols = 2048;        //                    columns per band: one 16 KiB slice of x, two slices in LDS

struct TiledCsr {
  bool built = false;
  bool ldsx = false;           // geometry of the LDS-staged kernel (W <= kLdsxCols, R <= kLdsxRows)
  int R = 0, W = 0, P = 0, J = 0, lcol_bits = kTiledColBits;
  float entries_per_tile = 0.f;  // nnz / (P * J): how full the (panel, band) tiles are on average
  unsigned *pk = nullptr;      // nnz packed (local row, local col)
  double *vals = nullptr;      // nnz permuted values (nullptr: pattern-only)
  int4 *items = nullptr;       // nitems: {first entry, count, band, 0}
  int *item_ptr = nullptr;     // P + 1
  int nitems = 0;
  int *panel_row = nullptr;    // P + 1: first (virtual) row of every panel
  // rows longer than `split` entries are cut into virtual rows of at most `split` consecutive entries
  int split = 0;               // 0: no row was cut, virtual rows = rows
  int nvrow = 0;               // number of virtual rows
  int *vfirst = nullptr;       // nrow + 1: first virtual row of every row (only when split > 0)
  double *yv = nullptr;        // nvrow: sums of the virtual rows, combined per row after the kernel
  int slots = 256;             // workgroups resident together (1 per CU)
  int *h_panel_row = nullptr;  // host mirrors of panel_row / chunk_panel, fetched by the first product with host vectors
  int *h_chunk_panel = nullptr;
  int *h_chunk_need = nullptr; // nchunks: columns of x the chunks 0 .. w of the launch order read (chunks sharing panels)
  // LDS-staged kernel only: a workgroup takes a CHUNK = a contiguous range of one panel's work items.  Normally a
  // panel is one chunk; a panel that holds far more than its share of the entries (few, long rows; a monster row) is
  // cut into several, whose y slices are then added up in HBM.  Rows are never cut into virtual rows here.
  int nchunks = 0;
  int *chunk_panel = nullptr;  // nchunks: panel of the chunk; bit 31 set when the panel has more than one chunk
  int *chunk_item = nullptr;   // 2 * nchunks: [first, one past the last) work item of every chunk
  bool shared = false;         // some panel has more than one chunk: products go through the zeroed scratch vector yv
  // fixed-order sums on this copy (option "reproducible", the solvers): the builder keeps the entries of one row inside a work
  // item with ONE wave (ldsx_reorder_kernel), whose LDS adds execute in program order; `orderable` = it managed to for every
  // item.  Chunks sharing a panel then add their slices to HBM one after the other, in the order chunk_ord gives (a ticket per
  // panel), instead of in arrival order.
  bool orderable = false;
  int *chunk_ord = nullptr;    // nchunks: ordinal of the chunk inside its panel
  int *ticket = nullptr;       // P: ordinal whose turn it is (zeroed by the launcher)
};

// Two-pass copy of a CSR ("expand, then reduce"; see DESIGN.md "spmv_expand_kernel / spmv_reduce_kernel").
// Columns are cut into bands of kBinCols columns (the x slice of a band lives in LDS), (virtual) rows into panels
// of at most kBinRowsMax rows (the y slice of a panel lives in LDS).  A RUN is the set of entries of one
// (band, panel) pair, kept in CSR storage order and padded to a multiple of kBinGroup entries.
//   pass 1 streams the runs in (band, panel) order: local column ids in, products out -- each group of kBinGroup
//          products goes to the place of its run in (panel, band) order (gdst), i.e. whole 128-byte lines;
//   pass 2 streams the products of one panel, which are now contiguous, with their local row ids, and adds them
//          into the y slice.
// No access of either pass leaves LDS except the two sequential streams.
constexpr int kBinBlock = 1024;        // threads per workgroup, both passes
constexpr int kBinCols = 16384;        // columns per band: 128 KiB of x in LDS, one pass-1 workgroup per CU
constexpr int kBinRowsMax = 16384;     // rows per panel: 128 KiB of y in LDS, one pass-2 workgroup per CU (measured equal to
                                       // 8192 rows x two workgroups; larger panels mean longer runs, less padding)
// short runs (a power-law shard with a very wide x: config 5, 66 entries per run) pay 7.5 padding entries per run: such
// matrices get bands and panels as large as LDS allows, 19 % fewer bands and panels, 29 % fewer runs (config-5 shard 2.84 ->
// 2.72 ms; config 2, 344 entries per run, was measured 1 % slower with them and keeps the power-of-two sizes)
constexpr int kBinColsBig = 19456, kBinRowsBig = 19456;   // 152 KiB of x / of y in LDS
constexpr int kBinBigRunEntries = 192;                     // chosen below this many entries per run (single-vector copies)
static_assert(kBinColsBig % 1024 == 0 && kBinColsBig < 65536, "band loads are 1024 threads wide; 16-bit local ids");
static_assert(kBinCols % 1024 == 0 && kBinCols % 4 == 0 && kBinRowsMax % 4 == 0 && kBinCols < 65536 && kBinRowsMax <= 65536,
              "band loads are 1024 threads wide; 16-bit local ids; k-column copies divide both by 2 and 4");
#ifndef FS_BIN_GROUP_LOG          // (experiment builds only, FS_HIPCC_EXTRA=-DFS_BIN_GROUP_LOG=5: 256-byte groups, profiles/r05_c2_group32_ab.txt)
#define FS_BIN_GROUP_LOG 4
#endif
constexpr int kBinGroupLog = FS_BIN_GROUP_LOG;
constexpr int kBinGroup = 1 << kBinGroupLog;  // entries per group = one 128-byte L2 line of products (runs that start on half
                                               // lines were measured 19 % slower in pass 1: 0.459 vs 0.386 ms)
constexpr int kBinShareMin = 8192;     // a pass-1 workgroup streams at least this many entries

// The longest rows of a heavy-tailed matrix (BASELINE config 5: power-law lengths up to 10^6), taken OUT of the two-pass copy.
// A row with many more entries than there are column bands has several entries per band; the two-pass pair would ship each of
// them through the product stream (16 bytes written and read back per entry).  A few thousand such rows hold 40 % of a config-5
// shard's entries, and their accumulators -- 8 bytes each -- fit into LDS NEXT TO a band of x: spmv_longrows_kernel sweeps the
// bands like pass 1 and adds every product straight into its row's LDS accumulator, no intermediate at all (10 bytes per entry).
// two geometries of the 152 KiB of LDS: a wide band with few accumulators, or a narrower band with four times as many rows
constexpr int kLongBandA = 16384, kLongRowsA = 3072;      // 128 KiB of x + 24 KiB of accumulators
constexpr int kLongBandB = 8192, kLongRowsB = 12032;      //  64 KiB of x + 94 KiB of accumulators (160 KiB with the zero slots)
constexpr int kLongOwners = kBinBlock / 64;               // the waves of a workgroup: every long row belongs to one of them
struct LongRows {
  int nlong = 0;                // rows taken out
  int64_t n = 0;                // their entries, every (band, owner) segment padded to an even count
  int bcols = kLongBandA;       // columns per band (kLongBandA or kLongBandB)
  int B = 0;                    // bands of bcols columns
  // The long rows are numbered so that the rows of owner w (one of the kLongOwners waves of a workgroup, dealt out by length so
  // that the owners carry equal numbers of entries) are the indices [own_first[w], own_first[w + 1]): inside a band the
  // entries are sorted by this index, so every owner's entries are one contiguous SEGMENT of the band.
  int *row = nullptr;           // nlong: row id of long row i
  uint16_t *lcol = nullptr;     // n, (band, long row index) order: column - band * bcols; padding = bcols (the zero slot)
  uint16_t *lrow = nullptr;     // n: index of the long row; padding = the previous entry's
  double *vals = nullptr;       // n (nullptr: pattern-only)
  int64_t *band_ptr = nullptr;  // B + 1: first entry of every band
  unsigned *seg_ptr = nullptr;  // B * (kLongOwners + 1): first entry of every owner's segment, relative to its band
  double *ylong = nullptr;      // nlong: the sums of one product, zeroed before and scattered into y after
  double *ypart = nullptr;      // nwg * nlong: the workgroups' sums, added up in workgroup order (fixed-order products)
  int nwg = 0;                  // persistent workgroups
};

struct BinnedCsr {
  bool built = false;
  LongRows *lr = nullptr;      // the rows that are NOT in this copy (their product: spmv_longrows_kernel)
  int kw = 1;                  // right-hand sides one sweep serves: bands of kBinCols / kw columns (kw * 8 bytes of X per
                               // column in LDS), panels of at most kBinRowsMax / kw rows, groups of kBinGroup / kw entries
                               // (a group is always kBinGroup products = one 128-byte line)
  int bcols = kBinCols;        // columns per band (kBinCols / kw, or kBinColsBig: see there)
  int B = 0, P = 0;            // bands, panels
  int64_t n = 0;               // padded entry count (multiple of kBinGroup)
  uint16_t *lcol = nullptr;    // n, pass-1 order: column - band*kBinCols; padding = kBinCols (a zero slot)
  double *vals = nullptr;      // n, pass-1 order (nullptr: pattern-only); padding = 0
  unsigned *gdst = nullptr;    // n / kBinGroup: pass-2 group index of every pass-1 group
  uint16_t *lrow = nullptr;    // n, pass-2 order: (virtual) row - first row of the panel; padding = 0
  // OR (kw = 1, dense cells: config 2 has 430 entries per (band, panel) cell, in ascending row order) one BYTE per entry: the step
  // from the entry before it, with the row in front of every group of 16 in gbase -- 1.125 bytes per entry instead of 2.  A step
  // above 255 is walked by dummy entries (zero slot of the band, value 0, step 255) in front of the entry, in BOTH orders (a group
  // holds the same entries in pass 1 and pass 2); the builder takes this form when the dummies stay below 1 % of the entries
  // (bin_flags bit 6: never, bit 7: always).  lrow is then nullptr until a fixed-order product asks for it.
  uint8_t *lrow8 = nullptr;    // n, pass-2 order: row - row of the slot before (first slot of a cell: 0); padding = 0
  uint16_t *gbase = nullptr;   // n / kBinGroup: the row in front of the group's first slot (first group of a cell: its first row)
  int64_t dummies = 0;         // entries added to walk steps above 255
  bool lrow_tried = false;     // the first fixed-order product writes the two-byte ids out once (lrow beside lrow8: the one-wave pass 2
                               // reads those; fs_kernels_twopass.hip, ensure_two_byte_ids)
  double *prod = nullptr;      // n * kw, pass-2 order: written by pass 1, read by pass 2
  unsigned *band_ptr = nullptr;  // B + 1: first pass-1 group of every band
  int nwg1 = 0;                // pass-1 workgroups (persistent, one per CU)
  int slots = 0;               // pass-2 workgroups resident together (one per CU: the y slice fills LDS)
  unsigned *bin_ptr = nullptr; // P + 1: first pass-2 group of every panel
  int *panel_row = nullptr;    // P + 1: first (virtual) row of every panel
  int split = 0, nvrow = 0;    // virtual rows, as in TiledCsr
  int *vfirst = nullptr;
  double *yv = nullptr;
  // host mirrors of band_ptr / panel_row / bin_ptr, fetched on the first product with host vectors (fs_spmv_host
  // cuts both passes into ranges of bands / panels so that the PCIe copies of x and y overlap them)
  unsigned *h_band_ptr = nullptr;
  int *h_panel_row = nullptr;
  int *h_vfirst = nullptr;     // host mirror of vfirst (cut rows: the row cuts of products in parts)
};

// staging vectors, stream and events of products with HOST vectors (fs_spmv_host); one set per handle, made on first use
struct HostPipe {
  double *sx = nullptr, *sy = nullptr;
  size_t cx = 0, cy = 0;
  hipStream_t stream = nullptr;
  static constexpr int kMaxChunks = 32;
  hipEvent_t ev[kMaxChunks] = {};
  int nev = 0;
};
void free_host_pipe(HostPipe &H);

}  // namespace fs

struct fs_matrix_s {
  fs::DeviceCsr a;             // A
  fs::DeviceCsr at;            // A' (built on demand)
  bool has_t = false;
  int device = 0;
  std::mutex lock;             // serialises products that share head/tail scratch
  hipStream_t last_stream = nullptr;   // stream of the handle's last asynchronous product (fs_spmv_host orders itself behind it)
  bool last_async = false;
  fs::HostPipe pipe;           // fs_spmv_host / fs_spmv_t_host
};

struct fs_cbcsr_s {
  int nrow = 0, ncol = 0, nblocks = 0, colblocksize = 0;
  int64_t nnz = 0;
  int *row_ptr = nullptr;      // nblocks*nrow + 1, cell = block*nrow + row
  int *cols = nullptr;
  int device = 0;
  // large matrices: the cell array IS a pattern-only CSR with nblocks*nrow rows; `cells` views it (borrowed
  // arrays + chunk schedule) so that the chunk-streaming kernel produces the cell sums, which are then added
  // block by block per row
  fs::DeviceCsr cells;
  double *cell_sums = nullptr;  // nblocks*nrow
  bool use_cells = false;
  // larger still: the same entries as an ordinary pattern-only CSR (rows keep the block-by-block order), so that the
  // product runs on whatever kernel the format builder measures fastest (the LDS-staged kernel for dense tiles)
  fs::DeviceCsr rows;
  bool use_rows = false;
  std::mutex lock;             // serialises products (cell sums and the chunk scratch are per handle)
};

namespace fs {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define FS_HIP(call)                                                   \
  do {                                                                 \
    hipError_t e_ = (call);                                            \
    if (e_ != hipSuccess) return ::fs::hip_fail(e_, #call, __FILE__, __LINE__); \
  } while (0)

// ---- roctx ranges around the C-ABI entry points ("roctx ranges around each C-ABI entry") -----------------
// librocprofiler-sdk-roctx.so.1 is dlopen'ed on first use when FS_ROCTX=1 or the process runs under rocprofv3 (its tool
// library is preloaded); otherwise a range costs one predictable branch.  `rocprofv3 --marker-trace -- <program>` shows them.
bool roctx_enabled();
void roctx_push(const char *name);
void roctx_pop();
struct Range {
  bool on;
  explicit Range(const char *name) : on(roctx_enabled()) { if (on) roctx_push(name); }
  ~Range() { if (on) roctx_pop(); }
  Range(const Range &) = delete;
  Range &operator=(const Range &) = delete;
};
#define FS_RANGE(name) ::fs::Range fs_range_(name)

struct Options {
  int strict_order = 0;
  int spmv_kernel = 0;   // 0 auto, 1 stream (nt loads), 2 lanes-per-row, 3 stream (cached loads), 6 tiled, 7 two-pass,
                         // 8 LDS-staged tiled
  int tiling = 1;        // 1: build the L2-tiled copy when the heuristic says it pays, 2: always, 0: never
  int tile_rows = 0;     // override R (0 = auto)
  int tile_cols = 0;     // override W (0 = auto)
  int tile_split = 0;    // rows longer than this are cut into virtual rows (0 = 256)
  int tiled_flags = 0;   // tuning switches of the tiled kernels (launch_spmv_tiled): bit 0 cached entry loads, bit 2 no LDS DMA
                         // for the x slices (spmv_ldsx_pipe_kernel); FS_TILED_FLAGS presets it
  int reproducible = 0;  // 1: only kernels whose sums are bit-identical run to run (the two-pass kernels add with LDS
                         // atomics in arrival order); read when a matrix is created and at every product
  int bin_wgs = 0;       // override the number of persistent pass-1 workgroups (0 = one per CU)
  int bin_flags = 0;     // tuning switches of the two-pass kernels (see launch_spmv_binned)
  int bin_rows = 0;      // override the rows per panel of the two-pass copy (0 = kBinRowsMax)
  int ldsx = 1;          // the copy for the LDS-staged kernel: 1 when the estimates do not rule it out, 2 always, 0 never
  int binning = 1;       // 1: build the two-pass copy when the heuristic says it pays, 2: always, 0: never
  int long_rows = 1;     // the longest rows of a heavy-tailed matrix outside the two-pass copy (LongRows): 1 when it pays, 2 always
                         // (every row of at least long_min_len entries, at most kLongRowsMax of them), 0 never
  int long_min_len = 0;  // override of the length from which a row counts as long (0 = auto)
  int long_geometry = 0; // 0 auto, 1 wide band / 3072 rows, 2 narrow band / 12288 rows (LongRows)
  int device_build = -1; // format constructors (new_csr, new_bcsr, new_cbcsr, new_bsbm, new_bsdm): -1 = FS_DEVICE_BUILD or 1;
                         // 0 host loops, 1 on the device from 4 M entries, 2 on the device whenever one is visible
  int ata_kernel = 0;    // fs_ata_mul: 0 / 1 two products (A, then the cached A'), 2 the fused single kernel (no copy of A')
  int spmm_wide = 0;     // row SpMM kernel with two columns per lane and 16-byte loads: 0 auto (even k from 4 to 14, 16-byte aligned X / Y), 1 wherever legal, -1 never
  int spmm_kernel = 0;   // multi-column products: 0 auto, 1 row kernel, 2 k-column two-pass sweep (k = 2..4), 3 one
                         // single-vector sweep per column, 4 the MFMA row kernel (experiment, see spmm_mfma_kernel)
  int cg_fixed_order = 1;  // fs_cg / fs_cg2 / fs_dist_cg run their products with fixed-order sums (as under "reproducible"), so
                           // that a solve is bit-identical from run to run like the reference's (cg.h:25-187); 0: the default kernels
  int release_csr = 0;     // 1: fs_csr_create / fs_coo_create / fs_matrix_build_transpose give the plain CSR arrays back once a re-ordered
                           // copy was kept (fs_matrix_release_csr); FS_RELEASE_CSR presets it; the drop-in layer never releases
  int dist_cg_scheme = 0;  // fs_dist_cg: 0 every device keeps whole vectors (no exchange for the dots), 1 every device keeps its
                           // slice of the unknowns (vector work divided by the devices; see fs_dist.hip)
};
Options &options();

// Fixed-order sums are wanted NOW, on this thread: the process-wide option, or the calling thread is inside a solver that asks
// for them (FixedOrderScope).  Read by the launchers at every product.
extern thread_local int tl_fixed_order;
extern thread_local int tl_keep_csr;      // > 0: creations on this thread ignore option release_csr (the drop-in layer's scope)
struct KeepCsrScope {
  KeepCsrScope() { ++tl_keep_csr; }
  ~KeepCsrScope() { --tl_keep_csr; }
  KeepCsrScope(const KeepCsrScope &) = delete;
  KeepCsrScope &operator=(const KeepCsrScope &) = delete;
};
// every fs_set_option moves this on: plans that depend on which kernel a product runs are checked again only when it moved
unsigned option_epoch();
inline bool reproducible_now() { return options().reproducible != 0 || tl_fixed_order > 0; }
struct FixedOrderScope {
  bool on;
  explicit FixedOrderScope(bool want) : on(want) { if (on) ++tl_fixed_order; }
  ~FixedOrderScope() { if (on) --tl_fixed_order; }
  FixedOrderScope(const FixedOrderScope &) = delete;
  FixedOrderScope &operator=(const FixedOrderScope &) = delete;
};

// ---- launchers implemented in fs_kernels.hip, fs_kernels_tiled.hip, fs_kernels_twopass.hip ------------
int launch_spmv(const DeviceCsr &A, double *y, const double *x, hipStream_t s, bool force_stream = false);
int debug_dma_trace(unsigned long long *out8, int reset);   // defined in -DFS_LAB -DFS_DMA_TRACE builds only (experiments/ldsx_dma_lab.inc)
int launch_spmm(DeviceCsr &A, double *Y, const double *X, int k, hipStream_t s);   // never builds, never waits: see prepare_spmm
int prepare_spmm(DeviceCsr &A, int k, hipStream_t s);   // k-column copy, scratch, measured choice: synchronous, idempotent
int spmm_plan(const DeviceCsr &A, int k, int *needs_prepare);   // which kernel launch_spmm runs for this k (kPlan* in fs_kernels.hip)
int launch_cbcsr(const fs_cbcsr_s &A, double *y, const double *x, hipStream_t s);
int spmv_choice(const DeviceCsr &A, const Options &o);   // 7 two-pass, 8 LDS-staged, 6 L2-tiled, 2 lanes per row, 1 chunk-streaming
int spmv_part_bounds(DeviceCsr &A, int nparts, const int **rows_out, const int **units_out, int *kind_out = nullptr, bool *cut_out = nullptr);
int launch_spmv_part(DeviceCsr &A, double *y, const double *x, int part, int nparts, hipStream_t s);
int spmm_part_bounds(DeviceCsr &A, int k, int nparts, const int **rows_out, const int **units_out, int *plan_out);
int launch_spmm_part(DeviceCsr &A, double *Y, const double *X, int k, int part, int nparts, hipStream_t s);
int launch_tiled_combine(int row_end, const int *vfirst, const double *yv, double *y, int ys, int row0, hipStream_t s);   // rows row0 .. row_end
int launch_strided_copy(int n, const double *v, double *y, int ys, hipStream_t s);
int launch_expand_groups(const DeviceCsr &A, const double *x, unsigned g0, unsigned g1, int wgs, hipStream_t s);   // pass 1, groups g0 .. g1
int launch_reduce_panels(const DeviceCsr &A, double *y, int p0, int p1, hipStream_t s);                             // pass 2, panels p0 .. p1
int launch_copy_segments(int nseg, const int64_t *tab_dev, int64_t max_count, const double *src, double *dst, hipStream_t s);
int launch_ata_fused(const DeviceCsr &A, double *y, const double *x, hipStream_t s);   // y[ncol] = A'A x, one kernel
// y_host = A x_host: copies and kernels overlapped where the kept copy allows it (two-pass copy without cut rows)
int spmv_host_vectors(const DeviceCsr &A, HostPipe &H, double *y_host, const double *x_host);
int last_host_path();

// ---- the vector steps of CG (fs_cg.hip) for callers with their own products; every step leaves its dot / norm in red[0]
constexpr int kCgPartDoubles = 3 * 1024;
constexpr int kCgStateDoubles = 16, kCgStateDone = 0, kCgStateIter = 1;   // device state of a solve: st[done], st[iterations], scalars
int cg_dev_init(int n, const double *b, double *x, double *r, double *p, double *part, double *red, double *st, double tol,
                hipStream_t s);                                                // x = 0, r = p = b; b.b, the stopping threshold
int cg_dev_steps(int n, double lambda, double *x, double *r, double *p, double *q, double *part, double *red, double *st,
                 hipStream_t s);                                               // everything of an iteration behind q = A'(A p)
// two right-hand sides, row-major n x 2 (bsbm_cg2, cg.h:85-187): init is synchronous and returns the column norms of B for finish
int cg2_dev_init(int n, const double *B, double *X, double *R, double *P, double *part, double *red, double *st, double tol,
                 double *norms, hipStream_t s);
int cg2_dev_steps(int n, double lambda, double *X, double *R, double *P, double *Q, double *part, double *red, double *st, hipStream_t s);
int cg2_dev_finish(int n, const double *norms, double *X, hipStream_t s);
// the same steps for a SLICE of the unknowns (fs_dist_cg, scheme "gather"): every step leaves this rank's partial dot in
// *red_out; the partials of all ranks, gathered, go through cg_dev_final, which adds them in rank order and does the scalar step
// `mode` of final_step_kernel (0 b.b and the threshold, 1 alpha, 2 convergence and beta)
int cg_dev_init_partial(int n, const double *b, double *x, double *r, double *p, double *part, double *red_out, hipStream_t s);
int cg_dev_step_a(int n, double lambda, const double *p, double *q, double *part, double *red_out, const double *st, hipStream_t s);
int cg_dev_step_b(int n, double *x, double *r, const double *p, const double *q, double *part, double *red_out, const double *st,
                  hipStream_t s);
int cg_dev_step_c(int n, double *p, const double *r, const double *st, hipStream_t s);
int cg_dev_final(int mode, const double *partials, int count, double *red_out, double *st, double arg, hipStream_t s);
// the host's view of a running solve: {done, iterations} of the two most recent iterations, in pinned memory
struct CgFlags {
  double *h = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int init();
  ~CgFlags();
  // behind iteration `iter`: its flags on their way to the host; then a look at the flags of iteration iter - 1 (they arrive
  // while this iteration runs).  *stop: that iteration had converged -- nothing further needs to be enqueued
  int after_iteration(int iter, const double *st, hipStream_t s, bool *stop);
};

// ---- format work implemented in fs_format.hip --------------------------------------------
int build_schedule(DeviceCsr &A, hipStream_t s, bool allow_tiled = true);
int build_tiled(DeviceCsr &A, hipStream_t s);       // no-op unless options/heuristic ask for it
int build_tiledx(DeviceCsr &A, hipStream_t s);      // the same for the LDS-staged kernel's geometry
int launch_spmv_tiled(const DeviceCsr &A, const TiledCsr &T, double *y, const double *x, hipStream_t s, int xs = 1,
                      int ys = 1, int c0 = 0, int c1 = -1);                  // T.ldsx selects the LDS-staged kernel
int build_binned(DeviceCsr &A, hipStream_t s);      // no-op unless options/heuristic ask for it
int build_binned_k(DeviceCsr &A, int kw, hipStream_t s);   // the k-column copy (kw = 2 or 4) into A.binned2 / A.binned4
int launch_spmm_binned(const DeviceCsr &A, const BinnedCsr &N, double *Y, const double *X, hipStream_t s, int xs, int ys, int p0 = 0,
                       int p1 = -1, int row0 = 0, int row1 = 0);
int choose_copy(DeviceCsr &A, hipStream_t s);       // times the candidates and keeps the fastest copy
int launch_spmv_binned(const DeviceCsr &A, double *y, const double *x, hipStream_t s, int xs = 1, int ys = 1, int p0 = 0,
                       int p1 = -1, int row0 = 0, int row1 = 0);   // p1 >= 0: pass 2 for panels p0 .. p1 only (pass 1 with p0 == 0)
int launch_spmv_tiled_trace(const DeviceCsr &A, double *y, const double *x, long long *times_dev, int *xcc_dev,
                            hipStream_t s);
int coo_to_csr_device(DeviceCsr &out, int nrow, int ncol, int64_t nnz, const int *rows_dev,
                      const int *cols_dev, const double *vals_dev, hipStream_t s);
int transpose_device(const DeviceCsr &A, DeviceCsr &At, hipStream_t s);
// row shards of A' from row shards of A on the devices (fs_dist_matrix_build_transpose_device): see fs_format.hip
int shard_column_counts(const DeviceCsr &A, int *counts_dev, hipStream_t s);     // counts[c] += entries of the shard in column c
int add_counts(int64_t n, int *acc_dev, const int *add_dev, hipStream_t s);
int cut_by_counts(int n_items, const int *counts_dev, int nparts, int *bounds_host, int64_t *total, hipStream_t s);
int shard_transpose_partition(const DeviceCsr &A, int row_lo, int nparts, const int *bounds_host, int **trow, int **tcol,
                              double **tval, int64_t *count_host, hipStream_t s);
int cbcsr_rows_device(DeviceCsr &out, int nrow, int ncol, int nblocks, int64_t nnz, const int *cell_ptr_dev,
                      const int *cols_dev, hipStream_t s);
int validate_indices(int nrow, int ncol, int64_t nnz, const int *row_ptr_dev, const int *rows_dev, const int *cols_dev,
                     hipStream_t s);   // FS_ERR_ARG (with a message) when an index is out of range
void pool_trim(bool everything = false);   // frees the format builders' idle scratch beyond FS_SCRATCH_POOL_MB (or all of it)
void free_csr(DeviceCsr &A);
int release_plain_csr(DeviceCsr &A);          // 1 released, 0 nothing to release (no kept copy / already released)
int release_prepared(DeviceCsr &A, int k);    // the k-column copy / scratch of k (k = 0: all of them)
int need_plain_csr(const DeviceCsr &A, const char *who);   // FS_OK, or FS_ERR_RELEASED with a message
void device_bytes(const DeviceCsr &A, int64_t out[3]);   // HBM held: CSR + schedule, kept single-vector copy, k-column copies + scratch

}  // namespace fs
