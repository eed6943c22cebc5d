// fs_format.hip -- one-time format work on the device: chunk schedule of the streaming
// SpMV kernel, stable COO -> CSR, CSR -> CSR of the transpose, synthetic generators.
//
// The reference builds its CSRs on the host with a stable counting sort (new_csr csr.h:375-422,
// new_bcsr csr.h:30-67).  Here the same result (entries of a row kept in input order) comes
// from a stable LSD radix sort of (row key, entry index) pairs -- rocPRIM's device radix sort is
// used for this one-time step; the products themselves run on the hand-written kernels of
// fs_kernels.hip.
#include <cstring>
#include <string.h>

#include <vector>

#include <rocprim/rocprim.hpp>

#include "fs_common.h"

namespace fs {

// ---- small helpers -------------------------------------------------------------------------
// device scratch that is released on every exit path
template <typename T>
struct Scratch {
  T *p = nullptr;
  Scratch() = default;
  Scratch(const Scratch &) = delete;
  Scratch &operator=(const Scratch &) = delete;
  ~Scratch() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc(&p, sizeof(T) * (n ? n : 1)); }
  operator T *() const { return p; }
};

static void free_tiled(DeviceCsr &A)
{
  if (!A.tiled) return;
  if (A.tiled->pk) (void)hipFree(A.tiled->pk);
  if (A.tiled->vals) (void)hipFree(A.tiled->vals);
  if (A.tiled->items) (void)hipFree(A.tiled->items);
  if (A.tiled->item_ptr) (void)hipFree(A.tiled->item_ptr);
  delete A.tiled;
  A.tiled = nullptr;
}

void free_csr(DeviceCsr &A)
{
  if (A.owns) {
    if (A.row_ptr) (void)hipFree(A.row_ptr);
    if (A.cols) (void)hipFree(A.cols);
    if (A.vals) (void)hipFree(A.vals);
  }
  if (A.first_row) (void)hipFree(A.first_row);
  if (A.head) (void)hipFree(A.head);
  if (A.tail) (void)hipFree(A.tail);
  free_tiled(A);
  A = DeviceCsr();
}

// first index r in [0, n] with a[r] >= key (a non-decreasing)
__device__ __forceinline__ int lower_bound_dev(const int *__restrict__ a, int n, int64_t key)
{
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if ((int64_t)a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// first_row[c] = first row whose first non-zero index is >= c*kChunk; first_row[nchunks] = nrow
__global__ void schedule_kernel(int nrow, int nchunks, const int *__restrict__ row_ptr, int *__restrict__ first_row)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nchunks) return;
  first_row[c] = (c == nchunks) ? nrow : lower_bound_dev(row_ptr, nrow, (int64_t)c * kChunk);
}

__global__ void count_spanning_kernel(int nchunks, int64_t nnz, const int *__restrict__ row_ptr,
                                      const int *__restrict__ first_row, int *__restrict__ count)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks) return;
  const int r0 = first_row[c], r1 = first_row[c + 1];
  if (r1 <= r0) return;
  int64_t e = (int64_t)(c + 1) * kChunk;
  if (e > nnz) e = nnz;
  if ((int64_t)row_ptr[r1] > e) atomicAdd(count, 1);
}

int build_schedule(DeviceCsr &A, hipStream_t s)
{
  A.nchunks = (int)((A.nnz + kChunk - 1) / kChunk);
  if (A.nchunks < 1) A.nchunks = 1;
  FS_HIP(hipMalloc(&A.first_row, sizeof(int) * ((size_t)A.nchunks + 1)));
  FS_HIP(hipMalloc(&A.head, sizeof(double) * (size_t)A.nchunks));
  FS_HIP(hipMalloc(&A.tail, sizeof(double) * (size_t)A.nchunks));
  FS_HIP(hipMemsetAsync(A.head, 0, sizeof(double) * (size_t)A.nchunks, s));
  FS_HIP(hipMemsetAsync(A.tail, 0, sizeof(double) * (size_t)A.nchunks, s));
  const int n = A.nchunks + 1;
  hipLaunchKernelGGL(schedule_kernel, dim3((n + 255) / 256), dim3(256), 0, s, A.nrow, A.nchunks, A.row_ptr,
                     A.first_row);
  FS_HIP(hipGetLastError());
  Scratch<int> cnt;
  FS_HIP(cnt.alloc(1));
  FS_HIP(hipMemsetAsync(cnt, 0, sizeof(int), s));
  hipLaunchKernelGGL(count_spanning_kernel, dim3((A.nchunks + 255) / 256), dim3(256), 0, s, A.nchunks, A.nnz,
                     A.row_ptr, A.first_row, cnt);
  FS_HIP(hipGetLastError());
  FS_HIP(hipMemcpyAsync(&A.spanning, cnt, sizeof(int), hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  return build_tiled(A, s);
}

// ---- stable COO -> CSR -----------------------------------------------------------------------
__global__ void iota_kernel(int64_t n, unsigned *__restrict__ idx)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (unsigned)i;
}

__global__ void permute_kernel(int64_t n, const unsigned *__restrict__ perm, const int *__restrict__ cols_in,
                               const double *__restrict__ vals_in, int *__restrict__ cols_out,
                               double *__restrict__ vals_out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned src = perm[i];
  cols_out[i] = cols_in[src];
  if (vals_in) vals_out[i] = vals_in[src];
}

// row_ptr[r] = first position in the sorted key array whose key is >= r
__global__ void row_ptr_kernel(int nrow, int64_t nnz, const int *__restrict__ sorted_rows, int *__restrict__ row_ptr)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nrow) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (sorted_rows[mid] < r) lo = mid + 1; else hi = mid;
  }
  row_ptr[r] = (int)lo;
}

static unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

int coo_to_csr_device(DeviceCsr &out, int nrow, int ncol, int64_t nnz, const int *rows_dev, const int *cols_dev,
                      const double *vals_dev, hipStream_t s)
{
  out = DeviceCsr();
  out.nrow = nrow; out.ncol = ncol; out.nnz = nnz; out.owns = true;
  const size_t n = (size_t)(nnz > 0 ? nnz : 1);
  FS_HIP(hipMalloc(&out.row_ptr, sizeof(int) * ((size_t)nrow + 1)));
  FS_HIP(hipMalloc(&out.cols, sizeof(int) * n));
  if (vals_dev) FS_HIP(hipMalloc(&out.vals, sizeof(double) * n));
  Scratch<int> keys_out;
  Scratch<unsigned> idx_in, idx_out;
  Scratch<char> tmp;
  size_t tmp_bytes = 0;
  FS_HIP(keys_out.alloc(n));
  FS_HIP(idx_in.alloc(n));
  FS_HIP(idx_out.alloc(n));
  if (nnz > 0) {
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(nnz)), dim3(256), 0, s, nnz, idx_in);
    FS_HIP(hipGetLastError());
    int bits = 1;
    while (bits < 31 && (1ll << bits) < (long long)nrow) ++bits;
    FS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, rows_dev, keys_out.p, idx_in.p, idx_out.p, (size_t)nnz, 0, bits, s));
    FS_HIP(tmp.alloc(tmp_bytes));
    FS_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmp_bytes, rows_dev, keys_out.p, idx_in.p, idx_out.p, (size_t)nnz, 0, bits, s));
    hipLaunchKernelGGL(permute_kernel, dim3(grid_for(nnz)), dim3(256), 0, s, nnz, idx_out.p, cols_dev, vals_dev,
                       out.cols, out.vals);
    FS_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(row_ptr_kernel, dim3(grid_for((int64_t)nrow + 1)), dim3(256), 0, s, nrow, nnz, keys_out.p,
                     out.row_ptr);
  FS_HIP(hipGetLastError());
  FS_HIP(hipStreamSynchronize(s));
  return build_schedule(out, s);
}

// ---- CSR -> CSR of the transpose -------------------------------------------------------------------
// row id of every stored entry (one thread per entry, binary search in row_ptr)
__global__ void expand_rows_kernel(int nrow, int64_t nnz, const int *__restrict__ row_ptr, int *__restrict__ rows)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  int lo = 0, hi = nrow;  // last r with row_ptr[r] <= i
  while (lo < hi) {
    const int mid = lo + ((hi - lo + 1) >> 1);
    if ((int64_t)row_ptr[mid] <= i) lo = mid; else hi = mid - 1;
  }
  rows[i] = lo;
}

int transpose_device(const DeviceCsr &A, DeviceCsr &At, hipStream_t s)
{
  Scratch<int> rows;
  const size_t n = (size_t)(A.nnz > 0 ? A.nnz : 1);
  FS_HIP(rows.alloc(n));
  if (A.nnz > 0) {
    hipLaunchKernelGGL(expand_rows_kernel, dim3(grid_for(A.nnz)), dim3(256), 0, s, A.nrow, A.nnz, A.row_ptr, rows.p);
    FS_HIP(hipGetLastError());
  }
  // A' in COO is (cols, rows, vals); stable sort by column keeps the row order inside each column,
  // i.e. the order in which the serial loops of At_mul_B (sparse.h:72-74) visit a column's entries
  // when the COO itself is row ordered.
  return coo_to_csr_device(At, A.ncol, A.nrow, A.nnz, A.cols, rows.p, A.vals, s);
}

// ---- L2-tiled copy ---------------------------------------------------------------------------------
// key of entry e = panel(row) * J + band(col); a stable sort by key starting from CSR order leaves every
// (panel, band) tile ordered by row and, inside a row, in CSR storage order.
__global__ void tile_key_kernel(int nrow, int64_t nnz, int R, int W, int J, const int *__restrict__ row_ptr,
                                const int *__restrict__ cols, int *__restrict__ rows, unsigned *__restrict__ keys)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  int lo = 0, hi = nrow;  // last r with row_ptr[r] <= i
  while (lo < hi) {
    const int mid = lo + ((hi - lo + 1) >> 1);
    if ((int64_t)row_ptr[mid] <= i) lo = mid; else hi = mid - 1;
  }
  rows[i] = lo;
  keys[i] = (unsigned)(lo / R) * (unsigned)J + (unsigned)(cols[i] / W);
}

__global__ void tile_pack_kernel(int64_t nnz, int R, int W, int J, int lcol_bits, const unsigned *__restrict__ skeys,
                                 const unsigned *__restrict__ perm, const int *__restrict__ rows,
                                 const int *__restrict__ cols, const double *__restrict__ vals,
                                 unsigned *__restrict__ pk, double *__restrict__ vals_out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const unsigned src = perm[i];
  const unsigned key = skeys[i];
  const int r = rows[src], c = cols[src];
  const unsigned p = key / (unsigned)J, j = key % (unsigned)J;
  const unsigned lrow = (unsigned)(r - (int)p * R), lcol = (unsigned)(c - (int)j * W);
  pk[i] = (lrow << lcol_bits) | lcol;
  if (vals) vals_out[i] = vals[src];
}

// tile_ptr[k] = first sorted position whose key is >= k, k = 0 .. ntiles
__global__ void tile_ptr_kernel(int64_t ntiles, int64_t nnz, const unsigned *__restrict__ skeys, int *__restrict__ tile_ptr)
{
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > ntiles) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if ((int64_t)skeys[mid] < k) lo = mid + 1; else hi = mid;
  }
  tile_ptr[k] = (int)lo;
}

__global__ void max_row_len_kernel(int nrow, const int *__restrict__ row_ptr, int *__restrict__ out)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  int len = r < nrow ? row_ptr[r + 1] - row_ptr[r] : 0;
  for (int m = 32; m > 0; m >>= 1) { const int o = __shfl_xor(len, m); len = o > len ? o : len; }
  if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(out, len);
}

static int build_tiled_impl(DeviceCsr &A, hipStream_t s);

// The tiled copy is an optimisation: if building it fails (typically: not enough HBM for the second copy) the
// matrix stays usable on the chunk-streaming kernel.
int build_tiled(DeviceCsr &A, hipStream_t s)
{
  const int rc = build_tiled_impl(A, s);
  if (rc != FS_OK || (A.tiled && !A.tiled->built)) {
    free_tiled(A);
    (void)hipGetLastError();
  }
  return FS_OK;
}

static int build_tiled_impl(DeviceCsr &A, hipStream_t s)
{
  const Options &o = options();
  if (o.tiling == 0 || A.nrow == 0 || A.nnz == 0) return FS_OK;
  int dev = 0, ncu = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
  const int slots = (ncu > 8 ? ncu : 256) / 8 * 8;  // one workgroup per CU, a multiple of the 8 XCDs
  // panel height: P a multiple of the resident workgroup count where the matrix is tall enough
  int R = o.tile_rows;
  if (R <= 0) {
    const int64_t g = ((int64_t)A.nrow + (int64_t)slots * kTiledRowsMax - 1) / ((int64_t)slots * kTiledRowsMax);
    R = (int)(((int64_t)A.nrow + slots * g - 1) / (slots * g));
    if (R < 256) R = A.nrow < 256 ? A.nrow : 256;
  }
  if (R > kTiledRowsMax) R = kTiledRowsMax;
  const int P = (A.nrow + R - 1) / R;
  // band width: tiles of about 0.9 work items on average (full 8-entries-per-thread items amortise the
  // two barriers per item), at most 2 MiB of x
  int W = o.tile_cols;
  if (W <= 0) {
    const double per_row = (double)A.nnz / A.nrow;
    double w = 0.9 * kTiledItem * (double)A.ncol / (per_row * R);
    if (w < 4096) w = 4096;
    if (w > (1 << kTiledColBits)) w = (1 << kTiledColBits);
    W = (int)w;
  }
  if (W > (1 << kTiledColBits)) W = 1 << kTiledColBits;
  if (W > A.ncol) W = A.ncol;
  const int J = (A.ncol + W - 1) / W;
  const int64_t ntiles = (int64_t)P * J;
  if (o.tiling == 1) {
    // pays when x does not fit the 32 KiB L1 of a CU many times over and tiles are not hopelessly thin ...
    // (measured, config-2 rows and non-zeros: x of 0.5-2 MB 0.75-0.82 ms tiled vs 0.92 ms streaming -- narrow
    // bands are L1 resident; x of 4-80 MB 0.70-1.06 ms vs 1.07-2.99 ms; x of 64 KB 1.3 ms vs 0.8 ms)
    const int64_t x_bytes = (int64_t)A.ncol * 8;
    if (x_bytes <= (256 << 10) || A.nnz < (4 << 20) || (double)A.nnz / ntiles < 256.0) return FS_OK;
    // ... and while re-reading x once per XCD and per generation of resident workgroups costs less than the
    // L2 misses it saves.  Measured rates: tiled ~150 G entries/s plus x refills at ~5 TB/s; streaming kernel
    // ~172 G entries/s while x stays L2 resident, ~53 G entries/s once every gather misses.
    const double gens = (double)((P + slots - 1) / slots);
    const double t_tiled = (double)A.nnz / 150e9 + gens * 8.0 * (double)A.ncol * 8.0 / 5e12;
    const double t_stream = (double)A.nnz / (x_bytes <= (3 << 20) ? 172e9 : 53e9);
    if (t_tiled > 0.95 * t_stream) return FS_OK;
    // ... and the matrix has no very long rows: a row's entries inside one tile are summed by one lane, and
    // a panel that holds a dense row falls behind the band sweep.  Heavy-tailed matrices (BASELINE config 5)
    // stay on the chunk-streaming kernel, whose work per workgroup does not depend on row lengths.
    Scratch<int> mx;
    int max_len = 0;
    FS_HIP(mx.alloc(1));
    FS_HIP(hipMemsetAsync(mx, 0, sizeof(int), s));
    hipLaunchKernelGGL(max_row_len_kernel, dim3(grid_for(A.nrow)), dim3(256), 0, s, A.nrow, A.row_ptr, mx.p);
    FS_HIP(hipMemcpyAsync(&max_len, mx, sizeof(int), hipMemcpyDeviceToHost, s));
    FS_HIP(hipStreamSynchronize(s));
    if ((double)max_len * W / A.ncol > 64.0 || (double)max_len > 16.0 * A.nnz / A.nrow + 4096.0) return FS_OK;
  }
  if (ntiles >= (1ll << 31)) return FS_OK;
  TiledCsr *T = new TiledCsr();
  T->R = R; T->W = W; T->P = P; T->J = J; T->lcol_bits = kTiledColBits;
  A.tiled = T;
  const size_t n = (size_t)A.nnz;
  Scratch<int> rows, tile_ptr;
  Scratch<unsigned> keys, skeys, idx_in, idx_out;
  Scratch<char> tmp;
  size_t tmp_bytes = 0;
  FS_HIP(rows.alloc(n));
  FS_HIP(keys.alloc(n));
  FS_HIP(skeys.alloc(n));
  FS_HIP(idx_in.alloc(n));
  FS_HIP(idx_out.alloc(n));
  FS_HIP(tile_ptr.alloc((size_t)ntiles + 1));
  FS_HIP(hipMalloc(&T->pk, sizeof(unsigned) * n));
  if (A.vals) FS_HIP(hipMalloc(&T->vals, sizeof(double) * n));
  hipLaunchKernelGGL(tile_key_kernel, dim3(grid_for(A.nnz)), dim3(256), 0, s, A.nrow, A.nnz, R, W, J, A.row_ptr, A.cols,
                     rows.p, keys.p);
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(A.nnz)), dim3(256), 0, s, A.nnz, idx_in.p);
  FS_HIP(hipGetLastError());
  int bits = 1;
  while (bits < 32 && (1ll << bits) < ntiles) ++bits;
  FS_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.p, skeys.p, idx_in.p, idx_out.p, n, 0, bits, s));
  FS_HIP(tmp.alloc(tmp_bytes));
  FS_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmp_bytes, keys.p, skeys.p, idx_in.p, idx_out.p, n, 0, bits, s));
  hipLaunchKernelGGL(tile_pack_kernel, dim3(grid_for(A.nnz)), dim3(256), 0, s, A.nnz, R, W, J, T->lcol_bits, skeys.p,
                     idx_out.p, rows.p, A.cols, A.vals, T->pk, T->vals);
  hipLaunchKernelGGL(tile_ptr_kernel, dim3(grid_for(ntiles + 1)), dim3(256), 0, s, ntiles, A.nnz, skeys.p, tile_ptr.p);
  FS_HIP(hipGetLastError());
  // work items are cut on the host from the tile pointers (P*J ints)
  std::vector<int> tp((size_t)ntiles + 1);
  FS_HIP(hipMemcpyAsync(tp.data(), tile_ptr.p, sizeof(int) * tp.size(), hipMemcpyDeviceToHost, s));
  FS_HIP(hipStreamSynchronize(s));
  std::vector<int4> items;
  std::vector<int> item_ptr((size_t)P + 1);
  items.reserve((size_t)(A.nnz / kTiledItem + ntiles / 4 + 16));
  for (int p = 0; p < P; ++p) {
    item_ptr[p] = (int)items.size();
    for (int j = 0; j < J; ++j) {
      const int a = tp[(size_t)p * J + j], b = tp[(size_t)p * J + j + 1];
      for (int off = a; off < b; off += kTiledItem) {
        int4 it;
        it.x = off; it.y = (b - off < kTiledItem) ? b - off : kTiledItem; it.z = j; it.w = 0;
        items.push_back(it);
      }
    }
  }
  item_ptr[P] = (int)items.size();
  T->nitems = (int)items.size();
  FS_HIP(hipMalloc(&T->items, sizeof(int4) * (items.size() ? items.size() : 1)));
  FS_HIP(hipMalloc(&T->item_ptr, sizeof(int) * item_ptr.size()));
  if (!items.empty()) FS_HIP(hipMemcpy(T->items, items.data(), sizeof(int4) * items.size(), hipMemcpyHostToDevice));
  FS_HIP(hipMemcpy(T->item_ptr, item_ptr.data(), sizeof(int) * item_ptr.size(), hipMemcpyHostToDevice));
  T->slots = slots;
  T->built = true;
  return FS_OK;
}

// ---- synthetic inputs (same arithmetic as oracle/fs_synth.c) -------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ void synth_entry(uint64_t seed, int64_t grow, int slot, int ncol, int *c, double *v)
{
  const uint64_t h = splitmix64(seed ^ ((uint64_t)grow * 0x100000001B3ull + (uint64_t)slot));
  *c = (int)__umul64hi(h, (uint64_t)ncol);
  const uint64_t h2 = splitmix64(h ^ 0xABCDEF0123456789ull);
  *v = (double)(h2 >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

__global__ void synth_uniform_kernel(int nrow, int ncol, int per_row, uint64_t seed, int64_t row_offset,
                                     int *__restrict__ row_ptr, int *__restrict__ cols, double *__restrict__ vals)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nnz = (int64_t)nrow * per_row;
  if (i <= nrow && row_ptr) row_ptr[i] = (int)(i * per_row);
  if (i >= nnz) return;
  const int64_t r = i / per_row;
  const int slot = (int)(i - r * per_row);
  int c; double v;
  synth_entry(seed, row_offset + r, slot, ncol, &c, &v);
  cols[i] = c;
  if (vals) vals[i] = v;
}

__global__ void synth_lengths_kernel(int nrow, double scale, int max_len, uint64_t seed, int64_t row_offset,
                                     int *__restrict__ len)
{
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrow) return;
  const uint64_t h = splitmix64(seed ^ (0xC0FFEEull + (uint64_t)(row_offset + r) * 0x9E3779B97F4A7C15ull));
  const double u = (double)((h >> 11) + 1) * (1.0 / 9007199254740992.0);  // (0, 1]
  double L = scale / u;
  if (L > (double)max_len) L = (double)max_len;
  int n = (int)L;
  len[r] = n < 1 ? 1 : n;
}

__global__ void synth_fill_kernel(int nrow, int ncol, uint64_t seed, int64_t row_offset,
                                  const int *__restrict__ row_ptr, int *__restrict__ cols, double *__restrict__ vals)
{
  // one wave per row, lanes stride over the row's slots
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (w >= nrow) return;
  const int a = row_ptr[w], b = row_ptr[w + 1];
  for (int i = a + lane; i < b; i += 64) {
    int c; double v;
    synth_entry(seed, row_offset + w, i - a, ncol, &c, &v);
    cols[i] = c;
    if (vals) vals[i] = v;
  }
}

}  // namespace fs

extern "C" {

int fs_synth_uniform(int nrow, int ncol, int per_row, uint64_t seed, int64_t row_offset, int *row_ptr_dev,
                     int *cols_dev, double *vals_dev, fs_stream_t stream)
{
  if (nrow < 0 || ncol < 1 || per_row < 0 || !cols_dev) { fs::set_error("fs_synth_uniform: bad argument"); return FS_ERR_ARG; }
  int64_t n = (int64_t)nrow * per_row;
  if (n < (int64_t)nrow + 1) n = (int64_t)nrow + 1;
  hipLaunchKernelGGL(fs::synth_uniform_kernel, dim3(fs::grid_for(n)), dim3(256), 0, (hipStream_t)stream, nrow, ncol,
                     per_row, seed, row_offset, row_ptr_dev, cols_dev, vals_dev);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_synth_powerlaw_lengths(int nrow, double scale, int max_len, uint64_t seed, int64_t row_offset, int *len_dev,
                              fs_stream_t stream)
{
  if (nrow < 0 || !len_dev || max_len < 1) { fs::set_error("fs_synth_powerlaw_lengths: bad argument"); return FS_ERR_ARG; }
  hipLaunchKernelGGL(fs::synth_lengths_kernel, dim3(fs::grid_for(nrow)), dim3(256), 0, (hipStream_t)stream, nrow,
                     scale, max_len, seed, row_offset, len_dev);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_synth_fill(int nrow, int ncol, uint64_t seed, int64_t row_offset, const int *row_ptr_dev, int *cols_dev,
                  double *vals_dev, fs_stream_t stream)
{
  if (nrow < 0 || ncol < 1 || !row_ptr_dev || !cols_dev) { fs::set_error("fs_synth_fill: bad argument"); return FS_ERR_ARG; }
  hipLaunchKernelGGL(fs::synth_fill_kernel, dim3(fs::grid_for((int64_t)nrow * 64)), dim3(256), 0, (hipStream_t)stream,
                     nrow, ncol, seed, row_offset, row_ptr_dev, cols_dev, vals_dev);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

}  // extern "C"
