// fs_dropin.hip -- the reference's entry points (include/sparse.h, dsparse.h, csr.h, cbcsr.h)
// dispatched onto the device layer (fs_abi.hip -> fs_kernels.hip).
//
// Contract of every product below:
//   * the matrix argument is the reference's host struct; its device copy is made on first
//     use and kept in a side table keyed by (struct address, variant).  An entry is reused
//     only while the struct's dimensions, array pointers and a sampled fingerprint of the
//     index arrays are unchanged; otherwise it is rebuilt.  fs_invalidate()/free_*() drop it.
//   * x / y may be host or device pointers (hipPointerGetAttributes decides); host vectors are
//     staged through per-thread device buffers.  The call returns after y is complete.
//   * y is overwritten, never accumulated into (SURVEY.md note N5).
//   * the functions return void like the reference's; a HIP failure prints the reason and
//     exits -- there is no CPU fallback.
#include <string.h>

#include <unordered_map>
#include <vector>

#include "cbcsr.h"
#include "cg.h"
#include "csr.h"
#include "dsparse.h"
#include "fs_common.h"
#include "sparse.h"

namespace {

[[noreturn]] void die(const char *who)
{
  fprintf(stderr, "libfastsparse_hip: %s failed: %s\n", who, fs_last_error());
  exit(1);
}

#define FS_MUST(expr, who) do { if ((expr) != FS_OK) die(who); } while (0)

// ---- fingerprints ------------------------------------------------------------------------------
uint64_t mix(uint64_t h, uint64_t v)
{
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  return h;
}

// up to 2048 strided samples plus both ends
uint64_t sample_ints(uint64_t h, const int *a, int64_t n)
{
  if (!a || n <= 0) return mix(h, 0);
  const int64_t step = n > 2048 ? n / 2048 : 1;
  for (int64_t i = 0; i < n; i += step) h = mix(h, (uint64_t)(unsigned)a[i]);
  return mix(h, (uint64_t)(unsigned)a[n - 1]);
}

uint64_t sample_doubles(uint64_t h, const double *a, int64_t n)
{
  if (!a || n <= 0) return mix(h, 0);
  const int64_t step = n > 2048 ? n / 2048 : 1;
  for (int64_t i = 0; i < n; i += step) { uint64_t b; memcpy(&b, a + i, 8); h = mix(h, b); }
  return h;
}

// ---- side table ------------------------------------------------------------------------------------
enum Variant { kDirect = 0, kTransposed = 1 };

struct Key {
  const void *host;
  int variant;
  bool operator==(const Key &o) const { return host == o.host && variant == o.variant; }
};
struct KeyHash {
  size_t operator()(const Key &k) const { return std::hash<const void *>()(k.host) * 31u + (size_t)k.variant; }
};
struct Entry {
  fs_matrix_t m = nullptr;
  fs_cbcsr_t cb = nullptr;
  uint64_t print = 0;
};

std::mutex g_table_lock;
std::unordered_map<Key, Entry, KeyHash> g_table;

void drop(Entry &e)
{
  if (e.m) fs_matrix_destroy(e.m);
  if (e.cb) fs_cbcsr_destroy(e.cb);
  e = Entry();
}

// returns the cached entry for (host, variant) when its fingerprint matches, else builds it
template <typename Build>
Entry lookup(const void *host, int variant, uint64_t print, Build build, const char *who)
{
  std::lock_guard<std::mutex> g(g_table_lock);
  Entry &e = g_table[Key{host, variant}];
  if ((e.m || e.cb) && e.print == print) return e;
  drop(e);
  build(e);
  if (!e.m && !e.cb) die(who);
  e.print = print;
  return e;
}

// ---- dense vectors ----------------------------------------------------------------------------------
bool on_device(const void *p)
{
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // plain malloc memory: not known to HIP
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

struct Staging {
  double *buf[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  double *get(int which, size_t n)
  {
    if (cap[which] < n) {
      if (buf[which]) (void)hipFree(buf[which]);
      buf[which] = nullptr; cap[which] = 0;
      if (hipMalloc(&buf[which], sizeof(double) * n) != hipSuccess) {
        fs::set_error("hipMalloc of a staging vector failed");
        die("vector staging");
      }
      cap[which] = n;
    }
    return buf[which];
  }
  ~Staging() { /* device memory is reclaimed with the context */ }
};
thread_local Staging g_stage;

// run `mul(y_dev, x_dev)` with host/device x and y of nx / ny doubles
template <typename Mul>
void with_vectors(double *y, size_t ny, const double *x, size_t nx, Mul mul, const char *who)
{
  const bool xd = on_device(x), yd = on_device(y);
  const double *xdev = x;
  double *ydev = y;
  if (!xd) {
    double *b = g_stage.get(0, nx ? nx : 1);
    if (nx && hipMemcpy(b, x, sizeof(double) * nx, hipMemcpyHostToDevice) != hipSuccess) {
      fs::set_error("copy of x to the device failed"); die(who);
    }
    xdev = b;
  }
  if (!yd) ydev = g_stage.get(1, ny ? ny : 1);
  FS_MUST(mul(ydev, xdev), who);
  if (!yd) {
    if (ny && hipMemcpy(y, ydev, sizeof(double) * ny, hipMemcpyDeviceToHost) != hipSuccess) {
      fs::set_error("copy of y from the device failed"); die(who);
    }
  } else if (hipStreamSynchronize(nullptr) != hipSuccess) {
    fs::set_error("stream synchronisation failed"); die(who);
  }
}

// ---- per-format uploads -----------------------------------------------------------------------------
uint64_t print_csr(int nrow, int ncol, long nnz, const int *row_ptr, const int *cols, const double *vals)
{
  uint64_t h = mix(mix(mix(1, (uint64_t)nrow), (uint64_t)ncol), (uint64_t)nnz);
  h = mix(mix(mix(h, (uint64_t)(uintptr_t)row_ptr), (uint64_t)(uintptr_t)cols), (uint64_t)(uintptr_t)vals);
  h = sample_ints(h, row_ptr, (int64_t)nrow + 1);
  h = sample_ints(h, cols, nnz);
  return sample_doubles(h, vals, nnz);
}

fs_matrix_t csr_handle(const void *host, int nrow, int ncol, long nnz, const int *row_ptr, const int *cols,
                       const double *vals, bool need_t, const char *who)
{
  const uint64_t p = print_csr(nrow, ncol, nnz, row_ptr, cols, vals);
  Entry e = lookup(host, kDirect, p, [&](Entry &n) { n.m = fs_csr_create(nrow, ncol, nnz, row_ptr, cols, vals, FS_HOST, 0); },
                   who);
  if (need_t) FS_MUST(fs_matrix_build_transpose(e.m, nullptr), who);
  return e.m;
}

// COO (optionally valued); variant kTransposed uploads (cols, rows) so that each output element
// keeps the entry order of the serial loop it replaces (sparse.h:72-74, dsparse.h:58-60)
fs_matrix_t coo_handle(const void *host, int variant, int nrow, int ncol, long nnz, const int *rows, const int *cols,
                       const double *vals, const char *who)
{
  uint64_t h = mix(mix(mix(2, (uint64_t)nrow), (uint64_t)ncol), (uint64_t)nnz);
  h = mix(mix(mix(h, (uint64_t)(uintptr_t)rows), (uint64_t)(uintptr_t)cols), (uint64_t)(uintptr_t)vals);
  h = sample_doubles(sample_ints(sample_ints(h, rows, nnz), cols, nnz), vals, nnz);
  return lookup(host, variant, h, [&](Entry &n) {
    n.m = variant == kDirect ? fs_coo_create(nrow, ncol, nnz, rows, cols, vals, FS_HOST)
                             : fs_coo_create(ncol, nrow, nnz, cols, rows, vals, FS_HOST);
  }, who).m;
}

// row-blocked COO: the per-block arrays are laid end to end and uploaded as one COO; a row lives in
// exactly one block, so its entries keep the order bsbm_A_mul_B (sparse.h:269-271) adds them in
fs_matrix_t blocked_handle(const void *host, int nrow, int ncol, int nblocks, const int *blk_nnz, int **brows,
                           int **bcols, double **bvals, const char *who)
{
  uint64_t h = mix(mix(mix(3, (uint64_t)nrow), (uint64_t)ncol), (uint64_t)nblocks);
  h = sample_ints(h, blk_nnz, nblocks);
  int64_t nnz = 0;
  for (int b = 0; b < nblocks; b++) {
    nnz += blk_nnz[b];
    h = mix(mix(h, (uint64_t)(uintptr_t)brows[b]), (uint64_t)(uintptr_t)bcols[b]);
    if (b % (nblocks / 64 + 1) == 0) h = sample_ints(sample_ints(h, brows[b], blk_nnz[b]), bcols[b], blk_nnz[b]);
  }
  return lookup(host, kDirect, h, [&](Entry &n) {
    std::vector<int> r((size_t)nnz), c((size_t)nnz);
    std::vector<double> v(bvals ? (size_t)nnz : 0);
    size_t o = 0;
    for (int b = 0; b < nblocks; b++) {
      const size_t m = (size_t)blk_nnz[b];
      if (m) {
        memcpy(r.data() + o, brows[b], sizeof(int) * m);
        memcpy(c.data() + o, bcols[b], sizeof(int) * m);
        if (bvals) memcpy(v.data() + o, bvals[b], sizeof(double) * m);
      }
      o += m;
    }
    n.m = fs_coo_create(nrow, ncol, nnz, r.data(), c.data(), bvals ? v.data() : nullptr, FS_HOST);
  }, who).m;
}

void bcsr_mul_k(double *Y, struct BinaryCSR *A, double *X, int k, const char *who)
{
  fs_matrix_t m = csr_handle(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, false, who);
  with_vectors(Y, (size_t)A->nrow * k, X, (size_t)A->ncol * k,
               [&](double *yd, const double *xd) { return fs_spmm(m, yd, xd, k, nullptr); }, who);
}

void bsbm_mul_k(double *Y, struct BlockedSBM *B, double *X, int k, const char *who)
{
  fs_matrix_t m = blocked_handle(B, B->nrow, B->ncol, B->nblocks, B->nnz, B->rows, B->cols, nullptr, who);
  with_vectors(Y, (size_t)B->nrow * k, X, (size_t)B->ncol * k,
               [&](double *yd, const double *xd) { return fs_spmm(m, yd, xd, k, nullptr); }, who);
}

}  // namespace

extern "C" {

void fs_invalidate(const void *host_struct)
{
  std::lock_guard<std::mutex> g(g_table_lock);
  for (int v = 0; v < 2; v++) {
    auto it = g_table.find(Key{host_struct, v});
    if (it != g_table.end()) { drop(it->second); g_table.erase(it); }
  }
}

void fs_release_all(void)
{
  std::lock_guard<std::mutex> g(g_table_lock);
  for (auto &kv : g_table) drop(kv.second);
  g_table.clear();
}

// ---- sparse.h ----------------------------------------------------------------------------------------
void A_mul_B(double *y, struct SparseBinaryMatrix *A, double *x)
{
  fs_matrix_t m = coo_handle(A, kDirect, A->nrow, A->ncol, A->nnz, A->rows, A->cols, nullptr, "A_mul_B");
  with_vectors(y, A->nrow, x, A->ncol, [&](double *yd, const double *xd) { return fs_spmv(m, yd, xd, nullptr); }, "A_mul_B");
}

void At_mul_B(double *y, struct SparseBinaryMatrix *A, double *x)
{
  fs_matrix_t m = coo_handle(A, kTransposed, A->nrow, A->ncol, A->nnz, A->rows, A->cols, nullptr, "At_mul_B");
  with_vectors(y, A->ncol, x, A->nrow, [&](double *yd, const double *xd) { return fs_spmv(m, yd, xd, nullptr); }, "At_mul_B");
}

void bsbm_A_mul_B(double *y, struct BlockedSBM *B, double *x) { bsbm_mul_k(y, B, x, 1, "bsbm_A_mul_B"); }
void bsbm_A_mul_B2(double *y, struct BlockedSBM *B, double *x) { bsbm_mul_k(y, B, x, 2, "bsbm_A_mul_B2"); }
void bsbm_A_mul_B4(double *y, struct BlockedSBM *B, double *x) { bsbm_mul_k(y, B, x, 4, "bsbm_A_mul_B4"); }
void bsbm_A_mul_Bn(double *y, struct BlockedSBM *B, double *x, int ncol) { bsbm_mul_k(y, B, x, ncol, "bsbm_A_mul_Bn"); }

// ---- cg.h --------------------------------------------------------------------------------------------------
void bsbm_AtA(double *y, struct BlockedSBM *A, struct BlockedSBM *At, double *x, double *tmp, double lambda)
{
  (void)tmp;  // host scratch of the CPU version; the intermediate A x stays in HBM here
  fs_matrix_t a = blocked_handle(A, A->nrow, A->ncol, A->nblocks, A->nnz, A->rows, A->cols, nullptr, "bsbm_AtA");
  fs_matrix_t at = blocked_handle(At, At->nrow, At->ncol, At->nblocks, At->nnz, At->rows, At->cols, nullptr, "bsbm_AtA");
  double *t = nullptr;
  if (hipMalloc(&t, sizeof(double) * (size_t)(A->nrow ? A->nrow : 1)) != hipSuccess) {
    fs::set_error("hipMalloc of the A x scratch failed"); die("bsbm_AtA");
  }
  with_vectors(y, At->nrow, x, A->ncol, [&](double *yd, const double *xd) {
    if (int rc = fs_spmv(a, t, xd, nullptr)) return rc;
    if (int rc = fs_spmv(at, yd, t, nullptr)) return rc;
    // y += lambda x (cg.h:17-21): reuse the SpMV-side axpy of the solver through a tiny CG-free path
    return fs_axpy(At->nrow, lambda, xd, yd, nullptr);
  }, "bsbm_AtA");
  (void)hipFree(t);
}

static void cg_common(double *x, struct BlockedSBM *A, struct BlockedSBM *At, double *b, double lambda, double tol,
                      int *out_iter, int k, const char *who)
{
  if (A->nrow != At->ncol || A->ncol != At->nrow) {  // cg.h:32-36
    printf("A (%d x %d) and At (%d x %d) must be transposes of each other.\n", A->nrow, A->ncol, At->nrow, At->ncol);
    exit(1);
  }
  fs_matrix_t a = blocked_handle(A, A->nrow, A->ncol, A->nblocks, A->nnz, A->rows, A->cols, nullptr, who);
  fs_matrix_t at = blocked_handle(At, At->nrow, At->ncol, At->nblocks, At->nnz, At->rows, At->cols, nullptr, who);
  const size_t n = (size_t)A->ncol * k;
  int iters = 0;
  with_vectors(x, n, b, n, [&](double *xd, const double *bd) {
    return k == 1 ? fs_cg(a, at, xd, bd, lambda, tol, &iters, nullptr) : fs_cg2(a, at, xd, bd, lambda, tol, &iters, nullptr);
  }, who);
  if (out_iter) *out_iter = iters;
}

void bsbm_cg(double *x, struct BlockedSBM *A, struct BlockedSBM *At, double *b, double lambda, double tol, int *out_iter)
{
  cg_common(x, A, At, b, lambda, tol, out_iter, 1, "bsbm_cg");
}

void bsbm_cg2(double *X, struct BlockedSBM *A, struct BlockedSBM *At, double *B, double lambda, double tol, int *out_iter)
{
  cg_common(X, A, At, B, lambda, tol, out_iter, 2, "bsbm_cg2");
}

// ---- dsparse.h ---------------------------------------------------------------------------------------
void sdm_A_mul_B(double *y, struct SparseDoubleMatrix *A, double *x)
{
  fs_matrix_t m = coo_handle(A, kDirect, A->nrow, A->ncol, A->nnz, A->rows, A->cols, A->vals, "sdm_A_mul_B");
  with_vectors(y, A->nrow, x, A->ncol, [&](double *yd, const double *xd) { return fs_spmv(m, yd, xd, nullptr); }, "sdm_A_mul_B");
}

void sdm_At_mul_B(double *y, struct SparseDoubleMatrix *A, double *x)
{
  fs_matrix_t m = coo_handle(A, kTransposed, A->nrow, A->ncol, A->nnz, A->rows, A->cols, A->vals, "sdm_At_mul_B");
  with_vectors(y, A->ncol, x, A->nrow, [&](double *yd, const double *xd) { return fs_spmv(m, yd, xd, nullptr); }, "sdm_At_mul_B");
}

void bsdm_A_mul_B(double *y, struct BlockedSDM *B, double *x)
{
  fs_matrix_t m = blocked_handle(B, B->nrow, B->ncol, B->nblocks, B->nnz, B->rows, B->cols, B->vals, "bsdm_A_mul_B");
  with_vectors(y, B->nrow, x, B->ncol, [&](double *yd, const double *xd) { return fs_spmv(m, yd, xd, nullptr); }, "bsdm_A_mul_B");
}

// ---- csr.h ---------------------------------------------------------------------------------------------
void free_bcsr(struct BinaryCSR *bcsr)
{
  fs_invalidate(bcsr);
  free(bcsr->row_ptr);
  free(bcsr->cols);
}

void free_csr(struct CSR *csr)
{
  fs_invalidate(csr);
  free(csr->row_ptr);
  free(csr->cols);
  free(csr->vals);
}

void bcsr_A_mul_B(double *y, struct BinaryCSR *A, double *x) { bcsr_mul_k(y, A, x, 1, "bcsr_A_mul_B"); }
void bcsr_A_mul_B2(double *Y, struct BinaryCSR *A, double *X) { bcsr_mul_k(Y, A, X, 2, "bcsr_A_mul_B2"); }
void bcsr_A_mul_B4(double *Y, struct BinaryCSR *A, double *X) { bcsr_mul_k(Y, A, X, 4, "bcsr_A_mul_B4"); }
void bcsr_A_mul_B8(double *Y, struct BinaryCSR *A, double *X) { bcsr_mul_k(Y, A, X, 8, "bcsr_A_mul_B8"); }
void bcsr_A_mul_B8_auto(double *Y, struct BinaryCSR *A, double *X) { bcsr_mul_k(Y, A, X, 8, "bcsr_A_mul_B8_auto"); }
void bcsr_A_mul_Bn(double *Y, struct BinaryCSR *A, double *X, const int ncol) { bcsr_mul_k(Y, A, X, ncol, "bcsr_A_mul_Bn"); }

void bcsr_A_mul_B32n(double *Y, struct BinaryCSR *A, double *X, const int ncol)
{
  if (ncol > 32) {  // the reference asserts this (csr.h:284)
    fprintf(stderr, "libfastsparse_hip: bcsr_A_mul_B32n: ncol = %d > 32\n", ncol);
    abort();
  }
  bcsr_mul_k(Y, A, X, ncol, "bcsr_A_mul_B32n");
}

void bcsr_AA_mul_B(double *y, struct BinaryCSR *A, double *x)
{
  fs_matrix_t m = csr_handle(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, true, "bcsr_AA_mul_B");
  double *tmp = nullptr;
  if (hipMalloc(&tmp, sizeof(double) * (size_t)(A->nrow ? A->nrow : 1)) != hipSuccess) {
    fs::set_error("hipMalloc of the A x scratch failed"); die("bcsr_AA_mul_B");
  }
  with_vectors(y, A->ncol, x, A->ncol, [&](double *yd, const double *xd) { return fs_ata_mul(m, yd, xd, tmp, nullptr); },
               "bcsr_AA_mul_B");
  (void)hipFree(tmp);
}

void parallel_bcsr_AA_mul_B(double *y, struct BinaryCSR *A, double *x, double *ytmp)
{
  (void)ytmp;  // per-thread replicas of y are a CPU device; not needed here
  bcsr_AA_mul_B(y, A, x);
}

void bcsr_At_mul_B(double *y, struct BinaryCSR *A, double *x)
{
  fs_matrix_t m = csr_handle(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, true, "bcsr_At_mul_B");
  with_vectors(y, A->ncol, x, A->nrow, [&](double *yd, const double *xd) { return fs_spmv_t(m, yd, xd, nullptr); }, "bcsr_At_mul_B");
}

void csr_A_mul_B(double *y, struct CSR *A, double *x)
{
  fs_matrix_t m = csr_handle(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, false, "csr_A_mul_B");
  with_vectors(y, A->nrow, x, A->ncol, [&](double *yd, const double *xd) { return fs_spmv(m, yd, xd, nullptr); }, "csr_A_mul_B");
}

void csr_At_mul_B(double *y, struct CSR *A, double *x)
{
  fs_matrix_t m = csr_handle(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, true, "csr_At_mul_B");
  with_vectors(y, A->ncol, x, A->nrow, [&](double *yd, const double *xd) { return fs_spmv_t(m, yd, xd, nullptr); }, "csr_At_mul_B");
}

void csr_A_mul_Bn(double *Y, struct CSR *A, double *X, const int ncol)
{
  fs_matrix_t m = csr_handle(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, false, "csr_A_mul_Bn");
  with_vectors(Y, (size_t)A->nrow * ncol, X, (size_t)A->ncol * ncol,
               [&](double *yd, const double *xd) { return fs_spmm(m, yd, xd, ncol, nullptr); }, "csr_A_mul_Bn");
}

// ---- cbcsr.h -------------------------------------------------------------------------------------------
void cbcsr_A_mul_B(double *y, struct ColBinaryCSR *A, double *x)
{
  const int64_t ncell = (int64_t)A->nblocks * A->nrow;
  uint64_t h = mix(mix(mix(4, (uint64_t)A->nrow), (uint64_t)A->ncol), (uint64_t)A->nnz);
  h = mix(mix(mix(h, (uint64_t)A->colblocksize), (uint64_t)(uintptr_t)A->row_ptr), (uint64_t)(uintptr_t)A->cols);
  h = sample_ints(sample_ints(h, A->row_ptr, ncell + 1), A->cols, A->nnz);
  Entry e = lookup(A, kDirect, h, [&](Entry &n) {
    n.cb = fs_cbcsr_create(A->nrow, A->ncol, A->nblocks, A->colblocksize, A->row_ptr, A->cols, FS_HOST);
  }, "cbcsr_A_mul_B");
  with_vectors(y, A->nrow, x, A->ncol, [&](double *yd, const double *xd) { return fs_cbcsr_spmv(e.cb, yd, xd, nullptr); },
               "cbcsr_A_mul_B");
}

}  // extern "C"
