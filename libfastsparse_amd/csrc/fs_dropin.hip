// fs_dropin.hip -- the reference's entry points (include/sparse.h, dsparse.h, csr.h, cbcsr.h)
// dispatched onto the device layer (fs_abi.hip -> fs_kernels.hip).
//
// Contract of every product below:
//   * the matrix argument is the reference's host struct; its device copy is made on first
//     use and kept in a side table keyed by (struct address, variant).  An entry is reused
//     only while the struct's dimensions, array pointers and a fingerprint of the arrays (full
//     for small matrices, sampled for large ones: see "fingerprints" below; FS_STRICT_CACHE=1 /
//     FS_DROPIN_CACHE=0 change that) are unchanged; otherwise it is rebuilt.  fs_invalidate(),
//     free_sbm/free_bcsr/free_csr, transpose and the sort_* functions drop it.  Entries are
//     ref-counted: a copy in use by another host thread outlives its removal from the table.
//   * x / y may be host or device pointers (hipPointerGetAttributes decides); host vectors are
//     staged through per-thread device buffers.  The call returns after y is complete.
//   * FASTSPARSE_NGPU=N in the environment: every product below runs row-sharded over N GPUs ("several GPUs" further down).
//   * y is overwritten, never accumulated into (SURVEY.md note N5).
//   * the functions return void like the reference's; a HIP failure prints the reason and
//     exits -- there is no CPU fallback.
#include <string.h>

#include <stdlib.h>

#include <memory>
#include <unordered_map>
#include <vector>

#include "cbcsr.h"
#include "cg.h"
#include "csr.h"
#include "dsparse.h"
#include "fs_common.h"
#include "sparse.h"

extern "C" long fs_debug_dist_products(void);

namespace {

[[noreturn]] void die(const char *who)
{
  fprintf(stderr, "libfastsparse_hip: %s failed: %s\n", who, fs_last_error());
  exit(1);
}

#define FS_MUST(expr, who) do { if ((expr) != FS_OK) die(who); } while (0)

// ---- fingerprints ------------------------------------------------------------------------------
// An entry of the side table is reused only while the host struct still describes the same matrix.  Dimensions and
// array pointers are always compared.  The CONTENTS of the arrays are hashed IN FULL while that is cheap (at most
// kFullHashBytes in all: re-reading 8 MB of host memory costs less than the PCIe copy of the vectors it rides with),
// otherwise at 2048 strided samples per array plus both ends -- enough to notice a re-sort, a transpose or a reload,
// not a handful of entries edited in place at the same addresses.  Callers who edit large matrices in place either
// call fs_invalidate(A) (what this library's own sort_* / transpose / free_* do) or set FS_STRICT_CACHE=1 (every
// array hashed in full on every call: a call then costs a pass over the host arrays) or FS_DROPIN_CACHE=0 (nothing
// is reused: every call uploads).  Both switches are read once.
constexpr int64_t kFullHashBytes = 8 << 20;

int env_int(const char *name, int dflt)
{
  const char *v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
bool strict_cache() { static const bool v = env_int("FS_STRICT_CACHE", 0) != 0; return v; }
bool cache_enabled() { static const bool v = env_int("FS_DROPIN_CACHE", 1) != 0; return v; }
size_t max_entries() { static const int v = env_int("FS_DROPIN_MAX_ENTRIES", 64); return (size_t)(v > 0 ? v : 1); }

uint64_t mix(uint64_t h, uint64_t v)
{
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  return h;
}

// every byte: four independent multiply-rotate lanes over 8-byte words (about 10 GB/s on one core)
uint64_t hash_bytes(uint64_t h, const void *p, size_t n)
{
  const unsigned char *b = (const unsigned char *)p;
  uint64_t l[4] = {h ^ 0x243F6A8885A308D3ull, h ^ 0x13198A2E03707344ull, h ^ 0xA4093822299F31D0ull, h ^ 0x082EFA98EC4E6C89ull};
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    memcpy(w, b + i, 32);
    for (int k = 0; k < 4; k++) {
      l[k] = (l[k] ^ w[k]) * 0x9E3779B97F4A7C15ull;
      l[k] = (l[k] << 29) | (l[k] >> 35);
    }
  }
  // the last n - i < 32 bytes: up to four zero-padded words, every one of them mixed in (every byte counts: an array of
  // 3 ints or 2 doubles is ALL tail)
  uint64_t w[4] = {0, 0, 0, 0};
  memcpy(w, b + i, n - i);
  uint64_t r = mix(mix(mix(mix(h, l[0]), l[1]), l[2]), l[3]);
  for (int k = 0; k < 4; k++) r = mix(r, w[k]);
  return mix(r, (uint64_t)n);
}

// `full`: hash everything; otherwise up to `samples` strided samples plus both ends
uint64_t print_ints(uint64_t h, const int *a, int64_t n, bool full, int samples = 2048)
{
  if (!a || n <= 0) return mix(h, 0);
  if (full) return hash_bytes(h, a, sizeof(int) * (size_t)n);
  const int64_t step = n > samples ? n / samples : 1;
  for (int64_t i = 0; i < n; i += step) h = mix(h, (uint64_t)(unsigned)a[i]);
  return mix(h, (uint64_t)(unsigned)a[n - 1]);
}

uint64_t print_doubles(uint64_t h, const double *a, int64_t n, bool full, int samples = 2048)
{
  if (!a || n <= 0) return mix(h, 0);
  if (full) return hash_bytes(h, a, sizeof(double) * (size_t)n);
  const int64_t step = n > samples ? n / samples : 1;
  for (int64_t i = 0; i < n; i += step) { uint64_t b; memcpy(&b, a + i, 8); h = mix(h, b); }
  uint64_t e; memcpy(&e, a + n - 1, 8);
  return mix(h, e);
}

bool hash_in_full(int64_t bytes) { return strict_cache() || bytes <= kFullHashBytes; }

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
// An entry is handed out as a shared_ptr and held by the caller across its launch: free_*() / fs_invalidate() /
// a rebuild from another host thread only take it out of the table, the device copy itself lives until the last
// product using it has returned (bench_a_mul_b.c:401-421 runs two host threads over shared matrices).
struct Entry {
  fs_matrix_t m = nullptr;
  fs_cbcsr_t cb = nullptr;
  fs_dist_matrix_t dm = nullptr;   // FASTSPARSE_NGPU > 1: the matrix row-sharded over the node's GPUs
  uint64_t print = 0;
  uint64_t tick = 0;          // last use, for eviction
  Entry() = default;
  Entry(const Entry &) = delete;
  Entry &operator=(const Entry &) = delete;
  ~Entry()
  {
    if (m) fs_matrix_destroy(m);
    if (cb) fs_cbcsr_destroy(cb);
    if (dm) fs_dist_matrix_destroy(dm);
  }
};
typedef std::shared_ptr<Entry> EntryP;

std::mutex g_table_lock;
std::unordered_map<Key, EntryP, KeyHash> g_table;
uint64_t g_tick = 0;

// drops the least recently used entries nobody is using right now until at most `keep` are left (table lock held)
void evict_lru(size_t keep)
{
  while (g_table.size() > keep) {
    auto victim = g_table.end();
    for (auto it = g_table.begin(); it != g_table.end(); ++it)
      if (it->second.use_count() == 1 && (victim == g_table.end() || it->second->tick < victim->second->tick)) victim = it;
    if (victim == g_table.end()) return;   // everything left is in use
    g_table.erase(victim);
  }
}

// returns the cached entry for (host, variant) when its fingerprint matches, else builds it.  A build that fails
// (typically: HBM full of other matrices' copies -- callers that construct matrices in a loop and never free them)
// is retried once after every idle entry has been dropped.
template <typename Build>
EntryP lookup(const void *host, int variant, uint64_t print, Build build, const char *who)
{
  std::lock_guard<std::mutex> g(g_table_lock);
  const Key key{host, variant};
  auto it = g_table.find(key);
  if (it != g_table.end()) {
    if (cache_enabled() && it->second->print == print) {
      it->second->tick = ++g_tick;
      return it->second;
    }
    g_table.erase(it);        // a stale copy: gone from the table now, destroyed when its last user returns
  }
  EntryP e = std::make_shared<Entry>();
  fs::KeepCsrScope keep;       // option release_csr is not for this layer: its callers may use any entry point next
  build(*e);
  if (!e->m && !e->cb && !e->dm) {
    evict_lru(0);
    (void)hipGetLastError();
    build(*e);
  }
  if (!e->m && !e->cb && !e->dm) die(who);
  e->print = print;
  e->tick = ++g_tick;
  g_table[key] = e;
  evict_lru(max_entries());
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

// y = A x (or A' x) of one handle: vectors both in host memory take fs_spmv_host, whose copies overlap the kernels
void product(fs_matrix_t m, bool transposed, double *y, size_t ny, const double *x, size_t nx, const char *who)
{
  if (ny && nx && !on_device(x) && !on_device(y)) {
    FS_MUST(transposed ? fs_spmv_t_host(m, y, x) : fs_spmv_host(m, y, x), who);
    return;
  }
  with_vectors(y, ny, x, nx, [&](double *yd, const double *xd) {
    return transposed ? fs_spmv_t(m, yd, xd, nullptr) : fs_spmv(m, yd, xd, nullptr); }, who);
}

// ---- fingerprints of the reference's containers (what an entry of the side table is compared by) -----------------------------
uint64_t csr_print(int nrow, int ncol, long nnz, const int *row_ptr, const int *cols, const double *vals)
{
  const bool full = hash_in_full((int64_t)nnz * (vals ? 12 : 4) + 4 * ((int64_t)nrow + 1));
  uint64_t h = mix(mix(mix(1, (uint64_t)nrow), (uint64_t)ncol), (uint64_t)nnz);
  h = mix(mix(mix(h, (uint64_t)(uintptr_t)row_ptr), (uint64_t)(uintptr_t)cols), (uint64_t)(uintptr_t)vals);
  h = print_ints(h, row_ptr, (int64_t)nrow + 1, full);
  h = print_ints(h, cols, nnz, full);
  return print_doubles(h, vals, nnz, full);
}

uint64_t coo_print(int nrow, int ncol, long nnz, const int *rows, const int *cols, const double *vals)
{
  const bool full = hash_in_full((int64_t)nnz * (vals ? 16 : 8));
  uint64_t h = mix(mix(mix(2, (uint64_t)nrow), (uint64_t)ncol), (uint64_t)nnz);
  h = mix(mix(mix(h, (uint64_t)(uintptr_t)rows), (uint64_t)(uintptr_t)cols), (uint64_t)(uintptr_t)vals);
  return print_doubles(print_ints(print_ints(h, rows, nnz, full), cols, nnz, full), vals, nnz, full);
}

// per call: nothing here may cost O(nblocks) once the blocks run into the millions (block size 8 on 10 M rows)
uint64_t blocked_print(int nrow, int ncol, int nblocks, const int *blk_nnz, int **brows, int **bcols, double **bvals)
{
  const bool few = nblocks <= (1 << 14);
  int64_t nnz = 0;
  if (few)
    for (int b = 0; b < nblocks; b++) nnz += blk_nnz[b];
  const bool full = few && hash_in_full(nnz * (bvals ? 16 : 8));
  uint64_t h = mix(mix(mix(3, (uint64_t)nrow), (uint64_t)ncol), (uint64_t)nblocks);
  h = print_ints(h, blk_nnz, nblocks, few);
  // sampled mode: the same budget as an unblocked matrix -- about 2048 samples per array in ALL, spread over at most 64
  // blocks (2048 per array in EACH of 64 blocks cost 0.85 ms of host time per call on a 2 M x 200 K matrix of 1954 blocks,
  // five times the product itself; the reference's bench loops over bsbm_A_mul_B)
  const int stride = nblocks / 64 + 1, sampled = (nblocks + stride - 1) / stride;
  const int per_block = 2048 / (sampled > 0 ? sampled : 1) + 1;     // nblocks == 0 (a matrix without rows): nothing to sample
  const int pstride = full ? 1 : nblocks / 4096 + 1;              // array pointers: all of them up to 4096 blocks
  for (int b = 0; b < nblocks; b += pstride) h = mix(mix(h, (uint64_t)(uintptr_t)brows[b]), (uint64_t)(uintptr_t)bcols[b]);
  for (int b = 0; b < nblocks; b += full ? 1 : stride) {
    h = print_ints(print_ints(h, brows[b], blk_nnz[b], full, per_block), bcols[b], blk_nnz[b], full, per_block);
    if (bvals) h = print_doubles(h, bvals[b], blk_nnz[b], full, per_block);
  }
  return h;
}

// the per-block arrays of a row-blocked matrix laid end to end: one COO whose rows keep the order bsbm_A_mul_B (sparse.h:269-271)
// adds them in (a row lives in exactly one block)
struct BlockedAsCoo {
  std::vector<int> r, c;
  std::vector<double> v;
  BlockedAsCoo(int nblocks, const int *blk_nnz, int **brows, int **bcols, double **bvals)
  {
    int64_t nnz = 0;
    for (int b = 0; b < nblocks; b++) nnz += blk_nnz[b];
    r.resize((size_t)nnz); c.resize((size_t)nnz);
    if (bvals) v.resize((size_t)nnz);
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
  }
};

// ---- several GPUs (FASTSPARSE_NGPU) ---------------------------------------------------------------------
// EVERY product entry point of a plain C caller across the GPUs of the node: FASTSPARSE_NGPU=N in the environment (and optionally
// FASTSPARSE_DEVICES=0,1,...: the device of every rank).  One context for the process.  Each container becomes the same CSR it
// becomes on one GPU -- so every row keeps the reference's order of additions -- cut into row shards by non-zeros (fs_dist.hip):
//   struct CSR / BinaryCSR            the arrays as they are; A' x: row shards of A' from the same arrays
//   COO (sbm / sdm)                   stable bucketing by row (fs_dist_coo_create); At_mul_B: the COO of (cols, rows)
//   BlockedSBM / BlockedSDM           the blocks end to end as one COO
//   ColBinaryCSR                      a CSR whose rows hold their cells' entries block by block
//   bsbm_cg / bsbm_cg2 / bsbm_AtA     the caller's A and At as a PAIR sharing both matrices' shards (fs_dist_matrix_pair)
// x / y in host memory or in HBM; a vector in HBM never touches the host (fs_dist.hip: vec_in / vec_out).
enum { kDist = 2, kDistT = 3, kDistPair = 4, kVariants = 5 };

int dist_ranks() { static const int v = env_int("FASTSPARSE_NGPU", 1); return v; }
bool dist_on() { return dist_ranks() > 1; }

fs_dist_t dist_context(const char *who)
{
  static std::mutex lock;
  static fs_dist_t D = nullptr;
  std::lock_guard<std::mutex> g(lock);
  if (D) return D;
  std::vector<int> devs;
  if (const char *list = getenv("FASTSPARSE_DEVICES"))
    for (const char *p = list; *p;) {
      devs.push_back(atoi(p));
      while (*p && *p != ',') ++p;
      if (*p == ',') ++p;
    }
  const int n = dist_ranks();
  if (!devs.empty() && (int)devs.size() != n) {
    fs::set_error("FASTSPARSE_DEVICES does not list FASTSPARSE_NGPU devices");
    die(who);
  }
  D = fs_dist_create(n, devs.empty() ? nullptr : devs.data());
  if (!D) die(who);
  if (env_int("FS_TRACE_DIST", 0))   // proof of the path taken for callers that cannot ask (the reference's own test program)
    atexit([] { fprintf(stderr, "[fastsparse] %ld sharded products on %d ranks\n", fs_debug_dist_products(), dist_ranks()); });
  return D;
}

EntryP dist_csr_entry(const void *host, int nrow, int ncol, long nnz, const int *row_ptr, const int *cols, const double *vals,
                      bool need_t, const char *who)
{
  fs_dist_t D = dist_context(who);
  EntryP e = lookup(host, kDist, mix(5, csr_print(nrow, ncol, nnz, row_ptr, cols, vals)),
                    [&](Entry &n) { n.dm = fs_dist_csr_create(D, nrow, ncol, nnz, row_ptr, cols, vals); }, who);
  // the transposed product: row shards of A' on the same devices, built from the same host arrays on first use
  if (need_t) FS_MUST(fs_dist_matrix_build_transpose(e->dm, row_ptr, cols, vals), who);
  return e;
}

EntryP dist_coo_entry(const void *host, int variant, int nrow, int ncol, long nnz, const int *rows, const int *cols,
                      const double *vals, const char *who)
{
  fs_dist_t D = dist_context(who);
  return lookup(host, variant, mix(5, coo_print(nrow, ncol, nnz, rows, cols, vals)), [&](Entry &n) {
    n.dm = variant == kDist ? fs_dist_coo_create(D, nrow, ncol, nnz, rows, cols, vals)
                            : fs_dist_coo_create(D, ncol, nrow, nnz, cols, rows, vals);
  }, who);
}

EntryP dist_blocked_entry(const void *host, int nrow, int ncol, int nblocks, const int *blk_nnz, int **brows, int **bcols,
                          double **bvals, const char *who)
{
  fs_dist_t D = dist_context(who);
  return lookup(host, kDist, mix(5, blocked_print(nrow, ncol, nblocks, blk_nnz, brows, bcols, bvals)), [&](Entry &n) {
    BlockedAsCoo coo(nblocks, blk_nnz, brows, bcols, bvals);
    n.dm = fs_dist_coo_create(D, nrow, ncol, (int64_t)coo.r.size(), coo.r.data(), coo.c.data(), bvals ? coo.v.data() : nullptr);
  }, who);
}

// A and At of bsbm_cg / bsbm_cg2 / bsbm_AtA (cg.h:9-187) as one handle: the pair shares the shards of the two matrices' own entries
EntryP dist_pair_entry(struct BlockedSBM *A, struct BlockedSBM *At, const char *who)
{
  EntryP ea = dist_blocked_entry(A, A->nrow, A->ncol, A->nblocks, A->nnz, A->rows, A->cols, nullptr, who);
  EntryP eat = dist_blocked_entry(At, At->nrow, At->ncol, At->nblocks, At->nnz, At->rows, At->cols, nullptr, who);
  const uint64_t h = mix(mix(mix(6, ea->print), eat->print), (uint64_t)(uintptr_t)At);
  return lookup(A, kDistPair, h, [&](Entry &n) { n.dm = fs_dist_matrix_pair(ea->dm, eat->dm); }, who);
}

// y = M x (transposed: M' x) for k row-major columns on the node's GPUs
void dist_mul(fs_dist_matrix_t dm, bool transposed, double *y, double *x, int k, const char *who)
{
  if (k == 1) FS_MUST(transposed ? fs_dist_spmv_t(dm, y, x) : fs_dist_spmv(dm, y, x), who);
  else FS_MUST(transposed ? fs_dist_spmm_t(dm, y, x, k) : fs_dist_spmm(dm, y, x, k), who);
}

// ---- one-time work of multi-column products ------------------------------------------------------------------
// fs_spmm never builds a copy or waits (include/fastsparse_hip.h): the k-column two-pass copy and the measured choice
// between column sweeps and the row kernel are made by fs_matrix_prepare.  A caller of the reference's bsbm_A_mul_B2(Y, B, X)
// knows nothing of that, so this layer prepares a matrix for k the first time it multiplies with that k -- and, for the
// ks listed in FS_PREPARE_K (e.g. "2,4"), already when it makes the device copy, so that no later call carries the work
// (the reference's harness times csr with _B4 cold, bench_a_mul_b.c:285-290).
const std::vector<int> &prepare_list()
{
  static const std::vector<int> v = [] {
    std::vector<int> q;
    if (const char *list = getenv("FS_PREPARE_K"))
      for (const char *p = list; *p;) {
        const int k = atoi(p);
        if (k >= 2 && k <= 64) q.push_back(k);
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
    return q;
  }();
  return v;
}

void prepare_listed(fs_matrix_t m, bool transposed_too, const char *who)
{
  for (int k : prepare_list()) {
    FS_MUST(fs_matrix_prepare(m, k, 0, nullptr), who);
    if (transposed_too && fs_matrix_has_transpose(m)) FS_MUST(fs_matrix_prepare(m, k, 1, nullptr), who);
  }
}

// ---- per-format uploads -----------------------------------------------------------------------------
EntryP csr_entry(const void *host, int nrow, int ncol, long nnz, const int *row_ptr, const int *cols,
                 const double *vals, bool need_t, const char *who)
{
  EntryP e = lookup(host, kDirect, csr_print(nrow, ncol, nnz, row_ptr, cols, vals), [&](Entry &n) {
    n.m = fs_csr_create(nrow, ncol, nnz, row_ptr, cols, vals, FS_HOST, 0);
    if (n.m) prepare_listed(n.m, false, who);
  }, who);
  if (need_t && !fs_matrix_has_transpose(e->m)) {
    FS_MUST(fs_matrix_build_transpose(e->m, nullptr), who);
    for (int k : prepare_list()) FS_MUST(fs_matrix_prepare(e->m, k, 1, nullptr), who);
  }
  return e;
}

// COO (optionally valued); variant kTransposed uploads (cols, rows) so that each output element
// keeps the entry order of the serial loop it replaces (sparse.h:72-74, dsparse.h:58-60)
EntryP coo_entry(const void *host, int variant, int nrow, int ncol, long nnz, const int *rows, const int *cols,
                 const double *vals, const char *who)
{
  return lookup(host, variant, coo_print(nrow, ncol, nnz, rows, cols, vals), [&](Entry &n) {
    n.m = variant == kDirect ? fs_coo_create(nrow, ncol, nnz, rows, cols, vals, FS_HOST)
                             : fs_coo_create(ncol, nrow, nnz, cols, rows, vals, FS_HOST);
    if (n.m) prepare_listed(n.m, false, who);
  }, who);
}

// row-blocked COO: the per-block arrays are laid end to end and uploaded as one COO; a row lives in
// exactly one block, so its entries keep the order bsbm_A_mul_B (sparse.h:269-271) adds them in
EntryP blocked_entry(const void *host, int nrow, int ncol, int nblocks, const int *blk_nnz, int **brows,
                     int **bcols, double **bvals, const char *who)
{
  return lookup(host, kDirect, blocked_print(nrow, ncol, nblocks, blk_nnz, brows, bcols, bvals), [&](Entry &n) {
    BlockedAsCoo coo(nblocks, blk_nnz, brows, bcols, bvals);
    n.m = fs_coo_create(nrow, ncol, (int64_t)coo.r.size(), coo.r.data(), coo.c.data(), bvals ? coo.v.data() : nullptr, FS_HOST);
    if (n.m) prepare_listed(n.m, false, who);
  }, who);
}

void bcsr_mul_k(double *Y, struct BinaryCSR *A, double *X, int k, const char *who)
{
  if (dist_on()) {
    dist_mul(dist_csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, false, who)->dm, false, Y, X, k, who);
    return;
  }
  EntryP e = csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, false, who);
  if (k == 1) { product(e->m, false, Y, A->nrow, X, A->ncol, who); return; }
  FS_MUST(fs_matrix_prepare(e->m, k, 0, nullptr), who);   // first product with this k: the one-time work; later: nothing
  with_vectors(Y, (size_t)A->nrow * k, X, (size_t)A->ncol * k,
               [&](double *yd, const double *xd) { return fs_spmm(e->m, yd, xd, k, nullptr); }, who);
}

void bsbm_mul_k(double *Y, struct BlockedSBM *B, double *X, int k, const char *who)
{
  if (dist_on()) {
    dist_mul(dist_blocked_entry(B, B->nrow, B->ncol, B->nblocks, B->nnz, B->rows, B->cols, nullptr, who)->dm, false, Y, X, k, who);
    return;
  }
  EntryP e = blocked_entry(B, B->nrow, B->ncol, B->nblocks, B->nnz, B->rows, B->cols, nullptr, who);
  if (k == 1) { product(e->m, false, Y, B->nrow, X, B->ncol, who); return; }
  FS_MUST(fs_matrix_prepare(e->m, k, 0, nullptr), who);
  with_vectors(Y, (size_t)B->nrow * k, X, (size_t)B->ncol * k,
               [&](double *yd, const double *xd) { return fs_spmm(e->m, yd, xd, k, nullptr); }, who);
}

}  // namespace

extern "C" {

void fs_invalidate(const void *host_struct)
{
  std::lock_guard<std::mutex> g(g_table_lock);
  for (int v = 0; v < kVariants; v++) g_table.erase(Key{host_struct, v});   // a product still running on the copy keeps it alive
}

void fs_release_all(void)
{
  {
    std::lock_guard<std::mutex> g(g_table_lock);
    g_table.clear();
  }
  fs::pool_trim(true);   // and the format builders' idle scratch
}

// number of device copies in the side table (tests, diagnostics)
int fs_cache_entries(void)
{
  std::lock_guard<std::mutex> g(g_table_lock);
  return (int)g_table.size();
}

// ---- sparse.h ----------------------------------------------------------------------------------------
void A_mul_B(double *y, struct SparseBinaryMatrix *A, double *x)
{
  if (dist_on()) { dist_mul(dist_coo_entry(A, kDist, A->nrow, A->ncol, A->nnz, A->rows, A->cols, nullptr, "A_mul_B")->dm, false, y, x, 1, "A_mul_B"); return; }
  EntryP e = coo_entry(A, kDirect, A->nrow, A->ncol, A->nnz, A->rows, A->cols, nullptr, "A_mul_B");
  product(e->m, false, y, A->nrow, x, A->ncol, "A_mul_B");
}

void At_mul_B(double *y, struct SparseBinaryMatrix *A, double *x)
{
  if (dist_on()) { dist_mul(dist_coo_entry(A, kDistT, A->nrow, A->ncol, A->nnz, A->rows, A->cols, nullptr, "At_mul_B")->dm, false, y, x, 1, "At_mul_B"); return; }
  EntryP e = coo_entry(A, kTransposed, A->nrow, A->ncol, A->nnz, A->rows, A->cols, nullptr, "At_mul_B");
  product(e->m, false, y, A->ncol, x, A->nrow, "At_mul_B");
}

void bsbm_A_mul_B(double *y, struct BlockedSBM *B, double *x) { bsbm_mul_k(y, B, x, 1, "bsbm_A_mul_B"); }
void bsbm_A_mul_B2(double *y, struct BlockedSBM *B, double *x) { bsbm_mul_k(y, B, x, 2, "bsbm_A_mul_B2"); }
void bsbm_A_mul_B4(double *y, struct BlockedSBM *B, double *x) { bsbm_mul_k(y, B, x, 4, "bsbm_A_mul_B4"); }
void bsbm_A_mul_Bn(double *y, struct BlockedSBM *B, double *x, int ncol) { bsbm_mul_k(y, B, x, ncol, "bsbm_A_mul_Bn"); }

// ---- cg.h --------------------------------------------------------------------------------------------------
void bsbm_AtA(double *y, struct BlockedSBM *A, struct BlockedSBM *At, double *x, double *tmp, double lambda)
{
  (void)tmp;  // host scratch of the CPU version; the intermediate A x stays in HBM here
  if (dist_on()) { FS_MUST(fs_dist_ata(dist_pair_entry(A, At, "bsbm_AtA")->dm, y, x, lambda), "bsbm_AtA"); return; }
  EntryP ea = blocked_entry(A, A->nrow, A->ncol, A->nblocks, A->nnz, A->rows, A->cols, nullptr, "bsbm_AtA");
  EntryP eat = blocked_entry(At, At->nrow, At->ncol, At->nblocks, At->nnz, At->rows, At->cols, nullptr, "bsbm_AtA");
  fs_matrix_t a = ea->m, at = eat->m;
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
  if (dist_on()) {
    fs_dist_matrix_t pair = dist_pair_entry(A, At, who)->dm;
    int it = 0;
    FS_MUST(k == 1 ? fs_dist_cg(pair, x, b, lambda, tol, &it) : fs_dist_cg2(pair, x, b, lambda, tol, &it), who);
    if (out_iter) *out_iter = it;
    return;
  }
  EntryP ea = blocked_entry(A, A->nrow, A->ncol, A->nblocks, A->nnz, A->rows, A->cols, nullptr, who);
  EntryP eat = blocked_entry(At, At->nrow, At->ncol, At->nblocks, At->nnz, At->rows, At->cols, nullptr, who);
  fs_matrix_t a = ea->m, at = eat->m;
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
  if (dist_on()) { dist_mul(dist_coo_entry(A, kDist, A->nrow, A->ncol, A->nnz, A->rows, A->cols, A->vals, "sdm_A_mul_B")->dm, false, y, x, 1, "sdm_A_mul_B"); return; }
  EntryP e = coo_entry(A, kDirect, A->nrow, A->ncol, A->nnz, A->rows, A->cols, A->vals, "sdm_A_mul_B");
  product(e->m, false, y, A->nrow, x, A->ncol, "sdm_A_mul_B");
}

void sdm_At_mul_B(double *y, struct SparseDoubleMatrix *A, double *x)
{
  if (dist_on()) { dist_mul(dist_coo_entry(A, kDistT, A->nrow, A->ncol, A->nnz, A->rows, A->cols, A->vals, "sdm_At_mul_B")->dm, false, y, x, 1, "sdm_At_mul_B"); return; }
  EntryP e = coo_entry(A, kTransposed, A->nrow, A->ncol, A->nnz, A->rows, A->cols, A->vals, "sdm_At_mul_B");
  product(e->m, false, y, A->ncol, x, A->nrow, "sdm_At_mul_B");
}

void bsdm_A_mul_B(double *y, struct BlockedSDM *B, double *x)
{
  if (dist_on()) { dist_mul(dist_blocked_entry(B, B->nrow, B->ncol, B->nblocks, B->nnz, B->rows, B->cols, B->vals, "bsdm_A_mul_B")->dm, false, y, x, 1, "bsdm_A_mul_B"); return; }
  EntryP e = blocked_entry(B, B->nrow, B->ncol, B->nblocks, B->nnz, B->rows, B->cols, B->vals, "bsdm_A_mul_B");
  product(e->m, false, y, B->nrow, x, B->ncol, "bsdm_A_mul_B");
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

void bcsr_A_mul_B(double *y, struct BinaryCSR *A, double *x)
{
  bcsr_mul_k(y, A, x, 1, "bcsr_A_mul_B");
}
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
  if (dist_on()) {
    FS_MUST(fs_dist_ata(dist_csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, true, "bcsr_AA_mul_B")->dm, y, x, 0.0), "bcsr_AA_mul_B");
    return;
  }
  EntryP e = csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, true, "bcsr_AA_mul_B");
  fs_matrix_t m = e->m;
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
  if (dist_on()) { dist_mul(dist_csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, true, "bcsr_At_mul_B")->dm, true, y, x, 1, "bcsr_At_mul_B"); return; }
  EntryP e = csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, nullptr, true, "bcsr_At_mul_B");
  fs_matrix_t m = e->m;
  product(m, true, y, A->ncol, x, A->nrow, "bcsr_At_mul_B");
}

void csr_A_mul_B(double *y, struct CSR *A, double *x)
{
  if (dist_on()) { dist_mul(dist_csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, false, "csr_A_mul_B")->dm, false, y, x, 1, "csr_A_mul_B"); return; }
  EntryP e = csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, false, "csr_A_mul_B");
  fs_matrix_t m = e->m;
  product(m, false, y, A->nrow, x, A->ncol, "csr_A_mul_B");
}

void csr_At_mul_B(double *y, struct CSR *A, double *x)
{
  if (dist_on()) { dist_mul(dist_csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, true, "csr_At_mul_B")->dm, true, y, x, 1, "csr_At_mul_B"); return; }
  EntryP e = csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, true, "csr_At_mul_B");
  fs_matrix_t m = e->m;
  product(m, true, y, A->ncol, x, A->nrow, "csr_At_mul_B");
}

void csr_A_mul_Bn(double *Y, struct CSR *A, double *X, const int ncol)
{
  if (dist_on()) { dist_mul(dist_csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, false, "csr_A_mul_Bn")->dm, false, Y, X, ncol, "csr_A_mul_Bn"); return; }
  EntryP e = csr_entry(A, A->nrow, A->ncol, A->nnz, A->row_ptr, A->cols, A->vals, false, "csr_A_mul_Bn");
  fs_matrix_t m = e->m;
  if (ncol > 1) FS_MUST(fs_matrix_prepare(m, ncol, 0, nullptr), "csr_A_mul_Bn");
  with_vectors(Y, (size_t)A->nrow * ncol, X, (size_t)A->ncol * ncol,
               [&](double *yd, const double *xd) { return fs_spmm(m, yd, xd, ncol, nullptr); }, "csr_A_mul_Bn");
}

// ---- cbcsr.h -------------------------------------------------------------------------------------------
void cbcsr_A_mul_B(double *y, struct ColBinaryCSR *A, double *x)
{
  const int64_t ncell = (int64_t)A->nblocks * A->nrow;
  const bool full = hash_in_full(4 * ((int64_t)A->nnz + ncell + 1));
  uint64_t h = mix(mix(mix(4, (uint64_t)A->nrow), (uint64_t)A->ncol), (uint64_t)A->nnz);
  h = mix(mix(mix(h, (uint64_t)A->colblocksize), (uint64_t)(uintptr_t)A->row_ptr), (uint64_t)(uintptr_t)A->cols);
  h = print_ints(print_ints(h, A->row_ptr, ncell + 1, full), A->cols, A->nnz, full);
  if (dist_on() && !fs::options().strict_order) {
    // across the GPUs: the same entries as a plain pattern-only CSR whose rows hold their cells' entries block by block (the order the
    // single-GPU kernels add large column-blocked matrices in; cbcsr.h:88-103 itself adds per-thread partial vectors in schedule order).
    // A row's terms are then ONE running sum, where the serial reference adds a partial sum per column block to y: the same bits for
    // integer data, within the bar otherwise -- so under strict_order (the reference's bits for arbitrary x) the call stays on ONE GPU,
    // whose kernels keep the per-block association (tools/fuzz_dropin.py under FS_STRICT_ORDER=1 FASTSPARSE_NGPU=3 found the difference).
    fs_dist_t D = dist_context("cbcsr_A_mul_B");
    EntryP e = lookup(A, kDist, mix(5, h), [&](Entry &n) {
      std::vector<int> rp((size_t)A->nrow + 1, 0), cc((size_t)A->nnz);
      for (int r = 0; r < A->nrow; ++r) {
        int64_t o = rp[(size_t)r];
        for (int b = 0; b < A->nblocks; ++b) {
          const int64_t cell = (int64_t)b * A->nrow + r;
          const int lo = A->row_ptr[cell], hi = A->row_ptr[cell + 1];
          if (hi > lo) memcpy(cc.data() + o, A->cols + lo, sizeof(int) * (size_t)(hi - lo));
          o += hi - lo;
        }
        rp[(size_t)r + 1] = (int)o;
      }
      n.dm = fs_dist_csr_create(D, A->nrow, A->ncol, A->nnz, rp.data(), cc.data(), nullptr);
    }, "cbcsr_A_mul_B");
    dist_mul(e->dm, false, y, x, 1, "cbcsr_A_mul_B");
    return;
  }
  EntryP e = lookup(A, kDirect, h, [&](Entry &n) {
    n.cb = fs_cbcsr_create(A->nrow, A->ncol, A->nblocks, A->colblocksize, A->row_ptr, A->cols, FS_HOST);
  }, "cbcsr_A_mul_B");
  with_vectors(y, A->nrow, x, A->ncol, [&](double *yd, const double *xd) { return fs_cbcsr_spmv(e->cb, yd, xd, nullptr); },
               "cbcsr_A_mul_B");
}

}  // extern "C"
