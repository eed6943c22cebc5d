// fs_abi.hip -- the device-resident layer of the C-ABI (include/fastsparse_hip.h, part 2).
#include <dlfcn.h>
#include <string.h>

#include <atomic>
#include <chrono>

#include "fs_common.h"

namespace fs {

static thread_local std::string g_err;
thread_local int tl_fixed_order = 0;
thread_local int tl_keep_csr = 0;
static std::atomic<unsigned> g_option_epoch{0};
unsigned option_epoch() { return g_option_epoch.load(std::memory_order_relaxed); }

void set_error(const std::string &msg) { g_err = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
  g_err = buf;
  return FS_ERR_HIP;
}

Options &options()
{
  // FS_TILED_FLAGS / FS_BIN_ROWS / FS_TILE_COLS: initial values of the options "tiled_flags" / "bin_rows" / "tile_cols" (A/B runs of bench.py, which sets
  // no tuning switches)
  static Options o = [] {
    Options q;
    if (const char *v = getenv("FS_TILED_FLAGS")) q.tiled_flags = atoi(v);
    if (const char *v = getenv("FS_BIN_ROWS")) q.bin_rows = atoi(v);
    if (const char *v = getenv("FS_BIN_FLAGS")) q.bin_flags = atoi(v);          // (A/B runs: 64 keeps two-byte row ids in the two-pass copy)
    if (const char *v = getenv("FS_TILE_COLS")) q.tile_cols = atoi(v);
    if (const char *v = getenv("FS_LONG_ROWS")) q.long_rows = atoi(v);
    if (const char *v = getenv("FS_LONG_GEOMETRY")) q.long_geometry = atoi(v);
    if (const char *v = getenv("FS_REPRODUCIBLE")) q.reproducible = atoi(v);     // drop-in callers: fixed-order sums without a code change
    if (const char *v = getenv("FS_STRICT_ORDER")) q.strict_order = atoi(v);     //                  the reference's own order of additions
    if (const char *v = getenv("FS_SPMM_WIDE")) q.spmm_wide = atoi(v);
    if (const char *v = getenv("FS_CG_FIXED_ORDER")) q.cg_fixed_order = atoi(v);
    if (const char *v = getenv("FS_DIST_CG_SCHEME")) q.dist_cg_scheme = atoi(v);
    if (const char *v = getenv("FS_RELEASE_CSR")) q.release_csr = atoi(v);
    return q;
  }();
  return o;
}

// ---- roctx (see fs_common.h) ---------------------------------------------------------------------------------
namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  bool on = false;
};
Roctx &roctx()
{
  static Roctx r = [] {
    Roctx q;
    const char *force = getenv("FS_ROCTX");
    const char *pre = getenv("LD_PRELOAD");
    const bool under_profiler = (pre && strstr(pre, "rocprofiler-sdk")) || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") ||
                                getenv("ROCPROF_OUTPUT_PATH") || getenv("ROCP_TOOL_LIBRARIES");
    if (force ? *force == '0' : !under_profiler) return q;
    void *lib = nullptr;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) return q;
    q.push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
    q.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    q.on = q.push && q.pop;
    return q;
  }();
  return r;
}
}  // namespace
bool roctx_enabled() { return roctx().on; }
void roctx_push(const char *name) { (void)roctx().push(name); }
void roctx_pop() { (void)roctx().pop(); }

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
static int to_device(T **dst, const T *src, size_t n, int space)
{
  FS_HIP(hipMalloc(dst, sizeof(T) * (n ? n : 1)));
  if (n) FS_HIP(hipMemcpy(*dst, src, sizeof(T) * n, space == FS_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  return FS_OK;
}

}  // namespace fs

using fs::set_error;

extern "C" {

const char *fs_version(void) { return "fastsparse-hip 0.1 (gfx950)"; }
const char *fs_last_error(void) { return fs::g_err.c_str(); }

int fs_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fs_set_device(int device)
{
  FS_HIP(hipSetDevice(device));
  return FS_OK;
}

void *fs_device_alloc(int64_t bytes)
{
  void *p = nullptr;
  if (bytes < 0 || hipMalloc(&p, (size_t)(bytes ? bytes : 1)) != hipSuccess) { set_error("fs_device_alloc failed"); return nullptr; }
  return p;
}

void fs_device_free(void *p) { if (p) (void)hipFree(p); }

int fs_copy_to_device(void *dst_dev, const void *src_host, int64_t bytes)
{
  FS_HIP(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
  return FS_OK;
}

int fs_copy_to_host(void *dst_host, const void *src_dev, int64_t bytes)
{
  FS_HIP(hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
  return FS_OK;
}

int fs_device_synchronize(void)
{
  FS_HIP(hipDeviceSynchronize());
  return FS_OK;
}

// option "release_csr": the plain arrays go once a copy is kept -- unless the products the options select NOW would read them (strict_order,
// spmv_kernel 1-3, or `reproducible` with a kept LDS-staged copy that cannot be ordered): releasing then would turn every product of the
// handle into an FS_ERR_RELEASED (tools/fuzz_parity.py under FS_REPRODUCIBLE=1 found exactly that); fs_matrix_release_csr stays unconditional
static void auto_release(fs::DeviceCsr &A)
{
  if (!fs::options().release_csr || fs::tl_keep_csr != 0) return;
  if (fs::spmv_choice(A, fs::options()) < 6) return;
  (void)fs::release_plain_csr(A);
}

int fs_set_option(const char *name, int value)
{
  if (!name) { set_error("fs_set_option: NULL name"); return FS_ERR_ARG; }
  fs::g_option_epoch.fetch_add(1, std::memory_order_relaxed);
  if (!strcmp(name, "release_csr")) { fs::options().release_csr = value; return FS_OK; }
  if (!strcmp(name, "strict_order")) { fs::options().strict_order = value; return FS_OK; }
  if (!strcmp(name, "spmv_kernel")) { fs::options().spmv_kernel = value; return FS_OK; }
  if (!strcmp(name, "tiling")) { fs::options().tiling = value; return FS_OK; }
  if (!strcmp(name, "tile_rows")) { fs::options().tile_rows = value; return FS_OK; }
  if (!strcmp(name, "tile_cols")) { fs::options().tile_cols = value; return FS_OK; }
  if (!strcmp(name, "tile_split")) { fs::options().tile_split = value; return FS_OK; }
  if (!strcmp(name, "tiled_flags")) { fs::options().tiled_flags = value; return FS_OK; }
  if (!strcmp(name, "reproducible")) { fs::options().reproducible = value; return FS_OK; }
  if (!strcmp(name, "bin_wgs")) { fs::options().bin_wgs = value; return FS_OK; }
  if (!strcmp(name, "bin_flags")) { fs::options().bin_flags = value; return FS_OK; }
  if (!strcmp(name, "bin_rows")) { fs::options().bin_rows = value; return FS_OK; }
  if (!strcmp(name, "ldsx")) { fs::options().ldsx = value; return FS_OK; }
  if (!strcmp(name, "binning")) { fs::options().binning = value; return FS_OK; }
  if (!strcmp(name, "long_rows")) { fs::options().long_rows = value; return FS_OK; }
  if (!strcmp(name, "long_min_len")) { fs::options().long_min_len = value; return FS_OK; }
  if (!strcmp(name, "long_geometry")) { fs::options().long_geometry = value; return FS_OK; }
  if (!strcmp(name, "spmm_kernel")) { fs::options().spmm_kernel = value; return FS_OK; }
  if (!strcmp(name, "spmm_wide")) { fs::options().spmm_wide = value; return FS_OK; }
  if (!strcmp(name, "ata_kernel")) { fs::options().ata_kernel = value; return FS_OK; }
  if (!strcmp(name, "device_build")) { fs::options().device_build = value; return FS_OK; }
  if (!strcmp(name, "cg_fixed_order")) { fs::options().cg_fixed_order = value; return FS_OK; }
  if (!strcmp(name, "dist_cg_scheme")) { fs::options().dist_cg_scheme = value; return FS_OK; }
  set_error(std::string("fs_set_option: unknown option ") + name);
  return FS_ERR_ARG;
}

int fs_get_option(const char *name)
{
  if (name && !strcmp(name, "strict_order")) return fs::options().strict_order;
  if (name && !strcmp(name, "spmv_kernel")) return fs::options().spmv_kernel;
  if (name && !strcmp(name, "tiling")) return fs::options().tiling;
  if (name && !strcmp(name, "tile_rows")) return fs::options().tile_rows;
  if (name && !strcmp(name, "tile_cols")) return fs::options().tile_cols;
  if (name && !strcmp(name, "tile_split")) return fs::options().tile_split;
  if (name && !strcmp(name, "binning")) return fs::options().binning;
  if (name && !strcmp(name, "long_rows")) return fs::options().long_rows;
  if (name && !strcmp(name, "ldsx")) return fs::options().ldsx;
  if (name && !strcmp(name, "reproducible")) return fs::options().reproducible;
  if (name && !strcmp(name, "spmm_kernel")) return fs::options().spmm_kernel;
  if (name && !strcmp(name, "spmm_wide")) return fs::options().spmm_wide;
  if (name && !strcmp(name, "ata_kernel")) return fs::options().ata_kernel;
  if (name && !strcmp(name, "device_build")) return fs::options().device_build;
  if (name && !strcmp(name, "cg_fixed_order")) return fs::options().cg_fixed_order;
  if (name && !strcmp(name, "dist_cg_scheme")) return fs::options().dist_cg_scheme;
  if (name && !strcmp(name, "release_csr")) return fs::options().release_csr;
  return FS_ERR_ARG;
}

fs_matrix_t fs_csr_create(int nrow, int ncol, int64_t nnz, const int *row_ptr, const int *cols, const double *vals,
                          int space, int borrow)
{
  FS_RANGE("fs_csr_create");
  if (nrow < 0 || ncol < 0 || nnz < 0 || !row_ptr || (nnz > 0 && !cols)) {
    set_error("fs_csr_create: bad argument");
    return nullptr;
  }
  fs_matrix_t M = new fs_matrix_s();
  if (hipGetDevice(&M->device) != hipSuccess) { set_error("fs_csr_create: no HIP device"); delete M; return nullptr; }
  fs::DeviceCsr &A = M->a;
  A.nrow = nrow; A.ncol = ncol; A.nnz = nnz;
  int rc = FS_OK;
  const auto t0 = std::chrono::steady_clock::now();
  if (space == FS_DEVICE && borrow && fs::aligned16(cols) && (!vals || fs::aligned16(vals))) {
    A.owns = false;
    A.row_ptr = const_cast<int *>(row_ptr);
    A.cols = const_cast<int *>(cols);
    A.vals = const_cast<double *>(vals);
  } else {
    A.owns = true;
    rc = fs::to_device(&A.row_ptr, row_ptr, (size_t)nrow + 1, space);
    if (!rc) rc = fs::to_device(&A.cols, cols, (size_t)nnz, space);
    if (!rc && vals) rc = fs::to_device(&A.vals, vals, (size_t)nnz, space);
  }
  if (!rc) rc = fs::validate_indices(nrow, ncol, nnz, A.row_ptr, nullptr, A.cols, nullptr);
  const float up_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();   // (validation ends in a synchronisation)
  if (!rc) rc = fs::build_schedule(A, nullptr);
  A.build_ms[0] = up_ms;
  fs::pool_trim();
  if (rc) { fs::free_csr(A); delete M; return nullptr; }
  auto_release(A);
  return M;
}

fs_matrix_t fs_coo_create(int nrow, int ncol, int64_t nnz, const int *rows, const int *cols, const double *vals,
                          int space)
{
  FS_RANGE("fs_coo_create");
  if (nrow < 0 || ncol < 0 || nnz < 0 || (nnz > 0 && (!rows || !cols))) {
    set_error("fs_coo_create: bad argument");
    return nullptr;
  }
  fs_matrix_t M = new fs_matrix_s();
  if (hipGetDevice(&M->device) != hipSuccess) { set_error("fs_coo_create: no HIP device"); delete M; return nullptr; }
  int *r = nullptr, *c = nullptr;
  double *v = nullptr;
  int rc = FS_OK;
  const auto t0 = std::chrono::steady_clock::now();
  if (space == FS_DEVICE) {
    r = const_cast<int *>(rows); c = const_cast<int *>(cols); v = const_cast<double *>(vals);
  } else {
    rc = fs::to_device(&r, rows, (size_t)nnz, FS_HOST);
    if (!rc) rc = fs::to_device(&c, cols, (size_t)nnz, FS_HOST);
    if (!rc && vals) rc = fs::to_device(&v, vals, (size_t)nnz, FS_HOST);
  }
  if (!rc) rc = fs::validate_indices(nrow, ncol, nnz, nullptr, r, c, nullptr);
  const float up_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (!rc) rc = fs::coo_to_csr_device(M->a, nrow, ncol, nnz, r, c, v, nullptr);
  M->a.build_ms[0] = up_ms;
  if (space != FS_DEVICE) {
    if (r) (void)hipFree(r);
    if (c) (void)hipFree(c);
    if (v) (void)hipFree(v);
  }
  fs::pool_trim();
  if (rc) { fs::free_csr(M->a); delete M; return nullptr; }
  auto_release(M->a);
  return M;
}

void fs_matrix_destroy(fs_matrix_t A)
{
  if (!A) return;
  fs::free_csr(A->a);
  if (A->has_t) fs::free_csr(A->at);
  fs::free_host_pipe(A->pipe);
  delete A;
}

int fs_matrix_build_transpose(fs_matrix_t A, fs_stream_t stream)
{
  FS_RANGE("fs_matrix_build_transpose");
  if (!A) { set_error("fs_matrix_build_transpose: NULL handle"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(A->lock);
  if (A->has_t) return FS_OK;
  const int rc = fs::transpose_device(A->a, A->at, (hipStream_t)stream);
  fs::pool_trim();
  if (rc) { fs::free_csr(A->at); return rc; }
  A->has_t = true;
  auto_release(A->at);
  return FS_OK;
}

int fs_matrix_has_transpose(fs_matrix_t A) { return A && A->has_t; }

int fs_matrix_spmv_kernel(fs_matrix_t A, int transposed)
{
  if (!A || (transposed && !A->has_t)) return FS_ERR_ARG;
  return fs::spmv_choice(transposed ? A->at : A->a, fs::options());
}

int fs_matrix_candidate_ms(fs_matrix_t A, int transposed, float *ms4)
{
  if (!A || !ms4 || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  for (int i = 0; i < 4; ++i) ms4[i] = a.candidate_ms[i];
  return FS_OK;
}

int fs_matrix_build_ms(fs_matrix_t A, int transposed, float *ms8)
{
  if (!A || !ms8 || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  for (int i = 0; i < 8; ++i) ms8[i] = a.build_ms[i];
  return FS_OK;
}

int fs_matrix_nrow(fs_matrix_t A) { return A ? A->a.nrow : FS_ERR_ARG; }
int fs_matrix_ncol(fs_matrix_t A) { return A ? A->a.ncol : FS_ERR_ARG; }
int64_t fs_matrix_nnz(fs_matrix_t A) { return A ? A->a.nnz : FS_ERR_ARG; }

int64_t fs_matrix_algorithmic_bytes(fs_matrix_t A, int k)
{
  if (!A || k < 1) return FS_ERR_ARG;
  const fs::DeviceCsr &a = A->a;
  return (a.has_vals() ? 12 : 4) * a.nnz + 4 * ((int64_t)a.nrow + 1) + 8ll * k * a.nrow + 8ll * k * a.ncol;
}

int fs_matrix_download(fs_matrix_t A, int transposed, int *row_ptr, int *cols, double *vals)
{
  if (!A) { set_error("fs_matrix_download: NULL handle"); return FS_ERR_ARG; }
  if (transposed && !A->has_t) { set_error("fs_matrix_download: transpose not built"); return FS_ERR_NO_TRANSPOSE; }
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (int rc = fs::need_plain_csr(a, "fs_matrix_download")) return rc;
  if (row_ptr) FS_HIP(hipMemcpy(row_ptr, a.row_ptr, sizeof(int) * ((size_t)a.nrow + 1), hipMemcpyDeviceToHost));
  if (cols && a.nnz) FS_HIP(hipMemcpy(cols, a.cols, sizeof(int) * (size_t)a.nnz, hipMemcpyDeviceToHost));
  if (vals && a.vals && a.nnz) FS_HIP(hipMemcpy(vals, a.vals, sizeof(double) * (size_t)a.nnz, hipMemcpyDeviceToHost));
  return FS_OK;
}

static int check_mul(fs_matrix_t A, const void *y, const void *x, const char *who)
{
  if (!A || !y || !x) { set_error(std::string(who) + ": NULL argument"); return FS_ERR_ARG; }
  return FS_OK;
}

int fs_spmv(fs_matrix_t A, double *y, const double *x, fs_stream_t stream)
{
  FS_RANGE("fs_spmv");
  if (int rc = check_mul(A, y, x, "fs_spmv")) return rc;
  std::lock_guard<std::mutex> g(A->lock);
  A->last_stream = (hipStream_t)stream; A->last_async = true;
  return fs::launch_spmv(A->a, y, x, (hipStream_t)stream);
}

// y = A x (transposed != 0: A' x) in nparts parts: see include/fastsparse_hip.h and launch_spmv_part
int fs_spmv_part_rows(fs_matrix_t A, int transposed, int nparts, int *rows)
{
  if (!A || !rows || nparts < 1 || nparts > 64) { set_error("fs_spmv_part_rows: bad argument (1 <= nparts <= 64)"); return FS_ERR_ARG; }
  if (transposed && !A->has_t) { set_error("fs_spmv_part_rows: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(A->lock);
  const int *r = nullptr;
  if (int rc = fs::spmv_part_bounds(transposed ? A->at : A->a, nparts, &r, nullptr)) return rc;
  for (int p = 0; p <= nparts; ++p) rows[p] = r[p];
  return FS_OK;
}

int fs_spmv_part(fs_matrix_t A, int transposed, double *y, const double *x, int part, int nparts, fs_stream_t stream)
{
  FS_RANGE("fs_spmv_part");
  if (int rc = check_mul(A, y, x, "fs_spmv_part")) return rc;
  if (nparts < 1 || nparts > 64 || part < 0 || part >= nparts) { set_error("fs_spmv_part: bad part"); return FS_ERR_ARG; }
  if (transposed && !A->has_t) { set_error("fs_spmv_part: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(A->lock);
  A->last_stream = (hipStream_t)stream; A->last_async = true;
  return fs::launch_spmv_part(transposed ? A->at : A->a, y, x, part, nparts, (hipStream_t)stream);
}

int fs_spmm_part_rows(fs_matrix_t A, int transposed, int k, int nparts, int *rows)
{
  if (!A || !rows || k < 1 || nparts < 1 || nparts > 64) { set_error("fs_spmm_part_rows: bad argument"); return FS_ERR_ARG; }
  if (k == 1) return fs_spmv_part_rows(A, transposed, nparts, rows);
  if (transposed && !A->has_t) { set_error("fs_spmm_part_rows: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(A->lock);
  fs::DeviceCsr &a = transposed ? A->at : A->a;
  const int *r = nullptr;
  if (k == 2 || k == 4) {
    if (int rc = fs::spmm_part_bounds(a, k, nparts, &r, nullptr, nullptr)) return rc;
    for (int p = 0; p <= nparts; ++p) rows[p] = r[p];
  } else {
    rows[0] = 0;
    for (int p = 1; p <= nparts; ++p) rows[p] = a.nrow;
  }
  return FS_OK;
}

int fs_spmm_part(fs_matrix_t A, int transposed, double *Y, const double *X, int k, int part, int nparts, fs_stream_t stream)
{
  if (k == 1) return fs_spmv_part(A, transposed, Y, X, part, nparts, stream);
  FS_RANGE("fs_spmm_part");
  if (int rc = check_mul(A, Y, X, "fs_spmm_part")) return rc;
  if (k < 1 || nparts < 1 || nparts > 64 || part < 0 || part >= nparts) { set_error("fs_spmm_part: bad argument"); return FS_ERR_ARG; }
  if (transposed && !A->has_t) { set_error("fs_spmm_part: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(A->lock);
  A->last_stream = (hipStream_t)stream; A->last_async = true;
  return fs::launch_spmm_part(transposed ? A->at : A->a, Y, X, k, part, nparts, (hipStream_t)stream);
}

int fs_copy_segments(int nseg, const int64_t *table_dev, int64_t max_count, const double *src, double *dst, fs_stream_t stream)
{
  if (nseg < 0 || (nseg > 0 && (!table_dev || !src || !dst))) { set_error("fs_copy_segments: bad argument"); return FS_ERR_ARG; }
  return fs::launch_copy_segments(nseg, table_dev, max_count, src, dst, (hipStream_t)stream);
}

// ---- diagnostics (not part of include/fastsparse_hip.h; used by tools/trace_tiled.py and the tests) --------
int fs_debug_last_host_path(void) { return fs::last_host_path(); }

// rows taken out of the two-pass copy (LongRows) and their entries incl. padding; 0 / 0 when the copy has none
int fs_debug_long_rows(fs_matrix_t A, int transposed, int64_t *out2)
{
  if (!A || !out2 || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  const fs::LongRows *L = (a.binned && a.binned->built) ? a.binned->lr : nullptr;
  out2[0] = L ? L->nlong : 0;
  out2[1] = L ? L->n : 0;
  return FS_OK;
}

// 1 when the LDS-staged copy of A (transposed != 0: of A') can give fixed-order sums (TiledCsr::orderable), 0 when not, < 0 without one
int fs_debug_ldsx_orderable(fs_matrix_t A, int transposed)
{
  if (!A || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (!a.tiledx || !a.tiledx->built) return FS_ERR_ARG;
  return a.tiledx->orderable ? 1 : 0;
}

// 1 when a product on A (transposed != 0: on A') inside a solver's fixed-order scope (cg_fixed_order) really adds in a fixed order,
// 0 when the kept LDS-staged copy is not orderable and keeps adding in arrival order (the solve is then correct to rounding but
// not bit-identical from run to run; option "reproducible" = 1 forces a fixed-order kernel at the price of speed)
int fs_debug_fixed_order_honoured(fs_matrix_t A, int transposed)
{
  if (!A || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  fs::Options o = fs::options();
  if (fs::spmv_choice(a, o) != 8) return 1;
  return a.tiledx->orderable ? 1 : 0;
}

// chunks of the LDS-staged copy that ever gave up waiting for their turn under fixed-order sums and added out of turn (a wrong order
// of additions, not a wrong sum): 0 on a healthy run; < 0 without such a copy
int fs_debug_ldsx_ticket_giveups(fs_matrix_t A, int transposed)
{
  if (!A || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (!a.tiledx || !a.tiledx->built || !a.tiledx->ticket) return FS_ERR_ARG;
  int n = 0;
  FS_HIP(hipDeviceSynchronize());
  FS_HIP(hipMemcpy(&n, a.tiledx->ticket - 1, sizeof(int), hipMemcpyDeviceToHost));
  return n;
}

// the two-pass copy of A (transposed != 0: of A'): device addresses of lcol, vals, gdst, lrow, prod, then n (padded entries), B, P
// (tools/placement_probe.py: identical copies run at different speeds depending on where their arrays land)
int fs_debug_two_pass_layout(fs_matrix_t A, int transposed, unsigned long long *out8)
{
  if (!A || !out8 || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (!a.binned || !a.binned->built) { set_error("no two-pass copy"); return FS_ERR_ARG; }
  const fs::BinnedCsr &N = *a.binned;
  out8[0] = (unsigned long long)(uintptr_t)N.lcol; out8[1] = (unsigned long long)(uintptr_t)N.vals;
  out8[2] = (unsigned long long)(uintptr_t)N.gdst; out8[3] = (unsigned long long)(uintptr_t)(N.lrow ? (const void *)N.lrow : (const void *)N.lrow8);
  out8[4] = (unsigned long long)(uintptr_t)N.prod; out8[5] = (unsigned long long)N.n; out8[6] = (unsigned long long)N.B;
  out8[7] = (unsigned long long)N.P;
  return FS_OK;
}

// the row ids of the two-pass copy: -1 two bytes per entry (or no such copy); >= 0 one byte per entry (BinnedCsr::lrow8), the value
// = the dummy entries that walk steps above 255 (diagnostics, not in include/fastsparse_hip.h)
long long fs_debug_two_pass_rows8(fs_matrix_t A, int transposed)
{
  if (!A || (transposed && !A->has_t)) return -1;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (!a.binned || !a.binned->built || !a.binned->lrow8) return -1;
  return (long long)a.binned->dummies;
}

// one pass of the two-pass pair alone (which = 1: pass 1, x -> the product stream; 2: pass 2, the product stream -> y)
int fs_debug_two_pass_run(fs_matrix_t A, int transposed, int which, double *y, const double *x, fs_stream_t stream)
{
  if (!A || (transposed && !A->has_t)) return FS_ERR_ARG;
  const fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (!a.binned || !a.binned->built || a.binned->split || a.binned->lr) { set_error("no plain two-pass copy"); return FS_ERR_ARG; }
  const fs::BinnedCsr &N = *a.binned;
  if (which == 1) return fs::launch_expand_groups(a, x, 0u, (unsigned)(N.n >> fs::kBinGroupLog), N.nwg1, (hipStream_t)stream);
  return fs::launch_reduce_panels(a, y, 0, N.P, (hipStream_t)stream);
}

#ifdef FS_LAB
// one array of the two-pass copy moved to a fresh allocation (which: 0 lcol, 1 vals, 2 gdst, 3 lrow, 4 prod), the old block
// freed only afterwards so that the new one lands elsewhere -- tools/placement_probe.py
int fs_debug_two_pass_realloc(fs_matrix_t A, int transposed, int which_and_flags)
{
  if (which_and_flags & 32) {     // all five arrays into ONE fresh block (the old blocks and earlier arenas are dropped, not freed)
    if (!A || (transposed && !A->has_t)) return FS_ERR_ARG;
    fs::DeviceCsr &a0 = transposed ? A->at : A->a;
    if (!a0.binned || !a0.binned->built) return FS_ERR_ARG;
    fs::BinnedCsr &N0 = *a0.binned;
    void **slots[5] = {(void **)&N0.lcol, (void **)&N0.vals, (void **)&N0.gdst, (void **)&N0.lrow, (void **)&N0.prod};
    const size_t sizes[5] = {(size_t)N0.n * 2, N0.vals ? (size_t)N0.n * 8 : 0, ((size_t)N0.n >> fs::kBinGroupLog) * 4, N0.lrow ? (size_t)N0.n * 2 : 0,
                             (size_t)N0.n * 8 * (size_t)N0.kw};
    const int order_a[5] = {4, 1, 0, 3, 2}, order_b[5] = {1, 4, 3, 0, 2};
    const int *order = (which_and_flags & 64) ? order_b : order_a;
    const size_t al = (size_t)2 << 20;
    size_t total = 0;
    for (int i = 0; i < 5; ++i) total += (sizes[i] + al - 1) / al * al;
    char *arena = nullptr;
    FS_HIP(hipDeviceSynchronize());
    FS_HIP(hipMalloc(&arena, total));
    size_t off = 0;
    for (int j = 0; j < 5; ++j) {
      const int i = order[j];
      if (!sizes[i]) continue;
      FS_HIP(hipMemcpy(arena + off, *slots[i], sizes[i], hipMemcpyDeviceToDevice));
      *slots[i] = arena + off;
      off += (sizes[i] + al - 1) / al * al;
    }
    return FS_OK;
  }
  const int which = which_and_flags & 15;
  const bool leak_old = (which_and_flags & 16) != 0;     // the probe keeps the old block allocated (distinct placements) and drops it
  if (!A || (transposed && !A->has_t)) return FS_ERR_ARG;
  fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (!a.binned || !a.binned->built) { set_error("no two-pass copy"); return FS_ERR_ARG; }
  fs::BinnedCsr &N = *a.binned;
  void **slot = which == 0 ? (void **)&N.lcol : which == 1 ? (void **)&N.vals : which == 2 ? (void **)&N.gdst : which == 3 ? (void **)&N.lrow : (void **)&N.prod;
  const size_t bytes = which == 0 || which == 3 ? (size_t)N.n * 2 : which == 2 ? ((size_t)N.n >> fs::kBinGroupLog) * 4 : (size_t)N.n * 8 * (which == 4 ? (size_t)N.kw : 1);
  if (!*slot) return FS_OK;
  void *fresh = nullptr;
  FS_HIP(hipDeviceSynchronize());
  FS_HIP(hipMalloc(&fresh, bytes));
  FS_HIP(hipMemcpy(fresh, *slot, bytes, hipMemcpyDeviceToDevice));
  if (!leak_old) FS_HIP(hipFree(*slot));
  *slot = fresh;
  return FS_OK;
}
#endif

int fs_debug_tiled_geometry(fs_matrix_t A, int *out6)
{
  const bool hx = A && A->a.tiledx && A->a.tiledx->built;     // the LDS-staged copy when that is the one built
  if (!A || !(hx || (A->a.tiled && A->a.tiled->built))) { set_error("no tiled copy"); return FS_ERR_ARG; }
  const fs::TiledCsr &T = hx ? *A->a.tiledx : *A->a.tiled;
  out6[0] = T.R; out6[1] = T.W; out6[2] = T.P; out6[3] = T.J; out6[4] = T.nitems; out6[5] = T.lcol_bits;
  return FS_OK;
}

#if defined(FS_LAB) && defined(FS_DMA_TRACE) && FS_DMA_TRACE
// instrumented builds only (-DFS_LAB -DFS_DMA_TRACE=1|2, tools/dma_phase_trace.py): clock sums of wave 0 of every workgroup of the DMA kernel
int fs_debug_dma_trace(unsigned long long *out8, int reset) { return fs::debug_dma_trace(out8, reset); }
#endif

int fs_debug_tiled_trace(fs_matrix_t A, double *y, const double *x, long long *times_host, int *xcc_host,
                         int *items_host, int *item_ptr_host)
{
  if (!A || !A->a.tiled || !A->a.tiled->built) { set_error("no tiled copy"); return FS_ERR_ARG; }
  const fs::TiledCsr &T = *A->a.tiled;
  long long *td = nullptr;
  int *xd = nullptr;
  FS_HIP(hipMalloc(&td, sizeof(long long) * (size_t)(T.nitems + 1)));
  FS_HIP(hipMalloc(&xd, sizeof(int) * (size_t)T.P));
  FS_HIP(hipMemset(td, 0, sizeof(long long) * (size_t)(T.nitems + 1)));
  for (int rep = 0; rep < 3; ++rep)  // the last of three back-to-back launches is the one kept
    if (int rc = fs::launch_spmv_tiled_trace(A->a, y, x, td, xd, nullptr)) return rc;
  FS_HIP(hipDeviceSynchronize());
  FS_HIP(hipMemcpy(times_host, td, sizeof(long long) * (size_t)T.nitems, hipMemcpyDeviceToHost));
  FS_HIP(hipMemcpy(xcc_host, xd, sizeof(int) * (size_t)T.P, hipMemcpyDeviceToHost));
  FS_HIP(hipMemcpy(items_host, T.items, sizeof(int) * 4 * (size_t)T.nitems, hipMemcpyDeviceToHost));
  FS_HIP(hipMemcpy(item_ptr_host, T.item_ptr, sizeof(int) * (size_t)(T.P + 1), hipMemcpyDeviceToHost));
  FS_HIP(hipFree(td));
  FS_HIP(hipFree(xd));
  return FS_OK;
}

// The host-vector path launches on the handle's own non-blocking stream but shares the handle's scratch (product stream of
// the two-pass copy, sums of cut rows) with the device-vector products, which return right after an asynchronous launch:
// wait for the stream the last of those went to (a non-blocking stream is not ordered against it, not even the null stream).
static int order_behind_last(fs_matrix_t A)
{
  if (!A->last_async) return FS_OK;
  A->last_async = false;
  if (hipStreamSynchronize(A->last_stream) != hipSuccess) {   // e.g. the caller has destroyed that stream since
    (void)hipGetLastError();
    FS_HIP(hipDeviceSynchronize());
  }
  return FS_OK;
}

// products with HOST vectors: synchronous; the copies of x and y overlap the kernels where the kept copy allows it
int fs_spmv_host(fs_matrix_t A, double *y_host, const double *x_host)
{
  FS_RANGE("fs_spmv_host");
  if (int rc = check_mul(A, y_host, x_host, "fs_spmv_host")) return rc;
  std::lock_guard<std::mutex> g(A->lock);
  if (int rc = order_behind_last(A)) return rc;
  return fs::spmv_host_vectors(A->a, A->pipe, y_host, x_host);
}

int fs_spmv_t_host(fs_matrix_t A, double *y_host, const double *x_host)
{
  if (int rc = check_mul(A, y_host, x_host, "fs_spmv_t_host")) return rc;
  FS_RANGE("fs_spmv_t_host");
  if (!A->has_t) { set_error("fs_spmv_t_host: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(A->lock);
  if (int rc = order_behind_last(A)) return rc;
  return fs::spmv_host_vectors(A->at, A->pipe, y_host, x_host);
}

int fs_spmv_t(fs_matrix_t A, double *y, const double *x, fs_stream_t stream)
{
  if (int rc = check_mul(A, y, x, "fs_spmv_t")) return rc;
  FS_RANGE("fs_spmv_t");
  if (!A->has_t) { set_error("fs_spmv_t: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(A->lock);
  A->last_stream = (hipStream_t)stream; A->last_async = true;
  return fs::launch_spmv(A->at, y, x, (hipStream_t)stream);
}

int fs_spmm(fs_matrix_t A, double *Y, const double *X, int k, fs_stream_t stream)
{
  if (int rc = check_mul(A, Y, X, "fs_spmm")) return rc;
  if (k < 1) { set_error("fs_spmm: k < 1"); return FS_ERR_ARG; }
  if (k == 1) return fs_spmv(A, Y, X, stream);
  FS_RANGE("fs_spmm");
  std::lock_guard<std::mutex> g(A->lock);   // the k-column sweeps use the handle's product scratch
  A->last_stream = (hipStream_t)stream; A->last_async = true;
  return fs::launch_spmm(A->a, Y, X, k, (hipStream_t)stream);
}

int fs_spmm_t(fs_matrix_t A, double *Y, const double *X, int k, fs_stream_t stream)
{
  if (int rc = check_mul(A, Y, X, "fs_spmm_t")) return rc;
  if (k < 1) { set_error("fs_spmm_t: k < 1"); return FS_ERR_ARG; }
  if (!A->has_t) { set_error("fs_spmm_t: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  if (k == 1) return fs_spmv_t(A, Y, X, stream);
  FS_RANGE("fs_spmm_t");
  std::lock_guard<std::mutex> g(A->lock);
  A->last_stream = (hipStream_t)stream; A->last_async = true;
  return fs::launch_spmm(A->at, Y, X, k, (hipStream_t)stream);
}

// Everything a k-column product on this handle needs beyond a launch -- the k-column two-pass copy (k = 2..4), the
// column-major scratch and the measured sweeps-or-row-kernel choice (LDS-staged copy, k = 3..16) -- done now, so that
// fs_spmm / fs_spmm_t never build, allocate a copy or wait inside a product.  Synchronous, idempotent.
int fs_matrix_prepare(fs_matrix_t A, int k, int transposed, fs_stream_t stream)
{
  FS_RANGE("fs_matrix_prepare");
  if (!A || k < 1) { set_error("fs_matrix_prepare: bad argument"); return FS_ERR_ARG; }
  if (transposed && !A->has_t) { set_error("fs_matrix_prepare: call fs_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(A->lock);
  fs::DeviceCsr &a = transposed ? A->at : A->a;
  int needs = 0;
  (void)fs::spmm_plan(a, k, &needs);
  const int rc = fs::prepare_spmm(a, k, (hipStream_t)stream);   // nothing left to do: a plan lookup (the drop-in asks on every call)
  if (needs) fs::pool_trim();
  return rc;
}

int fs_matrix_spmm_plan(fs_matrix_t A, int k, int transposed)
{
  if (!A || k < 1 || (transposed && !A->has_t)) return FS_ERR_ARG;
  if (k == 1) return 0;
  std::lock_guard<std::mutex> g(A->lock);
  return fs::spmm_plan(transposed ? A->at : A->a, k, nullptr);
}

int fs_matrix_device_bytes(fs_matrix_t A, int64_t *bytes3)
{
  if (!A || !bytes3) return FS_ERR_ARG;
  std::lock_guard<std::mutex> g(A->lock);
  int64_t a[3] = {0, 0, 0}, t[3] = {0, 0, 0};
  fs::device_bytes(A->a, a);
  if (A->has_t) fs::device_bytes(A->at, t);
  for (int i = 0; i < 3; ++i) bytes3[i] = a[i] + t[i];
  return FS_OK;
}

int fs_matrix_release_csr(fs_matrix_t A)
{
  if (!A) { set_error("fs_matrix_release_csr: NULL handle"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(A->lock);
  FS_HIP(hipDeviceSynchronize());        // products in flight may still read the arrays
  int n = fs::release_plain_csr(A->a);
  if (A->has_t) n += fs::release_plain_csr(A->at);
  return n;
}

int fs_matrix_restore_csr(fs_matrix_t A, int transposed, const int *row_ptr, const int *cols, const double *vals, int space, int borrow)
{
  if (!A || !row_ptr || (transposed && !A->has_t)) { set_error("fs_matrix_restore_csr: bad argument"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(A->lock);
  fs::DeviceCsr &a = transposed ? A->at : A->a;
  if (!a.released) return FS_OK;
  if ((a.nnz > 0 && !cols) || (a.released_valued != (vals != nullptr))) { set_error("fs_matrix_restore_csr: these are not the arrays that were released"); return FS_ERR_ARG; }
  int rc = FS_OK;
  if (space == FS_DEVICE && borrow && fs::aligned16(cols) && (!vals || fs::aligned16(vals))) {
    a.owns = false;
    a.row_ptr = const_cast<int *>(row_ptr); a.cols = const_cast<int *>(cols); a.vals = const_cast<double *>(vals);
  } else {
    a.owns = true;
    rc = fs::to_device(&a.row_ptr, row_ptr, (size_t)a.nrow + 1, space);
    if (!rc) rc = fs::to_device(&a.cols, cols, (size_t)a.nnz, space);
    if (!rc && vals) rc = fs::to_device(&a.vals, vals, (size_t)a.nnz, space);
  }
  a.released = false;
  a.max_row_len = -1;
  if (!rc) rc = fs::build_schedule(a, nullptr, /*allow_tiled=*/false);     // the chunk schedule of the streaming kernel only: the kept copy stays
  if (rc) { (void)fs::release_plain_csr(a); return rc; }
  return FS_OK;
}

int fs_matrix_release_prepared(fs_matrix_t A, int k)
{
  if (!A || k < 0) { set_error("fs_matrix_release_prepared: bad argument"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(A->lock);
  FS_HIP(hipDeviceSynchronize());
  int n = fs::release_prepared(A->a, k);
  if (A->has_t) n += fs::release_prepared(A->at, k);
  fs::pool_trim();
  return n;
}

int fs_ata_mul(fs_matrix_t A, double *y, const double *x, double *tmp, fs_stream_t stream)
{
  FS_RANGE("fs_ata_mul");
  if (int rc = check_mul(A, y, x, "fs_ata_mul")) return rc;
  if (fs::options().ata_kernel == 2 && !fs::options().strict_order && !fs::reproducible_now()) {
    // the fused form (bcsr_AA_mul_B's own loop nest): one pass over A per phase, no copy of A'.  Measured slower than
    // the two products wherever A' fits: kept for callers short of HBM and as the measured answer to
    // SURVEY 8f-4.
    std::lock_guard<std::mutex> g(A->lock);
    return fs::launch_ata_fused(A->a, y, x, (hipStream_t)stream);
  }
  if (!tmp) { set_error("fs_ata_mul: NULL scratch"); return FS_ERR_ARG; }
  if (int rc = fs_matrix_build_transpose(A, stream)) return rc;
  if (int rc = fs_spmv(A, tmp, x, stream)) return rc;
  return fs_spmv_t(A, y, tmp, stream);
}

fs_cbcsr_t fs_cbcsr_create(int nrow, int ncol, int nblocks, int colblocksize, const int *row_ptr, const int *cols,
                           int space)
{
  if (nrow < 0 || ncol < 0 || nblocks < 0 || colblocksize < 1 || !row_ptr) {
    set_error("fs_cbcsr_create: bad argument");
    return nullptr;
  }
  fs_cbcsr_t M = new fs_cbcsr_s();
  M->nrow = nrow; M->ncol = ncol; M->nblocks = nblocks; M->colblocksize = colblocksize;
  const size_t ncell = (size_t)nblocks * (size_t)nrow;
  int rc = fs::to_device(&M->row_ptr, row_ptr, ncell + 1, space);
  int last = 0;
  if (!rc && hipMemcpy(&last, M->row_ptr + ncell, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = FS_ERR_HIP;
  M->nnz = last;
  if (!rc) rc = fs::to_device(&M->cols, cols, (size_t)M->nnz, space);
  if (!rc && ncell < (size_t)0x7fffffff) rc = fs::validate_indices((int)ncell, ncol, M->nnz, M->row_ptr, nullptr, M->cols, nullptr);
  // big enough to pay for a temporary of cell sums: run the cells through the chunk-streaming kernel
  if (!rc && M->nnz >= (1 << 20) && ncell < (size_t)0x7fffffff) {
    fs::DeviceCsr &c = M->cells;
    c.nrow = (int)ncell; c.ncol = ncol; c.nnz = M->nnz;
    c.row_ptr = M->row_ptr; c.cols = M->cols; c.vals = nullptr; c.owns = false;
    rc = fs::build_schedule(c, nullptr, /*allow_tiled=*/false);  // the cell view is only ever streamed
    if (!rc && hipMalloc(&M->cell_sums, sizeof(double) * (ncell ? ncell : 1)) != hipSuccess) rc = FS_ERR_HIP;
    M->use_cells = !rc;
  }
  // large enough for the copies of the general SpMV path to pay: a plain CSR of the same entries (an optimisation:
  // if it cannot be built the cell path above stays)
  if (!rc && M->nnz >= (4 << 20) && ncell < (size_t)0x7fffffff) {
    if (fs::cbcsr_rows_device(M->rows, nrow, ncol, nblocks, M->nnz, M->row_ptr, M->cols, nullptr) == FS_OK) {
      M->use_rows = true;
    } else {
      fs::free_csr(M->rows);
      (void)hipGetLastError();
    }
  }
  fs::pool_trim();
  if (rc) { fs_cbcsr_destroy(M); return nullptr; }
  return M;
}

void fs_cbcsr_destroy(fs_cbcsr_t A)
{
  if (!A) return;
  fs::free_csr(A->cells);        // borrowed arrays: frees the schedule only
  fs::free_csr(A->rows);
  if (A->cell_sums) (void)hipFree(A->cell_sums);
  if (A->row_ptr) (void)hipFree(A->row_ptr);
  if (A->cols) (void)hipFree(A->cols);
  delete A;
}

int fs_cbcsr_spmv(fs_cbcsr_t A, double *y, const double *x, fs_stream_t stream)
{
  FS_RANGE("fs_cbcsr_spmv");
  if (!A || !y || !x) { set_error("fs_cbcsr_spmv: NULL argument"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(A->lock);
  return fs::launch_cbcsr(*A, y, x, (hipStream_t)stream);
}

}  // extern "C"
