// fs_dist.hip -- the row-sharded product on the GPUs of one node, inside the C-ABI (include/fastsparse_hip.h, "several
// GPUs"): one process, N devices, rows cut by non-zeros, x replicated, y all-gathered over RCCL (xGMI).
//
// This is the north_star's "host C dispatching through a thin C-ABI ... rows range-partitioned across the 8 GPUs of one
// node with y gathered via RCCL" for a plain C caller: csr_A_mul_B / bcsr_A_mul_B take this path when FASTSPARSE_NGPU > 1
// (fs_dropin.hip).  The Python bench uses one process per GPU and torch.distributed over the same RCCL
// (libfastsparse_amd/dist.py); both shard the same way.
//
// RCCL is loaded with dlopen when the first context with more than one distinct device is created, so that
// single-GPU users never load it and a process that already holds a copy (PyTorch ships one) shares it.
// A context whose device list names the same device more than once ("virtual ranks": RCCL refuses duplicates)
// exchanges the y shards with device-to-device copies instead; that form exists so that the sharding logic can be
// exercised on a one-GPU machine (tests/test_gpu_parity.py) and is not a substitute for RCCL on real devices.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "fs_common.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

Rccl &rccl()
{
  static Rccl r = [] {
    Rccl q;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      q.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (q.lib) break;
    }
    if (!q.lib) return q;
#define FS_SYM(f) q.f = reinterpret_cast<decltype(q.f)>(dlsym(q.lib, "nccl" #f))
    FS_SYM(CommInitAll); FS_SYM(CommDestroy); FS_SYM(GroupStart); FS_SYM(GroupEnd); FS_SYM(AllGather); FS_SYM(Broadcast);
    FS_SYM(GetErrorString);
#undef FS_SYM
    q.ok = q.CommInitAll && q.CommDestroy && q.GroupStart && q.GroupEnd && q.AllGather && q.Broadcast && q.GetErrorString;
    return q;
  }();
  return r;
}

int nccl_fail(ncclResult_t e, const char *what)
{
  fs::set_error(std::string("RCCL error in ") + what + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "?"));
  return FS_ERR_HIP;
}

#define FS_NCCL(call, what)                                       \
  do {                                                            \
    ncclResult_t e_ = (call);                                     \
    if (e_ != ncclSuccess) return nccl_fail(e_, what);            \
  } while (0)

// the calling thread gets its current device back however the function leaves
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

}  // namespace

struct fs_dist_s {
  int n = 0;
  std::vector<int> dev;               // device of every rank
  std::vector<hipStream_t> stream;    // one stream per rank, on its device
  bool use_rccl = false;
  std::vector<ncclComm_t> comm;
};

struct fs_dist_matrix_s {
  fs_dist_t D = nullptr;
  int nrow = 0, ncol = 0;
  int64_t nnz = 0;
  std::vector<int> bounds;            // rank r owns rows [bounds[r], bounds[r + 1])
  std::vector<int64_t> shard_nnz;
  std::vector<fs_matrix_t> shard;
  std::vector<double *> x, y;         // per rank: the whole x (ncol) and the whole y (nrow) on its device
  bool equal = false;                 // all shards hold the same number of rows: one ncclAllGather
};

extern "C" {

fs_dist_t fs_dist_create(int ndev, const int *devices)
{
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { fs::set_error("fs_dist_create: no HIP device"); return nullptr; }
  if (ndev < 1) ndev = visible;
  DeviceGuard guard;
  fs_dist_t D = new fs_dist_s();
  D->n = ndev;
  bool distinct = true;
  for (int r = 0; r < ndev; ++r) {
    const int d = devices ? devices[r] : r;
    if (d < 0 || d >= visible) { fs::set_error("fs_dist_create: device out of range"); delete D; return nullptr; }
    for (int q : D->dev) distinct = distinct && q != d;
    D->dev.push_back(d);
  }
  for (int r = 0; r < ndev; ++r) {
    hipStream_t s = nullptr;
    if (hipSetDevice(D->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      fs::set_error("fs_dist_create: cannot create a stream");
      fs_dist_destroy(D);
      return nullptr;
    }
    D->stream.push_back(s);
  }
  // FS_DIST_FORCE_RCCL=1 also takes a one-device context through RCCL (communicator, group call, collective on the
  // rank's stream): the only way to run that code on a one-GPU machine
  const char *force = getenv("FS_DIST_FORCE_RCCL");
  if (distinct && (ndev > 1 || (force && *force == '1'))) {
    if (!rccl().ok) { fs::set_error("fs_dist_create: librccl.so could not be loaded"); fs_dist_destroy(D); return nullptr; }
    D->comm.assign((size_t)ndev, nullptr);
    const ncclResult_t e = rccl().CommInitAll(D->comm.data(), ndev, D->dev.data());
    if (e != ncclSuccess) { nccl_fail(e, "ncclCommInitAll"); D->comm.clear(); fs_dist_destroy(D); return nullptr; }
    D->use_rccl = true;
  }
  return D;
}

void fs_dist_destroy(fs_dist_t D)
{
  if (!D) return;
  DeviceGuard guard;
  for (ncclComm_t c : D->comm)
    if (c) (void)rccl().CommDestroy(c);
  for (size_t r = 0; r < D->stream.size(); ++r) {
    (void)hipSetDevice(D->dev[r]);
    (void)hipStreamDestroy(D->stream[r]);
  }
  delete D;
}

int fs_dist_ndev(fs_dist_t D) { return D ? D->n : FS_ERR_ARG; }
int fs_dist_uses_rccl(fs_dist_t D) { return D ? (int)D->use_rccl : FS_ERR_ARG; }

void fs_dist_matrix_destroy(fs_dist_matrix_t M)
{
  if (!M) return;
  DeviceGuard guard;
  for (size_t r = 0; r < M->shard.size(); ++r) {
    (void)hipSetDevice(M->D->dev[r]);
    if (M->shard[r]) fs_matrix_destroy(M->shard[r]);
    if (r < M->x.size() && M->x[r]) (void)hipFree(M->x[r]);
    if (r < M->y.size() && M->y[r]) (void)hipFree(M->y[r]);
  }
  delete M;
}

// Row shards with (almost) equal numbers of non-zeros: bounds[r] = first row whose row_ptr is >= r/n of nnz -- the
// cut libfastsparse_amd/dist.py nnz_balanced_partition makes, essential for power-law matrices (BASELINE config 5).
fs_dist_matrix_t fs_dist_csr_create(fs_dist_t D, int nrow, int ncol, int64_t nnz, const int *row_ptr, const int *cols,
                                    const double *vals)
{
  if (!D || nrow < 0 || ncol < 0 || nnz < 0 || !row_ptr || (nnz > 0 && !cols)) { fs::set_error("fs_dist_csr_create: bad argument"); return nullptr; }
  DeviceGuard guard;
  fs_dist_matrix_t M = new fs_dist_matrix_s();
  M->D = D; M->nrow = nrow; M->ncol = ncol; M->nnz = nnz;
  const int n = D->n;
  M->bounds.assign((size_t)n + 1, 0);
  for (int r = 1; r < n; ++r) {
    const int64_t target = nnz * r / n;
    int b = (int)(std::lower_bound(row_ptr, row_ptr + nrow + 1, target, [](int a, int64_t t) { return (int64_t)a < t; }) - row_ptr);
    if (b > nrow) b = nrow;
    M->bounds[r] = b < M->bounds[r - 1] ? M->bounds[r - 1] : b;
  }
  M->bounds[n] = nrow;
  M->equal = true;
  for (int r = 0; r < n; ++r) M->equal = M->equal && (int64_t)(M->bounds[r + 1] - M->bounds[r]) * n == nrow;
  M->shard.assign((size_t)n, nullptr);
  M->x.assign((size_t)n, nullptr);
  M->y.assign((size_t)n, nullptr);
  M->shard_nnz.assign((size_t)n, 0);
  std::vector<int> lrp;
  for (int r = 0; r < n; ++r) {
    const int lo = M->bounds[r], hi = M->bounds[r + 1];
    const int64_t a = row_ptr[lo], b = row_ptr[hi];
    lrp.resize((size_t)(hi - lo) + 1);
    for (int i = lo; i <= hi; ++i) lrp[(size_t)(i - lo)] = (int)(row_ptr[i] - a);
    M->shard_nnz[r] = b - a;
    bool ok = hipSetDevice(D->dev[r]) == hipSuccess;
    if (ok) {
      M->shard[r] = fs_csr_create(hi - lo, ncol, b - a, lrp.data(), cols ? cols + a : nullptr, vals ? vals + a : nullptr, FS_HOST, 0);
      ok = M->shard[r] != nullptr;
    }
    ok = ok && hipMalloc(&M->x[r], sizeof(double) * (size_t)(ncol ? ncol : 1)) == hipSuccess;
    ok = ok && hipMalloc(&M->y[r], sizeof(double) * (size_t)(nrow ? nrow : 1)) == hipSuccess;
    if (!ok) {
      if (M->shard[r]) fs::set_error("fs_dist_csr_create: out of device memory");
      fs_dist_matrix_destroy(M);
      return nullptr;
    }
  }
  return M;
}

int fs_dist_matrix_bounds(fs_dist_matrix_t M, int *bounds)
{
  if (!M || !bounds) return FS_ERR_ARG;
  for (size_t i = 0; i < M->bounds.size(); ++i) bounds[i] = M->bounds[i];
  return FS_OK;
}

int64_t fs_dist_matrix_shard_nnz(fs_dist_matrix_t M, int rank)
{
  if (!M || rank < 0 || rank >= M->D->n) return FS_ERR_ARG;
  return M->shard_nnz[(size_t)rank];
}

// every rank: local product into its rows of its own y, then the exchange that completes y everywhere
static int dist_spmv_on_device(fs_dist_matrix_t M)
{
  fs_dist_t D = M->D;
  const int n = D->n;
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    if (M->bounds[r + 1] > M->bounds[r])
      if (int rc = fs_spmv(M->shard[r], M->y[r] + M->bounds[r], M->x[r], D->stream[r])) return rc;
  }
  if (n == 1 && !D->use_rccl) return FS_OK;
  if (D->use_rccl) {
    FS_NCCL(rccl().GroupStart(), "ncclGroupStart");
    if (M->equal) {
      const size_t count = (size_t)(M->nrow / n);
      for (int d = 0; d < n; ++d)
        FS_NCCL(rccl().AllGather(M->y[d] + (size_t)d * count, M->y[d], count, ncclDouble, D->comm[d], D->stream[d]), "ncclAllGather");
    } else {
      // unequal shards (the nnz-balanced cut of a power-law matrix): one in-place broadcast per shard, all in one group
      for (int r = 0; r < n; ++r) {
        const size_t count = (size_t)(M->bounds[r + 1] - M->bounds[r]);
        if (!count) continue;
        for (int d = 0; d < n; ++d)
          FS_NCCL(rccl().Broadcast(M->y[d] + M->bounds[r], M->y[d] + M->bounds[r], count, ncclDouble, r, D->comm[d], D->stream[d]),
                  "ncclBroadcast");
      }
    }
    FS_NCCL(rccl().GroupEnd(), "ncclGroupEnd");
    return FS_OK;
  }
  // virtual ranks on one device (see the header comment): every shard is copied to every other rank's y
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    FS_HIP(hipStreamSynchronize(D->stream[r]));
  }
  for (int r = 0; r < n; ++r) {
    const size_t bytes = sizeof(double) * (size_t)(M->bounds[r + 1] - M->bounds[r]);
    if (!bytes) continue;
    for (int d = 0; d < n; ++d)
      if (d != r) FS_HIP(hipMemcpyAsync(M->y[d] + M->bounds[r], M->y[r] + M->bounds[r], bytes, hipMemcpyDeviceToDevice, D->stream[d]));
  }
  return FS_OK;
}

static int dist_sync(fs_dist_matrix_t M)
{
  for (int r = 0; r < M->D->n; ++r) {
    FS_HIP(hipSetDevice(M->D->dev[r]));
    FS_HIP(hipStreamSynchronize(M->D->stream[r]));
  }
  return FS_OK;
}

int fs_dist_spmv(fs_dist_matrix_t M, double *y_host, const double *x_host)
{
  if (!M || !y_host || !x_host) { fs::set_error("fs_dist_spmv: NULL argument"); return FS_ERR_ARG; }
  fs_dist_t D = M->D;
  DeviceGuard guard;
  // x to every device over its own PCIe link.  A copy from pageable memory returns when it is done, so one host thread
  // per device: N uploads at once instead of N in a row (config 5: 800 MB per device)
  if (D->n == 1) {
    FS_HIP(hipSetDevice(D->dev[0]));
    FS_HIP(hipMemcpyAsync(M->x[0], x_host, sizeof(double) * (size_t)M->ncol, hipMemcpyHostToDevice, D->stream[0]));
  } else {
    std::vector<hipError_t> err((size_t)D->n, hipSuccess);
    std::vector<std::thread> up;
    for (int r = 0; r < D->n; ++r)
      up.emplace_back([&, r] {
        hipError_t e = hipSetDevice(D->dev[r]);
        if (e == hipSuccess) e = hipMemcpyAsync(M->x[r], x_host, sizeof(double) * (size_t)M->ncol, hipMemcpyHostToDevice, D->stream[r]);
        if (e == hipSuccess) e = hipStreamSynchronize(D->stream[r]);
        err[(size_t)r] = e;
      });
    for (std::thread &t : up) t.join();
    for (hipError_t e : err) FS_HIP(e);
  }
  if (int rc = dist_spmv_on_device(M)) return rc;
  FS_HIP(hipSetDevice(D->dev[0]));
  FS_HIP(hipMemcpyAsync(y_host, M->y[0], sizeof(double) * (size_t)M->nrow, hipMemcpyDeviceToHost, D->stream[0]));
  return dist_sync(M);
}

// device-resident form: the caller fills fs_dist_x(M, r) on every rank (or iterates: y of one product is the next x
// when nrow == ncol), runs the product, reads fs_dist_y(M, r); returns after every device has the whole y
int fs_dist_spmv_resident(fs_dist_matrix_t M)
{
  if (!M) { fs::set_error("fs_dist_spmv_resident: NULL handle"); return FS_ERR_ARG; }
  DeviceGuard guard;
  if (int rc = dist_spmv_on_device(M)) return rc;
  return dist_sync(M);
}

double *fs_dist_x(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->x[(size_t)rank] : nullptr; }
double *fs_dist_y(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->y[(size_t)rank] : nullptr; }

}  // extern "C"
