// fs_dist.hip -- the row-sharded products on the GPUs of one node, inside the C-ABI (include/fastsparse_hip.h, "several
// GPUs"): one process, N devices, rows cut by non-zeros, the input vector replicated, the output all-gathered over RCCL
// (xGMI) WHILE the product still runs.
//
// This is the north_star's "host C dispatching through a thin C-ABI ... rows range-partitioned across the 8 GPUs of one
// node with y gathered via RCCL" for a plain C caller: csr_A_mul_B / bcsr_A_mul_B / csr_At_mul_B / bcsr_At_mul_B take this
// path when FASTSPARSE_NGPU > 1 (fs_dropin.hip).  The Python bench uses one process per GPU and torch.distributed over the
// same RCCL (libfastsparse_amd/dist.py); both shard the same way.
//
// Both directions are "rows + all-gather" ("prefer holding CSR' row-sharded too"):
//   y = A x   rank r owns rows [bounds[r], bounds[r+1]) of A (equal non-zeros), x replicated;
//   z = A' u  rank r owns rows of A' (= columns of A, again cut by non-zeros), built on demand from the host arrays
//             (fs_dist_matrix_build_transpose), u replicated -- and u IS y after a product, so A then A' needs no copy.
// One product (dist_product): the local product runs in parts (fs_spmv_part: pass 2 of the two-pass pair by ranges of panels)
// on the rank's compute stream; behind every part an event lets the rank's COMMUNICATION stream all-gather the rows that
// part finished -- one ncclAllGather per part on a padded buffer (counts differ between ranks), all ranks' calls of a part
// in one ncclGroupStart/End -- while the next part computes; one fs_copy_segments launch per rank unpacks the padded buffer
// into the full vector at the end.  No n^2 broadcasts, no host round trip.
//
// RCCL is loaded with dlopen when the first context with more than one distinct device is created, so that
// single-GPU users never load it and a process that already holds a copy (PyTorch ships one) shares it.
// A context whose device list names the same device more than once ("virtual ranks": RCCL refuses duplicates)
// exchanges the parts with device-to-device copies instead; that form exists so that the sharding, the parts and the
// padded layout can be exercised on a one-GPU machine (tests/test_gpu_parity.py) and is not a substitute for RCCL on
// real devices.  RCCL with MORE THAN ONE rank has not run anywhere yet (no multi-GPU machine was available to the
// builder); FS_DIST_FORCE_RCCL=1 takes a one-device context through the same group calls.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
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
    FS_SYM(CommInitAll); FS_SYM(CommDestroy); FS_SYM(GroupStart); FS_SYM(GroupEnd); FS_SYM(AllGather); FS_SYM(GetErrorString);
#undef FS_SYM
    q.ok = q.CommInitAll && q.CommDestroy && q.GroupStart && q.GroupEnd && q.AllGather && q.GetErrorString;
    return q;
  }();
  return r;
}

int nccl_fail(ncclResult_t e, const char *what)
{
  fs::set_error(std::string("RCCL error in ") + what + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "?"));
  return FS_ERR_HIP;
}

// the calling thread gets its current device back however the function leaves
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

int env_parts()
{
  static const int v = [] {
    const char *e = getenv("FS_DIST_PARTS");
    const int p = e && *e ? atoi(e) : 4;
    return p < 1 ? 1 : (p > 16 ? 16 : p);
  }();
  return v;
}

}  // namespace

struct fs_dist_s {
  int n = 0;
  std::vector<int> dev;               // device of every rank
  std::vector<hipStream_t> stream;    // compute stream per rank, on its device
  std::vector<hipStream_t> comm_stream;  // communication stream per rank
  bool use_rccl = false;
  std::vector<ncclComm_t> comm;
  std::mutex lock;                    // the streams and communicators serve one product at a time (a communicator must not be
                                      // used from two host threads at once): taken behind the matrix's own lock
};

// one direction of a distributed matrix: row shards of M (A, or A'), input replicated, output gathered
struct DistSide {
  int nrow = 0, ncol = 0;             // of M
  std::vector<int> bounds;            // rank r owns rows [bounds[r], bounds[r + 1]) of M
  std::vector<int64_t> shard_nnz;
  std::vector<fs_matrix_t> shard;
  // per rank, on its device: the shard's rows of the output (room for a padded part behind the last row), the padded
  // receive buffer of the parts, the table of the unpack launch, one event per part
  std::vector<double *> local, pad;
  std::vector<int64_t *> table;
  std::vector<std::vector<hipEvent_t>> ev;   // [rank][part]
  std::vector<hipEvent_t> done;              // [rank]: the unpack on the communication stream
  int nparts = 0;
  std::vector<std::vector<int>> cut;  // [rank][part]: row cuts of the local product (fs_spmv_part_rows)
  std::vector<int> maxc;              // [part]: the largest count of any rank
  std::vector<int64_t> off;           // [part]: first element of the part's region of the padded buffer
  int nseg = 0;
  int64_t max_seg = 0;
  bool built = false;
};

struct fs_dist_matrix_s {
  fs_dist_t D = nullptr;
  int nrow = 0, ncol = 0;
  int64_t nnz = 0;
  DistSide a, t;                      // A (always), A' (fs_dist_matrix_build_transpose)
  // per rank: the whole x (ncol), y (nrow) and z (ncol) on its device.  u of z = A' u is y.
  std::vector<double *> x, y, z;
  // host copies of the CSR are NOT kept; the transpose is built from the arrays the caller passes again
  double *pin = nullptr;              // pinned staging of host vectors
  size_t pin_doubles = 0;
  std::mutex lock;                    // products on one matrix are serialised (x, y, z and the part buffers are per matrix)
};

namespace {

void free_plan(fs_dist_t D, DistSide &S);

void free_side(fs_dist_t D, DistSide &S)
{
  free_plan(D, S);
  for (size_t r = 0; r < S.shard.size(); ++r) {
    (void)hipSetDevice(D->dev[r]);
    if (S.shard[r]) fs_matrix_destroy(S.shard[r]);
  }
  S = DistSide();
}

// row cuts with (almost) equal numbers of non-zeros: bounds[r] = first row whose row_ptr is >= r/n of nnz -- the cut
// libfastsparse_amd/dist.py nnz_balanced_partition makes, essential for power-law matrices (BASELINE config 5)
template <typename RP>
void nnz_cut(std::vector<int> &bounds, int n, int nrow, const RP *row_ptr)
{
  const int64_t nnz = (int64_t)row_ptr[nrow];
  bounds.assign((size_t)n + 1, 0);
  for (int r = 1; r < n; ++r) {
    const int64_t target = nnz * r / n;
    int b = (int)(std::lower_bound(row_ptr, row_ptr + nrow + 1, target, [](RP a, int64_t t) { return (int64_t)a < t; }) - row_ptr);
    if (b > nrow) b = nrow;
    bounds[(size_t)r] = b < bounds[(size_t)r - 1] ? bounds[(size_t)r - 1] : b;
  }
  bounds[(size_t)n] = nrow;
}

int dist_sync(fs_dist_t D);

void free_plan(fs_dist_t D, DistSide &S)
{
  for (size_t r = 0; r < S.local.size(); ++r) {
    (void)hipSetDevice(D->dev[r]);
    if (S.local[r]) (void)hipFree(S.local[r]);
    if (r < S.pad.size() && S.pad[r]) (void)hipFree(S.pad[r]);
    if (r < S.table.size() && S.table[r]) (void)hipFree(S.table[r]);
    if (r < S.ev.size()) for (hipEvent_t e : S.ev[r]) if (e) (void)hipEventDestroy(e);
    if (r < S.done.size() && S.done[r]) (void)hipEventDestroy(S.done[r]);
  }
  S.local.clear(); S.pad.clear(); S.table.clear(); S.ev.clear(); S.done.clear();
  S.nseg = 0; S.max_seg = 0;
}

// after the shards exist: the parts of every rank's local product, the padded layout, buffers, events
int plan_side(fs_dist_t D, DistSide &S, int nparts)
{
  const int n = D->n;
  S.nparts = nparts;
  S.cut.assign((size_t)n, std::vector<int>((size_t)nparts + 1, 0));
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
    if (nl > 0) {
      if (int rc = fs_spmv_part_rows(S.shard[(size_t)r], 0, nparts, S.cut[(size_t)r].data())) return rc;
    }
  }
  S.maxc.assign((size_t)nparts, 0);
  S.off.assign((size_t)nparts + 1, 0);
  int max_rows = 0;
  for (int r = 0; r < n; ++r) max_rows = std::max(max_rows, S.bounds[(size_t)r + 1] - S.bounds[(size_t)r]);
  for (int p = 0; p < nparts; ++p) {
    for (int r = 0; r < n; ++r) S.maxc[(size_t)p] = std::max(S.maxc[(size_t)p], S.cut[(size_t)r][(size_t)p + 1] - S.cut[(size_t)r][(size_t)p]);
    S.off[(size_t)p + 1] = S.off[(size_t)p] + (int64_t)n * S.maxc[(size_t)p];
  }
  std::vector<int64_t> dst, src, cnt;
  for (int p = 0; p < nparts; ++p)
    for (int r = 0; r < n; ++r) {
      const int c = S.cut[(size_t)r][(size_t)p + 1] - S.cut[(size_t)r][(size_t)p];
      if (!c) continue;
      dst.push_back((int64_t)S.bounds[(size_t)r] + S.cut[(size_t)r][(size_t)p]);
      src.push_back(S.off[(size_t)p] + (int64_t)r * S.maxc[(size_t)p]);
      cnt.push_back(c);
      S.max_seg = std::max<int64_t>(S.max_seg, c);
    }
  S.nseg = (int)cnt.size();
  std::vector<int64_t> tab;
  tab.insert(tab.end(), dst.begin(), dst.end());
  tab.insert(tab.end(), src.begin(), src.end());
  tab.insert(tab.end(), cnt.begin(), cnt.end());
  S.local.assign((size_t)n, nullptr);
  S.pad.assign((size_t)n, nullptr);
  S.table.assign((size_t)n, nullptr);
  S.ev.assign((size_t)n, std::vector<hipEvent_t>());
  S.done.assign((size_t)n, nullptr);
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    // the send window of a part starts at its first row and is maxc[p] long: it may run past the shard's last row
    FS_HIP(hipMalloc(&S.local[(size_t)r], sizeof(double) * (size_t)(2 * (int64_t)max_rows + 1)));
    FS_HIP(hipMalloc(&S.pad[(size_t)r], sizeof(double) * (size_t)(S.off[(size_t)nparts] + 1)));
    FS_HIP(hipMalloc(&S.table[(size_t)r], sizeof(int64_t) * (tab.size() + 1)));
    if (!tab.empty()) FS_HIP(hipMemcpy(S.table[(size_t)r], tab.data(), sizeof(int64_t) * tab.size(), hipMemcpyHostToDevice));
    for (int p = 0; p < nparts; ++p) {
      hipEvent_t e = nullptr;
      FS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      S.ev[(size_t)r].push_back(e);
    }
    FS_HIP(hipEventCreateWithFlags(&S.done[(size_t)r], hipEventDisableTiming));
  }
  S.built = true;
  return FS_OK;
}

// out = M in on every rank: local products in parts, the all-gather of every finished part under the later parts, one unpack
// per rank.  in[r] / out[r]: the replicated input and the gathered output on rank r's device.  Asynchronous: the caller
// waits for the compute streams (dist_sync), which wait for the unpacks.
int dist_product(fs_dist_t D, DistSide &S, const std::vector<double *> &in, const std::vector<double *> &out)
{
  const int n = D->n;
  // the cuts belong to the kernel the options select NOW (strict_order / reproducible / spmv_kernel set since the plan was
  // made move the product to another kernel, which finishes its rows elsewhere): plan again when they moved
  {
    bool same = true;
    std::vector<int> now((size_t)S.nparts + 1);
    for (int r = 0; same && r < n; ++r) {
      if (S.bounds[(size_t)r + 1] == S.bounds[(size_t)r]) continue;
      FS_HIP(hipSetDevice(D->dev[r]));
      if (int rc = fs_spmv_part_rows(S.shard[(size_t)r], 0, S.nparts, now.data())) return rc;
      same = now == S.cut[(size_t)r];
    }
    if (!same) {
      if (int rc = dist_sync(D)) return rc;
      free_plan(D, S);
      if (int rc = plan_side(D, S, S.nparts)) return rc;
    }
  }
  const int np = S.nparts;
  for (int p = 0; p < np; ++p) {
    for (int r = 0; r < n; ++r) {
      const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
      FS_HIP(hipSetDevice(D->dev[r]));
      if (nl > 0)
        if (int rc = fs_spmv_part(S.shard[(size_t)r], 0, S.local[(size_t)r], in[(size_t)r], p, np, D->stream[r])) return rc;
      FS_HIP(hipEventRecord(S.ev[(size_t)r][(size_t)p], D->stream[r]));
      FS_HIP(hipStreamWaitEvent(D->comm_stream[r], S.ev[(size_t)r][(size_t)p], 0));
    }
    const size_t count = (size_t)S.maxc[(size_t)p];
    if (!count) continue;
    if (D->use_rccl) {
      // every rank's call of this part in one group; an error inside the group still closes it
      ncclResult_t first = rccl().GroupStart();
      if (first != ncclSuccess) return nccl_fail(first, "ncclGroupStart");
      for (int r = 0; r < n && first == ncclSuccess; ++r)
        first = rccl().AllGather(S.local[(size_t)r] + S.cut[(size_t)r][(size_t)p], S.pad[(size_t)r] + S.off[(size_t)p], count, ncclDouble,
                                 D->comm[(size_t)r], D->comm_stream[r]);
      const ncclResult_t end = rccl().GroupEnd();
      if (first != ncclSuccess) return nccl_fail(first, "ncclAllGather");
      if (end != ncclSuccess) return nccl_fail(end, "ncclGroupEnd");
    } else {
      // virtual ranks on one device (see the header comment): every rank's window is copied into every rank's padded buffer
      // -- the layout an all-gather of `count` elements per rank leaves
      for (int d = 0; d < n; ++d) {
        FS_HIP(hipSetDevice(D->dev[d]));
        for (int r = 0; r < n; ++r) {
          FS_HIP(hipStreamWaitEvent(D->comm_stream[d], S.ev[(size_t)r][(size_t)p], 0));
          FS_HIP(hipMemcpyAsync(S.pad[(size_t)d] + S.off[(size_t)p] + (int64_t)r * (int64_t)count, S.local[(size_t)r] + S.cut[(size_t)r][(size_t)p],
                                sizeof(double) * count, hipMemcpyDeviceToDevice, D->comm_stream[d]));
        }
      }
    }
  }
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    if (int rc = fs_copy_segments(S.nseg, S.table[(size_t)r], S.max_seg, S.pad[(size_t)r], out[(size_t)r], D->comm_stream[r])) return rc;
    FS_HIP(hipEventRecord(S.done[(size_t)r], D->comm_stream[r]));
    FS_HIP(hipStreamWaitEvent(D->stream[r], S.done[(size_t)r], 0));   // whatever follows on the compute stream sees the whole vector
  }
  return FS_OK;
}

int dist_sync(fs_dist_t D)
{
  for (int r = 0; r < D->n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    FS_HIP(hipStreamSynchronize(D->stream[r]));
    FS_HIP(hipStreamSynchronize(D->comm_stream[r]));
  }
  return FS_OK;
}

// host vector -> every rank's device vector through ONE pinned staging buffer: the host copy of chunk c + 1 runs while the
// N asynchronous uploads of chunk c are in flight (each over its own PCIe link)
int upload_all(fs_dist_matrix_t M, const std::vector<double *> &dst, const double *src_host, size_t count)
{
  fs_dist_t D = M->D;
  constexpr size_t kChunk = (size_t)4 << 20;     // doubles per chunk: 32 MB
  const size_t need = std::min(count, 2 * kChunk);
  if (M->pin_doubles < need) {
    if (M->pin) (void)hipHostFree(M->pin);
    M->pin = nullptr; M->pin_doubles = 0;
    FS_HIP(hipHostMalloc(&M->pin, sizeof(double) * (need ? need : 1), hipHostMallocDefault));
    M->pin_doubles = need;
  }
  int slot = 0;
  for (size_t a = 0; a < count; a += kChunk, slot ^= 1) {
    const size_t c = std::min(kChunk, count - a);
    double *stage = M->pin + (size_t)slot * kChunk;
    if (a >= 2 * kChunk)                          // the uploads that read this half two chunks ago have to be done
      for (int r = 0; r < D->n; ++r) { FS_HIP(hipSetDevice(D->dev[r])); FS_HIP(hipStreamSynchronize(D->stream[r])); }
    memcpy(stage, src_host + a, sizeof(double) * c);
    for (int r = 0; r < D->n; ++r) {
      FS_HIP(hipSetDevice(D->dev[r]));
      FS_HIP(hipMemcpyAsync(dst[(size_t)r] + a, stage, sizeof(double) * c, hipMemcpyHostToDevice, D->stream[r]));
    }
  }
  return FS_OK;
}

int download_from(fs_dist_matrix_t M, int rank, double *dst_host, const double *src_dev, size_t count)
{
  fs_dist_t D = M->D;
  FS_HIP(hipSetDevice(D->dev[rank]));
  FS_HIP(hipMemcpyAsync(dst_host, src_dev, sizeof(double) * count, hipMemcpyDeviceToHost, D->stream[rank]));
  return FS_OK;
}

// shards of one direction from host CSR arrays (row_ptr of the direction's matrix; rp may be 64-bit for A')
template <typename RP>
int make_shards(fs_dist_t D, DistSide &S, int nrow, int ncol, const RP *row_ptr, const int *cols, const double *vals)
{
  const int n = D->n;
  S.nrow = nrow; S.ncol = ncol;
  nnz_cut(S.bounds, n, nrow, row_ptr);
  S.shard.assign((size_t)n, nullptr);
  S.shard_nnz.assign((size_t)n, 0);
  std::vector<int> lrp;
  for (int r = 0; r < n; ++r) {
    const int lo = S.bounds[(size_t)r], hi = S.bounds[(size_t)r + 1];
    const int64_t a = (int64_t)row_ptr[lo], b = (int64_t)row_ptr[hi];
    if (b - a > 0x7fffffffll) { fs::set_error("a shard holds more than 2^31-1 non-zeros: use more devices"); return FS_ERR_ARG; }
    lrp.resize((size_t)(hi - lo) + 1);
    for (int i = lo; i <= hi; ++i) lrp[(size_t)(i - lo)] = (int)((int64_t)row_ptr[i] - a);
    S.shard_nnz[(size_t)r] = b - a;
    FS_HIP(hipSetDevice(D->dev[r]));
    S.shard[(size_t)r] = fs_csr_create(hi - lo, ncol, b - a, lrp.data(), cols ? cols + a : nullptr, vals ? vals + a : nullptr, FS_HOST, 0);
    if (!S.shard[(size_t)r]) return FS_ERR_HIP;
  }
  return plan_side(D, S, env_parts());
}

}  // namespace

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
    hipStream_t s = nullptr, c = nullptr;
    if (hipSetDevice(D->dev[r]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) {
      fs::set_error("fs_dist_create: cannot create a stream");
      if (s) (void)hipStreamDestroy(s);
      fs_dist_destroy(D);
      return nullptr;
    }
    D->stream.push_back(s);
    D->comm_stream.push_back(c);
  }
  // FS_DIST_FORCE_RCCL=1 also takes a one-device context through RCCL (communicator, group call, collective on the
  // rank's communication stream): the only way to run that code on a one-GPU machine
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
    if (r < D->comm_stream.size()) (void)hipStreamDestroy(D->comm_stream[r]);
  }
  delete D;
}

int fs_dist_ndev(fs_dist_t D) { return D ? D->n : FS_ERR_ARG; }
int fs_dist_uses_rccl(fs_dist_t D) { return D ? (int)D->use_rccl : FS_ERR_ARG; }

void fs_dist_matrix_destroy(fs_dist_matrix_t M)
{
  if (!M) return;
  DeviceGuard guard;
  (void)dist_sync(M->D);
  free_side(M->D, M->a);
  free_side(M->D, M->t);
  for (size_t r = 0; r < (size_t)M->D->n; ++r) {
    (void)hipSetDevice(M->D->dev[r]);
    if (r < M->x.size() && M->x[r]) (void)hipFree(M->x[r]);
    if (r < M->y.size() && M->y[r]) (void)hipFree(M->y[r]);
    if (r < M->z.size() && M->z[r]) (void)hipFree(M->z[r]);
  }
  if (M->pin) (void)hipHostFree(M->pin);
  delete M;
}

fs_dist_matrix_t fs_dist_csr_create(fs_dist_t D, int nrow, int ncol, int64_t nnz, const int *row_ptr, const int *cols,
                                    const double *vals)
{
  if (!D || nrow < 0 || ncol < 0 || nnz < 0 || !row_ptr || (nnz > 0 && !cols)) { fs::set_error("fs_dist_csr_create: bad argument"); return nullptr; }
  DeviceGuard guard;
  fs_dist_matrix_t M = new fs_dist_matrix_s();
  M->D = D; M->nrow = nrow; M->ncol = ncol; M->nnz = nnz;
  const int n = D->n;
  M->x.assign((size_t)n, nullptr);
  M->y.assign((size_t)n, nullptr);
  M->z.assign((size_t)n, nullptr);
  bool ok = make_shards(D, M->a, nrow, ncol, row_ptr, cols, vals) == FS_OK;
  for (int r = 0; ok && r < n; ++r) {
    ok = hipSetDevice(D->dev[r]) == hipSuccess;
    ok = ok && hipMalloc(&M->x[(size_t)r], sizeof(double) * (size_t)(ncol ? ncol : 1)) == hipSuccess;
    ok = ok && hipMalloc(&M->y[(size_t)r], sizeof(double) * (size_t)(nrow ? nrow : 1)) == hipSuccess;
    if (!ok) fs::set_error("fs_dist_csr_create: out of device memory");
  }
  if (!ok) { fs_dist_matrix_destroy(M); return nullptr; }
  return M;
}

// Row shards of A' (cut by non-zeros of the COLUMNS of A) from the same host arrays the matrix was created from: one pass
// counts the columns, one pass per device -- in parallel, one host thread each -- collects its columns' entries in ascending
// row order (the order a stable column sort of A gives: rows of A' keep it, like fs_matrix_build_transpose on one GPU).
int fs_dist_matrix_build_transpose(fs_dist_matrix_t M, const int *row_ptr, const int *cols, const double *vals)
{
  if (!M || !row_ptr || (M->nnz > 0 && !cols)) { fs::set_error("fs_dist_matrix_build_transpose: bad argument"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(M->lock);
  if (M->t.built) return FS_OK;
  DeviceGuard guard;
  fs_dist_t D = M->D;
  const int n = D->n, nrow = M->nrow, ncol = M->ncol;
  std::vector<int64_t> tptr((size_t)ncol + 1, 0);           // row_ptr of A'
  for (int64_t i = 0; i < M->nnz; ++i) {
    if ((unsigned)cols[i] >= (unsigned)ncol) { fs::set_error("fs_dist_matrix_build_transpose: column out of range"); return FS_ERR_ARG; }
    ++tptr[(size_t)cols[i] + 1];
  }
  for (int c = 0; c < ncol; ++c) tptr[(size_t)c + 1] += tptr[(size_t)c];
  DistSide &T = M->t;
  T.nrow = ncol; T.ncol = nrow;
  nnz_cut(T.bounds, n, ncol, tptr.data());
  T.shard.assign((size_t)n, nullptr);
  T.shard_nnz.assign((size_t)n, 0);
  std::vector<int> rcs((size_t)n, FS_OK);
  std::vector<std::string> errs((size_t)n);
  std::mutex create_lock;                                    // the host passes run in parallel, the device builds one at a time
  std::vector<std::thread> th;
  for (int r = 0; r < n; ++r)
    th.emplace_back([&, r] {
      const int lo = T.bounds[(size_t)r], hi = T.bounds[(size_t)r + 1];
      const int64_t base = tptr[(size_t)lo], cnt = tptr[(size_t)hi] - base;
      T.shard_nnz[(size_t)r] = cnt;
      if (cnt > 0x7fffffffll) { rcs[(size_t)r] = FS_ERR_ARG; errs[(size_t)r] = "a shard of A' holds more than 2^31-1 non-zeros"; return; }
      std::vector<int> lrp((size_t)(hi - lo) + 1), lc((size_t)cnt);
      std::vector<double> lv(vals ? (size_t)cnt : 0);
      for (int c = lo; c <= hi; ++c) lrp[(size_t)(c - lo)] = (int)(tptr[(size_t)c] - base);
      std::vector<int> fill(lrp.begin(), lrp.end() - 1);
      for (int row = 0; row < nrow; ++row)
        for (int64_t i = row_ptr[row]; i < row_ptr[row + 1]; ++i) {
          const int c = cols[i];
          if (c < lo || c >= hi) continue;
          const int at = fill[(size_t)(c - lo)]++;
          lc[(size_t)at] = row;
          if (vals) lv[(size_t)at] = vals[i];
        }
      std::lock_guard<std::mutex> cg(create_lock);
      if (hipSetDevice(D->dev[r]) != hipSuccess) { rcs[(size_t)r] = FS_ERR_HIP; errs[(size_t)r] = "hipSetDevice failed"; return; }
      T.shard[(size_t)r] = fs_csr_create(hi - lo, nrow, cnt, lrp.data(), lc.data(), vals ? lv.data() : nullptr, FS_HOST, 0);
      if (!T.shard[(size_t)r]) { rcs[(size_t)r] = FS_ERR_HIP; errs[(size_t)r] = fs_last_error(); }
    });
  for (std::thread &t : th) t.join();
  int rc = FS_OK;
  for (int r = 0; r < n; ++r)
    if (rcs[(size_t)r] != FS_OK) { rc = rcs[(size_t)r]; fs::set_error("fs_dist_matrix_build_transpose: " + errs[(size_t)r]); }
  if (rc == FS_OK) rc = plan_side(D, T, env_parts());
  for (int r = 0; rc == FS_OK && r < n; ++r) {
    if (hipSetDevice(D->dev[r]) != hipSuccess || hipMalloc(&M->z[(size_t)r], sizeof(double) * (size_t)(ncol ? ncol : 1)) != hipSuccess) {
      fs::set_error("fs_dist_matrix_build_transpose: out of device memory");
      rc = FS_ERR_HIP;
    }
  }
  if (rc != FS_OK) {
    for (int r = 0; r < n; ++r)
      if (M->z[(size_t)r]) { (void)hipSetDevice(D->dev[r]); (void)hipFree(M->z[(size_t)r]); M->z[(size_t)r] = nullptr; }
    free_side(D, T);
  }
  return rc;
}

int fs_dist_matrix_has_transpose(fs_dist_matrix_t M) { return M && M->t.built; }

int fs_dist_matrix_bounds(fs_dist_matrix_t M, int *bounds)
{
  if (!M || !bounds) return FS_ERR_ARG;
  for (size_t i = 0; i < M->a.bounds.size(); ++i) bounds[i] = M->a.bounds[i];
  return FS_OK;
}

int64_t fs_dist_matrix_shard_nnz(fs_dist_matrix_t M, int rank)
{
  if (!M || rank < 0 || rank >= M->D->n) return FS_ERR_ARG;
  return M->a.shard_nnz[(size_t)rank];
}

int fs_dist_spmv(fs_dist_matrix_t M, double *y_host, const double *x_host)
{
  if (!M || !y_host || !x_host) { fs::set_error("fs_dist_spmv: NULL argument"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  if (int rc = upload_all(M, M->x, x_host, (size_t)M->ncol)) return rc;
  if (int rc = dist_product(M->D, M->a, M->x, M->y)) return rc;
  if (int rc = download_from(M, 0, y_host, M->y[0], (size_t)M->nrow)) return rc;
  return dist_sync(M->D);
}

int fs_dist_spmv_t(fs_dist_matrix_t M, double *z_host, const double *u_host)
{
  if (!M || !z_host || !u_host) { fs::set_error("fs_dist_spmv_t: NULL argument"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_spmv_t: call fs_dist_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  if (int rc = upload_all(M, M->y, u_host, (size_t)M->nrow)) return rc;   // u lives where y does: A then A' chains without a copy
  if (int rc = dist_product(M->D, M->t, M->y, M->z)) return rc;
  if (int rc = download_from(M, 0, z_host, M->z[0], (size_t)M->ncol)) return rc;
  return dist_sync(M->D);
}

// device-resident forms: the caller fills fs_dist_x(M, r) on every rank once and iterates -- y = A x, z = A' y, and (square
// matrices) fs_dist_swap_xy to make y the next x -- without anything crossing PCIe; each call returns after every device has
// the whole output vector
int fs_dist_spmv_resident(fs_dist_matrix_t M)
{
  if (!M) { fs::set_error("fs_dist_spmv_resident: NULL handle"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  if (int rc = dist_product(M->D, M->a, M->x, M->y)) return rc;
  return dist_sync(M->D);
}

int fs_dist_spmv_t_resident(fs_dist_matrix_t M)
{
  if (!M) { fs::set_error("fs_dist_spmv_t_resident: NULL handle"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_spmv_t_resident: call fs_dist_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  if (int rc = dist_product(M->D, M->t, M->y, M->z)) return rc;
  return dist_sync(M->D);
}

int fs_dist_swap_xy(fs_dist_matrix_t M)
{
  if (!M) { fs::set_error("fs_dist_swap_xy: NULL handle"); return FS_ERR_ARG; }
  if (M->nrow != M->ncol) { fs::set_error("fs_dist_swap_xy: the matrix is not square"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(M->lock);
  M->x.swap(M->y);
  return FS_OK;
}

// (A'A + lambda I) x = b on the row-sharded matrix: bsbm_cg (cg.h:25-82) across the GPUs, everything resident.  Every device
// keeps the WHOLE x, r, p, q (fs_dist_x is p, fs_dist_z is q) and runs the same fused vector kernels on them -- identical
// inputs, identical kernels, so every device holds identical vectors and the dots need no exchange: the host reads 8 bytes
// from device 0 per reduction.  Per iteration: y = A p and q = A' y, each with its all-gather inside the product.  The O(F)
// vector work is replicated, not divided, by the number of devices (libfastsparse_amd/dist.py ShardedCG's "gather" scheme
// divides it; here the products dominate).  b_host / x_host: F = ncol doubles on the host.
int fs_dist_cg(fs_dist_matrix_t M, double *x_host, const double *b_host, double lambda, double tol, int *out_iter)
{
  if (!M || !x_host || !b_host) { fs::set_error("fs_dist_cg: NULL argument"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_cg: call fs_dist_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  fs_dist_t D = M->D;
  const int n = D->n, F = M->ncol;
  struct Work {
    fs_dist_t D;
    std::vector<double *> sol, r, b, part, red, st;
    ~Work()
    {
      for (size_t d = 0; d < sol.size(); ++d) {
        (void)hipSetDevice(D->dev[d]);
        for (double *p : {sol[d], r[d], b[d], part[d], red[d], st[d]})
          if (p) (void)hipFree(p);
      }
    }
  } W{D, std::vector<double *>((size_t)n, nullptr), std::vector<double *>((size_t)n, nullptr), std::vector<double *>((size_t)n, nullptr),
      std::vector<double *>((size_t)n, nullptr), std::vector<double *>((size_t)n, nullptr), std::vector<double *>((size_t)n, nullptr)};
  for (int d = 0; d < n; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    FS_HIP(hipMalloc(&W.sol[(size_t)d], sizeof(double) * (size_t)(F ? F : 1)));
    FS_HIP(hipMalloc(&W.r[(size_t)d], sizeof(double) * (size_t)(F ? F : 1)));
    FS_HIP(hipMalloc(&W.b[(size_t)d], sizeof(double) * (size_t)(F ? F : 1)));
    FS_HIP(hipMalloc(&W.part[(size_t)d], sizeof(double) * fs::kCgPartDoubles));
    FS_HIP(hipMalloc(&W.red[(size_t)d], sizeof(double) * 4));
    FS_HIP(hipMalloc(&W.st[(size_t)d], sizeof(double) * fs::kCgStateDoubles));
  }
  if (int rc = upload_all(M, W.b, b_host, (size_t)F)) return rc;
  // the scalars of the iteration stay on every device (the same kernels on the same data: every device decides alike); the
  // host watches device 0's done flag one iteration behind (see fs_cg.hip)
  fs::CgFlags fl;
  FS_HIP(hipSetDevice(D->dev[0]));
  if (int rc = fl.init()) return rc;
  for (int d = 0; d < n; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    if (int rc = fs::cg_dev_init(F, W.b[(size_t)d], W.sol[(size_t)d], W.r[(size_t)d], M->x[(size_t)d], W.part[(size_t)d], W.red[(size_t)d],
                                 W.st[(size_t)d], tol, D->stream[d])) return rc;
  }
  for (int iter = 0; iter < F; iter++) {
    if (int rc = dist_product(D, M->a, M->x, M->y)) return rc;      // y = A p
    if (int rc = dist_product(D, M->t, M->y, M->z)) return rc;      // q = A' y
    for (int d = 0; d < n; ++d) {
      FS_HIP(hipSetDevice(D->dev[d]));
      if (int rc = fs::cg_dev_steps(F, lambda, W.sol[(size_t)d], W.r[(size_t)d], M->x[(size_t)d], M->z[(size_t)d], W.part[(size_t)d],
                                    W.red[(size_t)d], W.st[(size_t)d], D->stream[d])) return rc;
    }
    bool stop = false;
    FS_HIP(hipSetDevice(D->dev[0]));
    if (int rc = fl.after_iteration(iter, W.st[0], D->stream[0], &stop)) return rc;
    if (stop) break;
  }
  double fin[2] = {0.0, 0.0};
  FS_HIP(hipSetDevice(D->dev[0]));
  FS_HIP(hipMemcpyAsync(fin, W.st[0] + fs::kCgStateDone, sizeof(fin), hipMemcpyDeviceToHost, D->stream[0]));
  FS_HIP(hipStreamSynchronize(D->stream[0]));
  const int iter = (int)fin[1];
  if (int rc = download_from(M, 0, x_host, W.sol[0], (size_t)F)) return rc;
  if (int rc = dist_sync(D)) return rc;
  if (out_iter) *out_iter = iter;
  return FS_OK;
}

double *fs_dist_x(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->x[(size_t)rank] : nullptr; }
double *fs_dist_y(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->y[(size_t)rank] : nullptr; }
double *fs_dist_z(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->z[(size_t)rank] : nullptr; }

}  // extern "C"
