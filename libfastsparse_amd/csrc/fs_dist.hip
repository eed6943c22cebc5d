// fs_dist.hip -- the row-sharded products on the GPUs of one node, inside the C-ABI (include/fastsparse_hip.h, "several
// GPUs"): one process, N devices, rows cut by non-zeros, the input vector replicated, the output all-gathered over RCCL
// (xGMI) WHILE the product still runs.
//
// This is the north_star's "host C dispatching through a thin C-ABI ... rows range-partitioned across the 8 GPUs of one
// node with y gathered via RCCL" for a plain C caller: EVERY product entry point of sparse.h / dsparse.h / csr.h / cbcsr.h / cg.h
// takes this path when FASTSPARSE_NGPU > 1 (fs_dropin.hip, "several GPUs").  The Python bench uses one process per GPU and torch.distributed over the
// same RCCL (libfastsparse_amd/dist.py); both shard the same way.
//
// Both directions are "rows + all-gather" ("prefer holding CSR' row-sharded too"):
//   y = A x   rank r owns rows [bounds[r], bounds[r+1]) of A (equal non-zeros), x replicated;
//   z = A' u  rank r owns rows of A' (= columns of A, again cut by non-zeros), u replicated -- and u IS y after a product, so
//             A then A' needs no copy.  The shards of A' are built from host arrays (fs_dist_matrix_build_transpose) or, with no
//             whole-matrix host array anywhere, from the device-resident shards of A (fs_dist_matrix_build_transpose_device).
// A matrix comes from ONE host CSR (fs_dist_csr_create: int row_ptr, < 2^31 entries in total) or from per-rank shards
// (fs_dist_csr_create_from_shards: every shard < 2^31 entries, the total is 64-bit -- BASELINE config 5 only exists this way).
//
// One product (dist_product): the local product runs in parts (fs_spmv_part: pass 2 of the two-pass pair by ranges of panels)
// on the rank's compute stream; behind every part an event lets the rank's COMMUNICATION stream all-gather the rows that
// part finished -- one ncclAllGather per part on a padded buffer (counts differ between ranks), all ranks' calls of a part
// in one ncclGroupStart/End -- while the next part computes; one fs_copy_segments launch per rank unpacks the padded buffer
// into the full vector at the end.  No n^2 broadcasts, no host round trip.
// The CONSERVATIVE mode (FS_DIST_PARTS=1; on virtual ranks also after an exchange of the overlapped mode failed) is one whole-shard
// all-gather behind the finished local product: no send window ever reaches into rows that are still being written.  With RCCL a
// group call that failed aborts the communicators (no second collective on a half-issued group): the context then returns errors.
// The caller's vectors may live in host memory or in the HBM of any device (vec_in / vec_out): a vector in HBM is read in place by
// the ranks of its device, travels device to device to the others, and an output in HBM is written by the unpack launch itself.
//
// RCCL is loaded with dlopen when the first context with more than one distinct device is created, so that
// single-GPU users never load it and a process that already holds a copy (PyTorch ships one) shares it.
// A context whose device list names the same device more than once ("virtual ranks": RCCL refuses duplicates)
// exchanges the parts with device-to-device copies instead; that form exists so that the sharding, the parts and the
// padded layout can be exercised on a one-GPU machine (tests/test_gpu_parity.py) and is not a substitute for RCCL on
// real devices.  UNVERIFIED ON HARDWARE: RCCL with MORE THAN ONE rank has not run anywhere yet (no multi-GPU machine was
// available to the builder); FS_DIST_FORCE_RCCL=1 takes a one-device context through the same group calls.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "csr.h"
#include "fs_common.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
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
    FS_SYM(CommInitAll); FS_SYM(CommDestroy); FS_SYM(CommAbort); FS_SYM(GroupStart); FS_SYM(GroupEnd); FS_SYM(AllGather); FS_SYM(GetErrorString);
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

// FS_DIST_FAIL_PART (tests): the exchange of part p fails, once per process -- whichever product gets there first
bool take_injection(int p);
std::atomic<long> g_products{0};     // sharded products launched by this process (fs_debug_dist_products: the tests' proof of the path taken)

int env_parts()
{
  static const int v = [] {
    const char *e = getenv("FS_DIST_PARTS");
    const int p = e && *e ? atoi(e) : 4;
    return p < 1 ? 1 : (p > 16 ? 16 : p);
  }();
  return v;
}

// FS_DIST_FAIL_PART=p (tests only): the all-gather of part p of the next overlapped product reports an error once, which takes
// the context to the conservative mode -- the only way to walk that path without a broken fabric
int env_fail_part()
{
  static const int v = [] { const char *e = getenv("FS_DIST_FAIL_PART"); return e && *e ? atoi(e) : -1; }();
  return v;
}

bool take_injection(int p)
{
  static std::atomic<bool> injected{false};
  return env_fail_part() == p && !injected.exchange(true);
}

}  // namespace

// One ISSUING thread per rank (FS_DIST_THREADS=1; experimental, off by default).  One host thread pays about 50 us per rank and
// product for the launches, events and exchange calls of the ranks one after the other (profiles/r05_dist_host_overhead.txt): at
// N = 8 the one-process path is issue-bound for products below ~0.5 ms.  With this switch every rank's calls come from a thread of
// its own -- its device current once and for all -- and a product is a few rounds of "every rank does its piece, then all meet".
// With RCCL each thread issues ITS communicator's ncclAllGather (no group: the many-threads-one-device-each form of the API).
// UNVERIFIED on more than one GPU, like the rest of this file; libfastsparse_amd/native_dist_bench.py times it beside the default.
struct RankWorkers {
  int n = 0;
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable go, done;
  std::function<int(int)> job;
  unsigned long gen = 0;
  int pending = 0, fixed_order = 0;
  bool stop = false;
  std::vector<int> rc;
  std::vector<std::string> err;

  void start(const std::vector<int> &dev)
  {
    n = (int)dev.size();
    rc.assign((size_t)n, 0);
    err.assign((size_t)n, std::string());
    for (int r = 0; r < n; ++r)
      th.emplace_back([this, r, d = dev[(size_t)r]] {
        (void)hipSetDevice(d);
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          go.wait(lk, [&] { return stop || gen != seen; });
          if (stop) return;
          seen = gen;
          const std::function<int(int)> f = job;
          const int fo = fixed_order;
          lk.unlock();
          fs::tl_fixed_order = fo;              // the caller's solver scope (thread-local there) holds for its ranks' launches too
          const int code = f(r);
          const std::string why = code ? fs_last_error() : "";
          fs::tl_fixed_order = 0;
          lk.lock();
          rc[(size_t)r] = code;
          err[(size_t)r] = why;
          if (--pending == 0) done.notify_one();
        }
      });
  }
  // f(r) on every rank's thread; returns when all are back: the first error (its message becomes the caller's), or FS_OK
  int run(const std::function<int(int)> &f)
  {
    std::unique_lock<std::mutex> lk(m);
    job = f;
    fixed_order = fs::tl_fixed_order;
    pending = n;
    ++gen;
    go.notify_all();
    done.wait(lk, [&] { return pending == 0; });
    for (int r = 0; r < n; ++r)
      if (rc[(size_t)r]) { fs::set_error(err[(size_t)r]); return rc[(size_t)r]; }
    return FS_OK;
  }
  ~RankWorkers()
  {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    go.notify_all();
    for (std::thread &t : th) t.join();
  }
};

struct fs_dist_s {
  int n = 0;
  std::vector<int> dev;               // device of every rank
  std::vector<hipStream_t> stream;    // compute stream per rank, on its device
  std::vector<hipStream_t> comm_stream;  // communication stream per rank
  bool use_rccl = false;
  bool conservative = false;          // one whole-shard all-gather behind the product (FS_DIST_PARTS=1, or after an error)
  bool broken = false;                // an RCCL call failed: the communicators were aborted, every later product is an error return
  std::unique_ptr<RankWorkers> workers;   // FS_DIST_THREADS=1: one issuing thread per rank (else null: the calling thread issues)
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
  int max_rows = 0;                   // the tallest shard
  // per rank, on its device: the shard's rows of the output (room for a padded window behind the last row), the padded
  // receive buffer, the tables of the unpack launches, one event per part
  std::vector<double *> local, pad;
  std::vector<int64_t *> table;       // [rank]: the overlapped layout (nseg segments), then the whole-shard layout (nseg1)
  std::vector<std::vector<hipEvent_t>> ev;   // [rank][part]
  std::vector<hipEvent_t> done;              // [rank]: the unpack on the communication stream
  int nparts = 0;
  std::vector<std::vector<int>> cut;  // [rank][part]: row cuts of the local product (fs_spmv_part_rows)
  std::vector<int> maxc;              // [part]: the largest count of any rank
  std::vector<int64_t> off;           // [part]: first element of the part's region of the padded buffer
  int nseg = 0, nseg1 = 0;
  int64_t max_seg = 0;
  bool built = false;
  unsigned epoch = 0;                 // fs::option_epoch() when the cuts were last checked against the kernels (replan_if_moved)
};

struct CgWork {                       // vectors of fs_dist_cg, kept on the handle between solves
  int F = 0;
  bool ready = false;                 // every allocation below succeeded
  std::vector<double *> sol, r, b, p, q, part, red, redall, st;
};

// k row-major columns on the sharded matrix (fs_dist_spmm, fs_dist_cg2): the replicated X / Y / Z, the shards' local outputs, the
// padded receive buffer, the unpack tables (offsets and counts times k) and the work vectors of the block solver, per rank
// the k-column local product of one side in parts (fs_spmm_part: the one-sweep plan of k = 2, 4 is cut like the single-vector pair),
// with the padded layout of the exchange of every part -- the k-column twin of DistSide's plan; counts in ROWS (x k doubles)
struct KParts {
  int np = 1;
  bool real = false;                    // some rank finishes rows in more than one part: the overlapped exchange has something to hide
  std::vector<std::vector<int>> cut;    // [rank][part]
  std::vector<int> maxc;                // [part]
  std::vector<int64_t> off;             // [part]: first ROW of the part's region of the padded buffer
  int nseg = 0;
  int64_t max_seg = 0;                  // in doubles
  size_t tab_at = 0;                    // where this side's (dst, src, cnt) table starts in KWork::tab
};

struct KWork {
  int k = 0;
  bool with_t = false;
  bool ready = false;                   // every allocation and prepare below succeeded
  unsigned epoch = 0;                   // fs::option_epoch() when the parts were planned
  std::vector<double *> x, y, z, la, lt, pad;
  std::vector<int64_t *> tab;           // [A whole-shard: dst, src, cnt][A' whole-shard][A in parts][A' in parts]
  int nseg_a = 0, nseg_t = 0;
  KParts pa, pt;
  std::vector<double *> sol, r, b, part, red, st;
};

struct fs_dist_matrix_s {
  fs_dist_t D = nullptr;
  int nrow = 0, ncol = 0;
  int64_t nnz = 0;
  // A (always) and A' (fs_dist_matrix_build_transpose[_device], or ANOTHER matrix's A side: fs_dist_matrix_pair).  The sides are
  // shared, not copied, between a pair and the matrices it was made from; the last owner frees the shards (side_owner).
  std::shared_ptr<DistSide> pa, pt;
  DistSide &a, &t;
  fs_dist_matrix_s(std::shared_ptr<DistSide> A, std::shared_ptr<DistSide> T) : pa(std::move(A)), pt(std::move(T)), a(*pa), t(*pt) {}
  // per rank: the whole x (ncol), y (nrow) and z (ncol) on its device.  u of z = A' u is y.
  std::vector<double *> x, y, z;
  // host copies of the CSR are NOT kept
  double *pin = nullptr;              // pinned staging of host vectors
  size_t pin_doubles = 0;
  CgWork cg;
  KWork kw;
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

void free_cg(fs_dist_t D, CgWork &W)
{
  for (size_t d = 0; d < W.sol.size(); ++d) {
    (void)hipSetDevice(D->dev[d]);
    for (double *p : {W.sol[d], W.r[d], W.b[d], W.p[d], W.q[d], W.part[d], W.red[d], W.redall[d], W.st[d]})
      if (p) (void)hipFree(p);
  }
  W = CgWork();
}

void free_k(fs_dist_t D, KWork &W)
{
  for (size_t d = 0; d < W.x.size(); ++d) {
    (void)hipSetDevice(D->dev[d]);
    for (void *p : {(void *)W.x[d], (void *)W.y[d], (void *)W.z[d], (void *)W.la[d], (void *)W.lt[d], (void *)W.pad[d], (void *)W.tab[d], (void *)W.sol[d],
                    (void *)W.r[d], (void *)W.b[d], (void *)W.part[d], (void *)W.red[d], (void *)W.st[d]})
      if (p) (void)hipFree(p);
  }
  W = KWork();
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
int dist_product_threaded(fs_dist_t D, DistSide &S, const std::vector<double *> &in, const std::vector<double *> &out);

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
  S.nseg = 0; S.nseg1 = 0; S.max_seg = 0;
}

// after the shards exist: the parts of every rank's local product, the padded layouts (overlapped: one region per part;
// whole-shard: one region of max_rows per rank), buffers, events
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
  S.max_rows = 0;
  for (int r = 0; r < n; ++r) S.max_rows = std::max(S.max_rows, S.bounds[(size_t)r + 1] - S.bounds[(size_t)r]);
  for (int p = 0; p < nparts; ++p) {
    for (int r = 0; r < n; ++r) S.maxc[(size_t)p] = std::max(S.maxc[(size_t)p], S.cut[(size_t)r][(size_t)p + 1] - S.cut[(size_t)r][(size_t)p]);
    S.off[(size_t)p + 1] = S.off[(size_t)p] + (int64_t)n * S.maxc[(size_t)p];
  }
  std::vector<int64_t> dst, src, cnt;
  S.max_seg = 0;
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
  // the whole-shard layout: rank r's rows at r * max_rows
  dst.clear(); src.clear(); cnt.clear();
  for (int r = 0; r < n; ++r) {
    const int c = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
    if (!c) continue;
    dst.push_back(S.bounds[(size_t)r]);
    src.push_back((int64_t)r * S.max_rows);
    cnt.push_back(c);
  }
  S.nseg1 = (int)cnt.size();
  tab.insert(tab.end(), dst.begin(), dst.end());
  tab.insert(tab.end(), src.begin(), src.end());
  tab.insert(tab.end(), cnt.begin(), cnt.end());
  const int64_t pad_doubles = std::max<int64_t>(S.off[(size_t)nparts], (int64_t)n * S.max_rows);
  S.local.assign((size_t)n, nullptr);
  S.pad.assign((size_t)n, nullptr);
  S.table.assign((size_t)n, nullptr);
  S.ev.assign((size_t)n, std::vector<hipEvent_t>());
  S.done.assign((size_t)n, nullptr);
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    // the send window of a part starts at its first row and is maxc[p] long: it may run past the shard's last row
    FS_HIP(hipMalloc(&S.local[(size_t)r], sizeof(double) * (size_t)(2 * (int64_t)S.max_rows + 1)));
    FS_HIP(hipMalloc(&S.pad[(size_t)r], sizeof(double) * (size_t)(pad_doubles + 1)));
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
  S.epoch = fs::option_epoch();
  return FS_OK;
}

// `count` doubles from every rank's send[r] to every rank's recv[d] + r * count, on the streams `st` (ordered behind the
// events ready[r], one per rank, recorded on whatever stream produced send[r]).  RCCL: one ncclAllGather per rank in one group;
// virtual ranks: device-to-device copies leaving the same layout.  *failed_call (RCCL only): the all-gather call returned an
// error -- the group was closed, nothing of this exchange can be relied on.
int exchange_equal(fs_dist_t D, const std::vector<const double *> &send, const std::vector<double *> &recv, size_t count,
                   const std::vector<hipStream_t> &st, const std::vector<hipEvent_t> &ready, bool inject_failure = false)
{
  const int n = D->n;
  if (D->broken) { fs::set_error("fs_dist: this context is unusable after an RCCL error (its communicators were aborted)"); return FS_ERR_HIP; }
  if (!count) return FS_OK;
  if (D->use_rccl) {
    // every rank's call in one group; an error inside the group still closes it
    ncclResult_t first = rccl().GroupStart();
    if (first != ncclSuccess) return nccl_fail(first, "ncclGroupStart");
    for (int r = 0; r < n && first == ncclSuccess; ++r) {
      if (inject_failure && r == (n > 1 ? 1 : 0)) { first = ncclInternalError; break; }   // (tests: rank 0's call is already in the group)
      first = rccl().AllGather(send[(size_t)r], recv[(size_t)r], count, ncclDouble, D->comm[(size_t)r], st[(size_t)r]);
    }
    const ncclResult_t end = rccl().GroupEnd();
    if (first == ncclSuccess && end == ncclSuccess) return FS_OK;
    // A group call that failed may have launched the collective on some ranks only: nothing issued on these communicators can be
    // relied on to finish, and a new collective on them is undefined.  Abort them (that also ends what is in flight) and leave the
    // context unusable: this product and every later one return the error.  (no second collective on a failed group.)
    const int rc = first != ncclSuccess ? nccl_fail(first, "ncclAllGather") : nccl_fail(end, "ncclGroupEnd");
    const std::string why = fs_last_error();
    for (ncclComm_t &c : D->comm)
      if (c) { if (rccl().CommAbort) (void)rccl().CommAbort(c); c = nullptr; }
    D->broken = true;
    fs::set_error(why + "; the RCCL communicators were aborted: create a new context");
    return rc;
  }
  for (int d = 0; d < n; ++d) {
    // (tests: the failure comes with the first destination's copies already enqueued -- a half-issued exchange)
    if (inject_failure && d == (n > 1 ? 1 : 0)) { fs::set_error("injected failure of an exchange (FS_DIST_FAIL_PART)"); return FS_ERR_HIP; }
    FS_HIP(hipSetDevice(D->dev[d]));
    for (int r = 0; r < n; ++r) {
      if (ready[(size_t)r]) FS_HIP(hipStreamWaitEvent(st[(size_t)d], ready[(size_t)r], 0));
      FS_HIP(hipMemcpyAsync(recv[(size_t)d] + (int64_t)r * (int64_t)count, send[(size_t)r], sizeof(double) * count, hipMemcpyDeviceToDevice,
                            st[(size_t)d]));
    }
  }
  return FS_OK;
}

// the cuts belong to the kernel the options select NOW (strict_order / reproducible / spmv_kernel set since the plan was
// made move the product to another kernel, which finishes its rows elsewhere): plan again when they moved
int replan_if_moved(fs_dist_t D, DistSide &S)
{
  const int n = D->n;
  const unsigned epoch = fs::option_epoch();
  if (S.epoch == epoch) return FS_OK;     // no option was set since the last check: the products run on the same kernels
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
  S.epoch = epoch;
  return FS_OK;
}

// the whole-shard all-gather: rank r's rows [bounds[r], bounds[r + 1]) from src[r] (room for max_rows doubles) into out[d] on
// every rank d.  Runs on the communication streams behind the compute streams' work so far; the compute streams wait for it.
int dist_gather(fs_dist_t D, DistSide &S, const std::vector<double *> &src, const std::vector<double *> &out)
{
  const int n = D->n;
  std::vector<const double *> send((size_t)n);
  std::vector<hipEvent_t> ready((size_t)n);
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    FS_HIP(hipEventRecord(S.ev[(size_t)r][0], D->stream[r]));
    FS_HIP(hipStreamWaitEvent(D->comm_stream[r], S.ev[(size_t)r][0], 0));
    send[(size_t)r] = src[(size_t)r];
    ready[(size_t)r] = S.ev[(size_t)r][0];
  }
  if (int rc = exchange_equal(D, send, S.pad, (size_t)S.max_rows, D->comm_stream, ready)) return rc;
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    if (int rc = fs_copy_segments(S.nseg1, S.table[(size_t)r] + 3 * (size_t)S.nseg, S.max_rows, S.pad[(size_t)r], out[(size_t)r], D->comm_stream[r])) return rc;
    FS_HIP(hipEventRecord(S.done[(size_t)r], D->comm_stream[r]));
    FS_HIP(hipStreamWaitEvent(D->stream[r], S.done[(size_t)r], 0));
  }
  return FS_OK;
}

// out = M in on every rank: local products in parts, the all-gather of every finished part under the later parts, one unpack
// per rank.  in[r] / out[r]: the replicated input and the gathered output on rank r's device.  Asynchronous: the caller
// waits for the compute streams (dist_sync), which wait for the unpacks.
int dist_product(fs_dist_t D, DistSide &S, const std::vector<double *> &in, const std::vector<double *> &out)
{
  const int n = D->n;
  if (D->broken) { fs::set_error("fs_dist: this context is unusable after an RCCL error (its communicators were aborted)"); return FS_ERR_HIP; }
  ++g_products;
  if (int rc = replan_if_moved(D, S)) return rc;
  if (D->workers) return dist_product_threaded(D, S, in, out);
  const int np = S.nparts;
  if (D->conservative || np == 1) {
    // one whole-shard all-gather behind the finished local product: no window reaches into rows still being written
    for (int r = 0; r < n; ++r) {
      const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
      FS_HIP(hipSetDevice(D->dev[r]));
      if (nl > 0)
        for (int p = 0; p < np; ++p)
          if (int rc = fs_spmv_part(S.shard[(size_t)r], 0, S.local[(size_t)r], in[(size_t)r], p, np, D->stream[r])) return rc;
    }
    return dist_gather(D, S, S.local, out);
  }
  for (int p = 0; p < np; ++p) {
    std::vector<const double *> send((size_t)n);
    std::vector<double *> recv((size_t)n);
    std::vector<hipEvent_t> ready((size_t)n);
    for (int r = 0; r < n; ++r) {
      const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
      FS_HIP(hipSetDevice(D->dev[r]));
      if (nl > 0)
        if (int rc = fs_spmv_part(S.shard[(size_t)r], 0, S.local[(size_t)r], in[(size_t)r], p, np, D->stream[r])) return rc;
      FS_HIP(hipEventRecord(S.ev[(size_t)r][(size_t)p], D->stream[r]));
      FS_HIP(hipStreamWaitEvent(D->comm_stream[r], S.ev[(size_t)r][(size_t)p], 0));
      send[(size_t)r] = S.local[(size_t)r] + S.cut[(size_t)r][(size_t)p];
      recv[(size_t)r] = S.pad[(size_t)r] + S.off[(size_t)p];
      ready[(size_t)r] = S.ev[(size_t)r][(size_t)p];
    }
    const bool inject = take_injection(p);
    if (int rc = exchange_equal(D, send, recv, (size_t)S.maxc[(size_t)p], D->comm_stream, ready, inject)) {
      static const bool trace = getenv("FS_TRACE_BUILD") != nullptr;
      D->conservative = true;
      if (D->use_rccl) {            // exchange_equal aborted the communicators: nothing to finish this product with
        if (trace) fprintf(stderr, "[fastsparse] the exchange of part %d failed (%s)\n", p, fs_last_error());
        return rc;
      }
      // virtual ranks (device-to-device copies, one GPU): whatever was enqueued completes by itself, so the product can be finished
      // with ONE whole-shard exchange behind the finished local product -- and the context stays conservative from now on
      if (trace) fprintf(stderr, "[fastsparse] the exchange of part %d failed (%s): conservative mode from here on\n", p, fs_last_error());
      for (int q = p + 1; q < np; ++q)
        for (int r = 0; r < n; ++r) {
          if (S.bounds[(size_t)r + 1] == S.bounds[(size_t)r]) continue;
          FS_HIP(hipSetDevice(D->dev[r]));
          if (int rc2 = fs_spmv_part(S.shard[(size_t)r], 0, S.local[(size_t)r], in[(size_t)r], q, np, D->stream[r])) return rc2;
        }
      if (int rc2 = dist_sync(D)) return rc2;        // whatever of the failed exchange was enqueued has drained
      return dist_gather(D, S, S.local, out);
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

// dist_product with one issuing thread per rank (RankWorkers): the same launches, events and exchange calls, every rank's from its
// own thread.  The rounds end in a meeting of all threads, so a rank's event is recorded before another rank's stream is told to
// wait for it (virtual ranks), and every thread reaches its all-gather of a part (RCCL).  No failure injection on this path.
int dist_product_threaded(fs_dist_t D, DistSide &S, const std::vector<double *> &in, const std::vector<double *> &out)
{
  RankWorkers &W = *D->workers;
  const int n = D->n, np = S.nparts;
  const bool whole = D->conservative || np == 1;
  auto exchange = [&](const std::function<const double *(int)> &send, const std::function<double *(int)> &recv, size_t count, int evp) -> int {
    if (!count) return FS_OK;
    if (D->use_rccl) {
      std::vector<ncclResult_t> res((size_t)n, ncclSuccess);
      (void)W.run([&](int r) -> int { res[(size_t)r] = rccl().AllGather(send(r), recv(r), count, ncclDouble, D->comm[(size_t)r], D->comm_stream[r]); return FS_OK; });
      for (int r = 0; r < n; ++r)
        if (res[(size_t)r] != ncclSuccess) {           // as in exchange_equal: no second collective on communicators in doubt
          const int rc = nccl_fail(res[(size_t)r], "ncclAllGather (issuing threads)");
          const std::string why = fs_last_error();
          for (ncclComm_t &c : D->comm)
            if (c) { if (rccl().CommAbort) (void)rccl().CommAbort(c); c = nullptr; }
          D->broken = true;
          fs::set_error(why + "; the RCCL communicators were aborted: create a new context");
          return rc;
        }
      return FS_OK;
    }
    return W.run([&](int d) -> int {
      for (int r = 0; r < n; ++r) {
        FS_HIP(hipStreamWaitEvent(D->comm_stream[d], S.ev[(size_t)r][(size_t)evp], 0));
        FS_HIP(hipMemcpyAsync(recv(d) + (int64_t)r * (int64_t)count, send(r), sizeof(double) * count, hipMemcpyDeviceToDevice, D->comm_stream[d]));
      }
      return FS_OK;
    });
  };
  for (int p = 0; p < (whole ? 1 : np); ++p) {
    if (int rc = W.run([&](int r) -> int {
          const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
          for (int q = whole ? 0 : p; q < (whole ? np : p + 1); ++q)
            if (nl > 0)
              if (int rc2 = fs_spmv_part(S.shard[(size_t)r], 0, S.local[(size_t)r], in[(size_t)r], q, np, D->stream[r])) return rc2;
          FS_HIP(hipEventRecord(S.ev[(size_t)r][(size_t)p], D->stream[r]));
          FS_HIP(hipStreamWaitEvent(D->comm_stream[r], S.ev[(size_t)r][(size_t)p], 0));
          return FS_OK;
        })) return rc;
    const int rc = whole ? exchange([&](int r) { return (const double *)S.local[(size_t)r]; }, [&](int r) { return S.pad[(size_t)r]; }, (size_t)S.max_rows, 0)
                         : exchange([&](int r) { return (const double *)(S.local[(size_t)r] + S.cut[(size_t)r][(size_t)p]); },
                                    [&](int r) { return S.pad[(size_t)r] + S.off[(size_t)p]; }, (size_t)S.maxc[(size_t)p], p);
    if (rc) return rc;
  }
  return W.run([&](int r) -> int {
    if (int rc = whole ? fs_copy_segments(S.nseg1, S.table[(size_t)r] + 3 * (size_t)S.nseg, S.max_rows, S.pad[(size_t)r], out[(size_t)r], D->comm_stream[r])
                       : fs_copy_segments(S.nseg, S.table[(size_t)r], S.max_seg, S.pad[(size_t)r], out[(size_t)r], D->comm_stream[r])) return rc;
    FS_HIP(hipEventRecord(S.done[(size_t)r], D->comm_stream[r]));
    FS_HIP(hipStreamWaitEvent(D->stream[r], S.done[(size_t)r], 0));
    return FS_OK;
  });
}

int dist_sync(fs_dist_t D)
{
  if (D->workers)
    return D->workers->run([D](int r) -> int {
      FS_HIP(hipStreamSynchronize(D->stream[r]));
      FS_HIP(hipStreamSynchronize(D->comm_stream[r]));
      return FS_OK;
    });
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

// ---- the caller's dense vectors: host memory or HBM of any device --------------------------------------------------------
// where a caller's vector lives: the HIP device ordinal, or -1 for host memory
int vector_device(const void *p)
{
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }   // plain malloc memory: not known to HIP
  return (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) ? a.device : -1;
}

// the caller's work on the legacy default stream of device d has to be done before the ranks' (non-blocking) streams touch its
// vectors: the entry points are synchronous, like the single-GPU drop-in's
int wait_for_caller(int d)
{
  FS_HIP(hipSetDevice(d));
  FS_HIP(hipStreamSynchronize(nullptr));
  return FS_OK;
}

// The input of a product on every rank: in[r].  Host memory goes up to own[r] through the pinned staging buffer (upload_all).  A
// vector in HBM never touches the host: the ranks of ITS device read it in place (in_place), the others get it device to device
// (xGMI) into own[r].
int vec_in(fs_dist_matrix_t M, const std::vector<double *> &own, const double *src, size_t count, std::vector<double *> &in, bool in_place)
{
  fs_dist_t D = M->D;
  in = own;
  const int d = vector_device(src);
  if (d < 0) return upload_all(M, own, src, count);
  if (int rc = wait_for_caller(d)) return rc;
  for (int r = 0; r < D->n; ++r) {
    if (in_place && D->dev[r] == d) { in[(size_t)r] = const_cast<double *>(src); continue; }
    FS_HIP(hipSetDevice(D->dev[r]));
    if (count) FS_HIP(hipMemcpyPeerAsync(own[(size_t)r], D->dev[r], src, d, sizeof(double) * count, D->stream[r]));
  }
  return FS_OK;
}

// The output of a product: every rank ends up with the whole vector (out[r]).  A caller's vector in HBM IS the output vector of the
// first rank on its device (*direct), so the unpack launch writes it and nothing is copied; otherwise out = own and vec_store
// brings rank 0's copy to the caller.
int vec_out(fs_dist_matrix_t M, const std::vector<double *> &own, double *dst, std::vector<double *> &out, int *direct)
{
  fs_dist_t D = M->D;
  out = own;
  *direct = -1;
  const int d = vector_device(dst);
  if (d < 0) return FS_OK;
  if (int rc = wait_for_caller(d)) return rc;
  for (int r = 0; r < D->n && *direct < 0; ++r)
    if (D->dev[r] == d) { out[(size_t)r] = dst; *direct = r; }
  return FS_OK;
}

// count doubles from rank `rank`'s device to the caller: device to host, or device to device when dst is in HBM.  Asynchronous on
// the rank's stream (the caller's dist_sync finishes it).
int vec_store(fs_dist_matrix_t M, int rank, double *dst, const double *src_dev, size_t count)
{
  fs_dist_t D = M->D;
  if (!count) return FS_OK;
  const int d = vector_device(dst);
  if (d < 0) return download_from(M, rank, dst, src_dev, count);
  FS_HIP(hipSetDevice(D->dev[rank]));
  FS_HIP(hipMemcpyPeerAsync(dst, d, src_dev, D->dev[rank], sizeof(double) * count, D->stream[rank]));
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
    fs::KeepCsrScope keep_plain_arrays;     // (option release_csr is not for shards: the exchange layer's own later work -- A' on the
                                            //  devices, a k-column prepare -- reads them)
    S.shard[(size_t)r] = fs_csr_create(hi - lo, ncol, b - a, lrp.data(), cols ? cols + a : nullptr, vals ? vals + a : nullptr, FS_HOST, 0);
    if (!S.shard[(size_t)r]) return FS_ERR_HIP;
  }
  return plan_side(D, S, env_parts());
}

// a side whose last owner -- the matrix it was made for, or a pair that shares it (fs_dist_matrix_pair) -- frees its shards
std::shared_ptr<DistSide> side_owner(fs_dist_t D)
{
  return std::shared_ptr<DistSide>(new DistSide(), [D](DistSide *S) {
    DeviceGuard guard;
    (void)dist_sync(D);
    free_side(D, *S);
    delete S;
  });
}

fs_dist_matrix_t new_dist_matrix(fs_dist_t D, int nrow, int ncol, std::shared_ptr<DistSide> A = nullptr, std::shared_ptr<DistSide> T = nullptr)
{
  fs_dist_matrix_t M = new fs_dist_matrix_s(A ? A : side_owner(D), T ? T : side_owner(D));
  M->D = D; M->nrow = nrow; M->ncol = ncol;
  const int n = D->n;
  M->x.assign((size_t)n, nullptr);
  M->y.assign((size_t)n, nullptr);
  M->z.assign((size_t)n, nullptr);
  return M;
}

bool alloc_xy(fs_dist_matrix_t M)
{
  fs_dist_t D = M->D;
  bool ok = true;
  for (int r = 0; ok && r < D->n; ++r) {
    ok = hipSetDevice(D->dev[r]) == hipSuccess;
    ok = ok && hipMalloc(&M->x[(size_t)r], sizeof(double) * (size_t)(M->ncol ? M->ncol : 1)) == hipSuccess;
    ok = ok && hipMalloc(&M->y[(size_t)r], sizeof(double) * (size_t)(M->nrow ? M->nrow : 1)) == hipSuccess;
    if (!ok) fs::set_error("fs_dist: out of device memory for the replicated vectors");
  }
  return ok;
}

// after the shards of A' exist: their plan and the z vectors
int finish_transpose(fs_dist_matrix_t M)
{
  fs_dist_t D = M->D;
  DistSide &T = M->t;
  int rc = plan_side(D, T, env_parts());
  for (int r = 0; rc == FS_OK && r < D->n; ++r) {
    if (hipSetDevice(D->dev[r]) != hipSuccess || hipMalloc(&M->z[(size_t)r], sizeof(double) * (size_t)(M->ncol ? M->ncol : 1)) != hipSuccess) {
      fs::set_error("fs_dist_matrix_build_transpose: out of device memory");
      rc = FS_ERR_HIP;
    }
  }
  if (rc != FS_OK) {
    for (int r = 0; r < D->n; ++r)
      if (M->z[(size_t)r]) { (void)hipSetDevice(D->dev[r]); (void)hipFree(M->z[(size_t)r]); M->z[(size_t)r] = nullptr; }
    free_side(D, T);
  }
  return rc;
}

// buffers, tables and prepared shards for products with k row-major columns (k >= 2); idempotent per (k, transposed side built)
int ensure_k(fs_dist_matrix_t M, int k)
{
  fs_dist_t D = M->D;
  const int n = D->n;
  KWork &W = M->kw;
  const bool with_t = M->t.built;
  if (W.ready && W.k == k && W.with_t == with_t && W.epoch == fs::option_epoch()) return FS_OK;
  if (int rc = dist_sync(D)) return rc;
  free_k(D, W);
  W.k = k; W.with_t = with_t; W.epoch = fs::option_epoch();
  for (auto *v : {&W.x, &W.y, &W.z, &W.la, &W.lt, &W.pad, &W.sol, &W.r, &W.b, &W.part, &W.red, &W.st}) v->assign((size_t)n, nullptr);
  W.tab.assign((size_t)n, nullptr);
  // the k-column copies of the shards first (fs_spmm itself never builds): the parts below belong to the plan they leave
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    if (M->a.shard[(size_t)r])
      if (int rc = fs_matrix_prepare(M->a.shard[(size_t)r], k, 0, D->stream[r])) return rc;
    if (with_t && M->t.shard[(size_t)r])
      if (int rc = fs_matrix_prepare(M->t.shard[(size_t)r], k, 0, D->stream[r])) return rc;
  }
  const int64_t mr = std::max<int64_t>(M->a.max_rows, with_t ? M->t.max_rows : 0);
  std::vector<int64_t> tab;
  auto side_table = [&](const DistSide &S, int *nseg) {
    std::vector<int64_t> dst, src, cnt;
    for (int r = 0; r < n; ++r) {
      const int64_t c = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
      if (!c) continue;
      dst.push_back((int64_t)S.bounds[(size_t)r] * k);
      src.push_back((int64_t)r * S.max_rows * k);
      cnt.push_back(c * k);
    }
    *nseg = (int)cnt.size();
    tab.insert(tab.end(), dst.begin(), dst.end());
    tab.insert(tab.end(), src.begin(), src.end());
    tab.insert(tab.end(), cnt.begin(), cnt.end());
  };
  side_table(M->a, &W.nseg_a);
  if (with_t) side_table(M->t, &W.nseg_t);
  // the local products in parts and the padded layout of every part's exchange (rows; the tables are in doubles)
  auto side_parts = [&](DistSide &S, KParts &P) -> int {
    P = KParts();
    P.np = S.nparts > 0 ? S.nparts : 1;
    P.cut.assign((size_t)n, std::vector<int>((size_t)P.np + 1, 0));
    for (int r = 0; r < n; ++r) {
      const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
      if (nl <= 0) continue;
      FS_HIP(hipSetDevice(D->dev[r]));
      if (int rc = fs_spmm_part_rows(S.shard[(size_t)r], 0, k, P.np, P.cut[(size_t)r].data())) return rc;
      int busy = 0;
      for (int p = 0; p < P.np; ++p) busy += P.cut[(size_t)r][(size_t)p + 1] > P.cut[(size_t)r][(size_t)p];
      P.real = P.real || busy > 1;
    }
    P.maxc.assign((size_t)P.np, 0);
    P.off.assign((size_t)P.np + 1, 0);
    for (int p = 0; p < P.np; ++p) {
      for (int r = 0; r < n; ++r) P.maxc[(size_t)p] = std::max(P.maxc[(size_t)p], P.cut[(size_t)r][(size_t)p + 1] - P.cut[(size_t)r][(size_t)p]);
      P.off[(size_t)p + 1] = P.off[(size_t)p] + (int64_t)n * P.maxc[(size_t)p];
    }
    std::vector<int64_t> dst, src, cnt;
    for (int p = 0; p < P.np; ++p)
      for (int r = 0; r < n; ++r) {
        const int64_t c = P.cut[(size_t)r][(size_t)p + 1] - P.cut[(size_t)r][(size_t)p];
        if (!c) continue;
        dst.push_back(((int64_t)S.bounds[(size_t)r] + P.cut[(size_t)r][(size_t)p]) * k);
        src.push_back((P.off[(size_t)p] + (int64_t)r * P.maxc[(size_t)p]) * k);
        cnt.push_back(c * k);
        P.max_seg = std::max<int64_t>(P.max_seg, c * k);
      }
    P.nseg = (int)cnt.size();
    P.tab_at = tab.size();
    tab.insert(tab.end(), dst.begin(), dst.end());
    tab.insert(tab.end(), src.begin(), src.end());
    tab.insert(tab.end(), cnt.begin(), cnt.end());
    return FS_OK;
  };
  if (int rc = side_parts(M->a, W.pa)) return rc;
  if (with_t)
    if (int rc = side_parts(M->t, W.pt)) return rc;
  const int64_t pad_rows = std::max<int64_t>((int64_t)n * mr, std::max<int64_t>(W.pa.off[(size_t)W.pa.np], with_t ? W.pt.off[(size_t)W.pt.np] : 0));
  const size_t F = (size_t)(M->ncol ? M->ncol : 1) * k, N = (size_t)(M->nrow ? M->nrow : 1) * k;
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    FS_HIP(hipMalloc(&W.x[(size_t)r], sizeof(double) * F));
    FS_HIP(hipMalloc(&W.y[(size_t)r], sizeof(double) * N));
    FS_HIP(hipMalloc(&W.z[(size_t)r], sizeof(double) * F));
    // (the send window of a part starts at its first row and is maxc[p] rows long: it may run past the shard's last row)
    FS_HIP(hipMalloc(&W.la[(size_t)r], sizeof(double) * (size_t)(2 * (int64_t)M->a.max_rows + 1) * k));
    FS_HIP(hipMalloc(&W.lt[(size_t)r], sizeof(double) * (size_t)(2 * (int64_t)(with_t ? M->t.max_rows : 0) + 1) * k));
    FS_HIP(hipMalloc(&W.pad[(size_t)r], sizeof(double) * (size_t)(pad_rows + 1) * k));
    FS_HIP(hipMalloc(&W.tab[(size_t)r], sizeof(int64_t) * (tab.size() + 1)));
    if (!tab.empty()) FS_HIP(hipMemcpy(W.tab[(size_t)r], tab.data(), sizeof(int64_t) * tab.size(), hipMemcpyHostToDevice));
    FS_HIP(hipMalloc(&W.sol[(size_t)r], sizeof(double) * F));
    FS_HIP(hipMalloc(&W.r[(size_t)r], sizeof(double) * F));
    FS_HIP(hipMalloc(&W.b[(size_t)r], sizeof(double) * F));
    FS_HIP(hipMalloc(&W.part[(size_t)r], sizeof(double) * fs::kCgPartDoubles));
    FS_HIP(hipMalloc(&W.red[(size_t)r], sizeof(double) * 4));
    FS_HIP(hipMalloc(&W.st[(size_t)r], sizeof(double) * fs::kCgStateDoubles));
  }
  W.ready = true;      // (a failure above leaves ready = false: the next call frees what exists and starts over)
  return FS_OK;
}

// the whole-shard exchange of a k-column product: ONE all-gather of max_rows * k doubles per rank behind the finished local product
int dist_gather_k(fs_dist_matrix_t M, bool transposed, const std::vector<double *> &out)
{
  fs_dist_t D = M->D;
  const int n = D->n;
  KWork &W = M->kw;
  DistSide &S = transposed ? M->t : M->a;
  const int k = W.k;
  const std::vector<double *> &loc = transposed ? W.lt : W.la;
  std::vector<const double *> send((size_t)n);
  std::vector<hipEvent_t> ready((size_t)n);
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    FS_HIP(hipEventRecord(S.ev[(size_t)r][0], D->stream[r]));
    FS_HIP(hipStreamWaitEvent(D->comm_stream[r], S.ev[(size_t)r][0], 0));
    send[(size_t)r] = loc[(size_t)r];
    ready[(size_t)r] = S.ev[(size_t)r][0];
  }
  // (the send window is max_rows * k doubles from the start of the local output; the buffer has that room)
  if (int rc = exchange_equal(D, send, W.pad, (size_t)S.max_rows * (size_t)k, D->comm_stream, ready)) return rc;
  const int nseg = transposed ? W.nseg_t : W.nseg_a;
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    const int64_t *tab = W.tab[(size_t)r] + (transposed ? 3 * (size_t)W.nseg_a : 0);
    if (int rc = fs_copy_segments(nseg, tab, (int64_t)S.max_rows * k, W.pad[(size_t)r], out[(size_t)r], D->comm_stream[r])) return rc;
    FS_HIP(hipEventRecord(S.done[(size_t)r], D->comm_stream[r]));
    FS_HIP(hipStreamWaitEvent(D->stream[r], S.done[(size_t)r], 0));
  }
  return FS_OK;
}

// out = M in for k row-major columns on every rank.  Where the local k-column product finishes its rows part by part (fs_spmm_part:
// the one-sweep plan of k = 2, 4 -- the block-CG products) the exchange runs INSIDE the product like dist_product's: the all-gather
// of the rows part p finished under part p + 1, one unpack at the end.  Every other plan (row kernel, column sweeps), FS_DIST_PARTS=1
// and a context gone conservative: the local product, then ONE whole-shard all-gather.
int dist_product_k(fs_dist_matrix_t M, bool transposed, const std::vector<double *> &in, const std::vector<double *> &out)
{
  fs_dist_t D = M->D;
  const int n = D->n;
  KWork &W = M->kw;
  DistSide &S = transposed ? M->t : M->a;
  KParts &P = transposed ? W.pt : W.pa;
  const int k = W.k;
  const std::vector<double *> &loc = transposed ? W.lt : W.la;
  if (D->broken) { fs::set_error("fs_dist: this context is unusable after an RCCL error (its communicators were aborted)"); return FS_ERR_HIP; }
  ++g_products;
  if (D->conservative || P.np <= 1 || !P.real) {
    for (int r = 0; r < n; ++r) {
      const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
      FS_HIP(hipSetDevice(D->dev[r]));
      if (nl > 0)
        if (int rc = fs_spmm(S.shard[(size_t)r], loc[(size_t)r], in[(size_t)r], k, D->stream[r])) return rc;
    }
    return dist_gather_k(M, transposed, out);
  }
  const int np = P.np;
  for (int p = 0; p < np; ++p) {
    std::vector<const double *> send((size_t)n);
    std::vector<double *> recv((size_t)n);
    std::vector<hipEvent_t> ready((size_t)n);
    for (int r = 0; r < n; ++r) {
      const int nl = S.bounds[(size_t)r + 1] - S.bounds[(size_t)r];
      FS_HIP(hipSetDevice(D->dev[r]));
      if (nl > 0)
        if (int rc = fs_spmm_part(S.shard[(size_t)r], 0, loc[(size_t)r], in[(size_t)r], k, p, np, D->stream[r])) return rc;
      FS_HIP(hipEventRecord(S.ev[(size_t)r][(size_t)p], D->stream[r]));
      FS_HIP(hipStreamWaitEvent(D->comm_stream[r], S.ev[(size_t)r][(size_t)p], 0));
      send[(size_t)r] = loc[(size_t)r] + (int64_t)P.cut[(size_t)r][(size_t)p] * k;
      recv[(size_t)r] = W.pad[(size_t)r] + P.off[(size_t)p] * k;
      ready[(size_t)r] = S.ev[(size_t)r][(size_t)p];
    }
    if (int rc = exchange_equal(D, send, recv, (size_t)P.maxc[(size_t)p] * (size_t)k, D->comm_stream, ready, take_injection(p))) {
      D->conservative = true;
      if (D->use_rccl) return rc;          // exchange_equal aborted the communicators
      // virtual ranks: finish the local product, then the whole-shard exchange (see dist_product)
      for (int q = p + 1; q < np; ++q)
        for (int r = 0; r < n; ++r) {
          if (S.bounds[(size_t)r + 1] == S.bounds[(size_t)r]) continue;
          FS_HIP(hipSetDevice(D->dev[r]));
          if (int rc2 = fs_spmm_part(S.shard[(size_t)r], 0, loc[(size_t)r], in[(size_t)r], k, q, np, D->stream[r])) return rc2;
        }
      if (int rc2 = dist_sync(D)) return rc2;
      return dist_gather_k(M, transposed, out);
    }
  }
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    if (int rc = fs_copy_segments(P.nseg, W.tab[(size_t)r] + P.tab_at, P.max_seg, W.pad[(size_t)r], out[(size_t)r], D->comm_stream[r])) return rc;
    FS_HIP(hipEventRecord(S.done[(size_t)r], D->comm_stream[r]));
    FS_HIP(hipStreamWaitEvent(D->stream[r], S.done[(size_t)r], 0));
  }
  return FS_OK;
}

}  // namespace

extern "C" {

fs_dist_t fs_dist_create(int ndev, const int *devices)
{
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { fs::set_error("fs_dist_create: no HIP device"); return nullptr; }
  if (ndev < 1) ndev = visible;
  if (ndev > 64) { fs::set_error("fs_dist_create: at most 64 ranks"); return nullptr; }
  DeviceGuard guard;
  fs_dist_t D = new fs_dist_s();
  D->n = ndev;
  D->conservative = env_parts() == 1;
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
  if (const char *t = getenv("FS_DIST_THREADS"))
    if (*t == '1' && ndev > 1) {
      D->workers.reset(new RankWorkers());
      D->workers->start(D->dev);
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

long fs_debug_dist_products(void) { return g_products.load(); }   // diagnostics, not in include/fastsparse_hip.h

// in how many parts the k-column product of the handle's current k finishes rows (the most any rank has; 1: the exchange simply
// follows the product; 0: no k-column product was made yet)
int fs_debug_dist_k_parts(fs_dist_matrix_t M, int transposed)
{
  if (!M || !M->kw.ready) return 0;
  const KParts &P = transposed ? M->kw.pt : M->kw.pa;
  int most = 0;
  for (const std::vector<int> &c : P.cut) {
    int busy = 0;
    for (size_t p = 0; p + 1 < c.size(); ++p) busy += c[p + 1] > c[p];
    most = std::max(most, busy);
  }
  return most;
}

// the same for the single-vector product of one side (0: no product made on that side yet)
int fs_debug_dist_parts(fs_dist_matrix_t M, int transposed)
{
  if (!M) return 0;
  const DistSide &S = transposed ? M->t : M->a;
  if (!S.built) return 0;
  int most = 0;
  for (const std::vector<int> &c : S.cut) {
    int busy = 0;
    for (size_t p = 0; p + 1 < c.size(); ++p) busy += c[p + 1] > c[p];
    most = std::max(most, busy);
  }
  return most;
}

int fs_dist_ndev(fs_dist_t D) { return D ? D->n : FS_ERR_ARG; }
int fs_dist_uses_rccl(fs_dist_t D) { return D ? (int)D->use_rccl : FS_ERR_ARG; }
int fs_debug_dist_issue_threads(fs_dist_t D) { return D && D->workers ? D->workers->n : 0; }   // diagnostics
int fs_dist_is_conservative(fs_dist_t D) { return D ? (int)D->conservative : FS_ERR_ARG; }

void fs_dist_matrix_destroy(fs_dist_matrix_t M)
{
  if (!M) return;
  DeviceGuard guard;
  (void)dist_sync(M->D);
  free_cg(M->D, M->cg);
  free_k(M->D, M->kw);
  for (size_t r = 0; r < (size_t)M->D->n; ++r) {
    (void)hipSetDevice(M->D->dev[r]);
    if (r < M->x.size() && M->x[r]) (void)hipFree(M->x[r]);
    if (r < M->y.size() && M->y[r]) (void)hipFree(M->y[r]);
    if (r < M->z.size() && M->z[r]) (void)hipFree(M->z[r]);
  }
  if (M->pin) (void)hipHostFree(M->pin);
  delete M;          // the sides go with their last owner (side_owner)
}

fs_dist_matrix_t fs_dist_csr_create(fs_dist_t D, int nrow, int ncol, int64_t nnz, const int *row_ptr, const int *cols,
                                    const double *vals)
{
  if (!D || nrow < 0 || ncol < 0 || nnz < 0 || !row_ptr || (nnz > 0 && !cols)) { fs::set_error("fs_dist_csr_create: bad argument"); return nullptr; }
  DeviceGuard guard;
  fs_dist_matrix_t M = new_dist_matrix(D, nrow, ncol);
  M->nnz = nnz;
  const bool ok = make_shards(D, M->a, nrow, ncol, row_ptr, cols, vals) == FS_OK && alloc_xy(M);
  if (!ok) { fs_dist_matrix_destroy(M); return nullptr; }
  return M;
}

// host COO arrays (optionally valued) -> the same row shards: the entries are bucketed stably by row first -- new_csr / new_bcsr
// (csr.h:375-422, 30-67; on the device from 4 M entries, fs_bucket_coo) -- so every row keeps the caller's entry order, the order
// the serial COO loops add in (sparse.h:58-65, dsparse.h:43-51): what fs_coo_create does on one GPU.  A_mul_B / sdm_A_mul_B /
// bsbm_* / bsdm_* across the GPUs enter here; At_mul_B passes (cols, rows).
fs_dist_matrix_t fs_dist_coo_create(fs_dist_t D, int nrow, int ncol, int64_t nnz, const int *rows, const int *cols, const double *vals)
{
  if (!D || nrow < 0 || ncol < 0 || nnz < 0 || nnz > 0x7fffffffll || (nnz > 0 && (!rows || !cols))) {
    fs::set_error("fs_dist_coo_create: bad argument (0 .. 2^31-1 entries)");
    return nullptr;
  }
  for (int64_t i = 0; i < nnz; ++i)
    if ((unsigned)rows[i] >= (unsigned)nrow) { fs::set_error("fs_dist_coo_create: row out of range"); return nullptr; }
  fs_dist_matrix_t M = nullptr;
  if (vals) {
    struct CSR c;
    new_csr(&c, (long)nnz, nrow, ncol, const_cast<int *>(rows), const_cast<int *>(cols), const_cast<double *>(vals));
    M = fs_dist_csr_create(D, nrow, ncol, nnz, c.row_ptr, c.cols, c.vals);
    free(c.row_ptr); free(c.cols); free(c.vals);
  } else {
    struct BinaryCSR c;
    new_bcsr(&c, (long)nnz, nrow, ncol, const_cast<int *>(rows), const_cast<int *>(cols));
    M = fs_dist_csr_create(D, nrow, ncol, nnz, c.row_ptr, c.cols, nullptr);
    free(c.row_ptr); free(c.cols);
  }
  return M;
}

// A and A' as the caller holds them -- two matrices (bsbm_cg(x, B, Bt, ...), cg.h:25: B and the blocked form of its transpose) --
// as ONE handle for fs_dist_cg / fs_dist_cg2 / fs_dist_ata / the *_t products: its direct side is A's, its transposed side is At's
// DIRECT side (so a row of A' adds in the order the caller's own At stores it, like the single-GPU solvers on two handles).
// Nothing is copied: the shards are shared with A and At and live until the last of the three handles is destroyed.  The pair
// has its own vectors and solver work space.
fs_dist_matrix_t fs_dist_matrix_pair(fs_dist_matrix_t A, fs_dist_matrix_t At)
{
  if (!A || !At || A->D != At->D) { fs::set_error("fs_dist_matrix_pair: NULL handle, or handles of two contexts"); return nullptr; }
  if (A->nrow != At->ncol || A->ncol != At->nrow) { fs::set_error("fs_dist_matrix_pair: the matrices are not each other's transposes in shape"); return nullptr; }
  DeviceGuard guard;
  std::lock_guard<std::mutex> ga(A->lock);
  fs_dist_matrix_t M = new_dist_matrix(A->D, A->nrow, A->ncol, A->pa, At->pa);
  M->nnz = A->nnz;
  bool ok = alloc_xy(M);
  for (int r = 0; ok && r < A->D->n; ++r)
    ok = hipSetDevice(A->D->dev[r]) == hipSuccess && hipMalloc(&M->z[(size_t)r], sizeof(double) * (size_t)(M->ncol ? M->ncol : 1)) == hipSuccess;
  if (!ok) { fs::set_error("fs_dist_matrix_pair: out of device memory for the vectors"); fs_dist_matrix_destroy(M); return nullptr; }
  return M;
}

// The matrix as the caller already holds it: one CSR per rank -- shard r = rows [sum of shard_rows[0 .. r), + shard_rows[r]) of
// A with a LOCAL row_ptr (shard_rows[r] + 1 ints, starting at 0), GLOBAL column ids and optional values.  Nothing here ever
// sees the whole matrix, so the total number of entries may exceed 2^31 - 1 (every shard stays below: int row_ptr,
// csr.h:358-366) -- BASELINE config 5 (3.2 G entries) comes in this way.  space = FS_HOST: host arrays; FS_DEVICE: shard r's
// arrays live on rank r's device (they are copied: the caller may free them).  The caller chooses the cuts (balance by
// non-zeros for power-law matrices).
fs_dist_matrix_t fs_dist_csr_create_from_shards(fs_dist_t D, int nrow, int ncol, const int *shard_rows, const int64_t *shard_nnz,
                                                const int *const *row_ptr, const int *const *cols, const double *const *vals, int space)
{
  if (!D || nrow < 0 || ncol < 0 || !shard_rows || !shard_nnz || !row_ptr || !cols) { fs::set_error("fs_dist_csr_create_from_shards: bad argument"); return nullptr; }
  const int n = D->n;
  int64_t rows = 0, nnz = 0;
  for (int r = 0; r < n; ++r) {
    if (shard_rows[r] < 0 || shard_nnz[r] < 0 || shard_nnz[r] > 0x7fffffffll || !row_ptr[r] || (shard_nnz[r] > 0 && !cols[r])) {
      fs::set_error("fs_dist_csr_create_from_shards: bad shard (each holds 0 .. 2^31-1 entries)");
      return nullptr;
    }
    rows += shard_rows[r];
    nnz += shard_nnz[r];
  }
  if (rows != nrow) { fs::set_error("fs_dist_csr_create_from_shards: the shards' rows do not add up to nrow"); return nullptr; }
  // valued or pattern-only: what the first NON-EMPTY shard says (the caller chooses the cuts: an empty first shard is legal, and an
  // empty shard's vals pointer means nothing)
  bool valued = false;
  for (int r = 0; r < n; ++r)
    if (shard_nnz[r] > 0) { valued = vals && vals[r] != nullptr; break; }
  for (int r = 0; r < n; ++r)
    if (shard_nnz[r] > 0 && (valued != (vals && vals[r] != nullptr))) { fs::set_error("fs_dist_csr_create_from_shards: values for some shards only"); return nullptr; }
  DeviceGuard guard;
  // a shard's row_ptr must end at its shard_nnz (the device kernels trust both): one int per shard, read where it lives
  for (int r = 0; r < n; ++r) {
    int last = 0;
    if (space == FS_DEVICE) {
      if (hipSetDevice(D->dev[r]) != hipSuccess || hipMemcpy(&last, row_ptr[r] + shard_rows[r], sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        fs::set_error("fs_dist_csr_create_from_shards: cannot read a shard's row_ptr on its device");
        return nullptr;
      }
    } else last = row_ptr[r][shard_rows[r]];
    if ((int64_t)last != shard_nnz[r]) { fs::set_error("fs_dist_csr_create_from_shards: row_ptr[shard_rows] of a shard is not its shard_nnz"); return nullptr; }
  }
  fs_dist_matrix_t M = new_dist_matrix(D, nrow, ncol);
  M->nnz = nnz;
  DistSide &S = M->a;
  S.nrow = nrow; S.ncol = ncol;
  S.bounds.assign((size_t)n + 1, 0);
  S.shard.assign((size_t)n, nullptr);
  S.shard_nnz.assign((size_t)n, 0);
  bool ok = true;
  for (int r = 0; ok && r < n; ++r) {
    S.bounds[(size_t)r + 1] = S.bounds[(size_t)r] + shard_rows[r];
    S.shard_nnz[(size_t)r] = shard_nnz[r];
    ok = hipSetDevice(D->dev[r]) == hipSuccess;
    fs::KeepCsrScope keep_plain_arrays;     // (option release_csr is not for shards: the exchange layer's own later work -- A' on the
                                            //  devices, a k-column prepare -- reads them)
    if (ok) S.shard[(size_t)r] = fs_csr_create(shard_rows[r], ncol, shard_nnz[r], row_ptr[r], cols[r], valued ? vals[r] : nullptr, space, 0);
    ok = ok && S.shard[(size_t)r] != nullptr;
  }
  ok = ok && plan_side(D, S, env_parts()) == FS_OK && alloc_xy(M);
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
      fs::KeepCsrScope keep_plain_arrays;     // (option release_csr is not for shards: the exchange layer's own later work -- A' on the
                                              //  devices, a k-column prepare -- reads them)
      T.shard[(size_t)r] = fs_csr_create(hi - lo, nrow, cnt, lrp.data(), lc.data(), vals ? lv.data() : nullptr, FS_HOST, 0);
      if (!T.shard[(size_t)r]) { rcs[(size_t)r] = FS_ERR_HIP; errs[(size_t)r] = fs_last_error(); }
    });
  for (std::thread &t : th) t.join();
  int rc = FS_OK;
  for (int r = 0; r < n; ++r)
    if (rcs[(size_t)r] != FS_OK) { rc = rcs[(size_t)r]; fs::set_error("fs_dist_matrix_build_transpose: " + errs[(size_t)r]); }
  if (rc != FS_OK) { free_side(D, T); return rc; }
  return finish_transpose(M);
}

// The same shards of A' with no host array of the matrix anywhere: built from the device-resident shards of A.
//   1  every rank counts its entries per column; the counts are added up on rank 0's device and cut by non-zeros there
//      (the cut of nnz_cut: the same bounds fs_dist_matrix_build_transpose finds on the host);
//   2  every rank partitions its entries stably by the rank that will own their column (fs::shard_transpose_partition);
//   3  the parts travel device to device (hipMemcpyPeerAsync: one process, so no collective is needed for a one-time build;
//      with virtual ranks these are copies inside one device), concatenated per destination in source-rank order;
//   4  every rank orders what it received by row of A' -- fs_coo_create's stable sort -- which leaves each row of A' in
//      ascending A-row order.
// Temporary HBM per rank: about 30 bytes per entry sent plus 16 per entry received.  Idempotent.
int fs_dist_matrix_build_transpose_device(fs_dist_matrix_t M)
{
  if (!M) { fs::set_error("fs_dist_matrix_build_transpose_device: NULL handle"); return FS_ERR_ARG; }
  std::lock_guard<std::mutex> g(M->lock);
  if (M->t.built) return FS_OK;
  DeviceGuard guard;
  fs_dist_t D = M->D;
  const int n = D->n, nrow = M->nrow, ncol = M->ncol;
  if (int rc = dist_sync(D)) return rc;
  const bool valued = [&] { for (int r = 0; r < n; ++r) if (M->a.shard_nnz[(size_t)r] > 0) return M->a.shard[(size_t)r]->a.has_vals(); return false; }();
  for (int r = 0; r < n; ++r)
    if (M->a.shard[(size_t)r])
      if (int rc = fs::need_plain_csr(M->a.shard[(size_t)r]->a, "fs_dist_matrix_build_transpose_device")) return rc;
  DistSide &T = M->t;
  T.nrow = ncol; T.ncol = nrow;
  T.bounds.assign((size_t)n + 1, 0);
  T.shard.assign((size_t)n, nullptr);
  T.shard_nnz.assign((size_t)n, 0);
  struct Tmp {
    fs_dist_t D;
    std::vector<int *> cnt, trow, tcol, rrow, rcol;
    std::vector<double *> tval, rval;
    ~Tmp()
    {
      for (size_t r = 0; r < cnt.size(); ++r) {
        (void)hipSetDevice(D->dev[r]);
        for (void *p : {(void *)cnt[r], (void *)trow[r], (void *)tcol[r], (void *)rrow[r], (void *)rcol[r], (void *)tval[r], (void *)rval[r]})
          if (p) (void)hipFree(p);
      }
    }
  } W{D};
  W.cnt.assign((size_t)n, nullptr); W.trow.assign((size_t)n, nullptr); W.tcol.assign((size_t)n, nullptr);
  W.rrow.assign((size_t)n, nullptr); W.rcol.assign((size_t)n, nullptr);
  W.tval.assign((size_t)n, nullptr); W.rval.assign((size_t)n, nullptr);
  // 1: column counts, added up on rank 0's device
  const size_t cbytes = sizeof(int) * (size_t)(ncol ? ncol : 1);
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    FS_HIP(hipMalloc(&W.cnt[(size_t)r], cbytes));
    FS_HIP(hipMemsetAsync(W.cnt[(size_t)r], 0, cbytes, D->stream[r]));
    if (int rc = fs::shard_column_counts(M->a.shard[(size_t)r]->a, W.cnt[(size_t)r], D->stream[r])) return rc;
  }
  if (int rc = dist_sync(D)) return rc;
  {
    int *tmp = nullptr;
    FS_HIP(hipSetDevice(D->dev[0]));
    if (n > 1) FS_HIP(hipMalloc(&tmp, cbytes));
    int rc = FS_OK;
    for (int r = 1; r < n && rc == FS_OK; ++r) {
      if (hipMemcpyPeerAsync(tmp, D->dev[0], W.cnt[(size_t)r], D->dev[r], cbytes, D->stream[0]) != hipSuccess) { fs::set_error("fs_dist_matrix_build_transpose_device: peer copy of the column counts failed"); rc = FS_ERR_HIP; break; }
      rc = fs::add_counts(ncol, W.cnt[0], tmp, D->stream[0]);
    }
    int64_t total = 0;
    if (rc == FS_OK) rc = fs::cut_by_counts(ncol, W.cnt[0], n, T.bounds.data(), &total, D->stream[0]);
    if (tmp) (void)hipFree(tmp);
    if (rc != FS_OK) { T = DistSide(); return rc; }
    if (total != M->nnz) { fs::set_error("fs_dist_matrix_build_transpose_device: the column counts do not add up to nnz"); T = DistSide(); return FS_ERR_ARG; }
  }
  // 2: every rank's entries by destination
  std::vector<std::vector<int64_t>> count((size_t)n, std::vector<int64_t>((size_t)n, 0));   // [source][destination]
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    (void)hipFree(W.cnt[(size_t)r]); W.cnt[(size_t)r] = nullptr;
    if (int rc = fs::shard_transpose_partition(M->a.shard[(size_t)r]->a, M->a.bounds[(size_t)r], n, T.bounds.data(), &W.trow[(size_t)r], &W.tcol[(size_t)r],
                                               &W.tval[(size_t)r], count[(size_t)r].data(), D->stream[r])) { T = DistSide(); return rc; }
  }
  // 3: the exchange
  for (int d = 0; d < n; ++d) {
    int64_t cnt = 0;
    for (int r = 0; r < n; ++r) cnt += count[(size_t)r][(size_t)d];
    T.shard_nnz[(size_t)d] = cnt;
    if (cnt > 0x7fffffffll) { fs::set_error("fs_dist_matrix_build_transpose_device: a shard of A' holds more than 2^31-1 non-zeros: use more devices"); T = DistSide(); return FS_ERR_ARG; }
    FS_HIP(hipSetDevice(D->dev[d]));
    const size_t c1 = (size_t)(cnt ? cnt : 1);
    FS_HIP(hipMalloc(&W.rrow[(size_t)d], sizeof(int) * c1));
    FS_HIP(hipMalloc(&W.rcol[(size_t)d], sizeof(int) * c1));
    if (valued) FS_HIP(hipMalloc(&W.rval[(size_t)d], sizeof(double) * c1));
  }
  for (int r = 0; r < n; ++r) {
    FS_HIP(hipSetDevice(D->dev[r]));
    int64_t soff = 0;
    for (int d = 0; d < n; ++d) {
      const int64_t c = count[(size_t)r][(size_t)d];
      int64_t doff = 0;
      for (int q = 0; q < r; ++q) doff += count[(size_t)q][(size_t)d];
      if (c > 0) {
        FS_HIP(hipMemcpyPeerAsync(W.rrow[(size_t)d] + doff, D->dev[d], W.trow[(size_t)r] + soff, D->dev[r], sizeof(int) * (size_t)c, D->stream[r]));
        FS_HIP(hipMemcpyPeerAsync(W.rcol[(size_t)d] + doff, D->dev[d], W.tcol[(size_t)r] + soff, D->dev[r], sizeof(int) * (size_t)c, D->stream[r]));
        if (valued) FS_HIP(hipMemcpyPeerAsync(W.rval[(size_t)d] + doff, D->dev[d], W.tval[(size_t)r] + soff, D->dev[r], sizeof(double) * (size_t)c, D->stream[r]));
      }
      soff += c;
    }
  }
  if (int rc = dist_sync(D)) { T = DistSide(); return rc; }
  // 4: the local builds
  for (int d = 0; d < n; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    for (void **p : {(void **)&W.trow[(size_t)d], (void **)&W.tcol[(size_t)d], (void **)&W.tval[(size_t)d]})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
    fs::KeepCsrScope keep_plain_arrays;     // (option release_csr is not for shards: the exchange layer's own later work -- A' on the
                                            //  devices, a k-column prepare -- reads them)
    T.shard[(size_t)d] = fs_coo_create(T.bounds[(size_t)d + 1] - T.bounds[(size_t)d], nrow, T.shard_nnz[(size_t)d], W.rrow[(size_t)d], W.rcol[(size_t)d],
                                       valued ? W.rval[(size_t)d] : nullptr, FS_DEVICE);
    if (!T.shard[(size_t)d]) { free_side(D, T); return FS_ERR_HIP; }
    for (void **p : {(void **)&W.rrow[(size_t)d], (void **)&W.rcol[(size_t)d], (void **)&W.rval[(size_t)d]})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
  }
  return finish_transpose(M);
}

int fs_dist_matrix_has_transpose(fs_dist_matrix_t M) { return M && M->t.built; }

int fs_dist_matrix_bounds(fs_dist_matrix_t M, int *bounds)
{
  if (!M || !bounds) return FS_ERR_ARG;
  for (size_t i = 0; i < M->a.bounds.size(); ++i) bounds[i] = M->a.bounds[i];
  return FS_OK;
}

int fs_dist_matrix_bounds_t(fs_dist_matrix_t M, int *bounds)
{
  if (!M || !bounds || !M->t.built) return FS_ERR_ARG;
  for (size_t i = 0; i < M->t.bounds.size(); ++i) bounds[i] = M->t.bounds[i];
  return FS_OK;
}

int64_t fs_dist_matrix_shard_nnz(fs_dist_matrix_t M, int rank)
{
  if (!M || rank < 0 || rank >= M->D->n) return FS_ERR_ARG;
  return M->a.shard_nnz[(size_t)rank];
}

int64_t fs_dist_matrix_nnz(fs_dist_matrix_t M) { return M ? M->nnz : FS_ERR_ARG; }

fs_matrix_t fs_dist_matrix_shard(fs_dist_matrix_t M, int rank, int transposed)
{
  if (!M || rank < 0 || rank >= M->D->n || (transposed && !M->t.built)) return nullptr;
  return (transposed ? M->t : M->a).shard[(size_t)rank];
}

// y = M x with the caller's vectors in host memory or in HBM (vec_in / vec_out): one direction of the matrix
static int dist_apply(fs_dist_matrix_t M, bool transposed, double *y, const double *x)
{
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  DistSide &S = transposed ? M->t : M->a;
  // (u of z = A' u lives where y does: A then A' chains without a copy)
  const std::vector<double *> &own_in = transposed ? M->y : M->x, &own_out = transposed ? M->z : M->y;
  std::vector<double *> in, out;
  int direct = -1;
  if (int rc = vec_in(M, own_in, x, (size_t)S.ncol, in, true)) return rc;
  if (int rc = vec_out(M, own_out, y, out, &direct)) return rc;
  if (int rc = dist_product(M->D, S, in, out)) return rc;
  if (direct < 0)
    if (int rc = vec_store(M, 0, y, own_out[0], (size_t)S.nrow)) return rc;
  return dist_sync(M->D);
}

int fs_dist_spmv(fs_dist_matrix_t M, double *y, const double *x)
{
  if (!M || !y || !x) { fs::set_error("fs_dist_spmv: NULL argument"); return FS_ERR_ARG; }
  return dist_apply(M, false, y, x);
}

int fs_dist_spmv_t(fs_dist_matrix_t M, double *z, const double *u)
{
  if (!M || !z || !u) { fs::set_error("fs_dist_spmv_t: NULL argument"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_spmv_t: call fs_dist_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  return dist_apply(M, true, z, u);
}

// z[ncol] = A'(A x[ncol]) + lambda x (bcsr_AA_mul_B, csr.h:305-319, with lambda = 0; bsbm_AtA, cg.h:9-22, across the GPUs): y = A x
// stays on the devices and is the u of the second product in place; two exchanges and one axpy on the rank that holds the output
int fs_dist_ata(fs_dist_matrix_t M, double *z, const double *x, double lambda)
{
  if (!M || !z || !x) { fs::set_error("fs_dist_ata: NULL argument"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_ata: call fs_dist_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  std::vector<double *> in, out;
  int direct = -1;
  if (int rc = vec_in(M, M->x, x, (size_t)M->ncol, in, true)) return rc;
  if (int rc = vec_out(M, M->z, z, out, &direct)) return rc;
  if (int rc = dist_product(M->D, M->a, in, M->y)) return rc;
  if (int rc = dist_product(M->D, M->t, M->y, out)) return rc;
  if (lambda != 0.0) {
    const int r = direct < 0 ? 0 : direct;
    FS_HIP(hipSetDevice(M->D->dev[r]));
    if (int rc = fs_axpy(M->ncol, lambda, in[(size_t)r], out[(size_t)r], M->D->stream[r])) return rc;
  }
  if (direct < 0)
    if (int rc = vec_store(M, 0, z, M->z[0], (size_t)M->ncol)) return rc;
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

// (A'A + lambda I) x = b on the row-sharded matrix: bsbm_cg (cg.h:25-82) across the GPUs, everything resident, the scalars of
// the iteration on the devices (fs_cg.hip).  Per iteration y = A p and q = A' y; two schemes for everything else (option
// "dist_cg_scheme"):
//   0 "replicate"  every device keeps the WHOLE x, r, p, q and runs the same fused vector kernels on them -- identical inputs,
//                  identical kernels, identical vectors: the dots need no exchange.  Both products carry their all-gather.
//                  The O(F) vector work is done N times over.
//   1 "gather"     (libfastsparse_amd/dist.py ShardedCG's scheme) rank r keeps ITS SLICE of x, r, p, q -- the rows of A' it owns.
//                  q_r = A'_r y needs no exchange; the dots are partial: every rank's partial is all-gathered (8 bytes per
//                  rank) and every rank adds the N values in rank order, so all ranks hold identical scalars; the new p slice is
//                  all-gathered (whole-shard exchange) for the next product.  The vector work is divided by N; two tiny
//                  exchanges per iteration are added.
// Either way EVERY device decides convergence for itself and the host compares the flags of all of them one iteration behind:
// a disagreement (which identical arithmetic rules out, and a faulty device or exchange does not) is an error return, never a
// rank waiting in a collective for one that left.  Products run with fixed-order sums unless option "cg_fixed_order" is 0.
// Work vectors are kept on the handle between solves.  CLOBBERS fs_dist_x / fs_dist_y / fs_dist_z of the handle.
int fs_dist_cg(fs_dist_matrix_t M, double *x_host, const double *b_host, double lambda, double tol, int *out_iter)
{
  if (!M || !x_host || !b_host) { fs::set_error("fs_dist_cg: NULL argument"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_cg: call fs_dist_matrix_build_transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  fs::FixedOrderScope fixed(fs::options().cg_fixed_order != 0);
  fs_dist_t D = M->D;
  const int n = D->n, F = M->ncol;
  const bool gather = fs::options().dist_cg_scheme == 1 && n > 1;
  DistSide &T = M->t;
  CgWork &W = M->cg;
  if (!W.ready || W.F != F) {
    free_cg(D, W);
    for (auto *v : {&W.sol, &W.r, &W.b, &W.p, &W.q, &W.part, &W.red, &W.redall, &W.st}) v->assign((size_t)n, nullptr);
    W.F = F;
    const size_t fl = (size_t)(F ? F : 1) + (size_t)T.max_rows + 1;       // (a slice is sent as a window of max_rows doubles)
    for (int d = 0; d < n; ++d) {
      FS_HIP(hipSetDevice(D->dev[d]));
      FS_HIP(hipMalloc(&W.sol[(size_t)d], sizeof(double) * fl));
      FS_HIP(hipMalloc(&W.r[(size_t)d], sizeof(double) * fl));
      FS_HIP(hipMalloc(&W.b[(size_t)d], sizeof(double) * fl));
      FS_HIP(hipMalloc(&W.p[(size_t)d], sizeof(double) * fl));
      FS_HIP(hipMalloc(&W.q[(size_t)d], sizeof(double) * fl));
      FS_HIP(hipMalloc(&W.part[(size_t)d], sizeof(double) * fs::kCgPartDoubles));
      FS_HIP(hipMalloc(&W.red[(size_t)d], sizeof(double) * 4));
      FS_HIP(hipMalloc(&W.redall[(size_t)d], sizeof(double) * (size_t)(n + 1)));
      FS_HIP(hipMalloc(&W.st[(size_t)d], sizeof(double) * fs::kCgStateDoubles));
    }
    W.ready = true;
  }
  {
    std::vector<double *> unused;
    if (int rc = vec_in(M, W.b, b_host, (size_t)F, unused, false)) return rc;    // (host memory or HBM; always a copy: b is read all solve long)
  }
  // one set of host-visible flags per device
  std::vector<fs::CgFlags> fl((size_t)n);
  for (int d = 0; d < n; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    if (int rc = fl[(size_t)d].init()) return rc;
  }
  std::vector<hipEvent_t> none((size_t)n, nullptr);
  // the sum over the ranks of every rank's red[0] (left there by a partial reduction), in rank order, then the scalar step `mode`
  // (send slots red[0] / red[1] alternate: a rank may be one exchange ahead of another that still reads its previous partial;
  // the scalar step leaves its sum in red[2])
  int slot = 0;
  auto combine = [&](int mode, double arg) -> int {
    std::vector<const double *> send((size_t)n);
    std::vector<hipEvent_t> ready((size_t)n);
    for (int d = 0; d < n; ++d) {
      FS_HIP(hipSetDevice(D->dev[d]));
      send[(size_t)d] = W.red[(size_t)d] + slot;
      FS_HIP(hipEventRecord(T.ev[(size_t)d][0], D->stream[d]));
      ready[(size_t)d] = T.ev[(size_t)d][0];
    }
    if (int rc = exchange_equal(D, send, W.redall, 1, D->stream, ready)) return rc;
    for (int d = 0; d < n; ++d) {
      FS_HIP(hipSetDevice(D->dev[d]));
      if (int rc = fs::cg_dev_final(mode, W.redall[(size_t)d], n, W.red[(size_t)d] + 2, W.st[(size_t)d], arg, D->stream[d])) return rc;
    }
    slot ^= 1;
    return FS_OK;
  };
  if (!gather) {
    for (int d = 0; d < n; ++d) {
      FS_HIP(hipSetDevice(D->dev[d]));
      if (int rc = fs::cg_dev_init(F, W.b[(size_t)d], W.sol[(size_t)d], W.r[(size_t)d], M->x[(size_t)d], W.part[(size_t)d], W.red[(size_t)d],
                                   W.st[(size_t)d], tol, D->stream[d])) return rc;
    }
  } else {
    for (int d = 0; d < n; ++d) {
      const int lo = T.bounds[(size_t)d], nl = T.bounds[(size_t)d + 1] - lo;
      FS_HIP(hipSetDevice(D->dev[d]));
      if (int rc = fs::cg_dev_init_partial(nl, W.b[(size_t)d] + lo, W.sol[(size_t)d], W.r[(size_t)d], W.p[(size_t)d], W.part[(size_t)d], W.red[(size_t)d] + slot,
                                           D->stream[d])) return rc;
    }
    if (int rc = combine(0, tol)) return rc;
    if (int rc = dist_gather(D, T, W.p, M->x)) return rc;                // the whole p on every rank
  }
  int rc_loop = FS_OK;
  for (int iter = 0; iter < F; iter++) {
    if (int rc = dist_product(D, M->a, M->x, M->y)) return rc;           // y = A p, its all-gather inside
    if (!gather) {
      if (int rc = dist_product(D, T, M->y, M->z)) return rc;            // q = A' y, its all-gather inside
      for (int d = 0; d < n; ++d) {
        FS_HIP(hipSetDevice(D->dev[d]));
        if (int rc = fs::cg_dev_steps(F, lambda, W.sol[(size_t)d], W.r[(size_t)d], M->x[(size_t)d], M->z[(size_t)d], W.part[(size_t)d],
                                      W.red[(size_t)d], W.st[(size_t)d], D->stream[d])) return rc;
      }
    } else {
      for (int d = 0; d < n; ++d) {                                      // q_r = A'_r y: this rank's rows of A', no exchange
        const int nl = T.bounds[(size_t)d + 1] - T.bounds[(size_t)d];
        FS_HIP(hipSetDevice(D->dev[d]));
        if (nl > 0)
          if (int rc = fs_spmv(T.shard[(size_t)d], W.q[(size_t)d], M->y[(size_t)d], D->stream[d])) return rc;
        if (int rc = fs::cg_dev_step_a(nl, lambda, W.p[(size_t)d], W.q[(size_t)d], W.part[(size_t)d], W.red[(size_t)d] + slot, W.st[(size_t)d], D->stream[d])) return rc;
      }
      if (int rc = combine(1, 0.0)) return rc;                            // alpha
      for (int d = 0; d < n; ++d) {
        const int nl = T.bounds[(size_t)d + 1] - T.bounds[(size_t)d];
        FS_HIP(hipSetDevice(D->dev[d]));
        if (int rc = fs::cg_dev_step_b(nl, W.sol[(size_t)d], W.r[(size_t)d], W.p[(size_t)d], W.q[(size_t)d], W.part[(size_t)d], W.red[(size_t)d] + slot,
                                       W.st[(size_t)d], D->stream[d])) return rc;
      }
      if (int rc = combine(2, 0.0)) return rc;                            // converged?  beta
      for (int d = 0; d < n; ++d) {
        const int nl = T.bounds[(size_t)d + 1] - T.bounds[(size_t)d];
        FS_HIP(hipSetDevice(D->dev[d]));
        if (int rc = fs::cg_dev_step_c(nl, W.p[(size_t)d], W.r[(size_t)d], W.st[(size_t)d], D->stream[d])) return rc;
      }
      if (int rc = dist_gather(D, T, W.p, M->x)) return rc;              // the new p on every rank
    }
    int stops = 0;
    for (int d = 0; d < n; ++d) {
      bool stop = false;
      FS_HIP(hipSetDevice(D->dev[d]));
      if (int rc = fl[(size_t)d].after_iteration(iter, W.st[(size_t)d], D->stream[d], &stop)) return rc;
      stops += stop ? 1 : 0;
    }
    if (stops != 0 && stops != n) {
      fs::set_error("fs_dist_cg: the devices disagree about convergence (a device or an exchange returned different bits)");
      rc_loop = FS_ERR_HIP;
      break;
    }
    if (stops == n) break;
  }
  if (int rc = dist_sync(D)) return rc;
  if (rc_loop != FS_OK) return rc_loop;
  // every device's final {done, iterations} must agree as well
  std::vector<double> fin(2 * (size_t)n, 0.0);
  for (int d = 0; d < n; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    FS_HIP(hipMemcpy(&fin[2 * (size_t)d], W.st[(size_t)d] + fs::kCgStateDone, sizeof(double) * 2, hipMemcpyDeviceToHost));
    if (fin[2 * (size_t)d] != fin[0] || fin[2 * (size_t)d + 1] != fin[1]) {
      fs::set_error("fs_dist_cg: the devices finished in different states");
      return FS_ERR_HIP;
    }
  }
  if (vector_device(x_host) >= 0)
    if (int rc = wait_for_caller(vector_device(x_host))) return rc;
  if (!gather) {
    if (int rc = vec_store(M, 0, x_host, W.sol[0], (size_t)F)) return rc;
  } else {
    for (int d = 0; d < n; ++d) {
      const int lo = T.bounds[(size_t)d], nl = T.bounds[(size_t)d + 1] - lo;
      if (nl > 0)
        if (int rc = vec_store(M, d, x_host + lo, W.sol[(size_t)d], (size_t)nl)) return rc;
    }
  }
  if (int rc = dist_sync(D)) return rc;
  if (out_iter) *out_iter = (int)fin[1];
  return FS_OK;
}

// Y[nrow, k] = A X[ncol, k], row-major host matrices (csr_A_mul_Bn / bcsr_A_mul_Bn / bsbm_A_mul_Bn across the GPUs, csr.h:441,
// 257; sparse.h:318): every device multiplies its shard (the k-column kernels of fs_spmm, prepared on first use of a k) and the
// Y shards are all-gathered, one whole-shard exchange behind the local product.  fs_dist_spmm_t: Z[ncol, k] = A' U[nrow, k].
static int dist_apply_k(fs_dist_matrix_t M, bool transposed, double *Y, const double *X, int k)
{
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  if (int rc = ensure_k(M, k)) return rc;
  DistSide &S = transposed ? M->t : M->a;
  KWork &W = M->kw;
  const std::vector<double *> &own_in = transposed ? W.y : W.x, &own_out = transposed ? W.z : W.y;
  std::vector<double *> in, out;
  int direct = -1;
  if (int rc = vec_in(M, own_in, X, (size_t)S.ncol * k, in, true)) return rc;
  if (int rc = vec_out(M, own_out, Y, out, &direct)) return rc;
  if (int rc = dist_product_k(M, transposed, in, out)) return rc;
  if (direct < 0)
    if (int rc = vec_store(M, 0, Y, own_out[0], (size_t)S.nrow * k)) return rc;
  return dist_sync(M->D);
}

int fs_dist_spmm(fs_dist_matrix_t M, double *Y, const double *X, int k)
{
  if (!M || !Y || !X || k < 1) { fs::set_error("fs_dist_spmm: bad argument"); return FS_ERR_ARG; }
  if (k == 1) return fs_dist_spmv(M, Y, X);
  return dist_apply_k(M, false, Y, X, k);
}

int fs_dist_spmm_t(fs_dist_matrix_t M, double *Z, const double *U, int k)
{
  if (!M || !Z || !U || k < 1) { fs::set_error("fs_dist_spmm_t: bad argument"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_spmm_t: build the transpose first"); return FS_ERR_NO_TRANSPOSE; }
  if (k == 1) return fs_dist_spmv_t(M, Z, U);
  return dist_apply_k(M, true, Z, U, k);
}

// (A'A + lambda I) X = B for two right-hand sides, row-major ncol x 2: bsbm_cg2 (cg.h:85-187) across the GPUs, everything
// resident.  Every device keeps whole X, R, P, Q and runs the same block-CG steps on them (identical inputs, identical kernels:
// the 2x2 algebra needs no exchange); per iteration TMP = A P and Q = A' TMP as two-column products with their all-gathers.
// Convergence is decided on every device and compared on the host like fs_dist_cg.  Products add in a fixed order unless
// option "cg_fixed_order" is 0.
int fs_dist_cg2(fs_dist_matrix_t M, double *X_host, const double *B_host, double lambda, double tol, int *out_iter)
{
  if (!M || !X_host || !B_host) { fs::set_error("fs_dist_cg2: NULL argument"); return FS_ERR_ARG; }
  if (!M->t.built) { fs::set_error("fs_dist_cg2: build the transpose first"); return FS_ERR_NO_TRANSPOSE; }
  std::lock_guard<std::mutex> g(M->lock);
  std::lock_guard<std::mutex> gd(M->D->lock);
  DeviceGuard guard;
  fs::FixedOrderScope fixed(fs::options().cg_fixed_order != 0);
  fs_dist_t D = M->D;
  const int n = D->n, F = M->ncol;
  if (int rc = ensure_k(M, 2)) return rc;
  KWork &W = M->kw;
  {
    std::vector<double *> unused;
    if (int rc = vec_in(M, W.b, B_host, (size_t)F * 2, unused, false)) return rc;
  }
  std::vector<fs::CgFlags> fl((size_t)n);
  std::vector<double> norms(2 * (size_t)n, 0.0);
  for (int d = 0; d < n; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    if (int rc = fl[(size_t)d].init()) return rc;
    if (int rc = fs::cg2_dev_init(F, W.b[(size_t)d], W.sol[(size_t)d], W.r[(size_t)d], W.x[(size_t)d], W.part[(size_t)d], W.red[(size_t)d],
                                  W.st[(size_t)d], tol, &norms[2 * (size_t)d], D->stream[d])) return rc;
    if (norms[2 * (size_t)d] != norms[0] || norms[2 * (size_t)d + 1] != norms[1]) {
      fs::set_error("fs_dist_cg2: the devices disagree about the norms of the right-hand sides");
      return FS_ERR_HIP;
    }
  }
  int rc_loop = FS_OK;
  for (int iter = 0; iter < F; iter++) {
    if (int rc = dist_product_k(M, false, W.x, W.y)) return rc;          // TMP = A P
    if (int rc = dist_product_k(M, true, W.y, W.z)) return rc;           // Q = A' TMP
    int stops = 0;
    for (int d = 0; d < n; ++d) {
      FS_HIP(hipSetDevice(D->dev[d]));
      if (int rc = fs::cg2_dev_steps(F, lambda, W.sol[(size_t)d], W.r[(size_t)d], W.x[(size_t)d], W.z[(size_t)d], W.part[(size_t)d],
                                     W.red[(size_t)d], W.st[(size_t)d], D->stream[d])) return rc;
      bool stop = false;
      if (int rc = fl[(size_t)d].after_iteration(iter, W.st[(size_t)d], D->stream[d], &stop)) return rc;
      stops += stop ? 1 : 0;
    }
    if (stops != 0 && stops != n) {
      fs::set_error("fs_dist_cg2: the devices disagree about convergence (a device or an exchange returned different bits)");
      rc_loop = FS_ERR_HIP;
      break;
    }
    if (stops == n) break;
  }
  for (int d = 0; d < n && rc_loop == FS_OK; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    if (int rc = fs::cg2_dev_finish(F, &norms[2 * (size_t)d], W.sol[(size_t)d], D->stream[d])) return rc;
  }
  if (int rc = dist_sync(D)) return rc;
  if (rc_loop != FS_OK) return rc_loop;
  std::vector<double> fin(2 * (size_t)n, 0.0);
  for (int d = 0; d < n; ++d) {
    FS_HIP(hipSetDevice(D->dev[d]));
    FS_HIP(hipMemcpy(&fin[2 * (size_t)d], W.st[(size_t)d] + fs::kCgStateDone, sizeof(double) * 2, hipMemcpyDeviceToHost));
    if (fin[2 * (size_t)d] != fin[0] || fin[2 * (size_t)d + 1] != fin[1]) {
      fs::set_error("fs_dist_cg2: the devices finished in different states");
      return FS_ERR_HIP;
    }
  }
  if (vector_device(X_host) >= 0)
    if (int rc = wait_for_caller(vector_device(X_host))) return rc;
  if (int rc = vec_store(M, 0, X_host, W.sol[0], (size_t)F * 2)) return rc;
  if (int rc = dist_sync(D)) return rc;
  if (out_iter) *out_iter = (int)fin[1];
  return FS_OK;
}

double *fs_dist_x(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->x[(size_t)rank] : nullptr; }
double *fs_dist_y(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->y[(size_t)rank] : nullptr; }
double *fs_dist_z(fs_dist_matrix_t M, int rank) { return (M && rank >= 0 && rank < M->D->n) ? M->z[(size_t)rank] : nullptr; }

}  // extern "C"
