#!/usr/bin/env python3
"""bench.py -- fp64 CSR SpMV (A_mul_B + At_mul_B) on BASELINE.json config 2, one rank per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload: per GPU a CSR shard of 10 000 000 rows x 16 non-zeros, columns uniform over
10 000 000 columns, values uniform(-1,1), generated on the device by the counter-based generator (seed
0x5EED0002, identical on CPU).  N = 1 is exactly config 2 (10M x 10M).  N > 1 is weak scaling by rows, the
partition config 5 names: the global matrix is (N*10M) x 10M, rank r owns rows [r*10M, (r+1)*10M) -- a
config-2 matrix of its own -- x (80 MB) is replicated, y = A x is a local SpMV followed by the RCCL
all-gather of the y shards, and z = A' u is the local transposed product of the rank's rows with its slice
of u followed by an RCCL all-reduce (sum) of the 80 MB partial results.  The two products of a step are
independent, so each exchange is started asynchronously and overlaps the next local product.

One step = y = A x  then  z = A' u  (two products per rank -- each one launch of the expand kernel and one of
the reduce kernel of the two-pass SpMV -- plus one all-gather and one all-reduce when N > 1).  value = algorithmic bytes of all ranks' products / max-over-ranks wall time.
Algorithmic bytes per product: 12*nnz + 4*(nrow+1) + 8*nrow + 8*ncol.

Prints ONE JSON line on rank 0 with `roofline` (one product = spmv_expand_kernel + spmv_reduce_kernel, HIP-event
timed inside the timed region; "launch" below means one product, i.e. that pair) and, at N = 1, `cpu_baseline` (the oracle's OpenMP restatement of csr_A_mul_B,
built with the reference's flags, timed on this box's host cores on the same matrix).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
SEED = 0x5EED0002
METRIC = "fp64 CSR SpMV effective GB/s (% HBM3E peak)"


def host_cpus():
    """CPUs this job may actually use: the affinity mask, cut down to the cgroup CPU quota when there is one"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(rows, ncol, per_row, reps=10):
    """csr_A_mul_B restated (oracle/fs_oracle.c: fso_csr_mul, omp parallel for schedule(dynamic,256) as csr.h:429),
    built here with the reference's flags (-O3 -march=native -fopenmp -ffast-math), on the same config-2 matrix."""
    import numpy as np
    from oracle import pyoracle, pysynth
    import ctypes as C
    out = os.path.join("/tmp", "liboracle_fast_%d.so" % os.getpid())
    fast = pyoracle.load(pyoracle.build_fast(out))
    # one OpenMP thread per host CPU this process may run on (the box gives a GPU job a share of the host's cores;
    # libgomp's default would be every core of the machine)
    ncpu = host_cpus()
    C.CDLL("libgomp.so.1").omp_set_num_threads(ncpu)
    rp, cc, vv = pysynth.uniform(rows, ncol, per_row, SEED)
    x = np.sin(7.0 * np.arange(ncol, dtype=np.float64) + 0.3)
    y = np.empty(rows)
    vp = vv.ctypes.data
    fast.fso_csr_mul(y, rows, rp, cc, vp, x)          # warm-up (bench_a_mul_b.c does one untimed call)
    t0 = time.time()
    for _ in range(reps):
        fast.fso_csr_mul(y, rows, rp, cc, vp, x)
    dt = (time.time() - t0) / reps
    nbytes = 12 * rows * per_row + 4 * (rows + 1) + 8 * rows + 8 * ncol
    strict = pyoracle.load()
    os.remove(out)
    return {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": int(strict.fso_threads()), "kind": "port",
            "host_cpus_available": ncpu,
            "ms_per_product": dt * 1e3,
            "sample": "csr_A_mul_B on the full config-2 matrix (%d x %d, %d nnz/row), 1 warm-up + mean of %d "
                      "repeats, OpenMP schedule(dynamic,256), gcc -O3 -march=native -ffast-math" % (rows, ncol, per_row, reps)}


def cpu_reference_serial(rows, ncol, per_row, reps=3):
    """The REAL reference's csr_A_mul_B (oracle/_ref/libfsref.so, compiled from the reference's own headers in the
    build container, without OpenMP -- see oracle/ref_shim.c) on the first `rows` rows of the config-2 matrix:
    one host core.  Present only when the prebuilt library travelled with the repo."""
    import ctypes as C
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_ref", "libfsref.so")
    if not os.path.exists(so):
        return None
    from oracle import pysynth
    lib = C.CDLL(so)
    rp, cc, vv = pysynth.uniform(rows, ncol, per_row, SEED)

    class CSR(C.Structure):      # csr.h:358-366
        _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("row_ptr", C.c_void_p),
                    ("cols", C.c_void_p), ("vals", C.c_void_p)]
    A = CSR(rows, ncol, len(cc), rp.ctypes.data, cc.ctypes.data, vv.ctypes.data)
    x = np.sin(7.0 * np.arange(ncol, dtype=np.float64) + 0.3)
    y = np.empty(rows)
    f = lib.csr_A_mul_B
    f.restype = None
    args = (C.c_void_p(y.ctypes.data), C.byref(A), C.c_void_p(x.ctypes.data))
    f(*args)
    t0 = time.time()
    for _ in range(reps):
        f(*args)
    dt = (time.time() - t0) / reps
    nbytes = 12 * rows * per_row + 4 * (rows + 1) + 8 * rows + 8 * ncol
    return {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference", "ms_per_product": dt * 1e3,
            "sample": "reference csr_A_mul_B (csr.h:425, serial build) on the first %d rows of the config-2 matrix, "
                      "x over all %d columns, 1 warm-up + mean of %d" % (rows, ncol, reps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU (config 2: 10M)")
    ap.add_argument("--per-row", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from libfastsparse_amd import capi
    from libfastsparse_amd import dist as fsd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # FS_BENCH_BACKEND=gloo rehearses the N > 1 code path with several ranks sharing one GPU (no RCCL there)
    backend = os.environ.get("FS_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    capi.lib()

    n_local, per = args.rows, args.per_row
    ncol = n_local                      # column space stays config 2's: every rank's shard is a config-2 matrix
    n_global = n_local * world          # global rows
    lo = rank * n_local
    st = capi.current_stream()

    # ---- this rank's shard: rows lo .. lo+n_local of the (N*10M) x 10M matrix ------------------------
    rp, cc, vv = capi.synth_uniform(n_local, ncol, per, SEED, row_offset=lo)
    A = capi.Matrix.from_csr(n_local, ncol, rp, cc, vv, borrow=True)
    A.build_transpose(st)
    bounds = fsd.even_row_partition(n_global, world)
    bytes_a = A.algorithmic_bytes()
    bytes_t = bytes_a                   # same entries, nrow and ncol swap roles (both 10M here)
    op_a = fsd.ShardedOperator(fsd.hip_local_spmv(A), bounds)
    op_t = fsd.TransposedShardedOperator(
        lambda z, u_local: A.spmv(z, u_local, capi.current_stream(), transposed=True), bounds)

    x = torch.sin(7.0 * torch.arange(ncol, device=dev, dtype=torch.float64) + 0.3)        # bench_a_mul_b.c:142
    u = torch.sin(11.0 * torch.arange(n_global, device=dev, dtype=torch.float64) - 0.2)   # 2nd column of X2col, :145
    y = torch.empty(n_global, dtype=torch.float64, device=dev)
    z = torch.empty(ncol, dtype=torch.float64, device=dev)

    # Two y buffers (and two operators with their own shard buffers) alternate, so that the all-gather of step k may
    # run until the end of step k+1: at 8 GPUs the 640 MB gather is longer than one local product, and only the
    # transposed product of the same step would otherwise be there to hide it.
    op_a2 = [op_a, fsd.ShardedOperator(fsd.hip_local_spmv(A), bounds)] if world > 1 else [op_a, op_a]
    ybufs = [y, torch.empty_like(y)] if world > 1 else [y, y]
    pending = [None]   # the all-reduce of the previous step's z, still in flight
    gather = [None]    # the all-gather of the previous step's y, still in flight
    count = [0]

    def step(ev=None):
        """one step: ev[0]|A x|ev[1]  start all-gather(y_k)  ev[2]|A' u|ev[3]  wait all-gather(y_k-1), start all-reduce(z).
        The two products are independent, so each exchange overlaps the following local products (RCCL runs on its
        own stream); every collective is waited for before its buffers are reused and before the clock stops."""
        b = count[0] & 1
        count[0] += 1
        if ev is not None:
            ev[0].record()
        yl = op_a2[b].local(ybufs[b], x)
        if ev is not None:
            ev[1].record()
        g = op_a2[b].gather_async(ybufs[b], yl)
        if pending[0] is not None:
            pending[0].wait()            # z of the previous step is complete before it is overwritten
            pending[0] = None
        if ev is not None:
            ev[2].record()
        op_t.apply_local(z, u)
        if ev is not None:
            ev[3].record()
        if gather[0] is not None:
            gather[0].wait()             # the other y buffer is complete before the next step writes its shard buffer
        gather[0] = g
        pending[0] = op_t.reduce_async(z)

    def drain():
        if gather[0] is not None:
            gather[0].wait()
            gather[0] = None
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(evs[k])
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-product durations inside the timed region (HIP events on the launch stream); a product is the kernel
    # pair expand + reduce of the two-pass SpMV (or one launch of the tiled / streaming kernel if the format
    # builder chose those)
    ka = [e[0].elapsed_time(e[1]) for e in evs]
    kt = [e[2].elapsed_time(e[3]) for e in evs]
    launches = 2 * args.steps
    avg_ms = (sum(ka) + sum(kt)) / launches
    bytes_per_launch = (bytes_a + bytes_t) / 2.0
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9

    total_bytes = float(world) * (bytes_a + bytes_t) * args.steps
    value = total_bytes / elapsed / 1e9

    # self-check outside the timed region (no oracle here: that is the tests' job): the products the timed loop left in
    # y and z against the storage-order chunk-streaming kernel on the same operands, and, at N > 1, every rank holding
    # the same gathered y / reduced z
    self_check = {}
    try:
        y_ref = torch.empty(n_local, dtype=torch.float64, device=dev)
        capi.set_option("strict_order", 1)
        try:
            A.spmv(y_ref, x, capi.current_stream())
            z_ref = torch.empty(ncol, dtype=torch.float64, device=dev)
            A.spmv(z_ref, u[lo:lo + n_local], capi.current_stream(), transposed=True)
        finally:
            capi.set_option("strict_order", 0)
        y_last = ybufs[(count[0] - 1) & 1]                       # the y the last timed step produced
        self_check["A_mul_B_max_abs_diff_vs_storage_order_kernel"] = float((y_last[lo:lo + n_local] - y_ref).abs().max())
        if world > 1:
            zr = z_ref.clone()
            fsd.all_reduce_sum(zr)                                    # sum of the ranks' partial products
            sums = torch.tensor([float(y_last.sum()), float(z.sum())], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            lo_, hi_ = sums.clone(), sums.clone()
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
            self_check["ranks_hold_identical_y_and_z"] = bool(torch.equal(lo_, hi_))
        else:
            zr = z_ref
        self_check["At_mul_B_max_abs_diff_vs_storage_order_kernel"] = float((z - zr).abs().max())
        self_check["ok"] = bool(self_check["A_mul_B_max_abs_diff_vs_storage_order_kernel"] <= 1e-11 and
                                self_check["At_mul_B_max_abs_diff_vs_storage_order_kernel"] <= 1e-11 * world and
                                self_check.get("ranks_hold_identical_y_and_z", True))
        del y_ref, z_ref, zr
    except Exception as ex:   # a failed check must show in the line, not kill it
        self_check = {"ok": False, "error": repr(ex)}

    # on-box streaming ceiling (SURVEY 8d asks for % of the measured stream peak next to % of the 8 TB/s spec):
    # a read-only pass over 1.92 GB, the size of config 2's cols + vals
    stream_gbs = None
    if rank == 0:
        probe = torch.empty(240_000_000, dtype=torch.float64, device=dev).fill_(1.0)
        for _ in range(2):
            probe.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            probe.sum()
        e1.record()
        torch.cuda.synchronize()
        stream_gbs = 5 * probe.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del probe

    if rank == 0:
        traffic = None
        kname, kname_t = A.kernel_name(), A.kernel_name(True)
        labels = {"two-pass": "fs::spmv_expand_kernel<valued> + fs::spmv_reduce_kernel (one product)",
                  "tiled": "fs::spmv_tiled_kernel<valued>", "lds-staged": "fs::spmv_ldsx_kernel<valued>",
                  "stream": "fs::spmv_stream_kernel<valued>"}
        klabel = labels.get(kname, kname)
        if kname_t != kname:   # the format builder's timed choice may differ between A and A' (they are within a few %)
            klabel = "A: %s; A': %s" % (klabel, labels.get(kname_t, kname_t))
        # PMC traffic per product (profiles/traffic_spmv_<kernel>.json, measured on this workload): the mean over
        # the step's two products, which may run on different kernels
        traffic = None
        if world == 1:
            try:
                vals_ = []
                for kn in (kname, kname_t):
                    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_spmv_%s.json" % kn.replace("-", "_"))))
                    if tj.get("rows") != n_local or tj.get("per_row") != per:
                        raise ValueError("traffic file is for another workload")
                    vals_.append(float(tj["hbm_bytes_per_launch"]))
                traffic = sum(vals_) / len(vals_)
            except Exception:
                traffic = None
        rec = {
            "metric": METRIC, "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: CSR %d x %d, %d nnz/row uniform, fp64, step = A_mul_B + At_mul_B"
                                   % (n_global, ncol, per) if world == 1 else
                                   "config-2 shards, weak scaling by rows: CSR %d x %d, %d rows/GPU, %d nnz/row, step = "
                                   "(A_mul_B + all-gather y) + (At_mul_B + all-reduce z)" % (n_global, ncol, n_local, per),
                       "rows_per_gpu": n_local, "nnz_per_gpu": n_local * per, "parallelism": "rows x%d" % world,
                       "pct_of_hbm_peak": 100.0 * value / (HBM_PEAK_GBS * world),
                       "stream_read_GBs_measured": stream_gbs,
                       "pct_of_measured_stream_read": 100.0 * achieved / stream_gbs if stream_gbs else None,
                       "A_mul_B_ms": sum(ka) / len(ka), "At_mul_B_ms": sum(kt) / len(kt),
                       "self_check": self_check,
                       "kernel_A": A.kernel_name(), "kernel_At": A.kernel_name(True),
                       "builder_timed_ms_A": A.candidate_ms(), "builder_timed_ms_At": A.candidate_ms(True)},
            "roofline": {"bound": "hbm", "kernel": klabel, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms,
                         "launches_timed": launches},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                rec["cpu_baseline"] = cpu_baseline(n_local, ncol, per)
                ref1 = cpu_reference_serial(min(n_local, 2_000_000), ncol, per)
                if ref1:
                    rec["cpu_baseline"]["reference_serial"] = ref1
            except Exception as ex:  # the baseline is a reported extra; its failure must not hide the GPU number
                rec["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
