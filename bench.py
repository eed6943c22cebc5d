#!/usr/bin/env python3
"""bench.py -- the A_mul_B / At_mul_B path on MI355X, one rank per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload all|c2|c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher starts the N ranks ITSELF (fresh child processes, before this
process touches the GPU) and fails loudly if the world it finds is not N.

ONE JSON line on rank 0.  The line itself is BASELINE config 2 (`metric`, `value`, `roofline`, `cpu_baseline` exactly as
before); with the default `--workload all` its key "also" holds one sub-record per other BASELINE config, each with its own
`value`, `roofline` (+ `traffic`), `self_check` and, at N = 1, `cpu_baseline`:
    N = 1:  c3, c4 (k = 32, with k = 2 / 4 / 8 beside it), c5 as ONE real shard, and the path's consumer: CG and
            2-right-hand-side block CG on config 2's pattern
    N > 1:  c2 strong scaling (the one 10 M-row matrix cut over the ranks), c5 across the N ranks (with and without the
            all-gather, and with A'u), and the row-sharded CG (ShardedCG) on config 2's pattern
so that one driver command reaches every config (`--workload c2` prints the headline alone; c3 / c4 / c5 print that
workload's record as the line).

Workloads (all data synthetic, generated on the device by the counter-based generators whose CPU twins
live in oracle/fs_synth.c):

  c2 (BASELINE.json configs[1], the configuration the metric is quoted on)
      fp64 CSR 10 M x 10 M, 16 nnz/row uniform.  Step = y = A x then z = A' u (two products).  N > 1: weak scaling by
      rows, the global matrix is (N*10M) x 10M, every rank owns a config-2 shard, x replicated.  The step is the ITERATING
      consumer's (cg.h:15-16: y of A x is the x of A'): y = local product in parts with the RCCL all-gather of the finished
      rows running under the later parts (ShardedOperator.apply_overlapped), y complete before A' starts; z = A' u as "row
      shards of A' + all-gather" (built once by an all-to-all of the entries; --z-scheme reduce: local A_r' u_r + all-reduce).
  c3  SparseBinaryMatrix 10 M x 1 M, 64 nnz/row, supplied as COO to the A_mul_B / At_mul_B handles (fs_coo_create),
      integer-valued x.  Step = A_mul_B + At_mul_B.  One GPU.
  c4  CSR x dense SpMM: the config-2 matrix times X (10 M x 32, row-major).  Step = one csr_A_mul_Bn (k = 32);
      k = 2, 4, 8 are timed beside it.  One GPU.
  c5  CSR 100 M x 100 M, power-law row lengths (mean 32, clipped at 1e6), rows cut by non-zeros over the N ranks, x
      (800 MB) replicated, y = local SpMV in parts + RCCL all-gather of the (unequal) y shards inside the product; with
      --transpose (always inside "also") a second timed loop adds z = A' u as "row shards of A' + all-gather".  The matrix
      does not fit one struct CSR (3.2 G non-zeros > 2^31-1), so N = 1 runs ONE real shard: the rows rank 3 of 8 owns under
      the nnz-balanced cut.

value = algorithmic bytes of all ranks' products / max-over-ranks wall time of the K timed steps (barrier +
synchronize on both sides), operands resident in HBM.  Algorithmic bytes per product:
(12 | 4)*nnz + 4*(nrow+1) + 8k*nrow + 8k*ncol.

`roofline` comes from HIP events inside the timed region: at N = 1 ONE pair around the K steps (events around every
product cost 1 % of a 0.86 ms product: profiles/r03_gap_probe.jsonl), at N > 1 a pair around every rank-local product
(the exchanges run on RCCL's stream).  `roofline.design_ceiling_frac` = algorithmic bytes / bytes the kept kernel really moves
(PMC, profiles/traffic_*.json) x the 6.29 TB/s a copy reaches on this chip / 8 TB/s: what THIS algorithm could reach.
`cpu_baseline` (N = 1): the oracle's OpenMP restatement built with the reference's flags and, where oracle/_ref travelled,
the real reference's serial loop, on a stated bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0          # what a float4 copy reaches on this chip (same guide): the rate a streaming kernel can hope for
SEED_C2 = 0x5EED0002
SEED_C3 = 0x5EED0003
SEED_C5 = 0x5EED0005
C5_SCALE, C5_MAXLEN = 2.3, 1_000_000     # P(len >= L) ~ 2.3 / L clipped at 1e6: mean 31.7 non-zeros per row
METRIC = "fp64 CSR SpMV effective GB/s (% HBM3E peak)"


# ------------------------------------------------------------------------------------------------------------
# launching
# ------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this process has not touched the
    GPU and never will), wait for them, return the worst exit code.  Rank 0 prints the JSON line."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), FS_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                c = p.poll()
                if c is None:
                    continue
                alive.remove(p)
                if c != 0:
                    rc = rc or c
                    for q in alive:          # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


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


# ------------------------------------------------------------------------------------------------------------
# the device side behind a small interface (tests/test_dist_gloo.py rehearses the multi-rank workloads on CPU with
# the oracle behind the same interface; bench.py itself only ever uses HipProvider)
# ------------------------------------------------------------------------------------------------------------
class HipProvider:
    name = "hip"

    def __init__(self, dev):
        import torch
        from libfastsparse_amd import capi
        self.torch, self.capi, self.dev = torch, capi, dev
        self._tables = {}
        # the library allocates and launches on the calling thread's current HIP device: make it this rank's, whatever
        # torch has done about it so far
        capi.check(capi.lib().fs_set_device(dev.index if dev.index is not None else 0), "fs_set_device")

    def stream(self):
        return self.capi.current_stream()

    def synchronize(self):
        self.torch.cuda.synchronize()

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def elapsed_ms(self, e0, e1):
        return e0.elapsed_time(e1)

    def empty(self, n, dtype=None):
        return self.torch.empty(n, dtype=dtype or self.torch.float64, device=self.dev)

    def sin_vector(self, n, a, b):
        t = self.torch
        return t.sin(a * t.arange(n, device=self.dev, dtype=t.float64) + b)

    def int_vector(self, n, seed):
        t = self.torch
        g = t.Generator(device=self.dev)
        g.manual_seed(seed)
        return t.randint(-1000, 1001, (n,), device=self.dev, generator=g).to(t.float64)

    def uniform(self, nrow, ncol, per, seed, row_offset=0, valued=True):
        return self.capi.synth_uniform(nrow, ncol, per, seed, row_offset=row_offset, valued=valued)

    def powerlaw_lengths(self, nrow, row_offset):
        t = self.torch
        lens = t.empty(nrow, dtype=t.int32, device=self.dev)
        self.capi.check(self.capi.lib().fs_synth_powerlaw_lengths(nrow, C5_SCALE, C5_MAXLEN, SEED_C5, row_offset,
                                                                    lens.data_ptr(), self.stream()))
        return lens

    def fill(self, row_ptr, ncol, row_offset, valued=True):
        t = self.torch
        nrow = row_ptr.numel() - 1
        nnz = int(row_ptr[-1].item())
        cols = t.empty(max(nnz, 1), dtype=t.int32, device=self.dev)[:nnz]
        vals = t.empty(max(nnz, 1), dtype=t.float64, device=self.dev)[:nnz] if valued else None
        self.capi.check(self.capi.lib().fs_synth_fill(nrow, ncol, SEED_C5, row_offset, row_ptr.data_ptr(), cols.data_ptr(),
                                                      vals.data_ptr() if valued else None, self.stream()))
        return cols, vals

    def csr(self, nrow, ncol, rp, cc, vv, transpose=False):
        M = self.capi.Matrix.from_csr(nrow, ncol, rp, cc, vv, borrow=True)
        if transpose:
            M.build_transpose(self.stream())
        return M

    def coo(self, nrow, ncol, rows, cols, vals):
        return self.capi.Matrix.from_coo(nrow, ncol, rows, cols, vals)

    def spmv(self, A, y, x, transposed=False):
        A.spmv(y, x, self.stream(), transposed=transposed)

    def spmv_strict(self, A, y, x, transposed=False):
        """the same product on the storage-order kernel (the self-checks' yardstick)"""
        self.capi.set_option("strict_order", 1)
        try:
            A.spmv(y, x, self.stream(), transposed=transposed)
        finally:
            self.capi.set_option("strict_order", 0)

    def parts(self, A, transposed=False):
        from libfastsparse_amd import dist as fsd
        return fsd.HipParts(A, transposed)

    def copy_segments(self, dst, src, dst_off, src_off, counts):
        """fs_copy_segments: the unpack of a padded all-gather buffer in one launch; the offset table lives in HBM, made once"""
        if not counts:
            return
        key = (tuple(dst_off), tuple(src_off), tuple(counts))
        tab = self._tables.get(key)
        if tab is None:
            tab = self.torch.tensor(list(dst_off) + list(src_off) + list(counts), dtype=self.torch.int64, device=self.dev)
            self._tables[key] = tab
        self.capi.check(self.capi.lib().fs_copy_segments(len(counts), tab.data_ptr(), max(counts), src.data_ptr(), dst.data_ptr(),
                                                        self.stream()), "fs_copy_segments")

    def kernel_name(self, A, transposed=False):
        return A.kernel_name(transposed)

    def candidate_ms(self, A, transposed=False):
        return A.candidate_ms(transposed)

    def trim(self):
        """inside a workload, after a matrix has been built: the builders' idle scratch and torch's cached blocks go back to the
        device (config 5 at N = 2 holds 1.6 G entries per rank: the next build needs the room)"""
        self.capi.lib().fs_release_all()        # side table of the drop-in (unused here) and the builders' scratch pool
        self.torch.cuda.empty_cache()

    def release(self):
        """between workloads: everything the previous one held goes back to the device"""
        import gc
        gc.collect()
        self._tables.clear()
        self.capi.lib().fs_release_all()
        self.torch.cuda.empty_cache()


def _multi(world):
    """more than one rank -- or FS_BENCH_FORCE_MULTI=1: the N > 1 code path with ONE rank, which is how the one-GPU box can run
    it on RCCL (RCCL refuses two ranks on one card): init_process_group("nccl"), the device-side collectives of the timing and
    of the exchanges, every sub-record of the N > 1 plan.  Use it with FS_DIST_FORCE_COLLECTIVES=1 and reduced sizes."""
    return world > 1 or os.environ.get("FS_BENCH_FORCE_MULTI", "0") == "1"


def in_turns(fn, prov, world, rank, nccl):
    """fn() on every rank -- one rank after the other when the ranks SHARE one GPU (the gloo rehearsal): the radix sorts of the
    format builders and of torch wait for other workgroups of their own grid (decoupled look-back), and several processes
    time-sliced on one card were seen to starve each other there for ever (GPU 100 % busy, no memory traffic).  One process
    per GPU, the real thing, has the card to itself."""
    if nccl or not _multi(world) or getattr(prov, "name", "") != "hip":
        return fn()
    import torch.distributed as dist
    out = None
    for r in range(world):
        if r == rank:
            out = fn()
            prov.synchronize()
        dist.barrier()
    return out


def timed_steps(prov, step, drain, steps, warmup, world, backend_is_nccl):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize, MAX over ranks (seconds); also the
    duration of the K steps between ONE pair of events on the launch stream (ms, this rank)"""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step(None)
    drain()
    prov.synchronize()
    if _multi(world):
        dist.barrier()
    evs = [[prov.event() for _ in range(4)] for _ in range(steps)]
    e0, e1 = prov.event(), prov.event()
    prov.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for k in range(steps):
        step(evs[k])
    drain()
    e1.record()
    prov.synchronize()
    if _multi(world):
        dist.barrier()
    elapsed = time.perf_counter() - t0
    region_ms = prov.elapsed_ms(e0, e1)
    if _multi(world):
        t = torch.tensor([elapsed], dtype=torch.float64, device=prov.dev if backend_is_nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, evs, region_ms


def stream_probe(prov):
    """on-box streaming ceiling: a read-only pass over 1.92 GB, the size of config 2's cols + vals"""
    t = prov.torch
    probe = t.empty(240_000_000, dtype=t.float64, device=prov.dev).fill_(1.0)
    for _ in range(2):
        probe.sum()
    e0, e1 = prov.event(), prov.event()
    e0.record()
    for _ in range(5):
        probe.sum()
    e1.record()
    prov.synchronize()
    gbs = 5 * probe.numel() * 8 / (prov.elapsed_ms(e0, e1) * 1e-3) / 1e9
    del probe
    return gbs


def csr_bytes(nnz, nrow, ncol, valued=True, k=1):
    return (12 if valued else 4) * nnz + 4 * (nrow + 1) + 8 * k * nrow + 8 * k * ncol


# ------------------------------------------------------------------------------------------------------------
# CPU baselines (the ONLY place bench.py touches oracle/: the checker timed as the reported baseline, never the product)
# ------------------------------------------------------------------------------------------------------------
def _fast_oracle():
    from oracle import pyoracle
    import ctypes as C
    out = os.path.join("/tmp", "liboracle_fast_%d.so" % os.getpid())
    fast = pyoracle.load(pyoracle.build_fast(out))
    os.remove(out)
    ncpu = host_cpus()
    # one OpenMP thread per host CPU this process may run on (the box gives a GPU job a share of the host's cores;
    # libgomp's default would be every core of the machine)
    C.CDLL("libgomp.so.1").omp_set_num_threads(ncpu)
    return fast, ncpu


def _time_calls(f, reps):
    f()                                   # warm-up (bench_a_mul_b.c does one untimed call)
    t0 = time.time()
    for _ in range(reps):
        f()
    return (time.time() - t0) / reps


def cpu_baseline_c2(rows, ncol, per_row, reps=10):
    """csr_A_mul_B restated (oracle/fs_oracle.c: fso_csr_mul, omp parallel for schedule(dynamic,256) as csr.h:429),
    built here with the reference's flags (-O3 -march=native -fopenmp -ffast-math), on the same config-2 matrix."""
    import numpy as np
    from oracle import pysynth
    fast, ncpu = _fast_oracle()
    rp, cc, vv = pysynth.uniform(rows, ncol, per_row, SEED_C2)
    x = np.sin(7.0 * np.arange(ncol, dtype=np.float64) + 0.3)
    y = np.empty(rows)
    dt = _time_calls(lambda: fast.fso_csr_mul(y, rows, rp, cc, vv.ctypes.data, x), reps)
    nbytes = csr_bytes(rows * per_row, rows, ncol)
    rec = {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": ncpu, "kind": "port", "ms_per_product": dt * 1e3,
           "sample": "csr_A_mul_B on the full config-2 matrix (%d x %d, %d nnz/row), 1 warm-up + mean of %d repeats, "
                     "OpenMP schedule(dynamic,256), gcc -O3 -march=native -ffast-math" % (rows, ncol, per_row, reps)}
    ref1 = cpu_reference_serial("csr_A_mul_B", min(rows, 2_000_000), ncol, per_row, SEED_C2, True)
    if ref1:
        rec["reference_serial"] = ref1
    return rec


def cpu_reference_serial(entry, rows, ncol, per_row, seed, valued, reps=3):
    """The REAL reference (oracle/_ref/libfsref.so, compiled from the reference's own headers in the build container,
    without OpenMP -- see oracle/ref_shim.c) on the first `rows` rows: one host core.  Present only when the prebuilt
    library travelled with the repo.  entry: csr_A_mul_B (csr.h:425), bcsr_A_mul_B (csr.h:149), A_mul_B (sparse.h:58)."""
    import ctypes as C
    import numpy as np
    so = os.path.join(ROOT, "oracle", "_ref", "libfsref.so")
    if not os.path.exists(so):
        return None
    from oracle import pysynth
    lib = C.CDLL(so)
    rp, cc, vv = pysynth.uniform(rows, ncol, per_row, seed, valued=valued)
    x = np.sin(7.0 * np.arange(ncol, dtype=np.float64) + 0.3)
    y = np.empty(rows)
    keep = [rp, cc, vv, x, y]
    if entry == "A_mul_B":         # struct SparseBinaryMatrix, sparse.h:11-18
        rows_arr = np.repeat(np.arange(rows, dtype=np.int32), per_row)
        keep.append(rows_arr)

        class SBM(C.Structure):
            _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("rows", C.c_void_p), ("cols", C.c_void_p)]
        A = SBM(rows, ncol, len(cc), rows_arr.ctypes.data, cc.ctypes.data)
        nbytes = csr_bytes(len(cc), rows, ncol, valued=False)
    elif entry == "bcsr_A_mul_B":  # struct BinaryCSR, csr.h:15-22
        class BCSR(C.Structure):
            _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("row_ptr", C.c_void_p), ("cols", C.c_void_p)]
        A = BCSR(rows, ncol, len(cc), rp.ctypes.data, cc.ctypes.data)
        nbytes = csr_bytes(len(cc), rows, ncol, valued=False)
    else:                          # struct CSR, csr.h:358-366
        class CSR(C.Structure):
            _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("row_ptr", C.c_void_p),
                        ("cols", C.c_void_p), ("vals", C.c_void_p)]
        A = CSR(rows, ncol, len(cc), rp.ctypes.data, cc.ctypes.data, vv.ctypes.data)
        nbytes = csr_bytes(len(cc), rows, ncol)
    f = getattr(lib, entry)
    f.restype = None
    args = (C.c_void_p(y.ctypes.data), C.byref(A), C.c_void_p(x.ctypes.data))
    dt = _time_calls(lambda: f(*args), reps)
    return {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference", "ms_per_product": dt * 1e3,
            "sample": "reference %s (serial build) on the first %d rows (%d x %d, %d nnz/row), 1 warm-up + mean of %d"
                      % (entry, rows, rows, ncol, per_row, reps)}


def cpu_baseline_c3(nrow, ncol, per_row, sample_rows=2_500_000, reps=3):
    """binary product: the oracle's bcsr_A_mul_B restatement (OpenMP as csr.h:152) on the first `sample_rows` rows, and
    the real reference's SERIAL A_mul_B (sparse.h:58-65 has no OpenMP pragma) on a smaller sample"""
    import numpy as np
    from oracle import pysynth
    fast, ncpu = _fast_oracle()
    rows = min(nrow, sample_rows)
    rp, cc, _ = pysynth.uniform(rows, ncol, per_row, SEED_C3, valued=False)
    x = np.random.default_rng(3).integers(-1000, 1001, ncol).astype(np.float64)
    y = np.empty(rows)
    dt = _time_calls(lambda: fast.fso_csr_mul(y, rows, rp, cc, None, x), reps)
    nbytes = csr_bytes(rows * per_row, rows, ncol, valued=False)
    rec = {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": ncpu, "kind": "port", "ms_per_product": dt * 1e3,
           "sample": "bcsr_A_mul_B restated (OpenMP schedule(dynamic,256)) on the first %d of %d rows (%d columns, %d "
                     "nnz/row), 1 warm-up + mean of %d, gcc -O3 -march=native -ffast-math" % (rows, nrow, ncol, per_row, reps)}
    ref1 = cpu_reference_serial("A_mul_B", min(nrow, 500_000), ncol, per_row, SEED_C3, False)
    if ref1:
        rec["reference_serial"] = ref1
    return rec


def cpu_baseline_c5(lo, ncol, sample_rows=1_000_000, reps=3):
    """csr_A_mul_B restated (OpenMP as csr.h:429) on the first `sample_rows` rows of the shard (power-law lengths, x over all
    100 M columns), generated by the CPU twin of the device generator"""
    import numpy as np
    from oracle import pysynth
    fast, ncpu = _fast_oracle()
    rp, cc, vv = pysynth.powerlaw(sample_rows, ncol, C5_SCALE, C5_MAXLEN, SEED_C5, row_offset=lo)
    x = np.sin(7.0 * np.arange(ncol, dtype=np.float64) + 0.3)
    y = np.empty(sample_rows)
    dt = _time_calls(lambda: fast.fso_csr_mul(y, sample_rows, rp, cc, vv.ctypes.data, x), reps)
    nbytes = csr_bytes(len(cc), sample_rows, ncol)
    return {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": ncpu, "kind": "port", "ms_per_product": dt * 1e3,
            "sample": "csr_A_mul_B restated (OpenMP schedule(dynamic,256)) on rows %d..%d of the config-5 matrix (%d non-zeros, x over "
                      "all %d columns: the 8 bytes per column are part of the byte count), 1 warm-up + mean of %d, gcc -O3 "
                      "-march=native -ffast-math" % (lo, lo + sample_rows, len(cc), ncol, reps)}


def cpu_baseline_c4(n, per_row, k, sample_rows=400_000, reps=2):
    """csr_A_mul_Bn on the first `sample_rows` rows (X over all columns), both ways SURVEY note N4 asks for: with the
    reference's own nested `omp parallel` + `omp parallel for` (csr.h:445-448: every thread walks all rows) and with
    the corrected single `omp for`"""
    import numpy as np
    from oracle import pysynth
    fast, ncpu = _fast_oracle()
    rows = min(n, sample_rows)
    rp, cc, vv = pysynth.uniform(rows, n, per_row, SEED_C2)
    i = np.arange(n, dtype=np.float64)[:, None]
    X = np.sin(7.0 * i + 17.0 * np.arange(k, dtype=np.float64)[None, :] + 0.3).reshape(-1)
    Y = np.empty(rows * k)
    nbytes = csr_bytes(rows * per_row, rows, n, k=k)
    dt_fix = _time_calls(lambda: fast.fso_csr_mul_n(Y, rows, rp, cc, vv.ctypes.data, X, k), reps)
    dt_ref = _time_calls(lambda: fast.fso_csr_mul_n_reference_schedule(Y, rows, rp, cc, vv.ctypes.data, X, k), reps)
    what = "csr_A_mul_Bn k=%d on the first %d rows of the config-2 matrix (X %d x %d), 1 warm-up + mean of %d" % (k, rows, n, k, reps)
    return {"value": nbytes / dt_fix / 1e9, "unit": "GB/s", "cores": ncpu, "kind": "port", "ms_per_product": dt_fix * 1e3,
            "sample": what + "; corrected schedule: one `omp for` over rows",
            "as_reference_runs_it": {"value": nbytes / dt_ref / 1e9, "unit": "GB/s", "cores": ncpu, "kind": "port",
                                     "ms_per_product": dt_ref * 1e3,
                                     "sample": what + "; the reference's nested parallel regions (csr.h:445-448): every one "
                                               "of the %d threads computes every row" % ncpu}}


# ------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------
KERNEL_LABELS = {"two-pass": "fs::spmv_expand_kernel + fs::spmv_reduce_kernel (one product)",
                 "tiled": "fs::spmv_tiled_kernel", "lds-staged": "fs::spmv_ldsx_dma_kernel", "stream": "fs::spmv_stream_kernel"}


def _klabel(ka, kt=None):
    lab = KERNEL_LABELS.get(ka, ka)
    if kt is not None and kt != ka:   # the format builder's timed choice may differ between A and A'
        lab = "A: %s; A': %s" % (lab, KERNEL_LABELS.get(kt, kt))
    return lab


def _traffic(kernels, workload, rows, per):
    """PMC traffic per product: NOT measured by this run (counters need rocprofv3 --pmc passes of their own) -- the mean over the
    step's products of profiles/traffic_<workload>_<kernel>.json, which tools/refresh_traffic.py writes from a round's PMC run of
    this same command; the file's round and source ride along (roofline.traffic_from_profiles)"""
    global _TRAFFIC_SOURCE
    try:
        vals_, src = [], []
        for kn in kernels:
            name = "traffic_spmv_%s.json" % kn.replace("-", "_") if workload == "c2" else \
                   "traffic_%s_%s.json" % (workload, kn.replace("-", "_"))
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            if tj.get("rows") != rows or tj.get("per_row") != per:
                raise ValueError("traffic file is for another workload")
            vals_.append(float(tj["hbm_bytes_per_launch"]))
            src.append({"file": "profiles/" + name, "round": tj.get("round"), "pmc_summary": tj.get("source")})
        _TRAFFIC_SOURCE = src
        return sum(vals_) / len(vals_)
    except Exception:
        _TRAFFIC_SOURCE = None
        return None


_TRAFFIC_SOURCE = None


_LIVE_PMC_BROKEN = None      # the first failure of a live PMC pass: later workloads do not try again (the run must stay within minutes)


def live_pmc_traffic(args, workload="c2", kernels=("fs::spmv_expand_kernel", "fs::spmv_reduce_")):
    """roofline.traffic MEASURED BY THIS RUN (N = 1): two short children of this process -- `rocprofv3 --kernel-trace --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE`, separate passes as MI355X_MICROARCH.md's HBM section prescribes, the program itself behind `--` -- run
    this file's `workload` for a few steps (`--lean`: device vectors only, no probes, no CPU baseline), and the per-dispatch sums of
    the product's kernels give HBM bytes per product = sum over its kernels of (2 x FETCH_SIZE + WRITE_SIZE) KB (the x 2 is the
    guide's gfx950 correction: FETCH_SIZE counts a 128-byte request as 64).  Outside the timed region, after it.  Any failure (no
    rocprofv3, a time-out, an unexpected CSV) returns (None, reason): the line then keeps the value of profiles/."""
    global _LIVE_PMC_BROKEN
    if _LIVE_PMC_BROKEN:
        return None, {"error": "not tried: an earlier live PMC pass of this run failed (%s)" % _LIVE_PMC_BROKEN}
    total, how = _live_pmc_traffic(args, workload, kernels)
    if not total:
        _LIVE_PMC_BROKEN = how.get("error", "?")[:200]
    return total, how


def _live_pmc_traffic(args, workload, kernels):
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, {"error": "rocprofv3 not found"}
    per_counter, t0 = {}, time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="fs_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", tmp, "-o", "p", "--", "python3",
                   os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "1", "--rows", str(args.rows), "--per-row", str(args.per_row),
                   "--c5-rows", str(args.c5_rows), "--lean", "--no-cpu-baseline", "--no-reproducible-cost"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
                env.pop(k, None)
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=float(os.environ.get("FS_BENCH_PMC_TIMEOUT_S", "90")))
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, {"error": "the %s pass failed (exit %d): %s" % (counter, p.returncode, (p.stderr or p.stdout)[-300:])}
            sums = {}                                   # kernel -> dispatch -> value (a dispatch has one row per counter instance)
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != counter:
                        continue
                    for kn in kernels:
                        if kn in row["Kernel_Name"]:
                            d = sums.setdefault(kn, {})
                            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if set(sums) != set(kernels):
                return None, {"error": "the %s pass saw %s, not the product's kernels %s" % (counter, sorted(sums), list(kernels))}
            per_counter[counter] = {kn: (sum(v.values()) / len(v), len(v)) for kn, v in sums.items()}
        except subprocess.TimeoutExpired:
            return None, {"error": "the %s pass timed out" % counter}
        except Exception as ex:
            return None, {"error": repr(ex)}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    durations = None
    if workload == "c2":          # the headline: the same child once more with the kernel trace alone -- rocprofv3's own clock on the kernels
        tmp = tempfile.mkdtemp(prefix="fs_trace_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py"), "--workload", workload,
                   "--steps", "10", "--warmup", "2", "--rows", str(args.rows), "--per-row", str(args.per_row), "--lean", "--no-cpu-baseline", "--no-reproducible-cost"]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
                env.pop(k, None)
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=float(os.environ.get("FS_BENCH_PMC_TIMEOUT_S", "90")))
            ns = {}
            for f in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    for kn in kernels:
                        if kn in row["Kernel_Name"]:
                            ns.setdefault(kn, []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
            if p.returncode == 0 and set(ns) == set(kernels):
                durations = {kn: {"mean_ms": sum(v) / len(v) * 1e-6, "dispatches": len(v)} for kn, v in ns.items()}
                try:          # the traced child's own HIP-event figure: the two clocks on the SAME launches
                    child = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
                    durations["hip_events_ms_per_launch_in_the_traced_child"] = child["roofline"]["avg_launch_ms"]
                except Exception:
                    pass
        except Exception:
            durations = None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    parts = {kn: (2.0 * per_counter["FETCH_SIZE"][kn][0] + per_counter["WRITE_SIZE"][kn][0]) * 1024.0 for kn in kernels}
    return sum(parts.values()), {
        "measured_by_this_run": True, "bytes_per_kernel": parts,
        "mean_FETCH_SIZE_KB": {kn: per_counter["FETCH_SIZE"][kn][0] for kn in kernels},
        "mean_WRITE_SIZE_KB": {kn: per_counter["WRITE_SIZE"][kn][0] for kn in kernels},
        "dispatches_averaged": {kn: per_counter["FETCH_SIZE"][kn][1] for kn in kernels},
        "kernel_trace_ms": durations,
        "how": "two children after the timed region: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
               "--workload %s --steps 3 --warmup 1 --lean; per product: sum over its kernels of the mean per dispatch of 2 x FETCH_SIZE + WRITE_SIZE (KB; "
               "the x 2 is the gfx950 correction of MI355X_MICROARCH.md); the dispatches include the builder's own timing runs of the same kernels" % workload,
        "seconds": round(time.perf_counter() - t0, 1)}


def apply_live_traffic(rec, args, workload, kernels, alg_bytes):
    """replace roofline.traffic (and the ceiling derived from it) by what two rocprofv3 --pmc children of THIS run measure; the value
    of profiles/ stays beside it for comparison; a failure leaves the record as it was and says why"""
    if getattr(args, "lean", False) or getattr(args, "no_live_traffic", False):
        return
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ):
        rec["roofline"]["traffic_live_measurement_failed"] = {"error": "not tried: this run is itself under a profiler"}
        return
    live, how = live_pmc_traffic(args, workload, kernels)
    r = rec["roofline"]
    if not live:
        r["traffic_live_measurement_failed"] = how
        return
    r["traffic_from_profiles_for_comparison"] = {"traffic": r.get("traffic"), "files": (r.pop("traffic_from_profiles", None) or {}).get("files")}
    r["traffic"] = live
    r["traffic_measured"] = how
    if how.get("kernel_trace_ms"):
        tr = sum(v["mean_ms"] for v in how["kernel_trace_ms"].values() if isinstance(v, dict))
        r["rocprofv3_kernel_trace_ms_per_launch"] = tr      # a third child: `rocprofv3 --kernel-trace` alone, the product's kernels added up
        r["hip_events_over_kernel_trace"] = r["avg_launch_ms"] / tr if tr else None
        inchild = how["kernel_trace_ms"].get("hip_events_ms_per_launch_in_the_traced_child")
        r["hip_events_over_kernel_trace_same_process"] = inchild / tr if (tr and inchild) else None
    r["design_ceiling_frac"] = alg_bytes / live * HBM_COPY_GBS / HBM_PEAK_GBS
    r["frac_of_design_ceiling"] = r["frac"] / r["design_ceiling_frac"]
    r["design_ceiling_source"] = "algorithmic bytes / the PMC traffic this run measured x 6.29 TB/s copy rate (MI355X_MICROARCH.md) / 8 TB/s"


def _roofline(kernel, achieved, traffic, alg_bytes, avg_ms, launches, traffic_file=None):
    """the `roofline` object; design_ceiling_frac: algorithmic bytes / the bytes the kept kernel really moves (PMC) x what a copy
    reaches on this chip: the fraction of peak THIS algorithm could reach if both its streams ran at the copy rate"""
    r = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
         "avg_launch_ms": avg_ms, "launches_timed": launches}
    if traffic:
        r["traffic_from_profiles"] = {"measured_by_this_run": False, "files": _TRAFFIC_SOURCE,
                                      "how": "rocprofv3 --pmc passes of this command (2 x FETCH_SIZE + WRITE_SIZE per launch), tools/refresh_traffic.py"}
        r["design_ceiling_frac"] = alg_bytes / traffic * HBM_COPY_GBS / HBM_PEAK_GBS
        r["frac_of_design_ceiling"] = r["frac"] / r["design_ceiling_frac"]
        r["design_ceiling_source"] = ("algorithmic bytes / PMC traffic of the kept kernel (%s) x 6.29 TB/s copy rate "
                                      "(MI355X_MICROARCH.md) / 8 TB/s" % (traffic_file or "profiles/traffic_*.json"))
    return r


def one_time_costs(handles, what, products_per_step=1):
    """What an unmodified caller's FIRST call pays and what the result holds: per handle the phases of
    fs_matrix_build_ms -- arrays into HBM + validation, ordering, schedule, every candidate copy built, the candidates timed, the
    losers freed -- the HBM the handle holds afterwards, and the break-even: products after which the kept copy has paid for the
    one-time work against the kernel that needs no copy (the chunk-streaming kernel, timed by the builder on the same matrix).
    The reference's one-time step is new_csr / new_bcsr (csr.h:375-422, 30-67).  `handles`: [(label, Matrix, transposed)]."""
    out = {"what": what}
    total_ms = saved = 0.0
    held = [0, 0, 0]
    seen = set()
    for label, M, tr in handles:
        try:
            b = M.build_ms(tr)
            c = M.candidate_ms(tr)
            b["total_ms"] = round(sum(b.values()), 3)
            kept = M.kernel_name(tr)
            kept_ms = c.get(kept, 0.0)
            b["kept"] = kept
            b["builder_timed_ms"] = c
            if kept != "stream" and c.get("stream", 0) > 0 and kept_ms > 0:
                b["saves_ms_per_product_vs_no_copy"] = round(c["stream"] - kept_ms, 4)
                saved += c["stream"] - kept_ms
            total_ms += b["total_ms"]
            out[label] = b
            if id(M) not in seen:
                seen.add(id(M))
                for i, v in enumerate(M.device_bytes()):
                    held[i] += v
        except Exception as ex:
            out[label] = {"error": repr(ex)}
    out["build_s"] = round(total_ms * 1e-3, 4)
    out["hbm_held_bytes"] = dict(zip(("csr_and_schedule", "kept_single_vector_copy", "k_column_copies_and_scratch"), held))
    out["hbm_held_bytes"]["total"] = sum(held)
    # one step = one product per handle listed: the copies together save `saved` ms per step
    out["break_even_steps"] = round(total_ms / saved, 1) if saved > 0 else None
    out["break_even_products"] = round(total_ms / saved * len(handles), 1) if saved > 0 else None
    return out


def rccl_info(prov, world, rank, nccl, backend):
    """Proof in the line that the group really had N ranks on N devices: the rank count by an all-reduce of
    ones over the group, every rank's device index / PCI bus id / name (all-gathered), the RCCL version torch was built with"""
    import torch
    import torch.distributed as dist
    cdev = prov.dev if nccl else "cpu"
    ones = torch.ones(1, dtype=torch.float64, device=cdev)
    dist.all_reduce(ones)
    me = {"rank": rank, "pid": os.getpid()}
    try:
        pr = torch.cuda.get_device_properties(prov.dev)
        me.update({"device_index": prov.dev.index, "name": pr.name,
                   "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xff, getattr(pr, "pci_device_id", 0)),
                   "uuid": str(getattr(pr, "uuid", ""))})
    except Exception as ex:
        me["device_error"] = repr(ex)
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if nccl else None
    except Exception:
        pass
    return {"backend": "nccl (RCCL over xGMI)" if nccl else backend, "ranks_counted_by_all_reduce_of_ones": int(ones.item()),
            "world_size": world, "ranks": everyone, "rccl_version": ver,
            "distinct_devices": len({(r or {}).get("pci_bus_id", (r or {}).get("rank")) for r in everyone})}


class ExchangeFault(RuntimeError):
    """the exchange inside the product disagreed with the plain one (verify_overlap) and --strict-exchange asked for a loud stop"""

    def __init__(self, info):
        RuntimeError.__init__(self, "overlapped exchange disagrees with the whole-shard all-gather: %r" % (info,))
        self.info = info


def error_record(ex, args, world, rank):
    """the OTHER shape of the line (N > 1 only): what went wrong, where, and how to tell an exchange fault from a kernel fault"""
    import traceback
    tb = traceback.format_exception(type(ex), ex, ex.__traceback__)
    return {"metric": METRIC, "value": None, "unit": "GB/s", "n_gpus": world, "error": repr(ex), "error_rank": rank,
            "error_kind": "exchange_fault" if isinstance(ex, ExchangeFault) else "exception in the headline workload",
            "exchange_fault": getattr(ex, "info", None),
            "exchange": {"parts": args.parts, "mode": "conservative" if args.parts <= 1 else "overlapped", "how": getattr(args, "exchange", "allgather")},
            "traceback_tail": "".join(tb).splitlines()[-6:],
            "hint": "--parts 1 runs the conservative exchange (one whole-shard all-gather behind the local product): a line that fails "
                    "with --parts 4 and passes with --parts 1 points at the overlapped exchange, not at the kernels"}


def _exchange_fault(exchange_check, args):
    """None, or what verify_overlap found: which vector, the part / rank / row range of the first bad segment"""
    bad = {k: v for k, v in (exchange_check or {}).items() if v.get("checked") and v.get("mode") == "conservative"}
    if not bad:
        return None
    fault = {"vectors": bad, "parts": args.parts,
             "consequence": "every rank switched to ONE whole-shard all-gather behind the finished local product for these operators; the "
                            "numbers of this line are the conservative exchange's"}
    if getattr(args, "strict_exchange", False):
        raise ExchangeFault(fault)
    return fault


def _same_on_all_ranks(values, dev_or_cpu):
    """True when every rank holds the same few doubles (checksums of gathered vectors)"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev_or_cpu)
    lo_, hi_ = t.clone(), t.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo_, hi_))


def fixed_order_cost(prov, capi, products, kernel_name, how):
    """What fixed-order (run-to-run bit-identical) sums cost on this workload: the SAME handles, the same loop of ten steps, option
    "reproducible" toggled -- arrival order timed before and after, the smaller taken as the base.  `products()` enqueues one
    step's products; `kernel_name()` names the kernel(s) that run under the option."""
    try:
        def ten(reproducible):
            capi.set_option("reproducible", reproducible)
            for _ in range(2):
                products()
            e0, e1 = prov.event(), prov.event()
            e0.record()
            for _ in range(10):
                products()
            e1.record()
            prov.synchronize()
            return prov.elapsed_ms(e0, e1) / 10
        try:
            ms0 = ten(0)
            ms1 = ten(1)
            kname1 = kernel_name()
            ms0b = ten(0)
        finally:
            capi.set_option("reproducible", 0)
        base = min(ms0, ms0b)
        return {"kernel": kname1, "step_ms": ms1, "step_ms_arrival_order_same_loop": base, "how": how,
                "reproducible_cost_pct": 100.0 * (ms1 / base - 1.0)}
    except Exception as ex:
        return {"error": repr(ex)}


def config2_bound(alg_bytes, prov=None):
    """Why 0.60 of peak is out of reach on config 2, as numbers in the line (r4 item 2a: MEASURED BY THIS RUN).
    Uniformly random columns over an 80 MB x leave two ways to touch x[col]: (a) ONE cache-line request per non-zero -- the chip
    answers at most ~250 G of them per second even when every one hits L2, next to which the 12-byte entry stream has to run
    (0.31 ms at the copy rate); the two do not overlap in one kernel; or (b) no gathers at all -- every random access in LDS, which
    costs a pass over an intermediate: 28 bytes per entry (the kept two-pass pair), 2.33 x the algorithmic bytes.
    The gather and gather + stream times come from the library's own probes (csrc/fs_probes.hip: fs_debug_probe_gather /
    fs_debug_probe_mix, millisecond kernels run here, after the timed region); where a probe cannot run the numbers of earlier
    rounds are quoted under names that say so (*_from_profiles)."""
    import ctypes as C
    live, err = {}, None
    try:
        L = prov.capi.lib()
        L.fs_debug_probe_gather.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.fs_debug_probe_mix.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
        ms = C.c_float(0)
        n, table = 160_000_000, 80 << 20
        prov.synchronize()
        best = None
        for window in (256 << 10, 1 << 20, 2 << 20):            # L2-hit gathers: per-XCD windows of an 80 MB table
            prov.capi.check(L.fs_debug_probe_gather(n, table, window, 64, 5, C.byref(ms)), "fs_debug_probe_gather")
            live["gather_l2_window_%dKiB_ms" % (window >> 10)] = float(ms.value)
            best = float(ms.value) if best is None else min(best, float(ms.value))
        live["gather_floor_ms"] = best
        prov.capi.check(L.fs_debug_probe_gather(n, table, table, 1, 5, C.byref(ms)), "fs_debug_probe_gather")
        live["gather_uniform_over_80MB_ms"] = float(ms.value)       # what a plain CSR kernel's gathers cost: one fabric request each
        prov.capi.check(L.fs_debug_probe_mix(n, 2 << 20, 256, 0, 5, C.byref(ms)), "fs_debug_probe_mix")
        live["gathers_plus_stream_measured_ms"] = float(ms.value)   # small workgroups, no y at all
        prov.capi.check(L.fs_debug_probe_mix(n, 2 << 20, 1024, 256, 5, C.byref(ms)), "fs_debug_probe_mix")
        live["gathers_plus_stream_one_workgroup_per_cu_ms"] = float(ms.value)   # the shape an LDS-resident y slice forces
    except Exception as ex:
        live, err = {}, repr(ex)
    measured = "gather_floor_ms" in live
    gather_floor_ms = live["gather_floor_ms"] if measured else 0.606
    stream_floor_ms = 1.92e9 / (HBM_COPY_GBS * 1e9) * 1e3     # the 12-byte entry stream at the copy rate
    overlapped = max(gather_floor_ms, stream_floor_ms)
    two_pass_floor_ms = (28.0 * 160e6 + 2 * 8e7 + 4e7) / (HBM_COPY_GBS * 1e9) * 1e3
    mix_ms = live.get("gathers_plus_stream_measured_ms")
    best = min(overlapped, two_pass_floor_ms)
    rec = {"bounds_measured_by_this_run": measured,
           "gather_kernel_ceiling_frac": alg_bytes / (overlapped * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "gather_kernel_ceiling_if_not_overlapped_frac": alg_bytes / ((gather_floor_ms + stream_floor_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "stream_floor_ms_at_copy_rate": stream_floor_ms, "two_pass_floor_ms_at_copy_rate": two_pass_floor_ms,
           "any_kernel_ceiling_frac": alg_bytes / (best * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "copy_rate_GBs_from_guide": HBM_COPY_GBS,
           "bound_statement": "uniform-random columns over an 80 MB x: this algorithm's ceiling (design_ceiling_frac) and the ceiling of any "
                              "exact-fp64 kernel on this chip (any_kernel_ceiling_frac: the better of perfectly overlapped L2-hit gathers "
                              "and 28 bytes per entry at the copy rate); the 0.60 target needs 364 G gathers/s or 14 bytes per entry"}
    if measured:
        rec.update(live)
        rec["gather_kernel_as_measured_frac"] = alg_bytes / (mix_ms * 1e-3) / 1e9 / HBM_PEAK_GBS   # gathers + stream in ONE kernel, no y
        rec["bound_sources"] = ["libfastsparse_amd/csrc/fs_probes.hip, run by this process after the timed region (160 M gathers per launch, "
                                "median of 5)", "MI355X_MICROARCH.md (6.29 TB/s copy rate)",
                                "not re-run here: profiles/r04_probe_overlap.jsonl, r02_probe_hybrid_cu_mask.jsonl (gather role and stream role "
                                "side by side take the SUM of their times), r01_probe_mall.jsonl"]
    else:
        rec.update({"gather_floor_ms_from_profiles": 0.606, "gathers_plus_stream_ms_from_profiles": 0.772,
                    "from_profiles": {"files": ["profiles/r04_probe_gather.jsonl", "profiles/r04_probe_mix.jsonl"], "round": 4,
                                      "why_not_measured": err}})
    return rec


def run_c2(args, prov, world, rank, nccl, strong=False, out=None):
    """BASELINE config 2.  N = 1: y = A x, z = A' u.  N > 1: every rank a row shard (weak: a config-2 matrix per rank; strong:
    the one matrix cut over the ranks), both products as "rows + all-gather" with the exchange INSIDE the product
    (`out`, a dict, receives the final vectors: the gloo rehearsal in tests/ checks them against the oracle)"""
    import torch
    import torch.distributed as dist
    from libfastsparse_amd import dist as fsd
    dev = prov.dev
    per = args.per_row
    ncol = args.rows                    # column space stays config 2's at every N
    if strong and _multi(world):            # strong scaling: the one 10 M-row matrix cut into equal row shards
        n_global = args.rows
        sb = fsd.even_row_partition(n_global, world)
        lo, n_local = sb[rank], sb[rank + 1] - sb[rank]
    else:                               # weak scaling: every rank's shard is a config-2 matrix
        n_local = args.rows
        n_global = n_local * world
        lo = rank * n_local
    cdev = dev if (nccl and _multi(world)) else "cpu"
    nparts = max(1, args.parts)
    z_scheme = "local" if not _multi(world) else args.z_scheme
    z_err = None

    # ---- this rank's shard: rows lo .. lo+n_local of the n_global x 10M matrix ---------------------------------
    rp, cc, vv = prov.uniform(n_local, ncol, per, SEED_C2, row_offset=lo)
    bounds = fsd.even_row_partition(n_global, world)
    cb = fsd.even_row_partition(ncol, world)
    At = None
    if _multi(world) and z_scheme == "gather":
        # z = A' u as "row shards of A' + all-gather": this rank owns rows cb[rank] .. cb[rank+1] of A' (= those columns of
        # A), built once by an all-to-all of the entries; moves F doubles per product where the all-reduce moves 2 F
        try:
            tr, tc, tv = fsd.build_transposed_shard(rp, cc, vv, lo, cb)
            At = in_turns(lambda: prov.coo(cb[rank + 1] - cb[rank], n_global, tr.to(torch.int32), tc.to(torch.int32), tv),
                          prov, world, rank, nccl)
            del tr, tc, tv
        except Exception as ex:          # never ran on more than one GPU before the driver's run: keep the number, say why
            z_scheme, z_err, At = "reduce", repr(ex), None
        if _multi(world):                    # every rank takes the same scheme
            flag = torch.tensor([1.0 if z_scheme == "gather" else 0.0], dtype=torch.float64, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) == 0.0 and z_scheme == "gather":
                z_scheme, z_err, At = "reduce", "another rank could not build its shard of A'", None
    A = in_turns(lambda: prov.csr(n_local, ncol, rp, cc, vv, transpose=(z_scheme != "gather")), prov, world, rank, nccl)
    nnz_local = n_local * per
    bytes_a = csr_bytes(nnz_local, n_local, ncol)
    if z_scheme == "gather":
        nnz_t = int(At.nnz)
        bytes_t = csr_bytes(nnz_t, cb[rank + 1] - cb[rank], n_global)
    else:
        bytes_t = csr_bytes(nnz_local, ncol, n_local)   # same entries, nrow and ncol swap roles

    op_a = fsd.ShardedOperator(lambda yl, xf: prov.spmv(A, yl, xf), bounds, parts=prov.parts(A), copy_segments=prov.copy_segments, exchange=getattr(args, "exchange", "allgather"))
    if z_scheme == "gather":
        op_t = fsd.ShardedOperator(lambda zl, uf: prov.spmv(At, zl, uf), cb, parts=prov.parts(At), copy_segments=prov.copy_segments, exchange=getattr(args, "exchange", "allgather"))
    elif z_scheme == "reduce":
        op_t = fsd.TransposedShardedOperator(lambda zf, ul: prov.spmv(A, zf, ul, transposed=True), bounds)
    else:
        op_t = None

    x = prov.sin_vector(ncol, 7.0, 0.3)             # bench_a_mul_b.c:142
    u = prov.sin_vector(n_global, 11.0, -0.2)       # 2nd column of X2col, :145
    y = prov.empty(n_global)
    z = prov.empty(ncol)
    # first contact: is the exchange inside the product sound on this fabric?  (it never ran on more than one RCCL rank before
    # the driver's run; an operator whose overlapped and plain exchanges disagree falls back to the plain one, on every rank)
    exchange_check = fault = None
    if _multi(world):
        exchange_check = {"y": op_a.verify_overlap(y, x, nparts)}
        if z_scheme == "gather":
            exchange_check["z"] = op_t.verify_overlap(z, u, nparts)
        fault = _exchange_fault(exchange_check, args)        # --strict-exchange: raises on every rank (the verdict is collective)

    def step(ev=None, exchange=True):
        """N = 1: A x, A' u.  N > 1, the iterating consumer's order (cg.h:15-16): [A x in parts, the all-gather of the finished
        rows under the later parts], y complete; [A' u likewise], z complete.  ev[0..1] / ev[2..3] bracket this rank's kernels
        of the two products on the launch stream (the exchanges run on RCCL's stream)."""
        if not _multi(world):
            prov.spmv(A, y, x)
            prov.spmv(A, z, u, transposed=True)
            return
        if ev is not None:
            ev[0].record()
        if exchange:
            hy = op_a.apply_overlapped_async(y, x, nparts)
        else:
            op_a.local(y, x)
        if ev is not None:
            ev[1].record()
        if exchange:
            hy.wait()
        if ev is not None:
            ev[2].record()
        if z_scheme == "gather":
            if exchange:
                hz = op_t.apply_overlapped_async(z, u, nparts)
            else:
                op_t.local(z, u)
        else:
            op_t.apply_local(z, u)
            if exchange:
                hz = op_t.reduce_async(z)
        if ev is not None:
            ev[3].record()
        if exchange:
            hz.wait()

    elapsed, evs, region_ms = timed_steps(prov, step, lambda: None, args.steps, args.warmup, world, nccl)
    launches = 2 * args.steps
    if not _multi(world):
        avg_ms = region_ms / launches                # ONE event pair around the K steps
    else:
        avg_ms = sum(prov.elapsed_ms(e[0], e[1]) + prov.elapsed_ms(e[2], e[3]) for e in evs) / launches
    bytes_per_launch = (bytes_a + bytes_t) / 2.0
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    tb = torch.tensor([float(bytes_a + bytes_t)], dtype=torch.float64, device=cdev)
    if _multi(world):
        dist.all_reduce(tb)             # shards may differ by a row under strong scaling
    total_bytes = float(tb.item()) * args.steps
    value = total_bytes / elapsed / 1e9

    # the two products apart (outside the timed region): A x and A' u, a few runs each
    split = {}
    if not _multi(world):
        e = [prov.event() for _ in range(3)]
        e[0].record()
        for _ in range(5):
            prov.spmv(A, y, x)
        e[1].record()
        for _ in range(5):
            prov.spmv(A, z, u, transposed=True)
        e[2].record()
        prov.synchronize()
        split = {"A_mul_B_ms": prov.elapsed_ms(e[0], e[1]) / 5, "At_mul_B_ms": prov.elapsed_ms(e[1], e[2]) / 5}
        # SURVEY 8(d)'s timing protocol: every launch with its own event pair, min / median / mean of 20
        per_launch = {}
        for name, tr, yy, xx in (("A_mul_B", False, y, x), ("At_mul_B", True, z, u)) if not getattr(args, "lean", False) else ():
            pairs = [(prov.event(), prov.event()) for _ in range(20)]
            for a, b in pairs:
                a.record()
                prov.spmv(A, yy, xx, transposed=tr)
                b.record()
            prov.synchronize()
            ts = sorted(prov.elapsed_ms(a, b) for a, b in pairs)
            per_launch[name] = {"min": ts[0], "median": 0.5 * (ts[9] + ts[10]), "mean": sum(ts) / len(ts), "max": ts[-1], "launches": 20}
        split["per_launch_ms"] = per_launch
        # the drop-in call as a reference caller makes it (csr_A_mul_B with malloc'ed vectors, csr.h:425): x goes up and y comes
        # down over PCIe inside the call, overlapped with the kernels part by part.  Never `value`: the PCIe-inclusive rate.
        if hasattr(A, "spmv_host") and not getattr(args, "lean", False):
            try:
                import numpy as np
                xh = x.cpu().numpy()
                yh = np.empty(n_local, dtype=np.float64)
                A.spmv_host(yh, xh)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    A.spmv_host(yh, xh)
                    ts.append((time.perf_counter() - t0) * 1e3)
                ts.sort()
                split["host_pointer_call"] = {
                    "csr_A_mul_B_ms": ts[1], "effective_GBs_incl_pcie": bytes_a / (ts[1] * 1e-3) / 1e9,
                    "max_abs_diff_vs_device_resident": float((torch.from_numpy(yh).to(dev) - y).abs().max()),
                    "what": "median wall time of 3 calls with pageable host vectors (80 MB up, 80 MB down per call)"}
                del xh, yh
            except Exception as ex:
                split["host_pointer_call"] = {"error": repr(ex)}
    else:
        split = {"A_mul_B_ms": sum(prov.elapsed_ms(e[0], e[1]) for e in evs) / len(evs),
                 "At_mul_B_ms": sum(prov.elapsed_ms(e[2], e[3]) for e in evs) / len(evs)}
    noex = piped = None
    if _multi(world):   # the same loop without any exchange: what the exchanges cost on top of the local products
        noex, _, _ = timed_steps(prov, lambda ev=None: step(None, exchange=False), lambda: None, args.steps, 1, world, nccl)
        # and with the two products of a step treated as what BASELINE config 2 defines -- INDEPENDENT (x and u are given vectors):
        # y's exchange then runs under the local product of A' u, and only the end of the step waits for both vectors.  Reported
        # beside `value`, which keeps the iterating consumer's order (y complete before A' starts, cg.h:15-16).
        if z_scheme == "gather":
            def step_piped(ev=None):
                hy = op_a.apply_overlapped_async(y, x, nparts)
                hz = op_t.apply_overlapped_async(z, u, nparts)
                hy.wait()
                hz.wait()
            try:
                piped, _, _ = timed_steps(prov, step_piped, lambda: None, args.steps, 1, world, nccl)
            except Exception as ex:
                piped = None
                z_err = (z_err or "") + " pipelined loop: %r" % (ex,)

    # self-check outside the timed region (no oracle here: that is the tests' job): the products the timed loop left in
    # y and z against the storage-order chunk-streaming kernel on the same operands, and, at N > 1, every rank holding
    # the same gathered y / z
    self_check = {}
    try:
        if _multi(world):
            step(None)                                                # y, z of a complete step (the no-exchange loop ran last)
        y_ref = prov.empty(n_local)
        prov.spmv_strict(A, y_ref, x)
        self_check["A_mul_B_max_abs_diff_vs_storage_order_kernel"] = float((y[lo:lo + n_local] - y_ref).abs().max())
        if z_scheme == "gather":
            z_ref = prov.empty(cb[rank + 1] - cb[rank])
            prov.spmv_strict(At, z_ref, u)
            dz = float((z[cb[rank]:cb[rank + 1]] - z_ref).abs().max()) if z_ref.numel() else 0.0
        else:
            z_ref = prov.empty(ncol)
            prov.spmv_strict(A, z_ref, u[lo:lo + n_local], transposed=True)
            if _multi(world):
                fsd.all_reduce_sum(z_ref)                             # sum of the ranks' partial products
            dz = float((z - z_ref).abs().max())
        self_check["At_mul_B_max_abs_diff_vs_storage_order_kernel"] = dz
        if _multi(world):
            self_check["ranks_hold_identical_y_and_z"] = _same_on_all_ranks([float(y.sum()), float(z.sum())], cdev)
        self_check["ok"] = bool(self_check["A_mul_B_max_abs_diff_vs_storage_order_kernel"] <= 1e-11 and
                                dz <= 1e-11 * max(1, world if z_scheme == "reduce" else 8) and
                                self_check.get("ranks_hold_identical_y_and_z", True))
        del y_ref, z_ref
    except Exception as ex:   # a failed check must show in the line, not kill it
        self_check = {"ok": False, "error": repr(ex)}
    if out is not None:
        out["y"], out["z"], out["bounds"], out["z_scheme"] = y, z, bounds, z_scheme

    # what fixed-order sums cost: the same matrix created under "reproducible" keeps a kernel whose additions
    # have a fixed order -- since round 3 the two-pass pair itself with a pass 2 that walks a panel's products in stream order, one
    # wave per panel (the default pass 2 adds with sixteen waves in arrival order)
    repro = None
    if not _multi(world) and hasattr(prov, "capi") and not args.no_reproducible_cost:
        repro = fixed_order_cost(prov, prov.capi, lambda: A.spmv(y, x, prov.stream()), lambda: A.kernel_name(),
                                 "the same handle with option reproducible = 1: pass 2 of the two-pass pair then runs one wave per panel "
                                 "and adds in stream order (bit-identical run to run); ten products each")

    rccl = per_rank = None
    if _multi(world):       # collective: every rank takes part, rank 0 keeps the answer
        rccl = rccl_info(prov, world, rank, nccl, os.environ.get("FS_BENCH_BACKEND", "nccl"))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "local_products_GBs": achieved, "frac_of_hbm_peak": achieved / HBM_PEAK_GBS,
                                          "avg_local_product_ms": avg_ms})
    if rank != 0:
        return None
    stream_gbs = stream_probe(prov) if hasattr(prov, "torch") else None
    kname = prov.kernel_name(A)
    kname_t = prov.kernel_name(At) if z_scheme == "gather" else prov.kernel_name(A, True)
    F8 = 8.0 * ncol
    if not _multi(world):
        what = "BASELINE config 2: CSR %d x %d, %d nnz/row uniform, fp64, step = A_mul_B + At_mul_B" % (n_global, ncol, per)
    else:
        what = ("%s: CSR %d x %d, %d rows/GPU, %d nnz/row, step = (A_mul_B in %d parts with the all-gather of y inside the product) "
                "then (At_mul_B, %s)" % ("BASELINE config 2 cut over %d GPUs (strong scaling)" % world if strong else
                                         "config-2 shards, WEAK scaling by rows with x fixed at %d columns: every rank receives %.0f MB of y "
                                         "per product (y all-gather = 8 B x %d rows x (N - 1) ranks) next to a ~0.9 ms local product, so the "
                                         "curve is exchange-bound by construction, not a kernel regression" % (ncol, 8e-6 * (n_global - n_local), n_local),
                                         n_global, ncol, n_local, per, nparts,
                                         "row shards of A' + all-gather of z inside the product" if z_scheme == "gather" else
                                         "local A_r' u_r + all-reduce of z"))
    traffic = _traffic((kname, kname_t), "c2", n_local, per) if not _multi(world) else None
    rec = {
        "metric": METRIC, "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if (strong and _multi(world)) else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": what,
                   "rows_per_gpu": n_local, "nnz_per_gpu": nnz_local, "parallelism": "rows x%d" % world,
                   "pct_of_hbm_peak": 100.0 * value / (HBM_PEAK_GBS * world),
                   "stream_read_GBs_measured": stream_gbs,
                   "pct_of_measured_stream_read": 100.0 * achieved / stream_gbs if stream_gbs else None,
                   "A_mul_B_ms": split["A_mul_B_ms"], "At_mul_B_ms": split["At_mul_B_ms"],
                   "per_launch_ms": split.get("per_launch_ms"), "host_pointer_call": split.get("host_pointer_call"),
                   "self_check": self_check,
                   "kernel_A": kname, "kernel_At": kname_t,
                   "builder_timed_ms_A": prov.candidate_ms(A),
                   "builder_timed_ms_At": prov.candidate_ms(At) if z_scheme == "gather" else prov.candidate_ms(A, True)},
        "roofline": _roofline(_klabel(kname, kname_t), achieved, traffic, bytes_per_launch, avg_ms, launches,
                              "profiles/traffic_spmv_%s.json" % kname.replace("-", "_")),
    }
    if repro is not None:
        rec["config"]["fixed_order_sums"] = repro
        rec["config"]["reproducible_cost_pct"] = repro.get("reproducible_cost_pct")
    if hasattr(A, "build_ms"):
        rec["config"]["one_time"] = one_time_costs(
            [("A", A, False), ("At", At, False) if z_scheme == "gather" else ("At", A, True)],
            "device arrays in (borrowed in place: phase 0 is validation only; a drop-in caller adds the PCIe upload of 12 B per entry); "
            "A' = fs_matrix_build_transpose (device sort by column) or, at N > 1, this rank's row shard of A' from exchanged entries; "
            "one step = A x + A' u")
    if not _multi(world) and not strong and n_global == 10_000_000 and per == 16 and not getattr(args, "lean", False):
        rec["roofline"].update(config2_bound(bytes_per_launch, prov))
    if not _multi(world) and not strong and kname == "two-pass" and kname_t == "two-pass" and hasattr(prov, "capi"):
        apply_live_traffic(rec, args, "c2", ("fs::spmv_expand_kernel", "fs::spmv_reduce_"), bytes_per_launch)
    if _multi(world):
        rec["config"].update({
            "exchange_check": exchange_check,
            "exchange": {"y": "all-gather of the y shards, started part by part inside the product (%d parts)" % nparts,
                         "how": getattr(args, "exchange", "allgather"),
                         "z_scheme": z_scheme, "z_scheme_fallback_reason": z_err,
                         "bytes_received_per_rank_per_step": {
                             "y_all_gather": 8.0 * (n_global - n_local),
                             "z_row_shards_of_At_plus_all_gather": F8 * (world - 1) / world,
                             "z_all_reduce_ring": 2.0 * F8 * (world - 1) / world}},
            "ms_per_step_without_exchanges": noex / args.steps * 1e3 if noex else None,
            "ms_per_step_products_independent": piped / args.steps * 1e3 if piped else None,
            "value_products_independent_GBs": total_bytes / piped / 1e9 if piped else None,
            "products_independent_what": "the same two products per step with y's all-gather running under the local product of A' u (the "
                                         "two products of BASELINE config 2 are independent); `value` keeps y complete before A' starts",
            "rccl_status": "first contact: no multi-GPU machine was available to the builder; numbers above are the driver's"})
        rec["config"]["exchange"].update({"parts": nparts, "mode": "conservative (one whole-shard all-gather behind the local product)"
                                          if nparts <= 1 or (exchange_check or {}).get("y", {}).get("mode") == "conservative"
                                          else "overlapped (the all-gather of part p under part p + 1)"})
        rec["rccl"] = rccl
        rec["per_rank"] = per_rank
        rec["aggregate_frac_of_hbm_peak"] = value / (HBM_PEAK_GBS * world)
        if fault:
            rec["exchange_fault"] = fault
    if not _multi(world) and not args.no_cpu_baseline and hasattr(prov, "capi"):
        try:
            rec["cpu_baseline"] = cpu_baseline_c2(n_local, ncol, per)
        except Exception as ex:  # the baseline is a reported extra; its failure must not hide the GPU number
            rec["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
    return rec


def run_c3(args, prov, world, rank, nccl):
    """BASELINE config 3 through its own entry points: the COO arrays of a SparseBinaryMatrix go to the A_mul_B handle
    (fs_coo_create: stable device sort by row = new_bcsr's order, csr.h:30-67) and, swapped, to the At_mul_B handle"""
    import torch
    from libfastsparse_amd import capi
    if world != 1:
        raise SystemExit("--workload c3 is a one-GPU workload (BASELINE configs[2])")
    dev = prov.dev
    nrow, ncol, per = args.rows, max(args.rows // 10, 1), 64
    st = capi.current_stream()
    _, cols, _ = capi.synth_uniform(nrow, ncol, per, SEED_C3, valued=False)
    rows = torch.arange(nrow, device=dev, dtype=torch.int32).repeat_interleave(per)   # row-major COO, as SURVEY C3 says
    nnz = nrow * per
    prov.synchronize()
    t0 = time.perf_counter()
    A = capi.Matrix.from_coo(nrow, ncol, rows, cols, None)
    prov.synchronize()
    t_build_a = time.perf_counter() - t0
    t0 = time.perf_counter()
    At = capi.Matrix.from_coo(ncol, nrow, cols, rows, None)        # each output keeps the COO entry order (sparse.h:72-74)
    prov.synchronize()
    t_build_t = time.perf_counter() - t0
    del rows
    x = prov.int_vector(ncol, 3)
    u = prov.int_vector(nrow, 4)
    y = torch.empty(nrow, dtype=torch.float64, device=dev)
    z = torch.empty(ncol, dtype=torch.float64, device=dev)
    bytes_a = csr_bytes(nnz, nrow, ncol, valued=False)
    bytes_t = csr_bytes(nnz, ncol, nrow, valued=False)

    def step(ev=None):
        A.spmv(y, x, st)
        At.spmv(z, u, st)

    elapsed, _, region_ms = timed_steps(prov, step, lambda: None, args.steps, args.warmup, 1, nccl)
    avg_ms = region_ms / (2 * args.steps)            # ONE event pair around the K steps
    e = [prov.event() for _ in range(3)]             # the two products apart, outside the timed region
    e[0].record()
    for _ in range(5):
        A.spmv(y, x, st)
    e[1].record()
    for _ in range(5):
        At.spmv(z, u, st)
    e[2].record()
    prov.synchronize()
    ka, kt = [prov.elapsed_ms(e[0], e[1]) / 5], [prov.elapsed_ms(e[1], e[2]) / 5]
    bpl = (bytes_a + bytes_t) / 2.0
    achieved = bpl / (avg_ms * 1e-3) / 1e9
    value = (bytes_a + bytes_t) * args.steps / elapsed / 1e9
    # self-check: integer data, so the storage-order kernel must give the same bits
    sc = {}
    try:
        y2, z2 = torch.empty_like(y), torch.empty_like(z)
        capi.set_option("strict_order", 1)
        try:
            A.spmv(y2, x, st)
            At.spmv(z2, u, st)
        finally:
            capi.set_option("strict_order", 0)
        # integer checksum of checksums: sum_r y[r] == sum over the entries of x[col] (exact below 2^53)
        xl, tot = x.to(torch.int64), 0
        for a in range(0, nnz, 64_000_000):
            tot += int(xl[cols[a:a + 64_000_000].long()].sum().item())
        sc = {"A_mul_B_bit_identical_to_storage_order_kernel": bool(torch.equal(y, y2)),
              "At_mul_B_bit_identical_to_storage_order_kernel": bool(torch.equal(z, z2)),
              "integer_checksum_of_checksums": int(y.to(torch.int64).sum().item()) == tot}
        sc["ok"] = all(sc.values())
    except Exception as ex:
        sc = {"ok": False, "error": repr(ex)}
    ka_name, kt_name = A.kernel_name(), At.kernel_name()
    repro = None
    if not args.no_reproducible_cost:
        xs_, us_ = prov.sin_vector(ncol, 7.0, 0.3), prov.sin_vector(nrow, 11.0, -0.2)      # non-integer data: the order matters

        def both():
            A.spmv(y, xs_, st)
            At.spmv(z, us_, st)
        repro = fixed_order_cost(prov, capi, both, lambda: "A: %s; A': %s" % (A.kernel_name(), At.kernel_name()),
                                 "the same two handles with option reproducible = 1, x = sin: the LDS-staged kernel then waits for a "
                                 "phase's adds before its barrier (the builder keeps a row's entries of a work item with one wave) and "
                                 "chunks that share a panel (A') add their slices in turn; ten steps each")
        if "error" not in repro:
            try:                        # and the point of it: two runs, the same bits
                capi.set_option("reproducible", 1)
                both(); y1, z1 = y.clone(), z.clone()
                both()
                repro["two_runs_bit_identical"] = bool(torch.equal(y, y1) and torch.equal(z, z1))
            finally:
                capi.set_option("reproducible", 0)
        del xs_, us_
    rec = {
        "metric": "binary SpMV (A_mul_B + At_mul_B) effective GB/s (% HBM3E peak)", "value": value, "unit": "GB/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 3: SparseBinaryMatrix %d x %d, %d nnz/row, COO -> A_mul_B + At_mul_B, "
                               "integer-valued x" % (nrow, ncol, per),
                   "A_mul_B_ms": sum(ka) / len(ka), "At_mul_B_ms": sum(kt) / len(kt),
                   "A_mul_B_GBs": bytes_a / (sum(ka) / len(ka) * 1e-3) / 1e9,
                   "At_mul_B_GBs": bytes_t / (sum(kt) / len(kt) * 1e-3) / 1e9,
                   "device_coo_to_csr_and_format_s": {"A": t_build_a, "At": t_build_t},
                   "kernel_A": ka_name, "kernel_At": kt_name,
                   "builder_timed_ms_A": A.candidate_ms(), "builder_timed_ms_At": At.candidate_ms(), "self_check": sc},
        "roofline": _roofline(_klabel(ka_name, kt_name), achieved, _traffic((ka_name,), "c3", nrow, per), bpl, avg_ms,
                              2 * args.steps, "profiles/traffic_c3_%s.json" % ka_name.replace("-", "_")),
    }
    if repro is not None:
        rec["config"]["fixed_order_sums"] = repro
        rec["config"]["reproducible_cost_pct"] = repro.get("reproducible_cost_pct")
    if ka_name == "lds-staged" and kt_name == "lds-staged":      # (mean over the dispatches of A and A': the two products of a step)
        apply_live_traffic(rec, args, "c3", ("fs::spmv_ldsx_dma_kernel",), bpl)
    rec["config"]["one_time"] = one_time_costs(
        [("A", A, False), ("At", At, False)],
        "COO arrays already in HBM -> fs_coo_create twice (A from (rows, cols), A' from (cols, rows)): validation, stable device sort by "
        "row, every candidate copy built, timed, the losers freed; wall seconds per handle in device_coo_to_csr_and_format_s; a "
        "drop-in caller adds the PCIe upload of 8 B per entry per handle; one step = A_mul_B + At_mul_B")
    if not args.no_cpu_baseline:
        try:
            rec["cpu_baseline"] = cpu_baseline_c3(nrow, ncol, per, sample_rows=args.cpu_sample_rows or 2_500_000)
        except Exception as ex:
            rec["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
    return rec


def run_c4(args, prov, world, rank, nccl):
    """BASELINE config 4: csr_A_mul_Bn, A as config 2, X 10 M x 32 row-major (X[i,c] = sin(7i + 17c + 0.3),
    bench_a_mul_b.c:149).  k = 2, 4, 8 beside it."""
    import torch
    from libfastsparse_amd import capi
    if world != 1:
        raise SystemExit("--workload c4 is a one-GPU workload (BASELINE configs[3])")
    dev = prov.dev
    n, per, k = args.rows, args.per_row, 32
    st = capi.current_stream()
    rp, cc, vv = capi.synth_uniform(n, n, per, SEED_C2)
    A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)

    def make_x(kk):
        i = torch.arange(n, device=dev, dtype=torch.float64)[:, None]
        c = torch.arange(kk, device=dev, dtype=torch.float64)[None, :]
        return torch.sin(7.0 * i + 17.0 * c + 0.3).contiguous()

    X = make_x(k)
    Y = torch.empty(n, k, dtype=torch.float64, device=dev)
    nbytes = A.algorithmic_bytes(k)
    capi.set_option("spmm_kernel", args.spmm_kernel)

    A.prepare(k, st)                       # nothing to build for k = 32 (row kernel); the call is part of the protocol

    def step(ev=None):
        A.spmm(Y, X, k, st)

    elapsed, _, region_ms = timed_steps(prov, step, lambda: None, args.steps, args.warmup, 1, nccl)
    avg_ms = region_ms / args.steps
    achieved = nbytes / (avg_ms * 1e-3) / 1e9
    value = nbytes * args.steps / elapsed / 1e9
    sc = {}
    try:     # column j of Y against the single-vector product of column j of X, both in storage order: identical bits
        yj = torch.empty(n, dtype=torch.float64, device=dev)
        capi.set_option("strict_order", 1)
        try:
            ok = True
            for j in (0, 17, 31):
                A.spmv(yj, X[:, j].contiguous(), st)
                ok = ok and bool(torch.equal(Y[:, j], yj))
        finally:
            capi.set_option("strict_order", 0)
        sc = {"columns_bit_identical_to_storage_order_spmv": ok, "ok": ok}
    except Exception as ex:
        sc = {"ok": False, "error": repr(ex)}
    del X, Y
    capi.set_option("spmm_kernel", 0)
    small = {}
    for kk in (2, 4, 8):
        Xk, Yk = make_x(kk), torch.empty(n, kk, dtype=torch.float64, device=dev)
        prov.synchronize()
        t0 = time.perf_counter()
        A.prepare(kk, st)                 # the k-column two-pass copy (k = 2, 4): one-time work, never inside a product
        prov.synchronize()
        t_prep = time.perf_counter() - t0
        for _ in range(2):
            A.spmm(Yk, Xk, kk, st)
        e0, e1 = prov.event(), prov.event()
        e0.record()
        for _ in range(5):
            A.spmm(Yk, Xk, kk, st)
        e1.record()
        prov.synchronize()
        ms = e0.elapsed_time(e1) / 5
        small["k%d" % kk] = {"ms": ms, "GBs": A.algorithmic_bytes(kk) / (ms * 1e-3) / 1e9,
                             "frac_of_peak": A.algorithmic_bytes(kk) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "kernel": A.spmm_plan(kk), "prepare_s": t_prep}
        del Xk, Yk
    rec = {
        "metric": "fp64 CSR SpMM (k = 32) effective GB/s (% HBM3E peak)", "value": value, "unit": "GB/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 4: CSR %d x %d, %d nnz/row, times X %d x %d row-major (csr_A_mul_Bn)" % (n, n, per, n, k),
                   "gather_traffic_model_bytes": n * per * 8 * k, "flops_per_product": 2 * k * n * per,
                   "small_k": small, "spmm_kernel_option": args.spmm_kernel,
                   "mfma": "none: the product's k = 32 kernel is VALU (profiles/r02_c4_*: SQ_INSTS_VALU_MFMA_F64 = 0); "
                           "--spmm-kernel 4 runs the v_mfma_f64_16x16x4_f64 experiment" if args.spmm_kernel != 4 else
                           "v_mfma_f64_16x16x4_f64 row kernel (1/16 of each instruction's multiply-adds useful)",
                   "self_check": sc},
        "roofline": _roofline("fs::spmm_kernel<valued, 5> (32 lanes per row)" if args.spmm_kernel != 4 else
                              "fs::spmm_mfma_kernel<valued>", achieved, _traffic(("spmm_k32",), "c4", n, per), nbytes, avg_ms,
                              args.steps, "profiles/traffic_c4_spmm_k32.json"),
    }
    if args.spmm_kernel != 4:
        apply_live_traffic(rec, args, "c4", ("fs::spmm_kernel<true, 5>",), nbytes)
    rec["config"]["hbm_held_by_the_handle_bytes"] = dict(zip(("csr_and_schedule", "kept_single_vector_copy", "k_column_copies_and_scratch"),
                                                             A.device_bytes()))
    rec["config"]["one_time"] = one_time_costs(
        [("A", A, False)], "device arrays in (borrowed); k = 32 runs on the row kernel and needs no copy of its own (the single-vector copy "
                           "the builder keeps serves k <= 4 and the solvers); the k-column copies of k = 2 / 4: small_k[*].prepare_s")
    released = A.release_prepared(0)
    rec["config"]["hbm_held_after_release_prepared_bytes"] = dict(zip(("csr_and_schedule", "kept_single_vector_copy", "k_column_copies_and_scratch"),
                                                                      A.device_bytes()), released=released)
    if not args.no_cpu_baseline:
        try:
            rec["cpu_baseline"] = cpu_baseline_c4(n, per, k, sample_rows=args.cpu_sample_rows or 400_000)
        except Exception as ex:
            rec["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
    return rec


def c5_partition(prov, n_global, parts):
    """row lengths of the whole power-law matrix (generated in slabs on this rank's device) -> cut points with equal
    non-zero counts (libfastsparse_amd/dist.py nnz_balanced_partition on the 64-bit prefix sums), total non-zeros"""
    import torch
    slab = 25_000_000
    sums = []
    total = 0
    for lo in range(0, n_global, slab):
        lens = prov.powerlaw_lengths(min(slab, n_global - lo), lo)
        c = torch.cumsum(lens.to(torch.int64), 0)
        sums.append(c + total)
        total += int(c[-1].item())
        del lens
    prefix = torch.cat(sums)                 # prefix[i] = non-zeros of rows 0 .. i
    del sums
    targets = torch.tensor([total * r // parts for r in range(1, parts)], dtype=torch.int64, device=prefix.device)
    # first row index whose exclusive prefix >= target  (== searchsorted on row_ptr, row_ptr[i+1] = prefix[i])
    cuts = torch.searchsorted(prefix, targets, right=False) + 1
    cuts = torch.clamp(cuts, 0, n_global)
    b = [0] + [int(v) for v in cuts.cpu()] + [n_global]
    for i in range(1, len(b)):
        b[i] = max(b[i], b[i - 1])
    cum = [0] + [int(prefix[b[i] - 1].item()) if b[i] > 0 else 0 for i in range(1, len(b))]   # non-zeros in front of each cut
    del prefix
    return b, cum, total


def c5_shard(prov, lo, hi, ncol):
    """rows [lo, hi) of the power-law matrix as a local CSR (local row_ptr, GLOBAL column ids)"""
    import torch
    lens = prov.powerlaw_lengths(hi - lo, lo)
    rp64 = torch.zeros(hi - lo + 1, dtype=torch.int64, device=lens.device)
    torch.cumsum(lens, 0, out=rp64[1:])
    nnz = int(rp64[-1].item())
    if nnz > 2**31 - 1:
        raise SystemExit("shard holds %d non-zeros: more than an int row_ptr (csr.h:363) can index; use more ranks" % nnz)
    rp = rp64.to(torch.int32)
    del rp64, lens
    cc, vv = prov.fill(rp, ncol, lo)
    return rp, cc, vv, nnz


def run_c5(args, prov, world, rank, nccl, out=None):
    """BASELINE config 5 (`out`, a dict, receives the final vectors: the gloo rehearsal in tests/ checks them against
    the oracle's product with the whole matrix).  Timed loop 1: y = A x (local product in parts, all-gather inside the product);
    with --transpose a second timed loop adds z = A' u on row shards of A'."""
    import torch
    import torch.distributed as dist
    from libfastsparse_amd import dist as fsd
    n_global = args.c5_rows
    ncol = n_global
    parts = world if _multi(world) else 8
    mine = rank if _multi(world) else min(3, parts - 1)     # N = 1: the shard rank 3 of 8 owns
    if not _multi(world) and os.environ.get("FS_C5_PARTS"):  # rehearsal: the shard a rank of an N = 2 / 4 run would hold, on one GPU
        parts = int(os.environ["FS_C5_PARTS"])
        mine = min(int(os.environ.get("FS_C5_RANK", "0")), parts - 1)
    nparts = max(1, args.parts)
    cdev = prov.dev if (nccl and _multi(world)) else "cpu"
    bounds, cum_nnz, total_nnz = c5_partition(prov, n_global, parts)
    lo, hi = bounds[mine], bounds[mine + 1]
    rp, cc, vv, nnz = c5_shard(prov, lo, hi, ncol)
    n_local = hi - lo
    A = in_turns(lambda: prov.csr(n_local, ncol, rp, cc, vv), prov, world, rank, nccl)
    bytes_local = csr_bytes(nnz, n_local, ncol)
    if hasattr(prov, "trim"):
        prov.trim()

    x = prov.sin_vector(ncol, 7.0, 0.3)
    if _multi(world):
        op = fsd.ShardedOperator(lambda yl, xf: prov.spmv(A, yl, xf), bounds, parts=prov.parts(A), copy_segments=prov.copy_segments, exchange=getattr(args, "exchange", "allgather"))
        y = prov.empty(n_global)
    else:
        op, y = None, prov.empty(n_local)

    # A' u as "row shards of A' + all-gather" (N > 1 with --transpose): columns are uniform, so an even cut of the
    # rows of A' is balanced; the shard is built once by an all-to-all of the entries
    opt = At = None
    t_err = None
    bytes_t = 0
    if _multi(world) and args.transpose:
        try:
            cb = fsd.even_row_partition(ncol, world)
            tr, tc, tv = fsd.build_transposed_shard(rp, cc, vv, lo, cb)
            At = in_turns(lambda: prov.coo(cb[rank + 1] - cb[rank], n_global, tr.to(torch.int32), tc.to(torch.int32), tv),
                          prov, world, rank, nccl)
            del tr, tc, tv
            if hasattr(prov, "trim"):
                prov.trim()
            opt = fsd.TransposedGatherOperator(lambda zl, uf: prov.spmv(At, zl, uf), cb)
            opt.parts, opt.copy_segments, opt.exchange = prov.parts(At), prov.copy_segments, getattr(args, "exchange", "allgather")
            u = prov.sin_vector(n_global, 11.0, -0.2)
            z = prov.empty(ncol)
            bytes_t = csr_bytes(int(At.nnz), cb[rank + 1] - cb[rank], n_global)
        except Exception as ex:      # the transposed direction is an extra: report, keep the config-5 number
            opt, t_err = None, repr(ex)
        ok_t = torch.tensor([1.0 if opt is not None else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if float(ok_t.item()) == 0.0 and opt is not None:
            opt, t_err = None, "another rank could not build its shard of A'"

    exchange_check = fault = None
    if _multi(world):
        exchange_check = {"y": op.verify_overlap(y, x, nparts)}
        if opt is not None:
            exchange_check["z"] = opt.verify_overlap(z, u, nparts)
        fault = _exchange_fault(exchange_check, args)

    def step(ev=None, exchange=True, transpose=False):
        if not _multi(world):
            ev = None                  # one GPU: ONE event pair around the K steps (timed_steps)
        if ev is not None:
            ev[0].record()
        if not _multi(world):
            prov.spmv(A, y, x)
        elif exchange:
            hy = op.apply_overlapped_async(y, x, nparts)
        else:
            op.local(y, x)
        if ev is not None:
            ev[1].record()
        if _multi(world) and exchange:
            hy.wait()
        if transpose:
            if ev is not None:
                ev[2].record()
            if exchange:
                hz = opt.apply_overlapped_async(z, u, nparts)
            else:
                opt.local(z, u)
            if ev is not None:
                ev[3].record()
            if exchange:
                hz.wait()

    elapsed, evs, region_ms = timed_steps(prov, step, lambda: None, args.steps, args.warmup, world, nccl)
    local_ms = region_ms / args.steps if not _multi(world) else sum(prov.elapsed_ms(e[0], e[1]) for e in evs) / len(evs)

    def all_ranks(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device=cdev)
        if _multi(world):
            dist.all_reduce(t)
        return float(t.item())

    bytes_all = all_ranks(bytes_local)
    value = bytes_all * args.steps / elapsed / 1e9
    noex = with_t = None
    if _multi(world):   # the same loop without any exchange: what the all-gather costs on top of the local products
        noex, _, _ = timed_steps(prov, lambda ev=None: step(None, exchange=False), lambda: None, args.steps, 1, world, nccl)
    if opt is not None:
        el_t, evs_t, _ = timed_steps(prov, lambda ev=None: step(ev, transpose=True), lambda: None, args.steps, 1, world, nccl)
        noex_t, _, _ = timed_steps(prov, lambda ev=None: step(None, exchange=False, transpose=True), lambda: None, args.steps, 1,
                                   world, nccl)
        bytes_both = bytes_all + all_ranks(bytes_t)
        with_t = {"value": bytes_both * args.steps / el_t / 1e9, "unit": "GB/s", "ms_per_step": el_t / args.steps * 1e3,
                  "ms_per_step_without_exchanges": noex_t / args.steps * 1e3,
                  "local_At_mul_B_ms_rank0": sum(prov.elapsed_ms(e[2], e[3]) for e in evs_t) / len(evs_t),
                  "pct_of_hbm_peak": 100.0 * bytes_both * args.steps / el_t / 1e9 / (HBM_PEAK_GBS * world),
                  "step": "y = A x, then z = A' u on row shards of A' (all-gather of y and of z inside the products)"}

    sc = {}
    try:
        if _multi(world):
            step(None, transpose=opt is not None)     # complete vectors (a no-exchange loop ran last)
        if out is not None:
            out["y"], out["z"], out["bounds"] = y, (z if opt is not None else None), bounds
        if _multi(world):
            sc["ranks_hold_identical_y"] = _same_on_all_ranks([float(y.sum())], cdev)
        y_ref = prov.empty(n_local)
        prov.spmv_strict(A, y_ref, x)
        mine_y = y[lo:hi] if _multi(world) else y
        lens = (rp[1:] - rp[:-1]).to(torch.float64).clamp(min=1.0)
        sc["rows_within_1e-12_x_row_length_of_storage_order_kernel"] = bool(((mine_y - y_ref).abs() <= 1e-12 * lens).all())
        sc["ok"] = all(v for v in sc.values())
    except Exception as ex:
        sc = {"ok": False, "error": repr(ex)}
    repro = None
    if not _multi(world) and hasattr(prov, "capi") and not args.no_reproducible_cost:
        repro = fixed_order_cost(prov, prov.capi, lambda: prov.spmv(A, y, x), lambda: prov.kernel_name(A),
                                 "the same handle with option reproducible = 1: one-wave pass 2 in stream order, the long rows each with "
                                 "one wave of a workgroup and the workgroups' sums added in workgroup order; ten products each")
        if "error" not in repro:
            try:
                prov.capi.set_option("reproducible", 1)
                prov.spmv(A, y, x); y1 = y.clone()
                prov.spmv(A, y, x)
                repro["two_runs_bit_identical"] = bool(torch.equal(y, y1))
                del y1
            finally:
                prov.capi.set_option("reproducible", 0)
    # what the handle holds, and what is left once the plain CSR is given back: the products of this matrix run on
    # the kept copy alone; strict_order -- the self-check above -- was the last user of the plain arrays
    release = None
    if hasattr(A, "release_csr") and not _multi(world):
        try:
            import torch
            before = A.device_bytes()
            free0 = torch.cuda.mem_get_info()[0]
            kn0 = prov.kernel_name(A)
            n_rel = A.release_csr()
            del rp, cc, vv                        # the borrowed arrays are the caller's: now they can go
            prov.trim()
            free1 = torch.cuda.mem_get_info()[0]
            after = A.device_bytes()
            y_chk = prov.empty(n_local)
            prov.spmv(A, y_chk, x)                 # still the same product
            same = bool(((y_chk - (y[lo:hi] if _multi(world) else y)).abs() <= 1e-12 * lens).all())
            release = {"sides_released": n_rel, "kernel": kn0, "hbm_held_bytes_before": {"caller_csr_arrays_borrowed": 16 * nnz - 4 * nnz + 4 * (n_local + 1),
                                                                                      "handle": sum(before)},
                       "hbm_held_bytes_after": {"caller_csr_arrays_borrowed": 0, "handle": sum(after)},
                       "device_memory_freed_bytes": int(free1 - free0), "handle_over_its_copy": sum(after) / max(after[1], 1),
                       "product_after_release_matches": same}
            del y_chk
        except Exception as ex:
            release = {"error": repr(ex)}
    per_rank = rccl = None
    if _multi(world):
        rccl = rccl_info(prov, world, rank, nccl, os.environ.get("FS_BENCH_BACKEND", "nccl"))
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "rows": n_local, "nnz": nnz, "local_A_mul_B_ms": local_ms,
                                          "frac_of_hbm_peak": bytes_local / (local_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    if rank != 0:
        return None
    kname = prov.kernel_name(A)
    achieved = bytes_local / (local_ms * 1e-3) / 1e9
    rec = {
        "metric": METRIC, "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("BASELINE config 5: CSR %d x %d power-law (mean %.1f nnz/row, clipped at %d), rows cut by "
                                "non-zeros over %d GPUs, x replicated, step = local SpMV in %d parts with the all-gather of y "
                                "inside the product" % (n_global, ncol, total_nnz / n_global, C5_MAXLEN, world, nparts)) if _multi(world) else
                               ("ONE shard of BASELINE config 5 (CSR %d x %d power-law, %d non-zeros in all): the rows rank %d of "
                                "%d owns under the nnz-balanced cut; one GPU cannot hold the matrix in a struct CSR "
                                "(2^31-1 non-zeros)" % (n_global, ncol, total_nnz, mine, parts)),
                   "total_nnz": total_nnz, "rows_this_rank": n_local, "nnz_this_rank": nnz,
                   "row_bounds": bounds if len(bounds) <= 17 else None,
                   "pct_of_hbm_peak": 100.0 * value / (HBM_PEAK_GBS * world),
                   "local_A_mul_B_ms_rank0": local_ms,
                   "ms_per_step_without_exchanges": noex / args.steps * 1e3 if noex else None,
                   "with_transpose": with_t, "transpose_error": t_err, "kernel": kname,
                   "builder_timed_ms": prov.candidate_ms(A), "self_check": sc},
        "roofline": _roofline(_klabel(kname), achieved, _traffic((kname,), "c5", n_local, 0) if not _multi(world) else None, bytes_local,
                              local_ms, args.steps, "profiles/traffic_c5_%s.json" % kname.replace("-", "_")),
    }
    if repro is not None:
        rec["config"]["fixed_order_sums"] = repro
        rec["config"]["reproducible_cost_pct"] = repro.get("reproducible_cost_pct")
    if not _multi(world) and kname == "two-pass" and hasattr(prov, "capi"):
        apply_live_traffic(rec, args, "c5", ("fs::spmv_expand_kernel", "fs::spmv_longrows_kernel", "fs::spmv_reduce_", "fs::tiled_combine_kernel"),
                           bytes_local)
    if hasattr(A, "build_ms"):
        rec["config"]["one_time"] = one_time_costs([("A", A, False)], "this rank's shard: device arrays in (borrowed), candidates built and timed")
    if release is not None:
        rec["config"]["release_csr"] = release
    if _multi(world):
        rec["config"]["rccl_status"] = "first contact: no multi-GPU machine was available to the builder"
        rec["config"]["exchange_check"] = exchange_check
        rec["config"]["exchange"] = {"parts": nparts, "mode": "conservative" if nparts <= 1 or exchange_check["y"].get("mode") == "conservative" else "overlapped"}
        rec["rccl"], rec["per_rank"] = rccl, per_rank
        rec["aggregate_frac_of_hbm_peak"] = value / (HBM_PEAK_GBS * world)
        if fault:
            rec["exchange_fault"] = fault
    if not _multi(world) and not args.no_cpu_baseline and hasattr(prov, "capi"):
        try:
            rec["cpu_baseline"] = cpu_baseline_c5(lo, ncol, sample_rows=args.cpu_sample_rows or 1_000_000)
        except Exception as ex:
            rec["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}
    return rec


def run_cg(args, prov, world, rank, nccl):
    """SURVEY 8f-1, the consumer of the path: bsbm_cg / bsbm_cg2 (cg.h:25-82, 85-187) device resident on config 2's PATTERN (the
    reference's CG runs on BlockedSBM, pattern-only): (A'A + 5 I) x = b to 1e-8, one and two right-hand sides.  Step = one solve;
    the record's value = the algorithmic bytes of the two products of every iteration / time (the vector kernels ride on top)."""
    import ctypes as C
    import torch
    from libfastsparse_amd import capi
    if world != 1:
        raise SystemExit("the CG sub-record is a one-GPU workload")
    dev = prov.dev
    n, per = args.rows, args.per_row
    st = capi.current_stream()
    rp, cc, _ = capi.synth_uniform(n, n, per, SEED_C2, valued=False)
    A = capi.Matrix.from_csr(n, n, rp, cc, None, borrow=True)
    rows = torch.arange(n, device=dev, dtype=torch.int32).repeat_interleave(per)
    At = capi.Matrix.from_coo(n, n, cc, rows, None)           # A' as its own handle: the reference passes B and Bt
    del rows
    i = torch.arange(n, device=dev, dtype=torch.float64)
    b1 = torch.sin(19.0 * i + 0.4)
    b2 = torch.stack([b1, torch.cos(23.0 * i + 0.7)], 1).contiguous()
    L = capi.lib()
    out = {}
    for k, b in ((1, b1), (2, b2)):
        x = torch.empty_like(b)
        it = C.c_int(0)
        f = L.fs_cg if k == 1 else L.fs_cg2
        capi.check(f(A.h, At.h, x.data_ptr(), b.data_ptr(), 5.0, 1e-8, C.byref(it), st), "fs_cg")      # warm (k = 2: prepares the copies)
        prov.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(1, args.steps // 10)):
            capi.check(f(A.h, At.h, x.data_ptr(), b.data_ptr(), 5.0, 1e-8, C.byref(it), st), "fs_cg")
        prov.synchronize()
        dt = (time.perf_counter() - t0) / max(1, args.steps // 10)
        iters = it.value + 1
        # self-check: the residual of the returned x, by two products and an axpy outside the solver
        bb, xx = b.reshape(n, k), x.reshape(n, k)
        worst = 0.0
        t = torch.empty(n, dtype=torch.float64, device=dev)
        q = torch.empty(n, dtype=torch.float64, device=dev)
        for j in range(k):
            xj = xx[:, j].contiguous()
            A.spmv(t, xj, st)
            At.spmv(q, t, st)
            r = bb[:, j] - (q + 5.0 * xj)
            worst = max(worst, float(r.norm() / bb[:, j].norm()))
        bytes_it = 2 * csr_bytes(n * per, n, n, valued=False, k=k)
        out[k] = {"iterations": iters, "ms_per_solve": dt * 1e3, "ms_per_iteration": dt * 1e3 / iters,
                  "products_GBs": bytes_it * iters / dt / 1e9, "relative_residual": worst, "ok": worst <= 2e-8,
                  "product_kernel": A.kernel_name() if k == 1 else A.spmm_plan(2)}
    v = out[2]
    return {"metric": "block CG (2 right-hand sides) on config 2's pattern: algorithmic GB/s of its products", "value": v["products_GBs"],
            "unit": "GB/s", "n_gpus": 1, "steps": max(1, args.steps // 10), "warmup": 1, "ms_per_step": v["ms_per_solve"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SURVEY 8f-1: bsbm_cg / bsbm_cg2 device resident, (A'A + 5 I) x = b, A = config 2's pattern %d x %d x %d"
                                   % (n, n, per), "cg": out[1], "cg2": out[2],
                       "self_check": {"relative_residuals": [out[1]["relative_residual"], out[2]["relative_residual"]],
                                      "ok": out[1]["ok"] and out[2]["ok"]}},
            "roofline": _roofline("fs::spmm_expand_kernel<false, 2> + fs::spmm_reduce_kernel<2> (the two products of an iteration)",
                                  v["products_GBs"], None, csr_bytes(n * per, n, n, valued=False, k=2),
                                  v["ms_per_iteration"] / 2.0, 2 * v["iterations"] * max(1, args.steps // 10))}


def run_cg_dist(args, prov, world, rank, nccl):
    """The iterating consumer across the ranks: ShardedCG -- bsbm_cg (cg.h:25-82) row-sharded, scheme "gather"
    (both directions rows + all-gather, the y exchange inside the product, slices of x / r / p per rank, dots as 1-double
    all-reduces) -- on config 2's PATTERN, weak scaling by rows: (A'A + 5 I) x = b to 1e-8.  value = algorithmic bytes of the two
    products of every iteration over all ranks / max-over-ranks time."""
    import torch
    import torch.distributed as dist
    from libfastsparse_amd import dist as fsd
    per = args.per_row
    n_local, ncol = args.rows, args.rows
    n_global, lo = n_local * world, rank * n_local
    cdev = prov.dev if nccl else "cpu"
    rp, cc, _ = prov.uniform(n_local, ncol, per, SEED_C2, row_offset=lo, valued=False)
    bounds = fsd.even_row_partition(n_global, world)
    cb = fsd.even_row_partition(ncol, world)
    tr, tc, _ = fsd.build_transposed_shard(rp, cc, None, lo, cb)
    At = in_turns(lambda: prov.coo(cb[rank + 1] - cb[rank], n_global, tr.to(torch.int32), tc.to(torch.int32), None), prov, world, rank, nccl)
    del tr, tc
    A = in_turns(lambda: prov.csr(n_local, ncol, rp, cc, None), prov, world, rank, nccl)
    op_a = fsd.ShardedOperator(lambda yl, xf: prov.spmv(A, yl, xf), bounds, parts=prov.parts(A), copy_segments=prov.copy_segments, exchange=getattr(args, "exchange", "allgather"))
    op_t = fsd.ShardedOperator(lambda zl, uf: prov.spmv(At, zl, uf), cb, copy_segments=prov.copy_segments, exchange=getattr(args, "exchange", "allgather"))
    cg = fsd.ShardedCG(op_a, op_t, scheme="gather", nparts=max(1, args.parts))
    b = prov.sin_vector(ncol, 19.0, 0.4)
    x, it = cg.solve(b, 5.0, 1e-8)                         # warm
    prov.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    x, it = cg.solve(b, 5.0, 1e-8)
    prov.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    iters = it + 1
    # residual of the returned x through the operators themselves
    y = prov.empty(n_global)
    op_a.apply(y, x)
    q = prov.empty(cb[rank + 1] - cb[rank])
    prov.spmv(At, q, y)
    sl = slice(cb[rank], cb[rank + 1])
    r = b[sl] - (q + 5.0 * x[sl])
    nn = torch.stack([torch.dot(r, r), torch.dot(b[sl], b[sl])]).to(torch.float64)
    nn = nn.to(cdev) if not nccl else nn
    dist.all_reduce(nn)
    rel = float((nn[0] / nn[1]).sqrt().item())
    same = _same_on_all_ranks([float(x.sum())], cdev)
    if rank != 0:
        return None
    bytes_it = world * (csr_bytes(n_local * per, n_local, ncol, valued=False) + csr_bytes(n_local * per, ncol // world, n_global, valued=False))
    return {"metric": "row-sharded CG on config 2's pattern: algorithmic GB/s of its products (all ranks)", "value": bytes_it * iters / dt / 1e9,
            "unit": "GB/s", "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SURVEY 8f-1 across the ranks: ShardedCG (bsbm_cg row-sharded, scheme gather, %d parts), (A'A + 5 I) x = b, A = "
                                   "config-2 pattern shards %d x %d, %d rows/GPU" % (max(1, args.parts), n_global, ncol, n_local),
                       "iterations": iters, "ms_per_iteration": dt * 1e3 / iters,
                       "exchange_per_iteration": "all-gather of y (%d doubles) inside the product, all-gather of p (%d doubles), two 1-double "
                                                 "all-reduces" % (n_global, ncol),
                       "self_check": {"relative_residual": rel, "ranks_hold_identical_x": same, "ok": rel <= 2e-8 and same},
                       "rccl_status": "first contact: no multi-GPU machine was available to the builder"},
            "roofline": _roofline(_klabel(prov.kernel_name(A), prov.kernel_name(At)), bytes_it / world * iters / dt / 1e9, None,
                                  bytes_it / world / 2.0, dt * 1e3 / iters / 2.0, 2 * iters)}


def run_also(args, prov, world, rank, nccl, recs=None, state=None):
    """the other BASELINE configs behind the same command: a list of sub-records, each a full record of its
    workload.  A workload that fails leaves {"workload": ..., "error": ...} -- it must not take the headline down.  `recs` fills
    as the workloads finish and `state["current"]` names the one running (the budget watchdog of main() prints what is there)."""
    import copy
    import torch
    import torch.distributed as dist
    recs = [] if recs is None else recs
    state = {} if state is None else state
    sub = copy.copy(args)
    sub.cpu_sample_rows = args.cpu_sample_rows or 0
    plan = [("c3", run_c3, {}), ("c4", run_c4, {}), ("c5", run_c5, {}), ("cg", run_cg, {})] if not _multi(world) else \
           [("c2-strong", run_c2, {"strong": True}), ("c5", run_c5, {}), ("cg", run_cg_dist, {})]
    for name, fn, kw in plan:
        prov.release()
        state["current"] = name
        t0 = time.perf_counter()
        a = copy.copy(sub)
        if not _multi(world):       # bounded samples for the CPU baselines of the sub-records: the default run stays within minutes
            a.cpu_sample_rows = args.cpu_sample_rows or {"c3": 1_000_000, "c4": 200_000, "c5": 500_000}.get(name, 0)
        else:
            a.transpose = True
        try:
            rec = fn(a, prov, world, rank, nccl, **kw)
            err = None
        except BaseException as ex:   # incl. SystemExit of a workload's own argument checks
            rec, err = None, repr(ex)
        if _multi(world):        # a rank that failed must not leave the others waiting in the next workload's collectives
            bad = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=prov.dev if nccl else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if float(bad.item()) and not err:
                rec, err = None, "another rank failed in this workload"
        if rank == 0:
            if rec is None:
                rec = {"workload": name, "error": err}
            rec["wall_s_incl_build"] = time.perf_counter() - t0
            recs.append(rec)
        if err and _multi(world):
            break            # collectives may be out of step after a failure: stop here
    state["current"] = None
    prov.release()
    return recs


def native_path_records(world, args):
    """The ONE-PROCESS path of the C-ABI (fs_dist_*: one host thread, N devices, ncclCommInitAll -- what FASTSPARSE_NGPU gives an
    unmodified C caller, the north_star's "host C dispatching through a thin C-ABI ... y gathered via RCCL") on the same GPUs, as a
    CHILD of rank 0 (libfastsparse_amd/native_dist_bench.py) after this process' own workloads, three times: the exchange inside the
    product (FS_DIST_PARTS=4), the conservative one (FS_DIST_PARTS=1), and the first again with one ISSUING THREAD per rank
    (FS_DIST_THREADS=1, experimental: the remedy for this path's ~50 us of serial issue work per rank and product).  A crash, a hang (time-out) or a launcher that shows
    every rank one device only costs this sub-record, never the line."""
    out = {}
    for tag, parts, threads in (("overlapped_4_parts", "4", "0"), ("conservative_1_part", "1", "0"), ("overlapped_4_parts_issue_threads", "4", "1")):
        env = dict(os.environ, FS_DIST_PARTS=parts, FS_DIST_THREADS=threads)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "FS_BENCH_SPAWNED"):
            env.pop(k, None)
        if world == 1:
            env["FS_DIST_FORCE_RCCL"] = "1"       # (the one-rank rehearsal: RCCL on one device)
        cmd = [sys.executable, "-m", "libfastsparse_amd.native_dist_bench", "--ndev", str(world), "--rows", str(args.rows), "--cols", str(args.rows),
               "--per-row", str(args.per_row), "--steps", str(args.steps), "--warmup", str(args.warmup)]
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=float(os.environ.get("FS_BENCH_NATIVE_TIMEOUT_S", "150")))
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            out[tag] = json.loads(lines[-1]) if lines else {"error": "no record (exit %d): %s" % (p.returncode, (p.stderr or p.stdout)[-600:])}
            if p.returncode != 0 and "error" not in out[tag]:
                out[tag]["exit_code"] = p.returncode
        except subprocess.TimeoutExpired:
            out[tag] = {"error": "timed out (a hung collective?)", "after_s": time.perf_counter() - t0}
            break                                  # do not start the second run behind a hang
        except Exception as ex:
            out[tag] = {"error": repr(ex)}
    a, b = out.get("overlapped_4_parts", {}), out.get("conservative_1_part", {})
    if a.get("self_check") and b.get("self_check"):
        ca, cb = a["self_check"]["y_checksum"], b["self_check"]["y_checksum"]
        out["both_exchanges_agree"] = bool(abs(ca - cb) <= 1e-9 * max(abs(ca), 1.0))
    out["what"] = ("fs_dist_spmv_resident + fs_dist_spmv_t_resident on config-2 shards made on the devices; one host thread issues for every "
                   "rank (profiles/r05_dist_host_overhead.txt); the number to compare with is this line's `value`")
    return out


_STDOUT_FD = None       # N > 1: the real stdout, kept aside while fd 1 points at stderr (see quiet_stdout)


def quiet_stdout():
    """N > 1: RCCL prints a version banner on STDOUT when a communicator is created (5 lines in front of the JSON line in round
    3's rehearsals), and anything else a library prints there would land next to the one line the driver reads.  From here on
    fd 1 is stderr; emit() writes the JSON line to the real stdout."""
    global _STDOUT_FD
    if _STDOUT_FD is None:
        sys.stdout.flush()
        _STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(rec):
    line = (json.dumps(rec) + "\n").encode()
    if _STDOUT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_STDOUT_FD, line)


def budget_watchdog(seconds, rank, headline, recs, state):
    """The headline must survive whatever the sub-workloads do: a rank that dies inside a collective of config 5 would leave the
    others waiting for RCCL's own timeout, longer than the driver waits.  After `seconds` (FS_BENCH_BUDGET_S, default 480) every
    rank ends itself; rank 0 first prints the line with the sub-records finished so far and the name of the one that was not."""
    import threading

    def fire():
        if rank == 0 and headline is not None:
            rec = dict(headline)
            rec["also"] = list(recs) + [{"workload": state.get("current"), "error": "not finished within the %d s budget" % seconds}]
            emit(rec)
        os._exit(0 if headline is not None or rank != 0 else 1)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "c2", "c3", "c4", "c5"],
                    help="all (default): the config-2 line with the other BASELINE configs under its key \"also\"")
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU (config 2 / 3 / 4: 10M)")
    ap.add_argument("--per-row", type=int, default=16)
    ap.add_argument("--c5-rows", type=int, default=100_000_000, help="rows = columns of the config-5 matrix")
    ap.add_argument("--transpose", action="store_true", help="c5, N > 1: also time z = A' u")
    ap.add_argument("--strong", action="store_true",
                    help="c2, N > 1: strong scaling -- the ONE config-2 matrix (--rows rows in all) cut over the ranks (default: weak "
                         "scaling, --rows rows per rank)")
    ap.add_argument("--parts", type=int, default=4, help="N > 1: parts of a local product; the all-gather of part p runs under part p+1")
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "direct"],
                    help="N > 1: how a part's rows travel: one padded all-gather per part (default) or direct point-to-point sends to "
                         "every peer (no padding, no unpack) -- to be A/B'ed on a machine with more than one GPU")
    ap.add_argument("--z-scheme", default="gather", choices=["gather", "reduce"],
                    help="c2, N > 1: z = A'u by row shards of A' + all-gather (default) or local A_r'u_r + all-reduce")
    ap.add_argument("--strict-exchange", action="store_true",
                    help="N > 1: when the exchange inside the product disagrees with the plain whole-shard all-gather (verify_overlap, before "
                         "the timed loops) print ONE JSON line with \"error\" and the failing part / rank / row range and exit 3; default: "
                         "say so in the line (exchange_fault), switch every rank to the conservative exchange and measure that")
    ap.add_argument("--lean", action="store_true",
                    help="profiling runs (tools/profile.sh): the timed products and the self-check only -- no per-launch events, no "
                         "host-pointer call, no probes, so that the per-kernel means of a PMC pass are the full products'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="config 2, N = 1: skip the two rocprofv3 --pmc children that measure roofline.traffic (then the value of profiles/ is reported)")
    ap.add_argument("--no-reproducible-cost", action="store_true", help="c2: skip timing the products with fixed-order sums")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the CPU baselines' samples (0: the workload's default)")
    ap.add_argument("--spmm-kernel", type=int, default=0,
                    help="c4: 0 the product's choice, 1 row kernel, 4 the matrix-core experiment (profiling runs)")
    args = ap.parse_args()
    t_start = time.perf_counter()

    # ---- ranks: a launcher's world must be the one asked for; without a launcher, start the ranks here, before this
    # process does anything with the GPU (a process that has initialised HIP must not be replaced or forked) -------
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args.gpus))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (torch.distributed.run "
                             "--nproc-per-node %d) or drop the launcher and let bench.py start them" % (args.gpus, world, args.gpus))

    # the host driver of this pool only supports dmabuf IPC: without this RCCL's peer mappings fail with
    # "hipIpcGetMemHandle: invalid argument" (set before anything loads the HIP runtime; a launcher's own value wins)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("FS_BENCH_WATCHDOG"):     # seconds: every rank then dumps its Python stacks to stderr (a hung collective)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FS_BENCH_WATCHDOG"]), repeat=False)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # FS_BENCH_BACKEND=gloo rehearses the N > 1 code path with several ranks sharing one GPU (no RCCL there)
    backend = os.environ.get("FS_BENCH_BACKEND", "nccl")
    nccl = backend == "nccl"
    ndev = max(torch.cuda.device_count(), 1)
    if nccl and world > ndev:
        raise SystemExit("--gpus %d but only %d GPU(s) visible" % (world, ndev))
    dev_index = local_rank if nccl else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if _multi(world):
        quiet_stdout()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "WORLD_SIZE" not in os.environ:      # FS_BENCH_FORCE_MULTI without a launcher: a one-rank group of its own
            os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if nccl:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    prov = HipProvider(dev)
    if _multi(world) and not os.environ.get("FS_BENCH_WATCHDOG"):
        # a collective that never returns (first contact with RCCL on more than one GPU) must not end in silence: after the
        # budget every rank dumps its Python stacks to stderr and exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("FS_BENCH_BUDGET_S", "480")), repeat=False, exit=True)
    if args.workload in ("all", "c2"):
        try:
            rec = run_c2(args, prov, world, rank, nccl, strong=args.strong)
        except Exception as ex:
            if not _multi(world):
                raise
            # first contact with RCCL on more than one GPU must be cheap to diagnose: ONE line, from whichever
            # rank met the error (rank 0 may be waiting in a collective: the launcher ends it), a normal exit path, no retry
            if rank == 0 or not isinstance(ex, ExchangeFault):      # (an ExchangeFault is raised on every rank: rank 0 speaks for all)
                emit(error_record(ex, args, world, rank))
            sys.stdout.flush()
            sys.stderr.write("bench.py: rank %d: %r\n" % (rank, ex))
            os._exit(3)
        if _multi(world) and not os.environ.get("FS_BENCH_WATCHDOG"):
            import faulthandler
            faulthandler.cancel_dump_traceback_later()      # the headline exists: from here on the budget watchdog prints it
        if args.workload == "all":
            also, state = [], {}
            budget = float(os.environ.get("FS_BENCH_BUDGET_S", "480")) - (time.perf_counter() - t_start)
            dog = budget_watchdog(max(budget, 5.0), rank, rec, also, state)
            run_also(args, prov, world, rank, nccl, recs=also, state=state)
            if _multi(world) and nccl and not os.environ.get("FS_BENCH_NO_NATIVE"):
                # the same GPUs once more, through the one-process C path (a child of rank 0; the other ranks wait at the barrier)
                state["current"] = "native one-process path"
                prov.release()
                prov.synchronize()
                dist.barrier()
                # the other ranks wait on the HOST (the rendezvous store), not in a collective: a barrier kernel spinning on their GPUs
                # would sit on CUs and links the child is being timed on
                native, store = None, None
                try:
                    store = dist.distributed_c10d._get_default_store()
                except Exception:
                    store = None
                if rank == 0:
                    native = native_path_records(world, args)
                    if store is not None:
                        store.set("fs_native_path_done", "1")
                elif store is not None:
                    import datetime
                    try:
                        store.wait(["fs_native_path_done"], datetime.timedelta(seconds=900))
                    except Exception:
                        pass
                dist.barrier()
                if rank == 0:
                    also.append({"workload": "native one-process path (fs_dist_*)", "native_one_process_path": native})
                state["current"] = None
            dog.cancel()
            if rank == 0 and rec is not None:
                rec["also"] = also
    else:
        if args.workload != "c5" and world != 1:
            raise SystemExit("--workload %s is a one-GPU workload" % args.workload)
        rec = {"c3": run_c3, "c4": run_c4, "c5": run_c5}[args.workload](args, prov, world, rank, nccl)
    if rank == 0 and rec is not None:
        emit(rec)
    if _multi(world):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
