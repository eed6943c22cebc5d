"""The ONE-PROCESS multi-GPU path of the C-ABI (fs_dist_*: what FASTSPARSE_NGPU gives an unmodified C caller) timed on the GPUs this
process can see -- a child of `bench.py --gpus N` (rank 0 starts it after its own workloads, the other ranks wait), so that the
driver's scaling run also measures the native RCCL path; a crash or hang here costs one sub-record, not the line.

    python -m libfastsparse_amd.native_dist_bench --ndev N [--rows R] [--per-row P] [--steps K] [--warmup W]

Workload = bench.py's config-2 weak scaling: every device owns R rows x 10 M columns, 16 per row, generated on ITS device
(fs_synth_uniform, the generator of bench.py) and handed over as per-rank shards (fs_dist_csr_create_from_shards, FS_DEVICE);
x replicated; step = y = A x (local product in FS_DIST_PARTS parts, ncclAllGather of each part's rows inside the product) then
z = A' y on row shards of A' built on the devices (fs_dist_matrix_build_transpose_device) -- everything resident, one host thread.
Prints ONE JSON line.  ctypes only: no torch in this process."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED_C2 = 0x5EED0002
HBM_PEAK_GBS = 8000.0


def csr_bytes(nnz, nrow, ncol, valued=True):
    return (12 if valued else 4) * nnz + 4 * (nrow + 1) + 8 * nrow + 8 * ncol


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ndev", type=int, required=True)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=10_000_000)
    ap.add_argument("--per-row", type=int, default=16)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--virtual", action="store_true", help="all ranks on device 0 (rehearsal on a one-GPU box: no RCCL)")
    args = ap.parse_args()
    import numpy as np
    from libfastsparse_amd import capi
    L = capi.lib()
    vp = C.c_void_p
    visible = L.fs_device_count()
    n, ncol, per, N = args.rows, args.cols, args.per_row, args.ndev
    rec = {"workload": "native one-process path (fs_dist_*, what FASTSPARSE_NGPU uses): config-2 shards, weak scaling, %d rows per device x %d "
                       "columns x %d per row; step = y = A x (all-gather inside the product) then z = A' y (row shards of A')" % (n, ncol, per),
           "n_devices": N, "devices_visible_to_this_process": visible, "parts": int(os.environ.get("FS_DIST_PARTS", "4") or 4)}
    if not args.virtual and visible < N:
        rec["error"] = "only %d device(s) visible to this process (the launcher restricts visibility per rank): not run" % visible
        print(json.dumps(rec))
        return 0
    devs = [0] * N if args.virtual else list(range(N))
    nnz = n * per
    t_all = time.perf_counter()
    # ---- every device generates its shard ----------------------------------------------------------------------
    rps, ccs, vvs = [], [], []
    for r in range(N):
        capi.check(L.fs_set_device(devs[r]), "fs_set_device")
        rp, cc, vv = L.fs_device_alloc(4 * (n + 1)), L.fs_device_alloc(4 * nnz), L.fs_device_alloc(8 * nnz)
        if not (rp and cc and vv):
            raise SystemExit("out of device memory for a shard")
        capi.check(L.fs_synth_uniform(n, ncol, per, SEED_C2, r * n, rp, cc, vv, None), "fs_synth_uniform")
        capi.check(L.fs_device_synchronize(), "synchronize")
        rps.append(rp); ccs.append(cc); vvs.append(vv)
    capi.check(L.fs_set_device(devs[0]), "fs_set_device")
    t0 = time.perf_counter()
    D = L.fs_dist_create(N, (C.c_int * N)(*devs))
    if not D:
        raise SystemExit("fs_dist_create: " + L.fs_last_error().decode())
    rec["uses_rccl"] = bool(L.fs_dist_uses_rccl(D))
    L.fs_debug_dist_issue_threads.argtypes = [vp]
    rec["issue_threads"] = int(L.fs_debug_dist_issue_threads(D))      # FS_DIST_THREADS=1: one issuing thread per rank (0: the caller issues)
    rec["context_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    M = L.fs_dist_csr_create_from_shards(D, N * n, ncol, (C.c_int * N)(*([n] * N)), (C.c_int64 * N)(*([nnz] * N)),
                                         (vp * N)(*rps), (vp * N)(*ccs), (vp * N)(*vvs), capi.FS_DEVICE)
    if not M:
        raise SystemExit("fs_dist_csr_create_from_shards: " + L.fs_last_error().decode())
    for p in rps + ccs + vvs:
        L.fs_device_free(p)
    rec["build_A_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    capi.check(L.fs_dist_matrix_build_transpose_device(M), "fs_dist_matrix_build_transpose_device")
    rec["build_At_on_the_devices_s"] = time.perf_counter() - t0
    bt = (C.c_int * (N + 1))()
    capi.check(L.fs_dist_matrix_bounds_t(M, bt), "bounds_t")
    x = np.sin(7.0 * np.arange(ncol, dtype=np.float64) + 0.3)
    for r in range(N):
        capi.check(L.fs_copy_to_device(L.fs_dist_x(M, r), x.ctypes.data, 8 * ncol), "x to device")

    def step():
        capi.check(L.fs_dist_spmv_resident(M), "fs_dist_spmv_resident")
        capi.check(L.fs_dist_spmv_t_resident(M), "fs_dist_spmv_t_resident")

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = (time.perf_counter() - t0) / args.steps
    ta = time.perf_counter()
    for _ in range(args.steps):
        capi.check(L.fs_dist_spmv_resident(M), "fs_dist_spmv_resident")
    dta = (time.perf_counter() - ta) / args.steps
    bytes_a = N * csr_bytes(nnz, n, ncol)
    bytes_t = sum(csr_bytes(int(L.fs_matrix_nnz(L.fs_dist_matrix_shard(M, r, 1))), bt[r + 1] - bt[r], N * n) for r in range(N))
    rec.update({"value": (bytes_a + bytes_t) / dt / 1e9, "unit": "GB/s", "ms_per_step": dt * 1e3, "A_mul_B_ms_with_its_exchange": dta * 1e3,
                "aggregate_frac_of_hbm_peak": (bytes_a + bytes_t) / dt / 1e9 / (HBM_PEAK_GBS * N), "steps": args.steps,
                "conservative_exchange": bool(L.fs_dist_is_conservative(D)),
                "kernel_of_shard_0": {1: "stream", 6: "tiled", 7: "two-pass", 8: "lds-staged"}.get(L.fs_matrix_spmv_kernel(L.fs_dist_matrix_shard(M, 0, 0), 0))})
    # self-check: every device holds the same y and z (bytes), and a window of y against the same rows recomputed on ONE device
    y0 = np.empty(N * n)
    capi.check(L.fs_copy_to_host(y0.ctypes.data, L.fs_dist_y(M, 0), 8 * N * n), "y from device 0")
    same = True
    for r in range(1, N):
        yr = np.empty(N * n)
        capi.check(L.fs_copy_to_host(yr.ctypes.data, L.fs_dist_y(M, r), 8 * N * n), "y from a device")
        same = same and bool(np.array_equal(y0, yr))
    rec["self_check"] = {"devices_hold_identical_y": same, "y_finite": bool(np.isfinite(y0).all()),
                         "y_checksum": float(np.abs(y0).sum())}
    # the last shard's rows once more as an ordinary single-GPU matrix on its own device: same generator, same kernels
    capi.check(L.fs_set_device(devs[N - 1]), "fs_set_device")
    rp, cc, vv = L.fs_device_alloc(4 * (n + 1)), L.fs_device_alloc(4 * nnz), L.fs_device_alloc(8 * nnz)
    capi.check(L.fs_synth_uniform(n, ncol, per, SEED_C2, (N - 1) * n, rp, cc, vv, None), "fs_synth_uniform")
    A1 = L.fs_csr_create(n, ncol, nnz, rp, cc, vv, capi.FS_DEVICE, 1)
    xd, yd = L.fs_device_alloc(8 * ncol), L.fs_device_alloc(8 * n)
    capi.check(L.fs_copy_to_device(xd, x.ctypes.data, 8 * ncol), "x")
    capi.check(L.fs_spmv(A1, yd, xd, None), "fs_spmv")
    capi.check(L.fs_device_synchronize(), "synchronize")
    y1 = np.empty(n)
    capi.check(L.fs_copy_to_host(y1.ctypes.data, yd, 8 * n), "y")
    diff = float(np.max(np.abs(y1 - y0[(N - 1) * n:])))
    rec["self_check"]["last_shard_vs_single_gpu_product_max_abs_diff"] = diff
    rec["self_check"]["ok"] = bool(same and rec["self_check"]["y_finite"] and diff <= 1e-11)
    L.fs_matrix_destroy(A1)
    for p in (rp, cc, vv, xd, yd):
        L.fs_device_free(p)
    L.fs_dist_matrix_destroy(M)
    L.fs_dist_destroy(D)
    rec["wall_s"] = time.perf_counter() - t_all
    print(json.dumps(rec))
    return 0


if __name__ == "__main__":
    sys.exit(main())
