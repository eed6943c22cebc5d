#!/usr/bin/env python3
"""Randomised differential run of the one-process multi-GPU layer (fs_dist_*) on VIRTUAL ranks of one GPU: random matrices, rank
counts, constructors (host CSR / host COO / per-rank shards in host or device memory / a pair of the caller's A and At), vectors
in host memory or in HBM, A x / A' u / k columns / A'A x, forced kernel families and options, every result against the oracle
(pattern-only with integer x: bit for bit; otherwise the row-scaled 1e-12 bound of tools/_fuzz_common.py) and -- every rank's
resident copy of the result -- against each other byte for byte.  FS_DIST_PARTS and FS_DIST_THREADS are read once per process:
run it once per setting.     python tools/fuzz_dist.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
from _fuzz_common import check, make  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4321
rng = np.random.default_rng(seed)
L = capi.lib()
vp = C.c_void_p
L.fs_debug_dist_k_parts.argtypes = [vp, C.c_int]
L.fs_debug_dist_parts.argtypes = [vp, C.c_int]
# option sets: small panels (bin_rows / tile_rows) are what lets a small matrix be cut into parts at all
FORCE = [{}, {}, {"binning": 2}, {"binning": 2, "bin_flags": 128}, {"binning": 2, "bin_flags": 128, "bin_rows": 256, "reproducible": 1}, {"binning": 2, "bin_flags": 64},
         {"binning": 2, "bin_rows": 64}, {"binning": 2, "bin_rows": 64, "reproducible": 1},
         {"binning": 2, "bin_rows": 256, "spmm_kernel": 2}, {"binning": 2, "long_rows": 2, "bin_rows": 128}, {"ldsx": 2}, {"ldsx": 2, "tile_rows": 64},
         {"tiling": 2, "tile_rows": 64}, {"tiling": 2, "tile_rows": 64, "reproducible": 1}, {"spmv_kernel": 1}, {"reproducible": 1},
         {"strict_order": 1}, {"release_csr": 1}]
RESET = {"binning": 1, "ldsx": 1, "tiling": 1, "long_rows": 1}          # every other option: 0
PRESET = {k: 1 for k, e in (("reproducible", "FS_REPRODUCIBLE"), ("strict_order", "FS_STRICT_ORDER")) if os.environ.get(e) == "1"}   # env presets stay


def ok(rc, what):
    assert rc == 0, (what, L.fs_last_error())


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def download(ptr, n):
    out = np.empty(n)
    ok(L.fs_copy_to_host(out.ctypes.data, ptr, 8 * n), "download")
    return out


def create(D, ranks, how, nrow, ncol, rp, cc, vals):
    nnz = len(cc)
    rows = np.repeat(np.arange(nrow, dtype=np.int32), np.diff(rp))
    vptr = None if vals is None else vals.ctypes.data
    if how == "csr":
        return L.fs_dist_csr_create(D, nrow, ncol, nnz, rp.ctypes.data, cc.ctypes.data, vptr), None
    if how == "coo":                                   # rows in a random order, every row's entries still in their own order
        order = np.argsort(rng.permutation(nrow)[rows], kind="stable")
        r2, c2 = np.ascontiguousarray(rows[order]), np.ascontiguousarray(cc[order])
        v2 = None if vals is None else np.ascontiguousarray(vals[order])
        return L.fs_dist_coo_create(D, nrow, ncol, nnz, r2.ctypes.data, c2.ctypes.data, None if v2 is None else v2.ctypes.data), None
    # per-rank shards with cuts of the caller's choosing (empty shards included)
    cuts = np.sort(rng.integers(0, nrow + 1, ranks - 1)) if ranks > 1 else np.zeros(0, np.int64)
    cuts = [0] + [int(c) for c in cuts] + [nrow]
    keep = []
    srows = (C.c_int * ranks)(*[cuts[r + 1] - cuts[r] for r in range(ranks)])
    snnz = (C.c_int64 * ranks)(*[int(rp[cuts[r + 1]] - rp[cuts[r]]) for r in range(ranks)])
    p_rp, p_cc, p_vv = (vp * ranks)(), (vp * ranks)(), (vp * ranks)()
    for r in range(ranks):
        a, b = int(rp[cuts[r]]), int(rp[cuts[r + 1]])
        lrp = (rp[cuts[r]:cuts[r + 1] + 1] - rp[cuts[r]]).astype(np.int32)
        lcc = np.ascontiguousarray(cc[a:b])
        lvv = None if vals is None else np.ascontiguousarray(vals[a:b])
        if how == "shards_device":
            lrp, lcc, lvv = dev(lrp), dev(lcc) if b > a else torch.empty(0, dtype=torch.int32, device="cuda"), \
                (None if lvv is None else (dev(lvv) if b > a else torch.empty(0, dtype=torch.float64, device="cuda")))
            p_rp[r], p_cc[r] = lrp.data_ptr(), lcc.data_ptr()
            p_vv[r] = None if lvv is None else lvv.data_ptr()
        else:
            p_rp[r], p_cc[r] = lrp.ctypes.data, lcc.ctypes.data
            p_vv[r] = None if lvv is None else lvv.ctypes.data
        keep += [lrp, lcc, lvv]
    torch.cuda.synchronize()
    M = L.fs_dist_csr_create_from_shards(D, nrow, ncol, srows, snnz, p_rp, p_cc, None if vals is None else p_vv,
                                         capi.FS_DEVICE if how == "shards_device" else capi.FS_HOST)
    return M, cuts


t_end = time.time() + budget
cases = 0
parts_seen = 0
cut_seen = 0
while time.time() < t_end:
    nrow, ncol, rp, cc, vv = make(rng, 3_000_000)
    valued = bool(rng.integers(0, 2))
    vals = vv if valued else None
    ranks = int(rng.choice([1, 2, 3, 5, 8]))
    how = str(rng.choice(["csr", "coo", "shards_host", "shards_device"]))
    force = FORCE[rng.integers(0, len(FORCE))]
    integer = bool(rng.integers(0, 2))
    in_hbm = bool(rng.integers(0, 2))
    k = int(rng.choice([1, 2, 3, 4, 8]))
    x = rng.integers(-1000, 1001, ncol).astype(np.float64) if integer else np.sin(7.0 * np.arange(ncol) + 0.3)
    u = rng.integers(-1000, 1001, nrow).astype(np.float64) if integer else np.sin(11.0 * np.arange(nrow) - 0.2)
    what = dict(nrow=nrow, ncol=ncol, nnz=len(cc), valued=valued, ranks=ranks, how=how, force=force, integer=integer, hbm=in_hbm, k=k,
                seed=seed, case=cases)
    exact = (not valued and integer) or "strict_order" in force or os.environ.get("FS_STRICT_ORDER") == "1"
    lens = np.diff(rp).astype(np.float64)
    clen = np.bincount(cc, minlength=ncol).astype(np.float64)
    rows = np.repeat(np.arange(nrow, dtype=np.int32), np.diff(rp))
    for name, value in force.items():
        capi.set_option(name, value)
    D = M = None
    try:
        # ranks == 1 with the default device list would be a one-rank RCCL group (tests/test_gpu_dist_rehearsal.py covers it)
        D = L.fs_dist_create(ranks, (C.c_int * ranks)(*([0] * ranks)))
        assert D, L.fs_last_error()
        M, cuts = create(D, ranks, how, nrow, ncol, rp, cc, vals)
        assert M, (what, L.fs_last_error())
        assert L.fs_dist_matrix_nnz(M) == len(cc), what
        b = (C.c_int * (ranks + 1))()
        ok(L.fs_dist_matrix_bounds(M, b), "bounds")
        assert b[0] == 0 and b[ranks] == nrow and all(b[i] <= b[i + 1] for i in range(ranks)), (what, list(b))
        if cuts is not None:
            assert list(b) == cuts, (what, list(b), cuts)

        def vec_in(a):
            if in_hbm:
                t = dev(a)
                torch.cuda.synchronize()
                return t, t.data_ptr()
            return a, a.ctypes.data

        def vec_out(shape):
            if in_hbm:
                t = torch.full(shape, -1.0, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                return t, t.data_ptr(), (lambda: t.cpu().numpy())
            a = np.full(shape, -1.0)
            return a, a.ctypes.data, (lambda: a)

        ref = O.csr_mul(nrow, rp, cc, vals, x)
        scale = O.csr_abs_scale(nrow, rp, cc, vals, x)
        xin, xp = vec_in(x)
        yo, yp, yget = vec_out((max(nrow, 1),))
        ok(L.fs_dist_spmv(M, yp, xp), ("fs_dist_spmv", what))
        y = yget()[:nrow]
        check(y, ref, scale, exact, ("spmv", what), lens)
        cut_seen += int(L.fs_debug_dist_parts(M, 0) > 1)
        for r in range(1 if in_hbm else 0, ranks):           # every rank's own copy of y: the same bytes (a y in HBM IS rank 0's)
            assert np.array_equal(download(L.fs_dist_y(M, r), nrow), y), (what, "y of rank", r)
        # the transposed side: built on the devices, or from the host arrays
        if how.startswith("shards") or rng.integers(0, 2):
            ok(L.fs_dist_matrix_build_transpose_device(M), ("build_transpose_device", what))
        else:
            ok(L.fs_dist_matrix_build_transpose(M, rp.ctypes.data, cc.ctypes.data, None if vals is None else vals.ctypes.data), "build_transpose")
        zref = O.coo_tmul(ncol, rows, cc, vals, u)
        zsc = O.coo_tmul(ncol, rows, cc, None if vals is None else np.abs(vals), np.abs(u))
        uin, up = vec_in(u if nrow else np.zeros(1))
        zo, zp, zget = vec_out((ncol,))
        ok(L.fs_dist_spmv_t(M, zp, up), ("fs_dist_spmv_t", what))
        check(zget(), zref, zsc, exact, ("spmv_t", what), clen)
        # A'A x (+ lambda x)
        lam = float(rng.choice([0.0, 0.5]))
        zo2, zp2, zget2 = vec_out((ncol,))
        ok(L.fs_dist_ata(M, zp2, xp, lam), ("fs_dist_ata", what))
        aref = O.coo_tmul(ncol, rows, cc, vals, ref) + lam * x
        asc = O.coo_tmul(ncol, rows, cc, None if vals is None else np.abs(vals), scale) + abs(lam) * np.abs(x)
        got = zget2()
        if exact and lam == 0.0:
            assert np.array_equal(got, aref), ("ata", what)
        else:                                                 # two chained sums: the first one's error is amplified by |A'|
            tol = np.maximum(1e-12, 2.0 * (clen + lens.max(initial=0.0)) * 2.0 ** -53) * 2.0
            assert np.all(np.abs(got - aref) <= tol * asc), ("ata", what, float(np.max(np.abs(got - aref) - tol * asc)))
        # k columns, host matrices (the API's contract for k > 1)
        if k > 1:
            X = rng.integers(-50, 51, (ncol, k)).astype(np.float64) if integer else np.sin(np.arange(ncol * k, dtype=np.float64)).reshape(ncol, k)
            Y = np.full((max(nrow, 1), k), -1.0)
            ok(L.fs_dist_spmm(M, Y.ctypes.data, X.ctypes.data, k), ("fs_dist_spmm", what))
            Yr = O.csr_mul_n(nrow, rp, cc, vals, X, k)
            Ys = O.csr_mul_n(nrow, rp, cc, None if vals is None else np.abs(vals), np.abs(X), k)
            check(Y[:nrow], Yr, Ys, exact, ("spmm", what), lens)
            U = rng.integers(-50, 51, (max(nrow, 1), k)).astype(np.float64) if integer else np.sin(np.arange(max(nrow, 1) * k, dtype=np.float64)).reshape(-1, k)
            Z = np.full((ncol, k), -1.0)
            ok(L.fs_dist_spmm_t(M, Z.ctypes.data, U.ctypes.data, k), ("fs_dist_spmm_t", what))
            Zr = np.stack([O.coo_tmul(ncol, rows, cc, vals, np.ascontiguousarray(U[:nrow, j])) for j in range(k)], 1)
            Zs = np.stack([O.coo_tmul(ncol, rows, cc, None if vals is None else np.abs(vals), np.abs(np.ascontiguousarray(U[:nrow, j]))) for j in range(k)], 1)
            check(Z, Zr, Zs, exact, ("spmm_t", what), clen)
            parts_seen += int(L.fs_debug_dist_k_parts(M, 0) > 1)
    finally:
        if M:
            L.fs_dist_matrix_destroy(M)
        if D:
            L.fs_dist_destroy(D)
        for name in force:
            capi.set_option(name, PRESET.get(name, RESET.get(name, 0)))
    cases += 1
    if cases % 20 == 0:
        print("%d cases ok" % cases, flush=True)
print("fuzz_dist: %d cases, all within the bars (seed %d, FS_DIST_PARTS=%s, FS_DIST_THREADS=%s; %d single-vector and %d k-column products "
      "cut into parts)" % (cases, seed, os.environ.get("FS_DIST_PARTS", "4"), os.environ.get("FS_DIST_THREADS", "0"), cut_seen, parts_seen))
