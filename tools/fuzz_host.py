#!/usr/bin/env python3
"""Randomised run of the HOST half (fs_host.c: the constructors an unmodified caller runs before any product) against the REAL
reference compiled from /root/reference (oracle/_ref/libfsref.so -- this container only; no GPU needed): random COO triples in
random entry order with duplicates and empty rows through new_csr / new_bcsr / new_cbcsr / new_bsbm / new_bsdm, every array of
the resulting structs byte for byte against the reference's.     python tools/fuzz_host.py [seconds] [seed] [device_build: 0 host loops, 2 on the device]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _hipbackend as H  # noqa: E402
import _refbind  # noqa: E402
from libfastsparse_amd import capi  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
if not _refbind.available():
    raise SystemExit("oracle/_ref/libfsref.so is missing: make -C oracle ref (needs /root/reference)")
# the host loops of fs_host.c by default; argv[3] = 2: the same constructors building ON THE DEVICE (fs_bucket_coo: needs a GPU)
how = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if how:
    import torch  # noqa: F401  (before the library: the HIP runtime the library binds to)
capi.set_option("device_build", how)
R, F = _refbind.Ref(), H.HostFormats()
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    nrow = int(rng.choice([1, 2, 17, 900, 20_000]))
    ncol = int(rng.choice([1, 3, 50, 4_000, 70_000]))
    nnz = int(rng.choice([0, 1, 5, 300, 20_000, 200_000]))
    rows = rng.integers(0, nrow, nnz).astype(np.int32)
    if rng.integers(0, 2):
        rows[rng.integers(0, 2, nnz).astype(bool)] = rng.integers(0, nrow)        # half of the entries in one row
    cols = rng.integers(0, min(ncol, int(rng.choice([ncol, 4]))), nnz).astype(np.int32)
    vals = rng.uniform(-1, 1, nnz)
    what = dict(nrow=nrow, ncol=ncol, nnz=nnz, seed=seed, case=cases)

    def same(a, b, n, dt, name):
        assert np.array_equal(F.arr(a, n, dt), R.arr(b, n, dt)), (what, name)

    a, b = F.csr(nrow, ncol, rows, cols, vals), R.csr(nrow, ncol, rows, cols, vals)
    assert (a.nrow, a.ncol, a.nnz) == (b.nrow, b.ncol, b.nnz), what
    same(a.row_ptr, b.row_ptr, nrow + 1, np.int32, "csr.row_ptr"); same(a.cols, b.cols, nnz, np.int32, "csr.cols"); same(a.vals, b.vals, nnz, np.float64, "csr.vals")
    a, b = F.bcsr(nrow, ncol, rows, cols), R.bcsr(nrow, ncol, rows, cols)
    same(a.row_ptr, b.row_ptr, nrow + 1, np.int32, "bcsr.row_ptr"); same(a.cols, b.cols, nnz, np.int32, "bcsr.cols")
    cbs = int(rng.choice([1, 7, 64, 100_000]))
    while -(-ncol // cbs) * (nrow + 1) > 5_000_000:
        cbs *= 4
    a, b = F.cbcsr(cbs, nrow, ncol, rows, cols), R.cbcsr(cbs, nrow, ncol, rows, cols)
    assert (a.nblocks, a.colblocksize, a.nnz) == (b.nblocks, b.colblocksize, b.nnz), what
    same(a.row_ptr, b.row_ptr, a.nblocks * nrow + 1, np.int32, "cbcsr.row_ptr"); same(a.cols, b.cols, nnz, np.int32, "cbcsr.cols")
    bs = int(rng.choice([1, 8, 48, 1024]))
    if nrow // bs <= 5_000:
        sa, sb = F.sbm(nrow, ncol, rows, cols), R.sbm(nrow, ncol, rows, cols)
        pa, pb = F.L.new_bsbm(C.byref(sa), bs).contents, R.lib.new_bsbm(C.byref(sb), bs).contents
        assert (pa.nrow, pa.ncol, pa.nblocks) == (pb.nrow, pb.ncol, pb.nblocks), what
        same(pa.start_row, pb.start_row, pa.nblocks + 1, np.int32, "bsbm.start_row"); same(pa.nnz, pb.nnz, pa.nblocks, np.int32, "bsbm.nnz")
        for k in range(pa.nblocks):
            same(pa.rows[k], pb.rows[k], pa.nnz[k], np.int32, "bsbm.rows"); same(pa.cols[k], pb.cols[k], pa.nnz[k], np.int32, "bsbm.cols")
        da, db = F.sdm(nrow, ncol, rows, cols, vals), R.sdm(nrow, ncol, rows, cols, vals)
        qa, qb = F.L.new_bsdm(C.byref(da), bs).contents, R.lib.new_bsdm(C.byref(db), bs).contents
        assert qa.nblocks == qb.nblocks, what
        same(qa.start_row, qb.start_row, qa.nblocks + 1, np.int32, "bsdm.start_row"); same(qa.nnz, qb.nnz, qa.nblocks, np.int32, "bsdm.nnz")
        for k in range(qa.nblocks):
            same(qa.rows[k], qb.rows[k], qa.nnz[k], np.int32, "bsdm.rows"); same(qa.cols[k], qb.cols[k], qa.nnz[k], np.int32, "bsdm.cols")
            same(qa.vals[k], qb.vals[k], qa.nnz[k], np.float64, "bsdm.vals")
    F._keep.clear(); R._keep.clear()
    cases += 1
    if cases % 100 == 0:
        print("%d cases ok" % cases, flush=True)
print("fuzz_host: %d cases, every array of every struct byte for byte the real reference's (seed %d, device_build %d)" % (cases, seed, how))
