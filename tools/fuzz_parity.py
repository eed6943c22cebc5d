#!/usr/bin/env python3
"""Randomised differential run: random shapes, densities, row-length profiles, forced kernel families, k and options through the
device-resident layer AND the reference-named entry points, every result against the oracle (pattern-only with integer x: bit for
bit; otherwise the row-scaled 1e-12 bound).  Not part of the test suite (open-ended); run on the GPU box:
    python tools/fuzz_parity.py [seconds] [seed]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import _hipbackend as H  # noqa: E402
from libfastsparse_amd import capi  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
from _fuzz_common import check, make  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
L = capi.lib()
st = capi.current_stream()
# option sets: small panels / tiles (bin_rows, tile_rows, tile_cols) take small matrices through the many-panel paths of the big ones
FORCE = [{}, {"binning": 2}, {"binning": 2, "bin_flags": 64}, {"binning": 2, "bin_flags": 128}, {"binning": 2, "bin_flags": 128, "reproducible": 1},
         {"binning": 2, "bin_rows": 64}, {"binning": 2, "bin_rows": 64, "reproducible": 1}, {"binning": 2, "long_rows": 2, "bin_rows": 128},
         {"binning": 2, "long_rows": 2, "long_min_len": 40, "reproducible": 1}, {"binning": 2, "bin_rows": 256, "spmm_kernel": 2}, {"ldsx": 2},
         {"ldsx": 2, "tile_rows": 64}, {"ldsx": 2, "reproducible": 1}, {"ldsx": 2, "tiled_flags": 4}, {"tiling": 2}, {"tiling": 2, "tile_rows": 64, "tile_cols": 512},
         {"tiling": 2, "tile_split": 8}, {"spmv_kernel": 1}, {"spmv_kernel": 2}, {"spmv_kernel": 3}, {"reproducible": 1}, {"strict_order": 1},
         {"spmm_kernel": 1}, {"spmm_kernel": 3}, {"spmm_wide": 1}, {"release_csr": 1}, {"release_csr": 1, "binning": 2},
         {"release_csr": 1, "ldsx": 2}, {"release_csr": 1, "tiling": 2, "tile_rows": 64}]
RESET = {"binning": 1, "ldsx": 1, "tiling": 1, "long_rows": 1}          # every other option: 0
PRESET = {k: 1 for k, e in (("reproducible", "FS_REPRODUCIBLE"), ("strict_order", "FS_STRICT_ORDER")) if os.environ.get(e) == "1"}   # env presets stay


t_end = time.time() + budget
cases = 0
released = 0
while time.time() < t_end:
    nrow, ncol, rp, cc, vv = make(rng)
    valued = bool(rng.integers(0, 2))
    vals = vv if valued else None
    force = FORCE[rng.integers(0, len(FORCE))]
    integer = bool(rng.integers(0, 2))
    x = rng.integers(-1000, 1001, ncol).astype(np.float64) if integer else np.sin(7.0 * np.arange(ncol) + 0.3)
    u = rng.integers(-1000, 1001, nrow).astype(np.float64) if integer else np.sin(11.0 * np.arange(nrow) - 0.2)
    k = int(rng.choice([1, 2, 3, 4, 5, 8, 17]))
    what = dict(nrow=nrow, ncol=ncol, nnz=int(rp[-1]), valued=valued, force=force, integer=integer, k=k, seed=seed, case=cases)
    for name, value in force.items():
        capi.set_option(name, value)
    try:
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        A = capi.Matrix.from_csr(nrow, ncol, d(rp), d(cc), None if vals is None else d(vals))
        exact = (not valued and integer) or "strict_order" in force
        ref = O.csr_mul(nrow, rp, cc, vals, x)
        scale = O.csr_abs_scale(nrow, rp, cc, vals, x)
        y = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
        A.spmv(y, d(x), st)
        lens = np.diff(rp).astype(np.float64)
        clen = np.bincount(cc, minlength=ncol).astype(np.float64)
        check(y.cpu().numpy(), ref, scale, exact, ("spmv", what), lens)
        fixed = L.fs_get_option(b"reproducible") == 1 or L.fs_get_option(b"strict_order") == 1
        if fixed:                                   # fixed-order sums: a second product gives the same bits
            y2 = torch.full((nrow,), -2.0, dtype=torch.float64, device="cuda")
            A.spmv(y2, d(x), st)
            assert torch.equal(y, y2), ("spmv twice", what)
        # transposed (a handle that gave its plain arrays back under release_csr says so, takes them back and goes on)
        try:
            A.build_transpose(st)
        except Exception as e:  # noqa: BLE001
            assert "release_csr" in force and "(-5)" in str(e), (what, str(e))      # FS_ERR_RELEASED
            A.restore_csr(d(rp), d(cc), None if vals is None else d(vals))
            A.build_transpose(st)
            released += 1
        rows = np.repeat(np.arange(nrow, dtype=np.int32), np.diff(rp))
        zref = O.coo_tmul(ncol, rows, cc, vals, u)
        zsc = O.coo_tmul(ncol, rows, cc, None if vals is None else np.abs(vals), np.abs(u))
        z = torch.full((ncol,), -1.0, dtype=torch.float64, device="cuda")
        A.spmv(z, d(u), st, transposed=True)
        check(z.cpu().numpy(), zref, zsc, exact, ("spmv_t", what), clen)
        if fixed:
            z2 = torch.full((ncol,), -2.0, dtype=torch.float64, device="cuda")
            A.spmv(z2, d(u), st, transposed=True)
            assert torch.equal(z, z2), ("spmv_t twice", what)
        # k columns
        if k > 1:
            X = rng.integers(-50, 51, (ncol, k)).astype(np.float64) if integer else np.sin(np.arange(ncol * k, dtype=np.float64)).reshape(ncol, k)
            A.prepare(k, st)
            Y = torch.full((nrow, k), -1.0, dtype=torch.float64, device="cuda")
            A.spmm(Y, d(X), k, st)
            Yr = O.csr_mul_n(nrow, rp, cc, vals, X, k)
            Ys = O.csr_mul_n(nrow, rp, cc, None if vals is None else np.abs(vals), np.abs(X), k)
            check(Y.cpu().numpy(), Yr, Ys, exact, ("spmm", what), lens)
            if fixed:
                Y2 = torch.full((nrow, k), -2.0, dtype=torch.float64, device="cuda")
                A.spmm(Y2, d(X), k, st)
                assert torch.equal(Y, Y2), ("spmm twice", what)
        A.close()
        # the reference-named entry point with host struct + host vectors
        if valued:
            S = H.CSR(nrow, ncol, int(rp[-1]), H._ip(rp), H._ip(cc), H._dp(vv))
            f = L.csr_A_mul_B
        else:
            S = H.BCSR(nrow, ncol, int(rp[-1]), H._ip(rp), H._ip(cc))
            f = L.bcsr_A_mul_B
        f.restype = None
        yh = np.full(max(nrow, 1), -1.0)
        f(H._dp(yh), C.byref(S), H._dp(x.copy()))
        L.fs_invalidate(C.byref(S))
        check(yh[:nrow], ref, scale, exact, ("dropin", what), lens)
    finally:
        for name in force:
            capi.set_option(name, PRESET.get(name, RESET.get(name, 0)))
    cases += 1
    if cases % 25 == 0:
        print("%d cases ok" % cases, flush=True)
print("fuzz: %d cases, all within the bars (seed %d; %d handles took their released arrays back)" % (cases, seed, released))
