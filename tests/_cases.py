"""Shared parity cases.

A *case* is a COO matrix plus input vectors; `run_case(backend, case)` pushes it
through every entry point of the hot path and returns
{output name: ndarray}.  Three backends implement the same small interface:

  RefBackend     tests/_refbind.py   the real reference (build container only)
  OracleBackend  oracle/pyoracle.py  the CPU restatement
  HipBackend     tests/_hipbackend.py the product, through the C-ABI on the GPU

make_golden.py stores run_case(RefBackend) under tests/golden/; the CPU tests
require OracleBackend == golden bit for bit; the GPU tests compare HipBackend
with both.
"""
import numpy as np

import _synth as S

BIN_SPMM = [("bcsr_A_mul_B2", 2), ("bcsr_A_mul_B4", 4), ("bcsr_A_mul_B8", 8), ("bcsr_A_mul_B8_auto", 8)]
BIN_SPMM_VAR = [("bcsr_A_mul_Bn", 3), ("bcsr_A_mul_Bn", 8), ("bcsr_A_mul_B32n", 5), ("bcsr_A_mul_B32n", 32)]
VAL_SPMM = [2, 4, 8, 32]


class Case:
    def __init__(self, name, nrow, ncol, rows, cols, vals, xs, block_sizes=(8, 1024), colblocks=(8,), kmax=32):
        self.name, self.nrow, self.ncol = name, nrow, ncol
        self.rows, self.cols, self.vals = rows, cols, vals
        self.xs = xs                      # {tag: x over columns}
        self.block_sizes, self.colblocks, self.kmax = block_sizes, colblocks, kmax

    def xt(self, tag):
        """matching vector over rows for the transposed products"""
        if tag == "int":
            return S.x_int(77, self.nrow)
        return S.x_sin(self.nrow, 11.0, -0.2)


def _kat_sbm():
    # make_sbm(), test_sparse.c:16-28
    rows = np.array([0, 3, 3, 1, 2], np.int32)
    cols = np.array([0, 2, 0, 2, 1], np.int32)
    return Case("kat_sbm_4x3", 4, 3, rows, cols, None,
                {"kat": np.array([0.5, -0.7, 1.9])}, block_sizes=(2,), colblocks=(2,), kmax=8)


def _kat_sdm():
    # make_sdm(), test_sparse.c:395-410
    rows = np.array([1, 1, 3, 4, 1, 4, 5, 0, 1, 2, 4], np.int32)
    cols = np.array([0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3], np.int32)
    vals = np.array([0.65, 0.84, 0.54, 0.59, 0.51, 0.27, 0.23, 0.94, 0.66, 0.31, 0.92])
    return Case("kat_sdm_6x4", 6, 4, rows, cols, vals,
                {"kat": np.array([0.5, -0.7, 1.9, 2.3])}, block_sizes=(4,), colblocks=(2,), kmax=8)


def all_cases():
    out = [_kat_sbm(), _kat_sdm()]
    nrow, ncol, rows, cols, _ = S.fixture_sbm()
    i = np.arange(ncol, dtype=np.float64)
    xs = {"t19": np.sin(i * 19 + 0.4) + np.cos(i * i * 3),      # test_sparse.c:51
          "t17": np.sin(i * 17 + 0.2),                           # test_sparse.c:306
          "bench": S.x_sin(ncol), "int": S.x_int(1, ncol)}
    out.append(Case("fix_sbm_100x50", nrow, ncol, rows, cols, None, xs))
    nrow, ncol, rows, cols, vals = S.fixture_sdm()
    out.append(Case("fix_sdm_100x50", nrow, ncol, rows, cols, vals, xs))

    def syn(name, seed, nrow, ncol, per_row, kmax=8, **kw):
        r, c, v = S.synth_coo(seed, nrow, ncol, per_row, **kw)
        return Case(name, nrow, ncol, r, c, v,
                    {"bench": S.x_sin(ncol), "int": S.x_int(seed, ncol)},
                    block_sizes=(64, 1024), colblocks=(512,), kmax=kmax)

    out.append(syn("syn_u16_2048", 0x5EED01, 2048, 2048, 16, kmax=32))
    out.append(syn("syn_empty_1500x900", 0x5EED02, 1500, 900, 7, empty_frac=0.1))
    out.append(syn("syn_dup_1024", 0x5EED03, 1024, 1024, 12, dup_frac=0.2))
    out.append(syn("syn_long_800x5000", 0x5EED04, 800, 5000, 3, long_row=(411, 50000)))
    out.append(syn("syn_rowmajor_3000x200", 0x5EED05, 3000, 200, 64, shuffle=False))
    # wide: 40 000 columns = 3 bands of the two-pass copy, 20 slices of the LDS-staged one; tall: 70 columns
    out.append(syn("syn_wide_300x40000", 0x5EED06, 300, 40000, 40))
    out.append(syn("syn_tall_4000x70", 0x5EED07, 4000, 70, 5))
    return out


def run_case(be, case, tags=None, light=False, absolute=False):
    """Push `case` through every hot-path entry point of backend `be`.

    absolute=True runs the same products on |A| and |x| (no solver outputs): with the oracle backend that is
    sum_j |a_ij||x_j| per output element, the scale of SURVEY N2's row-scaled bound."""
    c = case
    res = {}
    binary_vals = None
    ab = (lambda a: None if a is None else np.abs(a)) if absolute else (lambda a: a)
    cvals = ab(c.vals)
    for tag, x in c.xs.items():
        if tags and tag not in tags:
            continue
        xt = ab(c.xt(tag))
        x = ab(x)
        # --- pattern-only family: sparse.h / csr.h upper half / cbcsr.h ---
        res[f"A_mul_B/{tag}"] = be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, binary_vals, x)
        res[f"At_mul_B/{tag}"] = be.coo_tmul(c.nrow, c.ncol, c.rows, c.cols, binary_vals, xt)
        res[f"bcsr_A_mul_B/{tag}"] = be.csr_mul(c.nrow, c.ncol, c.rows, c.cols, binary_vals, x)
        res[f"bcsr_AA_mul_B/{tag}"] = be.aa_mul(c.nrow, c.ncol, c.rows, c.cols, x, False)
        res[f"parallel_bcsr_AA_mul_B/{tag}"] = be.aa_mul(c.nrow, c.ncol, c.rows, c.cols, x, True)
        for bs in c.block_sizes:
            res[f"bsbm_A_mul_B/bs{bs}/{tag}"] = be.blocked_mul(c.nrow, c.ncol, c.rows, c.cols, binary_vals, bs,
                                                               x, 1, "bsbm_A_mul_B")
        for cbs in c.colblocks:
            res[f"cbcsr_A_mul_B/cb{cbs}/{tag}"] = be.cbcsr_mul(c.nrow, c.ncol, c.rows, c.cols, cbs, x)
        # --- valued family: dsparse.h / csr.h lower half ---
        if c.vals is not None:
            res[f"sdm_A_mul_B/{tag}"] = be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, cvals, x)
            res[f"sdm_At_mul_B/{tag}"] = be.coo_tmul(c.nrow, c.ncol, c.rows, c.cols, cvals, xt)
            res[f"csr_A_mul_B/{tag}"] = be.csr_mul(c.nrow, c.ncol, c.rows, c.cols, cvals, x)
            for bs in c.block_sizes:
                res[f"bsdm_A_mul_B/bs{bs}/{tag}"] = be.blocked_mul(c.nrow, c.ncol, c.rows, c.cols, cvals, bs,
                                                                   x, 1, "bsdm_A_mul_B")
    if light:
        return res
    # --- multi right-hand-side products, X row-major (bench_a_mul_b.c:149) ---
    for name, k in BIN_SPMM + BIN_SPMM_VAR:
        if k > c.kmax:
            continue
        X = ab(S.X_sin(c.ncol, k))
        res[f"{name}/k{k}"] = be.csr_mul_n(c.nrow, c.ncol, c.rows, c.cols, None, X, k, name)
    bs = c.block_sizes[0]
    for name, k in [("bsbm_A_mul_B2", 2), ("bsbm_A_mul_B4", 4), ("bsbm_A_mul_Bn", 3)]:
        X = ab(S.X_sin(c.ncol, k))
        res[f"{name}/bs{bs}"] = be.blocked_mul(c.nrow, c.ncol, c.rows, c.cols, None, bs, X, k, name)
    if c.vals is not None:
        for k in VAL_SPMM:
            if k > c.kmax:
                continue
            X = ab(S.X_sin(c.ncol, k))
            res[f"csr_A_mul_Bn/k{k}"] = be.csr_mul_n(c.nrow, c.ncol, c.rows, c.cols, cvals, X, k, "csr_A_mul_Bn")
    # --- the consumers of the path: CG on (A'A + lambda I), one and two right-hand sides ---
    # (F >= 16: on a 3-column matrix the second 2-column search block is rank deficient and the 2x2 solves of
    #  bsbm_cg2 divide by rounding noise -- in the reference too)
    if 16 <= c.ncol <= 2100 and hasattr(be, "cg") and not absolute:
        b1, b2 = cg_rhs(c.ncol)
        x, it = be.cg(c.nrow, c.ncol, c.rows, c.cols, b1, CG_LAMBDA, CG_TOL, False)
        res["bsbm_cg/x"], res["bsbm_cg/iter"] = x, np.array([float(it)])
        X, it = be.cg(c.nrow, c.ncol, c.rows, c.cols, np.stack([b1, b2], axis=1).copy(), CG_LAMBDA, CG_TOL, True)
        res["bsbm_cg2/X"], res["bsbm_cg2/iter"] = X, np.array([float(it)])
    return res


CG_LAMBDA, CG_TOL = 5.0, 1e-6


def cg_rhs(F):
    """right-hand sides of test_cg (test_sparse.c:570-572, :594-597)"""
    i = np.arange(F, dtype=np.float64)
    return np.sin(i * 19 + 0.4) + np.cos(i * i * 3), np.cos(i * 23 + 0.7) + np.sin(i * i * 7)


# ----------------------------------------------------------------------------
class OracleBackend:
    """oracle/pyoracle.py behind the case interface."""

    def __init__(self, lib=None):
        from oracle import pyoracle
        self.O, self.lib = pyoracle, lib

    def coo_mul(self, nrow, ncol, rows, cols, vals, x):
        return self.O.coo_mul(nrow, rows, cols, vals, x, lib=self.lib)

    def coo_tmul(self, nrow, ncol, rows, cols, vals, x):
        return self.O.coo_tmul(ncol, rows, cols, vals, x, lib=self.lib)

    def csr_mul(self, nrow, ncol, rows, cols, vals, x):
        rp, cc, vv = self.O.coo_to_csr(nrow, rows, cols, vals, lib=self.lib)
        return self.O.csr_mul(nrow, rp, cc, vv, x, lib=self.lib)

    def csr_mul_n(self, nrow, ncol, rows, cols, vals, X, k, name):
        rp, cc, vv = self.O.coo_to_csr(nrow, rows, cols, vals, lib=self.lib)
        return self.O.csr_mul_n(nrow, rp, cc, vv, X, k, lib=self.lib)

    def aa_mul(self, nrow, ncol, rows, cols, x, parallel):
        rp, cc, _ = self.O.coo_to_csr(nrow, rows, cols, None, lib=self.lib)
        return self.O.bcsr_aa_mul(nrow, ncol, rp, cc, x, lib=self.lib)

    def blocked_mul(self, nrow, ncol, rows, cols, vals, bs, X, k, name):
        blk = self.O.coo_to_blocked(nrow, bs, rows, cols, vals, lib=self.lib)
        return self.O.blocked_mul_n(nrow, blk, X, k, lib=self.lib)

    def cbcsr_mul(self, nrow, ncol, rows, cols, cbs, x):
        nb, rp, cc = self.O.coo_to_cbcsr(cbs, nrow, ncol, rows, cols, lib=self.lib)
        return self.O.cbcsr_mul(nrow, nb, rp, cc, x, lib=self.lib)

    def cg(self, nrow, ncol, rows, cols, b, lam, tol, two):
        return self.O.cg_normal(nrow, ncol, rows, cols, b, lam, tol, two, lib=self.lib)


class RefBackend:
    """The real reference (oracle/_ref/libfsref.so) behind the case interface."""

    def __init__(self, fast=False):
        import ctypes as C
        import _refbind
        self.C, self.R = C, _refbind.Ref(fast=fast)

    def coo_mul(self, nrow, ncol, rows, cols, vals, x):
        if vals is None:
            return self.R.A_mul_B(self.R.sbm(nrow, ncol, rows, cols), x)
        return self.R.sdm_A_mul_B(self.R.sdm(nrow, ncol, rows, cols, vals), x)

    def coo_tmul(self, nrow, ncol, rows, cols, vals, x):
        if vals is None:
            return self.R.At_mul_B(self.R.sbm(nrow, ncol, rows, cols), x)
        return self.R.sdm_At_mul_B(self.R.sdm(nrow, ncol, rows, cols, vals), x)

    def csr_mul(self, nrow, ncol, rows, cols, vals, x):
        if vals is None:
            return self.R.bcsr_A_mul_B(self.R.bcsr(nrow, ncol, rows, cols), x)
        return self.R.csr_A_mul_B(self.R.csr(nrow, ncol, rows, cols, vals), x)

    def csr_mul_n(self, nrow, ncol, rows, cols, vals, X, k, name):
        if vals is not None:
            return self.R.csr_A_mul_Bn(self.R.csr(nrow, ncol, rows, cols, vals), X, k)
        A = self.R.bcsr(nrow, ncol, rows, cols)
        if name in ("bcsr_A_mul_Bn", "bcsr_A_mul_B32n"):
            return self.R.bcsr_var(name, A, X, k)
        return self.R.bcsr_fixed(name, A, X, k)

    def aa_mul(self, nrow, ncol, rows, cols, x, parallel):
        A = self.R.bcsr(nrow, ncol, rows, cols)
        return self.R.parallel_bcsr_AA_mul_B(A, x) if parallel else self.R.bcsr_AA_mul_B(A, x)

    def blocked_mul(self, nrow, ncol, rows, cols, vals, bs, X, k, name):
        if vals is not None:
            B = self.R.lib.new_bsdm(self.C.byref(self.R.sdm(nrow, ncol, rows, cols, vals)), bs)
            return self.R.bsdm_A_mul_B(B, X)
        B = self.R.lib.new_bsbm(self.C.byref(self.R.sbm(nrow, ncol, rows, cols)), bs)
        extra = (self.C.c_int(k),) if name == "bsbm_A_mul_Bn" else ()
        y = self.R.bsbm(name, B, X, k, *extra)
        return y.reshape(nrow, k) if k > 1 else y

    def cbcsr_mul(self, nrow, ncol, rows, cols, cbs, x):
        return self.R.cbcsr_A_mul_B(self.R.cbcsr(cbs, nrow, ncol, rows, cols), x)

    def cg(self, nrow, ncol, rows, cols, b, lam, tol, two):
        # as test_cg (test_sparse.c:560-566) without the Hilbert pre-sort: B = new_bsbm(A, 8), Bt = new_bsbm(A', 8)
        C = self.C
        s = self.R.sbm(nrow, ncol, rows, cols)
        B = self.R.lib.new_bsbm(C.byref(s), 8)
        st = self.R.sbm(ncol, nrow, cols, rows)
        Bt = self.R.lib.new_bsbm(C.byref(st), 8)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1).copy()
        x = np.full(b.size, -1.0)
        it = C.c_int(-1)
        f = self.R.lib.bsbm_cg2 if two else self.R.lib.bsbm_cg
        f.restype = None
        f(x.ctypes.data_as(C.POINTER(C.c_double)), B, Bt, b.ctypes.data_as(C.POINTER(C.c_double)),
          C.c_double(lam), C.c_double(tol), C.byref(it))
        return (x.reshape(ncol, 2) if two else x), it.value
