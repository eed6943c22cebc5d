"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module (see the header of fs_oracle.c).  The product package
libfastsparse_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so with gcc (strict IEEE flags, see oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("fs_oracle.c", "fs_synth.c", "fs_oracle_cg.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def build_fast(out):
    """CPU-baseline build with the reference's own flags (reference Makefile:2);
    must be built on the machine that times it (-march=native)."""
    subprocess.check_call(["make", "-C", _HERE, "fast", "OUT=" + out], stdout=subprocess.DEVNULL)
    return out


def _vp(a):
    """nullable double* argument"""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(C.c_void_p)


def load(path=None):
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    lib = C.CDLL(path or build())
    lib.fso_coo_to_csr.argtypes = [C.c_int64, C.c_int, _i32p, _i32p, C.c_void_p, _i32p, _i32p, C.c_void_p]
    lib.fso_num_blocks.argtypes = [C.c_int, C.c_int]
    lib.fso_num_blocks.restype = C.c_int
    lib.fso_coo_to_blocked.argtypes = [C.c_int64, C.c_int, C.c_int, _i32p, _i32p, C.c_void_p,
                                       _i32p, _i32p, _i64p, _i32p, _i32p, C.c_void_p]
    lib.fso_coo_to_cbcsr.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p]
    lib.fso_csr_mul.argtypes = [_f64p, C.c_int, _i32p, _i32p, C.c_void_p, _f64p]
    lib.fso_csr_mul_n.argtypes = [_f64p, C.c_int, _i32p, _i32p, C.c_void_p, _f64p, C.c_int]
    lib.fso_csr_mul_n_reference_schedule.argtypes = lib.fso_csr_mul_n.argtypes
    lib.fso_csr_mul_n_reference_schedule.restype = None
    lib.fso_bcsr_aa_mul.argtypes = [_f64p, C.c_int, C.c_int, _i32p, _i32p, _f64p]
    lib.fso_coo_mul.argtypes = [_f64p, C.c_int, C.c_int64, _i32p, _i32p, C.c_void_p, _f64p]
    lib.fso_coo_tmul.argtypes = [_f64p, C.c_int, C.c_int64, _i32p, _i32p, C.c_void_p, _f64p]
    lib.fso_blocked_mul_n.argtypes = [_f64p, C.c_int, _i32p, _i32p, _i64p, _i32p, _i32p, C.c_void_p, _f64p, C.c_int]
    lib.fso_cbcsr_mul.argtypes = [_f64p, C.c_int, C.c_int, _i32p, _i32p, _f64p]
    lib.fso_csr_abs_scale.argtypes = [_f64p, C.c_int, _i32p, _i32p, C.c_void_p, _f64p]
    lib.fso_threads.restype = C.c_int
    for f in ("fso_cg", "fso_cg2"):
        getattr(lib, f).argtypes = [_f64p, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p, C.c_double, C.c_double]
        getattr(lib, f).restype = C.c_int
    for f in ("fso_coo_to_csr", "fso_coo_to_blocked", "fso_coo_to_cbcsr", "fso_csr_mul", "fso_csr_mul_n",
              "fso_bcsr_aa_mul", "fso_coo_mul", "fso_coo_tmul", "fso_blocked_mul_n", "fso_cbcsr_mul",
              "fso_csr_abs_scale"):
        getattr(lib, f).restype = None
    if path is None:
        _LIB = lib
    return lib


# ----------------------------------------------------------------------------
# numpy-level wrappers (lib=None -> strict liboracle.so)
# ----------------------------------------------------------------------------

def coo_to_csr(nrow, rows, cols, vals=None, lib=None):
    lib = lib or load()
    nnz = len(rows)
    row_ptr = np.empty(nrow + 1, np.int32)
    oc = np.empty(nnz, np.int32)
    ov = np.empty(nnz, np.float64) if vals is not None else None
    lib.fso_coo_to_csr(nnz, nrow, rows, cols, _vp(vals), row_ptr, oc, _vp(ov))
    return row_ptr, oc, ov


def coo_to_blocked(nrow, block_size, rows, cols, vals=None, lib=None):
    lib = lib or load()
    nnz = len(rows)
    nb = lib.fso_num_blocks(nrow, block_size)
    start_row = np.empty(nb + 1, np.int32)
    blk_nnz = np.zeros(nb, np.int32)
    blk_off = np.empty(nb + 1, np.int64)
    orr = np.empty(nnz, np.int32)
    oc = np.empty(nnz, np.int32)
    ov = np.empty(nnz, np.float64) if vals is not None else None
    lib.fso_coo_to_blocked(nnz, nrow, block_size, rows, cols, _vp(vals),
                           start_row, blk_nnz, blk_off, orr, oc, _vp(ov))
    return dict(nblocks=nb, start_row=start_row, blk_nnz=blk_nnz, blk_off=blk_off,
                rows=orr, cols=oc, vals=ov)


def coo_to_cbcsr(colblocksize, nrow, ncol, rows, cols, lib=None):
    lib = lib or load()
    nb = lib.fso_num_blocks(ncol, colblocksize)
    row_ptr = np.empty(nb * nrow + 1, np.int32)
    oc = np.empty(len(rows), np.int32)
    lib.fso_coo_to_cbcsr(colblocksize, len(rows), nrow, ncol, rows, cols, row_ptr, oc)
    return nb, row_ptr, oc


def csr_mul(nrow, row_ptr, cols, vals, x, lib=None):
    lib = lib or load()
    y = np.full(nrow, -1.0)
    lib.fso_csr_mul(y, nrow, row_ptr, cols, _vp(vals), x)
    return y


def csr_mul_n(nrow, row_ptr, cols, vals, X, k, lib=None):
    lib = lib or load()
    Y = np.full(nrow * k, -1.0)
    lib.fso_csr_mul_n(Y, nrow, row_ptr, cols, _vp(vals), np.ascontiguousarray(X).reshape(-1), k)
    return Y.reshape(nrow, k)


def bcsr_aa_mul(nrow, ncol, row_ptr, cols, x, lib=None):
    lib = lib or load()
    y = np.full(ncol, -1.0)
    lib.fso_bcsr_aa_mul(y, nrow, ncol, row_ptr, cols, x)
    return y


def coo_mul(nrow, rows, cols, vals, x, lib=None):
    lib = lib or load()
    y = np.full(nrow, -1.0)
    lib.fso_coo_mul(y, nrow, len(rows), rows, cols, _vp(vals), x)
    return y


def coo_tmul(ncol, rows, cols, vals, x, lib=None):
    lib = lib or load()
    y = np.full(ncol, -1.0)
    lib.fso_coo_tmul(y, ncol, len(rows), rows, cols, _vp(vals), x)
    return y


def blocked_mul_n(nrow, blk, X, k, lib=None):
    lib = lib or load()
    Y = np.full(nrow * k, -1.0)
    lib.fso_blocked_mul_n(Y, blk["nblocks"], blk["start_row"], np.ascontiguousarray(blk["blk_nnz"]),
                          blk["blk_off"], blk["rows"], blk["cols"], _vp(blk["vals"]),
                          np.ascontiguousarray(X).reshape(-1), k)
    return Y.reshape(nrow, k) if k > 1 else Y


def cbcsr_mul(nrow, nblocks, row_ptr, cols, x, lib=None):
    lib = lib or load()
    y = np.full(nrow, -1.0)
    lib.fso_cbcsr_mul(y, nrow, nblocks, row_ptr, cols, x)
    return y


def csr_abs_scale(nrow, row_ptr, cols, vals, x, lib=None):
    lib = lib or load()
    s = np.empty(nrow)
    lib.fso_csr_abs_scale(s, nrow, row_ptr, cols, _vp(vals), x)
    return s


def cg_normal(nrow, ncol, rows, cols, b, lam, tol, two=False, lib=None):
    """(A'A + lam I) x = b on the pattern-only COO (rows, cols); b is F or F x 2 row-major.  Returns (x, iterations)."""
    lib = lib or load()
    a_rp, a_cc, _ = coo_to_csr(nrow, rows, cols, None, lib=lib)
    t_rp, t_cc, _ = coo_to_csr(ncol, cols, rows, None, lib=lib)     # A' rows keep the COO entry order, like new_bsbm(At)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
    x = np.full(b.size, -1.0)
    f = lib.fso_cg2 if two else lib.fso_cg
    it = f(x, nrow, ncol, a_rp, a_cc, t_rp, t_cc, b, float(lam), float(tol))
    return (x.reshape(ncol, 2) if two else x), it
