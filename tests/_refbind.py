"""ctypes binding of oracle/_ref/libfsref.so = the REAL reference compiled from
/root/reference (oracle/Makefile target `ref`).  Exists only in the build
container; tests that need it are marked `ref` and skip elsewhere."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libfsref.so")
REF_FAST_SO = os.path.join(ROOT, "oracle", "_ref", "libfsref_fast.so")

ip = C.POINTER(C.c_int)
dp = C.POINTER(C.c_double)


class SBM(C.Structure):      # sparse.h:11-18
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("rows", ip), ("cols", ip)]


class SDM(C.Structure):      # dsparse.h:11-19
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("rows", ip), ("cols", ip), ("vals", dp)]


class BCSR(C.Structure):     # csr.h:15-22
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("row_ptr", ip), ("cols", ip)]


class CSR(C.Structure):      # csr.h:358-366
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("row_ptr", ip), ("cols", ip), ("vals", dp)]


class CBCSR(C.Structure):    # cbcsr.h:5-14
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nblocks", C.c_int), ("colblocksize", C.c_int),
                ("nnz", C.c_int), ("row_ptr", ip), ("cols", ip)]


class BSBM(C.Structure):     # sparse.h:163-172
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nblocks", C.c_int), ("start_row", ip), ("nnz", ip),
                ("rows", C.POINTER(ip)), ("cols", C.POINTER(ip))]


class BSDM(C.Structure):     # dsparse.h:119-129
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nblocks", C.c_int), ("start_row", ip), ("nnz", ip),
                ("rows", C.POINTER(ip)), ("cols", C.POINTER(ip)), ("vals", C.POINTER(dp))]


def available():
    return os.path.exists(REF_SO)


def _ip(a):
    return a.ctypes.data_as(ip)


def _dp(a):
    return a.ctypes.data_as(dp)


class Ref:
    """Thin object API over the reference build; numpy in, numpy out."""

    def __init__(self, fast=False):
        self.lib = C.CDLL(REF_FAST_SO if fast else REF_SO)
        L = self.lib
        L.new_bsbm.restype = C.POINTER(BSBM)
        L.new_bsbm.argtypes = [C.POINTER(SBM), C.c_int]
        L.new_bsdm.restype = C.POINTER(BSDM)
        L.new_bsdm.argtypes = [C.POINTER(SDM), C.c_int]
        L.read_sbm.restype = C.POINTER(SBM)
        L.read_sbm.argtypes = [C.c_char_p]
        L.read_sdm.restype = C.POINTER(SDM)
        L.read_sdm.argtypes = [C.c_char_p]
        self._keep = []

    # --- containers -------------------------------------------------------
    def sbm(self, nrow, ncol, rows, cols):
        rows, cols = rows.copy(), cols.copy()
        self._keep += [rows, cols]
        return SBM(nrow, ncol, len(rows), _ip(rows), _ip(cols))

    def sdm(self, nrow, ncol, rows, cols, vals):
        rows, cols, vals = rows.copy(), cols.copy(), vals.copy()
        self._keep += [rows, cols, vals]
        return SDM(nrow, ncol, len(rows), _ip(rows), _ip(cols), _dp(vals))

    def bcsr(self, nrow, ncol, rows, cols):
        A = BCSR()
        r, c = rows.copy(), cols.copy()
        self.lib.ref_new_bcsr(C.byref(A), C.c_long(len(r)), nrow, ncol, _ip(r), _ip(c))
        return A

    def csr(self, nrow, ncol, rows, cols, vals):
        A = CSR()
        r, c, v = rows.copy(), cols.copy(), vals.copy()
        self.lib.ref_new_csr(C.byref(A), C.c_long(len(r)), nrow, ncol, _ip(r), _ip(c), _dp(v))
        return A

    def cbcsr(self, cbs, nrow, ncol, rows, cols):
        A = CBCSR()
        r, c = rows.copy(), cols.copy()
        self.lib.ref_new_cbcsr(C.byref(A), cbs, C.c_long(len(r)), nrow, ncol, _ip(r), _ip(c))
        return A

    @staticmethod
    def arr(ptr, n, dtype):
        if n == 0:
            return np.empty(0, dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)

    # --- kernels ------------------------------------------------------------
    def _call(self, name, nout, A, x, *extra):
        y = np.full(nout, -1.0)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        getattr(self.lib, name)(_dp(y), C.byref(A) if not isinstance(A, C._Pointer) else A, _dp(x), *extra)
        return y

    def A_mul_B(self, A, x):
        return self._call("A_mul_B", A.nrow, A, x)

    def At_mul_B(self, A, x):
        return self._call("At_mul_B", A.ncol, A, x)

    def sdm_A_mul_B(self, A, x):
        return self._call("sdm_A_mul_B", A.nrow, A, x)

    def sdm_At_mul_B(self, A, x):
        return self._call("sdm_At_mul_B", A.ncol, A, x)

    def csr_A_mul_B(self, A, x):
        return self._call("csr_A_mul_B", A.nrow, A, x)

    def csr_A_mul_Bn(self, A, X, k):
        return self._call("csr_A_mul_Bn", A.nrow * k, A, X, C.c_int(k)).reshape(A.nrow, k)

    def bcsr_A_mul_B(self, A, x):
        return self._call("bcsr_A_mul_B", A.nrow, A, x)

    def bcsr_fixed(self, name, A, X, k):
        return self._call(name, A.nrow * k, A, X).reshape(A.nrow, k)

    def bcsr_var(self, name, A, X, k):
        return self._call(name, A.nrow * k, A, X, C.c_int(k)).reshape(A.nrow, k)

    def bcsr_AA_mul_B(self, A, x):
        return self._call("bcsr_AA_mul_B", A.ncol, A, x)

    def parallel_bcsr_AA_mul_B(self, A, x):
        ytmp = np.zeros(A.ncol * 4)
        return self._call("parallel_bcsr_AA_mul_B", A.ncol, A, x, _dp(ytmp))

    def cbcsr_A_mul_B(self, A, x):
        return self._call("cbcsr_A_mul_B", A.nrow, A, x)

    def bsbm(self, name, B, X, k, *extra):
        nrow = B.contents.nrow
        return self._call(name, nrow * k, B, X, *extra)

    def bsdm_A_mul_B(self, B, x):
        return self._call("bsdm_A_mul_B", B.contents.nrow, B, x)
