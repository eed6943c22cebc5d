"""The product behind the case interface of tests/_cases.py, two ways:

HipDeviceBackend  device-resident layer (fs_* of include/fastsparse_hip.h) on torch CUDA tensors
HipDropinBackend  the reference-named entry points (include/sparse.h, dsparse.h, csr.h, cbcsr.h) with
                  the reference's host structs and host numpy vectors -- what an existing C caller does
"""
import ctypes as C

import numpy as np

from libfastsparse_amd import capi

ip = C.POINTER(C.c_int)
dp = C.POINTER(C.c_double)


class HipDeviceBackend:
    def __init__(self):
        import torch
        self.t = torch
        self.L = capi.lib()
        self.dev = "cuda"

    def _d(self, a, dtype=None):
        return self.t.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def _coo(self, nrow, ncol, rows, cols, vals):
        return capi.Matrix.from_coo(nrow, ncol, self._d(rows), self._d(cols), None if vals is None else self._d(vals))

    def _out(self, n):
        return self.t.full((n,), -1.0, dtype=self.t.float64, device=self.dev)

    def coo_mul(self, nrow, ncol, rows, cols, vals, x):
        m = self._coo(nrow, ncol, rows, cols, vals)
        y = self._out(nrow)
        m.spmv(y, self._d(x), capi.current_stream())
        return y.cpu().numpy()

    def coo_tmul(self, nrow, ncol, rows, cols, vals, x):
        # the serial loop order of At_mul_B: upload the COO with rows/cols swapped
        m = self._coo(ncol, nrow, cols, rows, vals)
        y = self._out(ncol)
        m.spmv(y, self._d(x), capi.current_stream())
        return y.cpu().numpy()

    def csr_mul(self, nrow, ncol, rows, cols, vals, x):
        return self.coo_mul(nrow, ncol, rows, cols, vals, x)

    def csr_mul_n(self, nrow, ncol, rows, cols, vals, X, k, name):
        m = self._coo(nrow, ncol, rows, cols, vals)
        Y = self._out(nrow * k)
        m.prepare(k, capi.current_stream())
        m.spmm(Y, self._d(np.ascontiguousarray(X).reshape(-1)), k, capi.current_stream())
        return Y.cpu().numpy().reshape(nrow, k)

    def aa_mul(self, nrow, ncol, rows, cols, x, parallel):
        m = self._coo(nrow, ncol, rows, cols, None)
        y = self._out(ncol)
        tmp = self._out(max(nrow, 1))
        m.ata(y, self._d(x), tmp, capi.current_stream())
        return y.cpu().numpy()

    def blocked_mul(self, nrow, ncol, rows, cols, vals, bs, X, k, name):
        # a row-blocked COO is the COO regrouped by row block; per-row entry order is unchanged
        order = np.argsort(rows // bs, kind="stable")
        m = self._coo(nrow, ncol, rows[order], cols[order], None if vals is None else vals[order])
        Y = self._out(nrow * k)
        m.prepare(k, capi.current_stream())
        m.spmm(Y, self._d(np.ascontiguousarray(X).reshape(-1)), k, capi.current_stream())
        Y = Y.cpu().numpy()
        return Y.reshape(nrow, k) if k > 1 else Y

    def cbcsr_mul(self, nrow, ncol, rows, cols, cbs, x):
        # format built by the product's own host constructor (new_cbcsr, fs_host.c), then uploaded
        F = HostFormats()
        K = F.cbcsr(cbs, nrow, ncol, np.ascontiguousarray(rows), np.ascontiguousarray(cols))
        rp = F.arr(K.row_ptr, K.nblocks * nrow + 1, np.int32)
        cc = F.arr(K.cols, len(rows), np.int32)
        m = capi.ColBlockMatrix(nrow, ncol, K.nblocks, cbs, self._d(rp), self._d(cc))
        y = self._out(nrow)
        m.spmv(y, self._d(x), capi.current_stream())
        return y.cpu().numpy()

    def cg(self, nrow, ncol, rows, cols, b, lam, tol, two):
        """fs_cg / fs_cg2 on device vectors; A and A' as two handles, like the reference passes B and Bt"""
        A = self._coo(nrow, ncol, rows, cols, None)
        At = self._coo(ncol, nrow, cols, rows, None)
        bd = self._d(np.ascontiguousarray(b, dtype=np.float64).reshape(-1))
        xd = self._out(bd.numel())
        it = C.c_int(-1)
        f = self.L.fs_cg2 if two else self.L.fs_cg
        capi.check(f(A.h, At.h, xd.data_ptr(), bd.data_ptr(), float(lam), float(tol), C.byref(it), capi.current_stream()))
        x = xd.cpu().numpy()
        return (x.reshape(ncol, 2) if two else x), it.value

    def transposed_csr_mul(self, nrow, ncol, rows, cols, vals, x):
        """A' x through fs_matrix_build_transpose + fs_spmv_t (the CSR At_mul_B of BASELINE config 2)"""
        m = self._coo(nrow, ncol, rows, cols, vals)
        m.build_transpose(capi.current_stream())
        y = self._out(ncol)
        m.spmv(y, self._d(x), capi.current_stream(), transposed=True)
        return y.cpu().numpy()


# ---- reference structs (same layouts as include/*.h) ------------------------------------------------
class SBM(C.Structure):
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("rows", ip), ("cols", ip)]


class SDM(C.Structure):
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("rows", ip), ("cols", ip), ("vals", dp)]


class BCSR(C.Structure):
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("row_ptr", ip), ("cols", ip)]


class CSR(C.Structure):
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_long), ("row_ptr", ip), ("cols", ip), ("vals", dp)]


class CBCSR(C.Structure):
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nblocks", C.c_int), ("colblocksize", C.c_int),
                ("nnz", C.c_int), ("row_ptr", ip), ("cols", ip)]


class BSBM(C.Structure):
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nblocks", C.c_int), ("start_row", ip), ("nnz", ip),
                ("rows", C.POINTER(ip)), ("cols", C.POINTER(ip))]


class BSDM(C.Structure):
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nblocks", C.c_int), ("start_row", ip), ("nnz", ip),
                ("rows", C.POINTER(ip)), ("cols", C.POINTER(ip)), ("vals", C.POINTER(dp))]


def _ip(a):
    return a.ctypes.data_as(ip)


def _dp(a):
    return a.ctypes.data_as(dp)


class HostFormats:
    """The product's host-side constructors (fs_host.c) -- usable without a GPU."""

    def __init__(self):
        self.L = capi.lib()
        self.L.new_bsbm.restype = C.POINTER(BSBM)
        self.L.new_bsbm.argtypes = [C.POINTER(SBM), C.c_int]
        self.L.new_bsdm.restype = C.POINTER(BSDM)
        self.L.new_bsdm.argtypes = [C.POINTER(SDM), C.c_int]
        self.L.read_sbm.restype = C.POINTER(SBM)
        self.L.read_sbm.argtypes = [C.c_char_p]
        self.L.read_sdm.restype = C.POINTER(SDM)
        self.L.read_sdm.argtypes = [C.c_char_p]
        self.L.new_transpose.restype = C.POINTER(SBM)
        self.L.new_transpose.argtypes = [C.POINTER(SBM)]
        self._keep = []

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
        self.L.new_bcsr(C.byref(A), C.c_long(len(rows)), nrow, ncol, _ip(rows), _ip(cols))
        return A

    def csr(self, nrow, ncol, rows, cols, vals):
        A = CSR()
        self.L.new_csr(C.byref(A), C.c_long(len(rows)), nrow, ncol, _ip(rows), _ip(cols), _dp(vals))
        return A

    def cbcsr(self, cbs, nrow, ncol, rows, cols):
        A = CBCSR()
        self.L.new_cbcsr(C.byref(A), cbs, C.c_long(len(rows)), nrow, ncol, _ip(rows), _ip(cols))
        return A

    @staticmethod
    def arr(ptr, n, dtype):
        if n == 0:
            return np.empty(0, dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class HipDropinBackend(HostFormats):
    """Reference-named entry points with host structs and host vectors."""

    def _call(self, name, nout, A, x, *extra):
        y = np.full(nout, -1.0)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        f = getattr(self.L, name)
        f.restype = None
        f(_dp(y), A if isinstance(A, C._Pointer) else C.byref(A), _dp(x), *extra)
        self.L.fs_invalidate(A if isinstance(A, C._Pointer) else C.byref(A))   # structs are short-lived here
        return y

    def coo_mul(self, nrow, ncol, rows, cols, vals, x):
        if vals is None:
            return self._call("A_mul_B", nrow, self.sbm(nrow, ncol, rows, cols), x)
        return self._call("sdm_A_mul_B", nrow, self.sdm(nrow, ncol, rows, cols, vals), x)

    def coo_tmul(self, nrow, ncol, rows, cols, vals, x):
        if vals is None:
            return self._call("At_mul_B", ncol, self.sbm(nrow, ncol, rows, cols), x)
        return self._call("sdm_At_mul_B", ncol, self.sdm(nrow, ncol, rows, cols, vals), x)

    def csr_mul(self, nrow, ncol, rows, cols, vals, x):
        if vals is None:
            return self._call("bcsr_A_mul_B", nrow, self.bcsr(nrow, ncol, rows, cols), x)
        return self._call("csr_A_mul_B", nrow, self.csr(nrow, ncol, rows, cols, vals), x)

    def csr_mul_n(self, nrow, ncol, rows, cols, vals, X, k, name):
        if vals is not None:
            return self._call("csr_A_mul_Bn", nrow * k, self.csr(nrow, ncol, rows, cols, vals), X, C.c_int(k)).reshape(nrow, k)
        A = self.bcsr(nrow, ncol, rows, cols)
        extra = (C.c_int(k),) if name in ("bcsr_A_mul_Bn", "bcsr_A_mul_B32n") else ()
        return self._call(name, nrow * k, A, X, *extra).reshape(nrow, k)

    def aa_mul(self, nrow, ncol, rows, cols, x, parallel):
        A = self.bcsr(nrow, ncol, rows, cols)
        if parallel:
            ytmp = np.zeros(max(ncol, 1))
            return self._call("parallel_bcsr_AA_mul_B", ncol, A, x, _dp(ytmp))
        return self._call("bcsr_AA_mul_B", ncol, A, x)

    def blocked_mul(self, nrow, ncol, rows, cols, vals, bs, X, k, name):
        if vals is not None:
            s = self.sdm(nrow, ncol, rows, cols, vals)
            return self._call("bsdm_A_mul_B", nrow, self.L.new_bsdm(C.byref(s), bs), X)
        s = self.sbm(nrow, ncol, rows, cols)
        B = self.L.new_bsbm(C.byref(s), bs)
        extra = (C.c_int(k),) if name == "bsbm_A_mul_Bn" else ()
        y = self._call(name, nrow * k, B, X, *extra)
        return y.reshape(nrow, k) if k > 1 else y

    def cbcsr_mul(self, nrow, ncol, rows, cols, cbs, x):
        return self._call("cbcsr_A_mul_B", nrow, self.cbcsr(cbs, nrow, ncol, rows, cols), x)

    def cg(self, nrow, ncol, rows, cols, b, lam, tol, two):
        """bsbm_cg / bsbm_cg2 with host structs and host vectors, as test_cg (test_sparse.c:560-608) calls them"""
        s = self.sbm(nrow, ncol, rows, cols)
        B = self.L.new_bsbm(C.byref(s), 8)
        st = self.sbm(ncol, nrow, cols, rows)
        Bt = self.L.new_bsbm(C.byref(st), 8)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1).copy()
        x = np.full(b.size, -1.0)
        it = C.c_int(-1)
        f = self.L.bsbm_cg2 if two else self.L.bsbm_cg
        f.restype = None
        f(_dp(x), B, Bt, _dp(b), C.c_double(lam), C.c_double(tol), C.byref(it))
        self.L.fs_invalidate(B)
        self.L.fs_invalidate(Bt)
        return (x.reshape(ncol, 2) if two else x), it.value

    def transposed_csr_mul(self, nrow, ncol, rows, cols, vals, x):
        if vals is None:
            return self._call("bcsr_At_mul_B", ncol, self.bcsr(nrow, ncol, rows, cols), x)
        return self._call("csr_At_mul_B", ncol, self.csr(nrow, ncol, rows, cols, vals), x)
