"""ctypes binding of libfastsparse_hip.so (include/fastsparse_hip.h + the reference-named entry points).

Loading fails loudly if the library has not been built (run `python -c "import __graft_entry__ as g; g.build()"`
or `python libfastsparse_amd/_build.py`); nothing here computes on the CPU.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FS_LIB_PATH: an experiment build of the same library (tools/build_variants.py: ablation / instrumented kernels), never a fallback
LIB_PATH = os.environ.get("FS_LIB_PATH") or os.path.join(HERE, "libfastsparse_hip.so")
# FS_HOST_ONLY_LIB=<path>: a library holding ONLY the host half (fs_host.c, fs_sort.c: constructors, loaders, sorters) -- the
# sanitizer build tests/test_host_sanitizers.py runs the host tests against; no device entry point is declared on it
HOST_ONLY_LIB = os.environ.get("FS_HOST_ONLY_LIB")

FS_HOST, FS_DEVICE = 0, 1

vp = C.c_void_p
_lib = None

# every symbol declared in include/*.h (tests check that the library exports all of them)
DEVICE_API = [
    "fs_version", "fs_last_error", "fs_device_count", "fs_set_device", "fs_set_option", "fs_get_option",
    "fs_device_alloc", "fs_device_free", "fs_copy_to_device", "fs_copy_to_host", "fs_device_synchronize",
    "fs_csr_create", "fs_coo_create", "fs_matrix_destroy", "fs_matrix_build_transpose", "fs_matrix_has_transpose", "fs_matrix_spmv_kernel", "fs_matrix_candidate_ms", "fs_matrix_build_ms",
    "fs_matrix_release_csr", "fs_matrix_restore_csr", "fs_matrix_release_prepared",
    "fs_matrix_prepare", "fs_matrix_spmm_plan", "fs_matrix_device_bytes", "fs_spmv_part", "fs_spmv_part_rows", "fs_spmm_part", "fs_spmm_part_rows", "fs_copy_segments",
    "fs_matrix_nrow", "fs_matrix_ncol", "fs_matrix_nnz", "fs_matrix_algorithmic_bytes", "fs_matrix_download",
    "fs_spmv", "fs_spmv_t", "fs_spmv_host", "fs_spmv_t_host", "fs_spmm", "fs_spmm_t", "fs_ata_mul", "fs_cg", "fs_cg2", "fs_axpy",
    "fs_cbcsr_create", "fs_cbcsr_destroy", "fs_cbcsr_spmv", "fs_invalidate", "fs_release_all", "fs_cache_entries",
    "fs_synth_uniform", "fs_synth_powerlaw_lengths", "fs_synth_fill", "fs_bucket_coo", "fs_device_build_wanted",
    "fs_dist_create", "fs_dist_destroy", "fs_dist_ndev", "fs_dist_uses_rccl", "fs_dist_csr_create", "fs_dist_matrix_destroy",
    "fs_dist_matrix_bounds", "fs_dist_matrix_shard_nnz", "fs_dist_spmv", "fs_dist_spmv_resident", "fs_dist_x", "fs_dist_y",
    "fs_dist_matrix_build_transpose", "fs_dist_matrix_has_transpose", "fs_dist_spmv_t", "fs_dist_spmv_t_resident", "fs_dist_swap_xy",
    "fs_dist_z", "fs_dist_cg", "fs_dist_is_conservative", "fs_dist_csr_create_from_shards", "fs_dist_matrix_build_transpose_device",
    "fs_dist_matrix_bounds_t", "fs_dist_matrix_nnz", "fs_dist_matrix_shard", "fs_dist_spmm", "fs_dist_spmm_t", "fs_dist_cg2",
    "fs_dist_coo_create", "fs_dist_matrix_pair", "fs_dist_ata",
]
REFERENCE_API = [
    # sparse.h
    "new_sbm", "free_sbm", "new_transpose", "transpose", "read_sbm", "new_bsbm", "read_long",
    "A_mul_B", "At_mul_B", "bsbm_A_mul_B", "bsbm_A_mul_B2", "bsbm_A_mul_B4", "bsbm_A_mul_Bn",
    # dsparse.h
    "new_sdm", "sdm_transpose", "read_sdm", "new_bsdm", "sdm_A_mul_B", "sdm_At_mul_B", "bsdm_A_mul_B",
    # csr.h
    "new_bcsr", "bcsr_from_sbm", "free_bcsr", "new_csr", "free_csr", "serialize_to_file", "deserialize_from_file",
    "bcsr_A_mul_B", "bcsr_A_mul_B2", "bcsr_A_mul_B4", "bcsr_A_mul_B8", "bcsr_A_mul_B8_auto", "bcsr_A_mul_Bn",
    "bcsr_A_mul_B32n", "bcsr_AA_mul_B", "parallel_bcsr_AA_mul_B", "csr_A_mul_B", "csr_A_mul_Bn",
    "csr_At_mul_B", "bcsr_At_mul_B",
    # cbcsr.h
    "new_cbcsr", "cbcsr_from_sbm", "cbcsr_A_mul_B",
    # cg.h, linalg.h
    "bsbm_AtA", "bsbm_cg", "bsbm_cg2", "dist", "pnormsq", "pnormsq2", "pouter2", "pdot", "pdot2sym", "solve2sym",
    # hilbert.h, quickSort.h, quickSortD.h and the sorters of sparse.h / dsparse.h
    "ceilPower2", "xy2d", "d2xy", "rot", "row_xy2d", "row_d2xy", "quickSort", "quickSortD",
    "sort_sbm", "sort_bsbm", "sort_bsbm_byrow", "sort_sdm", "sort_bsdm",
    # samplers of sparse.h, timing.h, omp_util.h
    "exprand", "randexp", "randsubseq", "timing", "thread_num", "nthreads", "thread_limit", "threads_init",
]


class FastsparseError(RuntimeError):
    pass


def lib():
    """The loaded library.  Import torch first when device tensors are shared with it, so that
    both use the same HIP runtime (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if HOST_ONLY_LIB:
        _lib = C.CDLL(HOST_ONLY_LIB)
        return _lib
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: compile the HIP sources in-tree (hipcc cross-compiles gfx950 without a GPU)
        try:
            from . import _build
            _build.build()
        except Exception as ex:
            raise FastsparseError(f"{LIB_PATH} is missing and could not be built ({ex!r}); "
                                  "there is no CPU fallback") from ex
    L = C.CDLL(LIB_PATH)   # RTLD_LOCAL: the reference-named symbols must not interpose other libraries
    L.fs_version.restype = C.c_char_p
    L.fs_last_error.restype = C.c_char_p
    L.fs_device_alloc.restype = vp
    L.fs_device_alloc.argtypes = [C.c_int64]
    L.fs_device_free.argtypes = [vp]
    L.fs_device_free.restype = None
    L.fs_copy_to_device.argtypes = [vp, vp, C.c_int64]
    L.fs_copy_to_host.argtypes = [vp, vp, C.c_int64]
    L.fs_set_option.argtypes = [C.c_char_p, C.c_int]
    L.fs_get_option.argtypes = [C.c_char_p]
    L.fs_csr_create.restype = vp
    L.fs_csr_create.argtypes = [C.c_int, C.c_int, C.c_int64, vp, vp, vp, C.c_int, C.c_int]
    L.fs_coo_create.restype = vp
    L.fs_coo_create.argtypes = [C.c_int, C.c_int, C.c_int64, vp, vp, vp, C.c_int]
    L.fs_matrix_destroy.argtypes = [vp]
    L.fs_matrix_destroy.restype = None
    L.fs_matrix_build_transpose.argtypes = [vp, vp]
    L.fs_matrix_has_transpose.argtypes = [vp]
    L.fs_matrix_spmv_kernel.argtypes = [vp, C.c_int]
    L.fs_matrix_candidate_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.fs_matrix_build_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.fs_matrix_release_csr.argtypes = [vp]
    L.fs_matrix_restore_csr.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int]
    L.fs_matrix_release_prepared.argtypes = [vp, C.c_int]
    L.fs_matrix_prepare.argtypes = [vp, C.c_int, C.c_int, vp]
    L.fs_matrix_spmm_plan.argtypes = [vp, C.c_int, C.c_int]
    L.fs_matrix_device_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.fs_spmv_part_rows.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.fs_spmv_part.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp]
    L.fs_spmm_part_rows.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.fs_spmm_part.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.fs_copy_segments.argtypes = [C.c_int, vp, C.c_int64, vp, vp, vp]
    L.fs_matrix_nrow.argtypes = [vp]
    L.fs_matrix_ncol.argtypes = [vp]
    L.fs_matrix_nnz.argtypes = [vp]
    L.fs_matrix_nnz.restype = C.c_int64
    L.fs_matrix_algorithmic_bytes.argtypes = [vp, C.c_int]
    L.fs_matrix_algorithmic_bytes.restype = C.c_int64
    L.fs_matrix_download.argtypes = [vp, C.c_int, vp, vp, vp]
    for f in ("fs_spmv", "fs_spmv_t"):
        getattr(L, f).argtypes = [vp, vp, vp, vp]
    for f in ("fs_spmv_host", "fs_spmv_t_host"):
        getattr(L, f).argtypes = [vp, vp, vp]
    for f in ("fs_spmm", "fs_spmm_t"):
        getattr(L, f).argtypes = [vp, vp, vp, C.c_int, vp]
    L.fs_ata_mul.argtypes = [vp, vp, vp, vp, vp]
    for f in ("fs_cg", "fs_cg2"):
        getattr(L, f).argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, C.POINTER(C.c_int), vp]
    L.fs_axpy.argtypes = [C.c_int, C.c_double, vp, vp, vp]
    L.fs_cbcsr_create.restype = vp
    L.fs_cbcsr_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
    L.fs_cbcsr_destroy.argtypes = [vp]
    L.fs_cbcsr_destroy.restype = None
    L.fs_cbcsr_spmv.argtypes = [vp, vp, vp, vp]
    L.fs_invalidate.argtypes = [vp]
    L.fs_invalidate.restype = None
    L.fs_release_all.restype = None
    L.fs_synth_uniform.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, vp, vp, vp, vp]
    L.fs_synth_powerlaw_lengths.argtypes = [C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_int64, vp, vp]
    L.fs_synth_fill.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64, vp, vp, vp, vp]
    L.fs_dist_create.restype = vp
    L.fs_dist_create.argtypes = [C.c_int, vp]
    L.fs_dist_destroy.argtypes = [vp]
    L.fs_dist_destroy.restype = None
    L.fs_dist_ndev.argtypes = [vp]
    L.fs_dist_uses_rccl.argtypes = [vp]
    L.fs_dist_csr_create.restype = vp
    L.fs_dist_csr_create.argtypes = [vp, C.c_int, C.c_int, C.c_int64, vp, vp, vp]
    L.fs_dist_matrix_destroy.argtypes = [vp]
    L.fs_dist_matrix_destroy.restype = None
    L.fs_dist_matrix_bounds.argtypes = [vp, vp]
    L.fs_dist_matrix_shard_nnz.argtypes = [vp, C.c_int]
    L.fs_dist_matrix_shard_nnz.restype = C.c_int64
    L.fs_dist_spmv.argtypes = [vp, vp, vp]
    L.fs_dist_spmv_t.argtypes = [vp, vp, vp]
    L.fs_dist_matrix_build_transpose.argtypes = [vp, vp, vp, vp]
    L.fs_dist_matrix_has_transpose.argtypes = [vp]
    L.fs_dist_is_conservative.argtypes = [vp]
    L.fs_dist_csr_create_from_shards.restype = vp
    L.fs_dist_csr_create_from_shards.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    L.fs_dist_matrix_build_transpose_device.argtypes = [vp]
    L.fs_dist_matrix_bounds_t.argtypes = [vp, vp]
    L.fs_dist_matrix_nnz.argtypes = [vp]
    L.fs_dist_matrix_nnz.restype = C.c_int64
    L.fs_dist_matrix_shard.argtypes = [vp, C.c_int, C.c_int]
    L.fs_dist_matrix_shard.restype = vp
    L.fs_dist_cg.argtypes = [vp, vp, vp, C.c_double, C.c_double, C.POINTER(C.c_int)]
    L.fs_dist_cg2.argtypes = [vp, vp, vp, C.c_double, C.c_double, C.POINTER(C.c_int)]
    L.fs_dist_coo_create.restype = vp
    L.fs_dist_coo_create.argtypes = [vp, C.c_int, C.c_int, C.c_int64, vp, vp, vp]
    L.fs_dist_matrix_pair.restype = vp
    L.fs_dist_matrix_pair.argtypes = [vp, vp]
    L.fs_dist_ata.argtypes = [vp, vp, vp, C.c_double]
    L.fs_dist_spmm.argtypes = [vp, vp, vp, C.c_int]
    L.fs_dist_spmm_t.argtypes = [vp, vp, vp, C.c_int]
    for f in ("fs_dist_spmv_resident", "fs_dist_spmv_t_resident", "fs_dist_swap_xy"):
        getattr(L, f).argtypes = [vp]
    for f in ("fs_dist_x", "fs_dist_y", "fs_dist_z"):
        getattr(L, f).restype = vp
        getattr(L, f).argtypes = [vp, C.c_int]
    _lib = L
    return L


def check(rc, what="call"):
    if rc != 0:
        raise FastsparseError(f"{what} failed ({rc}): {lib().fs_last_error().decode()}")


def _ptr(t):
    """device/host address of a torch tensor, numpy array, or None"""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        assert t.is_contiguous()
        return t.data_ptr()
    assert t.flags.c_contiguous
    return t.ctypes.data


def _space(t):
    return FS_DEVICE if hasattr(t, "is_cuda") and t.is_cuda else FS_HOST


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


class Matrix:
    """Device-resident CSR (valued or pattern-only) behind an fs_matrix_t handle."""

    def __init__(self, handle, keep=()):
        if not handle:
            raise FastsparseError("matrix creation failed: " + lib().fs_last_error().decode())
        self.h = handle
        self._keep = keep  # borrowed device arrays must outlive the handle

    @classmethod
    def from_csr(cls, nrow, ncol, row_ptr, cols, vals=None, borrow=False):
        nnz = int(cols.numel() if hasattr(cols, "numel") else cols.size)
        h = lib().fs_csr_create(nrow, ncol, nnz, _ptr(row_ptr), _ptr(cols), _ptr(vals), _space(cols), int(borrow))
        return cls(h, (row_ptr, cols, vals) if borrow else ())

    @classmethod
    def from_coo(cls, nrow, ncol, rows, cols, vals=None):
        nnz = int(cols.numel() if hasattr(cols, "numel") else cols.size)
        return cls(lib().fs_coo_create(nrow, ncol, nnz, _ptr(rows), _ptr(cols), _ptr(vals), _space(cols)))

    nrow = property(lambda s: lib().fs_matrix_nrow(s.h))
    ncol = property(lambda s: lib().fs_matrix_ncol(s.h))
    nnz = property(lambda s: lib().fs_matrix_nnz(s.h))

    def algorithmic_bytes(self, k=1):
        return lib().fs_matrix_algorithmic_bytes(self.h, k)

    def kernel_name(self, transposed=False):
        """the SpMV kernel the format builder chose for this matrix (under the current options)"""
        code = lib().fs_matrix_spmv_kernel(self.h, int(transposed))
        return {1: "stream", 2: "vector", 6: "tiled", 7: "two-pass", 8: "lds-staged"}.get(code, str(code))

    def candidate_ms(self, transposed=False):
        """ms per product the format builder measured for every candidate kernel (0 = not built)"""
        out = (C.c_float * 4)()
        check(lib().fs_matrix_candidate_ms(self.h, int(transposed), out), "fs_matrix_candidate_ms")
        return dict(zip(("stream", "tiled", "lds-staged", "two-pass"), (round(float(v), 4) for v in out)))

    BUILD_PHASES = ("upload_and_validate", "ordering", "chunk_schedule", "two_pass_copy", "l2_tiled_copy", "lds_staged_copy",
                    "candidates_timed", "losers_freed")

    def build_ms(self, transposed=False):
        """where the one-time work of this matrix went, ms by phase (fs_matrix_build_ms)"""
        out = (C.c_float * 8)()
        check(lib().fs_matrix_build_ms(self.h, int(transposed), out), "fs_matrix_build_ms")
        return dict(zip(self.BUILD_PHASES, (round(float(v), 3) for v in out)))

    def release_csr(self):
        """give back the plain CSR arrays once a re-ordered copy is kept (fs_matrix_release_csr); returns the sides released"""
        n = lib().fs_matrix_release_csr(self.h)
        if n < 0:
            check(n, "fs_matrix_release_csr")
        if n:
            self._keep = ()
        return n

    def restore_csr(self, row_ptr, cols, vals=None, transposed=False, borrow=False):
        check(lib().fs_matrix_restore_csr(self.h, int(transposed), _ptr(row_ptr), _ptr(cols), _ptr(vals), _space(cols), int(borrow)),
              "fs_matrix_restore_csr")
        if borrow:
            self._keep = tuple(self._keep) + (row_ptr, cols, vals)

    def release_prepared(self, k=0):
        n = lib().fs_matrix_release_prepared(self.h, int(k))
        if n < 0:
            check(n, "fs_matrix_release_prepared")
        return n

    SPMM_PLANS = {0: "spmv", 1: "row", 2: "k-column two-pass", 3: "two-pass per column", 4: "mfma", 5: "lds-staged per column",
                  6: "lds-staged strided", 7: "tiled strided"}

    def prepare(self, k, stream=None, transposed=False):
        """one-time work for k-column products (fs_matrix_prepare): fs_spmm itself never builds or waits"""
        check(lib().fs_matrix_prepare(self.h, int(k), int(transposed), stream), "fs_matrix_prepare")

    def spmm_plan(self, k, transposed=False):
        """name of the kernel fs_spmm runs for this k right now"""
        code = lib().fs_matrix_spmm_plan(self.h, int(k), int(transposed))
        return self.SPMM_PLANS.get(code, str(code))

    def device_bytes(self):
        """HBM held by the handle: (CSR + schedule, kept single-vector copy, k-column copies + scratch)"""
        out = (C.c_int64 * 3)()
        check(lib().fs_matrix_device_bytes(self.h, out), "fs_matrix_device_bytes")
        return tuple(int(v) for v in out)

    def build_transpose(self, stream=None):
        check(lib().fs_matrix_build_transpose(self.h, stream), "fs_matrix_build_transpose")

    def download(self, transposed=False):
        import numpy as np
        nrow, ncol, nnz = (self.ncol, self.nrow, self.nnz) if transposed else (self.nrow, self.ncol, self.nnz)
        rp = np.empty(nrow + 1, np.int32)
        cc = np.empty(nnz, np.int32)
        vv = np.full(nnz, np.nan)
        check(lib().fs_matrix_download(self.h, int(transposed), rp.ctypes.data, cc.ctypes.data, vv.ctypes.data))
        return rp, cc, vv

    def spmv(self, y, x, stream=None, transposed=False):
        f = lib().fs_spmv_t if transposed else lib().fs_spmv
        check(f(self.h, _ptr(y), _ptr(x), stream), "fs_spmv")

    def part_rows(self, nparts, transposed=False, k=1):
        """row cuts of the product (k columns) in `nparts` parts: rows [r[p], r[p+1]) are final after part p"""
        out = (C.c_int * (nparts + 1))()
        check(lib().fs_spmm_part_rows(self.h, int(transposed), k, nparts, out), "fs_spmm_part_rows")
        return [int(v) for v in out]

    def spmv_part(self, y, x, part, nparts, stream=None, transposed=False):
        check(lib().fs_spmv_part(self.h, int(transposed), _ptr(y), _ptr(x), part, nparts, stream), "fs_spmv_part")

    def spmm_part(self, Y, X, k, part, nparts, stream=None, transposed=False):
        check(lib().fs_spmm_part(self.h, int(transposed), _ptr(Y), _ptr(X), k, part, nparts, stream), "fs_spmm_part")

    def spmv_host(self, y, x, transposed=False):
        """y, x: contiguous float64 numpy arrays in host memory (fs_spmv_host: copies overlapped with the kernels)"""
        f = lib().fs_spmv_t_host if transposed else lib().fs_spmv_host
        check(f(self.h, y.ctypes.data, x.ctypes.data), "fs_spmv_host")

    def spmm(self, Y, X, k, stream=None, transposed=False):
        f = lib().fs_spmm_t if transposed else lib().fs_spmm
        check(f(self.h, _ptr(Y), _ptr(X), k, stream), "fs_spmm")

    def ata(self, y, x, tmp, stream=None):
        check(lib().fs_ata_mul(self.h, _ptr(y), _ptr(x), _ptr(tmp), stream), "fs_ata_mul")

    def close(self):
        if self.h:
            lib().fs_matrix_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ColBlockMatrix:
    """Device-resident column-blocked binary CSR (cbcsr.h)."""

    def __init__(self, nrow, ncol, nblocks, colblocksize, row_ptr, cols):
        self.h = lib().fs_cbcsr_create(nrow, ncol, nblocks, colblocksize, _ptr(row_ptr), _ptr(cols), _space(cols))
        if not self.h:
            raise FastsparseError("fs_cbcsr_create failed: " + lib().fs_last_error().decode())

    def spmv(self, y, x, stream=None):
        check(lib().fs_cbcsr_spmv(self.h, _ptr(y), _ptr(x), stream), "fs_cbcsr_spmv")

    def close(self):
        if self.h:
            lib().fs_cbcsr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_option_epoch = 0


def set_option(name, value):
    """process-wide option of the library (include/fastsparse_hip.h).  Every call moves the option epoch on: plans that depend on
    which kernel a product runs (the row cuts of products in parts, dist.ShardedOperator) are re-made when it has moved."""
    global _option_epoch
    check(lib().fs_set_option(name.encode(), int(value)), "fs_set_option")
    _option_epoch += 1


def option_epoch():
    return _option_epoch


def synth_uniform(nrow, ncol, per_row, seed, row_offset=0, valued=True, device="cuda"):
    """Uniform synthetic CSR generated on the device (BASELINE configs 2-4)."""
    import torch
    nnz = nrow * per_row
    row_ptr = torch.empty(nrow + 1, dtype=torch.int32, device=device)
    cols = torch.empty(max(nnz, 1), dtype=torch.int32, device=device)[:nnz]
    vals = torch.empty(max(nnz, 1), dtype=torch.float64, device=device)[:nnz] if valued else None
    check(lib().fs_synth_uniform(nrow, ncol, per_row, seed, row_offset, _ptr(row_ptr), _ptr(cols), _ptr(vals),
                                 current_stream()), "fs_synth_uniform")
    return row_ptr, cols, vals


def synth_powerlaw(nrow, ncol, scale, max_len, seed, row_offset=0, valued=True, device="cuda"):
    """Power-law row lengths, uniform columns (BASELINE config 5 shards)."""
    import torch
    lens = torch.empty(nrow, dtype=torch.int32, device=device)
    check(lib().fs_synth_powerlaw_lengths(nrow, float(scale), max_len, seed, row_offset, _ptr(lens), current_stream()))
    row_ptr = torch.zeros(nrow + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=row_ptr[1:])
    nnz = int(row_ptr[-1].item())
    if nnz > 2**31 - 1:
        raise FastsparseError("shard holds more than 2^31-1 non-zeros: int row_ptr cannot index it")
    row_ptr = row_ptr.to(torch.int32)
    cols = torch.empty(nnz, dtype=torch.int32, device=device)
    vals = torch.empty(nnz, dtype=torch.float64, device=device) if valued else None
    check(lib().fs_synth_fill(nrow, ncol, seed, row_offset, _ptr(row_ptr), _ptr(cols), _ptr(vals), current_stream()))
    return row_ptr, cols, vals
