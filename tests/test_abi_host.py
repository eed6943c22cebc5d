"""CPU-side checks of the product: the C-ABI library loads, exports every symbol the headers in
include/ declare, keeps the reference's struct layouts, builds formats on the host exactly as
the reference does, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import _cases
import _hipbackend as H
import _synth as S
from libfastsparse_amd import capi
from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def _declared_functions(header):
    txt = open(os.path.join(INC, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", txt)
    return [n for n in names if n not in ("defined",)]


@pytest.mark.parametrize("header", ["fastsparse_hip.h", "sparse.h", "dsparse.h", "csr.h", "cbcsr.h", "cg.h", "linalg.h", "hilbert.h",
                                    "quickSort.h", "quickSortD.h", "utils.h", "timing.h", "omp_util.h"])
def test_library_exports_every_declared_symbol(header):
    L = capi.lib()
    names = _declared_functions(header)
    assert len(names) >= 1
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_lists_match_headers():
    declared = set()
    for h in ("sparse.h", "dsparse.h", "csr.h", "cbcsr.h", "cg.h", "linalg.h", "hilbert.h", "quickSort.h", "quickSortD.h", "utils.h", "timing.h",
              "omp_util.h"):
        declared |= set(_declared_functions(h))
    assert declared == set(capi.REFERENCE_API)
    assert set(_declared_functions("fastsparse_hip.h")) == set(capi.DEVICE_API)


def test_struct_layouts_match_reference(tmp_path):
    """sizes/offsets probed from the reference headers in SURVEY.md 8(b)"""
    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stddef.h>
#include <stdio.h>
#include "sparse.h"
#include "dsparse.h"
#include "csr.h"
#include "cbcsr.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(struct SparseBinaryMatrix), sizeof(struct SparseDoubleMatrix),
         sizeof(struct BinaryCSR), sizeof(struct CSR), sizeof(struct ColBinaryCSR), sizeof(struct BlockedSBM),
         sizeof(struct BlockedSDM));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(struct CSR, nrow), offsetof(struct CSR, ncol), offsetof(struct CSR, nnz),
         offsetof(struct CSR, row_ptr), offsetof(struct CSR, cols), offsetof(struct CSR, vals));
  printf("%zu %zu %zu\n", offsetof(struct ColBinaryCSR, nnz), offsetof(struct ColBinaryCSR, row_ptr),
         offsetof(struct ColBinaryCSR, cols));
  printf("%zu %zu %zu %zu %zu\n", offsetof(struct BlockedSBM, nblocks), offsetof(struct BlockedSBM, start_row),
         offsetof(struct BlockedSBM, nnz), offsetof(struct BlockedSBM, rows), offsetof(struct BlockedSBM, cols));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=gnu99", "-I" + INC, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    assert out[0].split() == ["32", "40", "32", "40", "40", "48", "56"]
    assert out[1].split() == ["0", "4", "8", "16", "24", "32"]
    assert out[2].split() == ["16", "24", "32"]
    assert out[3].split() == ["8", "16", "24", "32", "40"]
    for cls, size in [(H.SBM, 32), (H.SDM, 40), (H.BCSR, 32), (H.CSR, 40), (H.CBCSR, 40), (H.BSBM, 48), (H.BSDM, 56)]:
        assert C.sizeof(cls) == size


@pytest.mark.parametrize("seed", [1, 2])
def test_host_constructors_match_oracle(seed):
    F = H.HostFormats()
    nrow, ncol = 900, 333
    r, c, v = S.synth_coo(seed, nrow, ncol, 6, empty_frac=0.2, dup_frac=0.1)
    nnz = len(r)
    A = F.csr(nrow, ncol, r, c, v)
    rp, cc, vv = O.coo_to_csr(nrow, r, c, v)
    assert (A.nrow, A.ncol, A.nnz) == (nrow, ncol, nnz)
    assert np.array_equal(F.arr(A.row_ptr, nrow + 1, np.int32), rp)
    assert np.array_equal(F.arr(A.cols, nnz, np.int32), cc)
    assert np.array_equal(F.arr(A.vals, nnz, np.float64), vv)
    B = F.bcsr(nrow, ncol, r, c)
    assert np.array_equal(F.arr(B.row_ptr, nrow + 1, np.int32), rp) and np.array_equal(F.arr(B.cols, nnz, np.int32), cc)
    K = F.cbcsr(50, nrow, ncol, r, c)
    nb, rp2, cc2 = O.coo_to_cbcsr(50, nrow, ncol, r, c)
    assert (K.nblocks, K.colblocksize, K.nnz) == (nb, 50, nnz)
    assert np.array_equal(F.arr(K.row_ptr, nb * nrow + 1, np.int32), rp2)
    assert np.array_equal(F.arr(K.cols, nnz, np.int32), cc2)
    sb = F.sdm(nrow, ncol, r, c, v)
    P = F.L.new_bsdm(C.byref(sb), 64).contents
    blk = O.coo_to_blocked(nrow, 64, r, c, v)
    assert P.nblocks == blk["nblocks"]
    assert np.array_equal(F.arr(P.start_row, P.nblocks + 1, np.int32), blk["start_row"])
    assert np.array_equal(F.arr(P.nnz, P.nblocks, np.int32), blk["blk_nnz"])
    for b in range(P.nblocks):
        n, o = blk["blk_nnz"][b], blk["blk_off"][b]
        assert np.array_equal(F.arr(P.rows[b], n, np.int32), blk["rows"][o:o + n])
        assert np.array_equal(F.arr(P.cols[b], n, np.int32), blk["cols"][o:o + n])
        assert np.array_equal(F.arr(P.vals[b], n, np.float64), blk["vals"][o:o + n])


def test_host_loaders_and_containers():
    F = H.HostFormats()
    A = F.L.read_sbm(os.path.join(S.GOLDEN, "sbm-100-50.data").encode()).contents
    nrow, ncol, rows, cols, _ = S.fixture_sbm()
    assert (A.nrow, A.ncol, A.nnz) == (100, 50, 504)
    assert np.array_equal(F.arr(A.rows, 504, np.int32), rows) and np.array_equal(F.arr(A.cols, 504, np.int32), cols)
    D = F.L.read_sdm(os.path.join(S.GOLDEN, "sdm-100-50.data").encode()).contents
    assert (D.nrow, D.ncol, D.nnz) == (100, 50, 470)
    assert np.array_equal(F.arr(D.vals, 470, np.float64), S.fixture_sdm()[4])
    T = F.L.new_transpose(C.byref(A)).contents          # aliases A's arrays, swapped (sparse.h:38-46)
    assert (T.nrow, T.ncol) == (50, 100)
    assert C.addressof(T.rows.contents) == C.addressof(A.cols.contents)
    F.L.transpose(C.byref(A))
    assert (A.nrow, A.ncol) == (50, 100)
    P = F.L.new_bsbm(C.byref(F.sbm(nrow, ncol, rows, cols)), 8).contents
    assert (P.nblocks, P.start_row[0], P.start_row[1], P.start_row[13]) == (13, 0, 8, 100)   # test_sparse.c:293-300


def test_no_cpu_fallback_without_gpu():
    """With no HIP device the device layer must refuse, not compute on the host."""
    L = capi.lib()
    if L.fs_device_count() > 0:
        pytest.skip("a GPU is present")
    rp = np.array([0, 1], np.int32)
    cc = np.array([0], np.int32)
    h = L.fs_csr_create(1, 1, 1, rp.ctypes.data, cc.ctypes.data, None, capi.FS_HOST, 0)
    assert not h
    assert L.fs_last_error()


def test_dropin_product_dies_loudly_without_gpu():
    L = capi.lib()
    if L.fs_device_count() > 0:
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, _hipbackend as H\n"
            "b = H.HipDropinBackend()\n"
            "b.coo_mul(2, 2, np.array([0, 1], np.int32), np.array([1, 0], np.int32), None, np.ones(2))\n"
            "print('COMPUTED')\n") % (ROOT, os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode != 0 and "COMPUTED" not in p.stdout
    assert "libfastsparse_hip" in p.stderr


@pytest.mark.ref
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference headers (build container only)")
def test_objects_built_against_original_headers_link(tmp_path):
    """C99 inline leaves the reference's API calls as undefined symbols; an object compiled
    against the ORIGINAL headers must link against libfastsparse_hip.so unchanged.  Link only: the binary
    is not run and does not leave this container."""
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include "sparse.h"
#include "dsparse.h"
#include "csr.h"
#include "cbcsr.h"
int main(int argc, char** argv) {
  struct SparseBinaryMatrix* A = read_sbm(argv[1]);
  struct SparseDoubleMatrix* D = read_sdm(argv[2]);
  double x[64], y[128];
  A_mul_B(y, A, x); At_mul_B(x, A, y); sdm_A_mul_B(y, D, x); sdm_At_mul_B(x, D, y);
  struct BlockedSBM* B = new_bsbm(A, 8); bsbm_A_mul_B(y, B, x); bsbm_A_mul_B2(y, B, x); bsbm_A_mul_B4(y, B, x); bsbm_A_mul_Bn(y, B, x, 3);
  struct BlockedSDM* E = new_bsdm(D, 8); bsdm_A_mul_B(y, E, x);
  struct BinaryCSR c; bcsr_from_sbm(&c, A);
  bcsr_A_mul_B(y, &c, x); bcsr_A_mul_B2(y, &c, x); bcsr_A_mul_B4(y, &c, x); bcsr_A_mul_B8(y, &c, x); bcsr_A_mul_B8_auto(y, &c, x);
  bcsr_A_mul_Bn(y, &c, x, 3); bcsr_A_mul_B32n(y, &c, x, 3); bcsr_AA_mul_B(x, &c, x); parallel_bcsr_AA_mul_B(x, &c, x, y); free_bcsr(&c);
  struct CSR v; new_csr(&v, D->nnz, D->nrow, D->ncol, D->rows, D->cols, D->vals); csr_A_mul_B(y, &v, x); csr_A_mul_Bn(y, &v, x, 2); free_csr(&v);
  struct ColBinaryCSR k; cbcsr_from_sbm(&k, A, 8); cbcsr_A_mul_B(y, &k, x);
  return 0;
}''')
    obj = tmp_path / "caller.o"
    # -O0 and no -fgnu89-inline: every API call stays an undefined reference, like `make test` of the reference
    subprocess.check_call(["gcc", "-std=gnu99", "-O0", "-w", "-I/root/reference", "-c", str(src), "-o", str(obj)])
    undef = subprocess.check_output(["nm", "-u", str(obj)]).decode()
    for sym in ("A_mul_B", "csr_A_mul_B", "bcsr_A_mul_B8_auto", "cbcsr_A_mul_B", "new_bsbm", "read_sdm"):
        assert sym in undef
    exe = tmp_path / "caller"
    lib_dir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", str(obj), "-o", str(exe), "-L" + lib_dir, "-lfastsparse_hip", "-lm",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)
