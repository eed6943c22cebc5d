"""Oracle vs the real reference, live (build container only: needs oracle/_ref).

Beyond the committed goldens this sweeps larger and odder shapes, and checks the
format constructors array for array."""
import ctypes as C

import numpy as np
import pytest

import _cases
import _refbind
import _synth as S
from oracle import pyoracle as O

pytestmark = [pytest.mark.ref,
              pytest.mark.skipif(not _refbind.available(), reason="oracle/_ref not built (no /root/reference here)")]


def _eq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.int64), b.view(np.int64))


@pytest.mark.parametrize("case", _cases.all_cases(), ids=lambda c: c.name)
def test_all_entry_points_bitwise(case):
    ref = _cases.run_case(_cases.RefBackend(), case)
    ora = _cases.run_case(_cases.OracleBackend(), case)
    assert ref.keys() == ora.keys()
    for k in ref:
        assert _eq(ref[k], ora[k]), (case.name, k)


@pytest.mark.parametrize("shape", [(100000, 50000, 16), (50000, 100000, 5), (3, 70000, 9000), (65537, 17, 3)])
def test_larger_shapes_bitwise(shape):
    nrow, ncol, per = shape
    r, c, v = S.synth_coo(99 + nrow, nrow, ncol, per, empty_frac=0.05, dup_frac=0.05)
    case = _cases.Case("big", nrow, ncol, r, c, v, {"bench": S.x_sin(ncol), "int": S.x_int(3, ncol)},
                       block_sizes=(1024,), colblocks=(4096,), kmax=8)
    ref = _cases.run_case(_cases.RefBackend(), case)
    ora = _cases.run_case(_cases.OracleBackend(), case)
    for k in ref:
        assert _eq(ref[k], ora[k]), k


def test_constructors_match_reference_arrays():
    R = _refbind.Ref()
    nrow, ncol = 700, 300
    r, c, v = S.synth_coo(5, nrow, ncol, 9, empty_frac=0.2, dup_frac=0.1)
    nnz = len(r)
    A = R.csr(nrow, ncol, r, c, v)
    rp, cc, vv = O.coo_to_csr(nrow, r, c, v)
    assert np.array_equal(R.arr(A.row_ptr, nrow + 1, np.int32), rp)
    assert np.array_equal(R.arr(A.cols, nnz, np.int32), cc)
    assert np.array_equal(R.arr(A.vals, nnz, np.float64), vv)
    B = R.cbcsr(64, nrow, ncol, r, c)
    nb, rp2, cc2 = O.coo_to_cbcsr(64, nrow, ncol, r, c)
    assert B.nblocks == nb and B.nnz == nnz
    assert np.array_equal(R.arr(B.row_ptr, nb * nrow + 1, np.int32), rp2)
    assert np.array_equal(R.arr(B.cols, nnz, np.int32), cc2)
    sb = R.sbm(nrow, ncol, r, c)
    P = R.lib.new_bsbm(C.byref(sb), 48).contents
    blk = O.coo_to_blocked(nrow, 48, r, c, None)
    assert P.nblocks == blk["nblocks"]
    assert np.array_equal(R.arr(P.start_row, P.nblocks + 1, np.int32), blk["start_row"])
    assert np.array_equal(R.arr(P.nnz, P.nblocks, np.int32), blk["blk_nnz"])
    for b in range(P.nblocks):
        n = blk["blk_nnz"][b]
        o = blk["blk_off"][b]
        assert np.array_equal(R.arr(P.rows[b], n, np.int32), blk["rows"][o:o + n])
        assert np.array_equal(R.arr(P.cols[b], n, np.int32), blk["cols"][o:o + n])


def test_fixture_loader_matches_read_sbm_read_sdm():
    import os
    R = _refbind.Ref()
    A = R.lib.read_sbm(os.path.join(S.GOLDEN, "sbm-100-50.data").encode()).contents
    nrow, ncol, rows, cols, _ = S.fixture_sbm()
    assert (A.nrow, A.ncol, A.nnz) == (nrow, ncol, len(rows)) == (100, 50, 504)
    assert np.array_equal(R.arr(A.rows, A.nnz, np.int32), rows) and np.array_equal(R.arr(A.cols, A.nnz, np.int32), cols)
    D = R.lib.read_sdm(os.path.join(S.GOLDEN, "sdm-100-50.data").encode()).contents
    nrow, ncol, rows, cols, vals = S.fixture_sdm()
    assert (D.nrow, D.ncol, D.nnz) == (100, 50, 470)
    assert np.array_equal(R.arr(D.vals, D.nnz, np.float64), vals)


def test_bcsr_file_format_interoperates_with_reference(tmp_path):
    """csr.h:97-146: a BinaryCSR written by the reference loads through the product's deserialize_from_file
    and the other way round; same bytes on disk (up to the two meaningless pointers inside the struct dump)"""
    import _hipbackend as H
    R = _refbind.Ref()
    F = H.HostFormats()
    nrow, ncol, rows, cols, _ = S.fixture_sbm()
    A_ref = R.bcsr(nrow, ncol, rows, cols)
    A_own = F.bcsr(nrow, ncol, rows.copy(), cols.copy())
    f1, f2 = str(tmp_path / "ref.csr.bin").encode(), str(tmp_path / "own.csr.bin").encode()
    R.lib.ref_serialize_to_file(C.byref(A_ref), f1)
    F.L.serialize_to_file(C.byref(A_own), f2)
    b1, b2 = open(f1, "rb").read(), open(f2, "rb").read()
    off = len(b"BINARY_CSR: struct BinaryCSR, int[nrow], int[nnz]\nstruct BinaryCSR\n")
    assert len(b1) == len(b2) and b1[:off + 16] == b2[:off + 16] and b1[off + 32:] == b2[off + 32:]
    B1, B2 = H.BCSR(), _refbind.BCSR()
    F.L.deserialize_from_file(C.byref(B1), f1)          # product reads the reference's file
    R.lib.ref_deserialize_from_file(C.byref(B2), f2)    # reference reads the product's file
    rp, cc, _ = O.coo_to_csr(nrow, rows, cols)
    for B, arr in ((B1, F.arr), (B2, R.arr)):
        assert (B.nrow, B.ncol, B.nnz) == (nrow, ncol, len(rows))
        assert np.array_equal(arr(B.row_ptr, nrow + 1, np.int32), rp) and np.array_equal(arr(B.cols, len(rows), np.int32), cc)
