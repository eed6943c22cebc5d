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


def test_locality_sorters_match_reference():
    """8f-3: sort_sbm / sort_sdm (Hilbert order), sort_bsbm / sort_bsdm (per-block Hilbert order), sort_bsbm_byrow and
    the curve helpers give the entry order the reference gives (matrices without repeated (row, col) pairs: array
    for array)"""
    import _hipbackend as H
    R = _refbind.Ref()
    F = H.HostFormats()
    L, RL = F.L, R.lib
    for f in ("xy2d", "row_xy2d"):
        getattr(L, f).restype = C.c_long
        getattr(RL, f).restype = C.c_long
    for n, x, y in ((8, 3, 5), (1024, 1000, 17), (128, 0, 127), (64, 63, 63)):
        assert L.xy2d(n, x, y) == RL.xy2d(n, x, y)
        assert L.row_xy2d(n, x % n, y * 7) == RL.row_xy2d(n, x % n, y * 7)
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.d2xy(n, C.c_long(L.xy2d(n, x, y)), C.byref(a), C.byref(b))
        assert (a.value, b.value) == (x, y)
        L.row_d2xy(n, C.c_long(L.row_xy2d(n, x % n, y * 7)), C.byref(c), C.byref(d))
        assert (c.value, d.value) == (x % n, y * 7)
    assert [L.ceilPower2(v) for v in (1, 2, 3, 100, 1024, 1025)] == [RL.ceilPower2(v) for v in (1, 2, 3, 100, 1024, 1025)]

    nrow, ncol, rows, cols, vals = S.fixture_sdm()       # distinct (row, col) pairs
    # whole-matrix Hilbert order
    a, b = F.sbm(nrow, ncol, rows, cols), R.sbm(nrow, ncol, rows, cols)
    L.sort_sbm(C.byref(a)); RL.ref_sort_sbm(C.byref(b))
    assert np.array_equal(F.arr(a.rows, len(rows), np.int32), R.arr(b.rows, len(rows), np.int32))
    assert np.array_equal(F.arr(a.cols, len(rows), np.int32), R.arr(b.cols, len(rows), np.int32))
    d1, d2 = F.sdm(nrow, ncol, rows, cols, vals), R.sdm(nrow, ncol, rows, cols, vals)
    L.sort_sdm(C.byref(d1)); RL.sort_sdm(C.byref(d2))
    for fld, dt in (("rows", np.int32), ("cols", np.int32), ("vals", np.float64)):
        assert np.array_equal(F.arr(getattr(d1, fld), len(rows), dt), R.arr(getattr(d2, fld), len(rows), dt)), fld
    # per-block orders
    for own_sort, ref_sort in ((L.sort_bsbm, RL.ref_sort_bsbm), (L.sort_bsbm_byrow, RL.ref_sort_bsbm_byrow)):
        B1 = L.new_bsbm(C.byref(F.sbm(nrow, ncol, rows, cols)), 8)
        B2 = RL.new_bsbm(C.byref(R.sbm(nrow, ncol, rows, cols)), 8)
        own_sort(B1); ref_sort(B2)
        for blk in range(B1.contents.nblocks):
            n = B1.contents.nnz[blk]
            assert np.array_equal(F.arr(B1.contents.rows[blk], n, np.int32), R.arr(B2.contents.rows[blk], n, np.int32))
            assert np.array_equal(F.arr(B1.contents.cols[blk], n, np.int32), R.arr(B2.contents.cols[blk], n, np.int32))
    E1 = L.new_bsdm(C.byref(F.sdm(nrow, ncol, rows, cols, vals)), 8)
    E2 = RL.new_bsdm(C.byref(R.sdm(nrow, ncol, rows, cols, vals)), 8)
    L.sort_bsdm(E1); RL.sort_bsdm(E2)
    for blk in range(E1.contents.nblocks):
        n = E1.contents.nnz[blk]
        assert np.array_equal(F.arr(E1.contents.rows[blk], n, np.int32), R.arr(E2.contents.rows[blk], n, np.int32))
        assert np.array_equal(F.arr(E1.contents.vals[blk], n, np.float64), R.arr(E2.contents.vals[blk], n, np.float64))
    # quickSort / quickSortD
    rng = np.random.default_rng(1)
    k = rng.integers(-10**12, 10**12, 5000).astype(np.int64)
    v = rng.uniform(size=5000)
    k1, v1 = k.copy(), v.copy()
    L.quickSortD(k1.ctypes.data_as(C.POINTER(C.c_long)), C.c_long(0), C.c_long(4999), v1.ctypes.data_as(C.POINTER(C.c_double)))
    o = np.argsort(k, kind="stable")
    assert np.array_equal(k1, k[o]) and np.array_equal(v1, v[o])


def test_samplers_match_reference():
    """exprand / randexp / randsubseq draw from drand48 as the reference's do: same seed, same samples"""
    import ctypes as C
    from libfastsparse_amd import capi
    L = capi.lib()
    R = _refbind.Ref().lib
    libc = C.CDLL(None)
    libc.srand48.argtypes = [C.c_long]
    for lib_ in (L, R):
        lib_.exprand.restype = C.c_double
        lib_.randexp.restype = C.c_double
        lib_.randsubseq.restype = C.c_long
        lib_.randsubseq.argtypes = [C.c_long, C.c_long, C.c_double, C.POINTER(C.c_long)]

    def draw(lib_, seed, N, cap, p):
        libc.srand48(seed)
        buf = (C.c_long * cap)()
        n = lib_.randsubseq(N, cap, p, buf)
        return list(buf[:n]), lib_.exprand(), lib_.randexp()

    for seed, N, cap, p in [(1234567890, 10000, 1000, 0.05), (7, 50, 100, 0.5), (3, 1000, 5, 0.3), (11, 1, 4, 0.9)]:
        got, want = draw(L, seed, N, cap, p), draw(R, seed, N, cap, p)
        assert got == want
        assert all(0 <= v < N for v in got[0]) and len(got[0]) <= cap
