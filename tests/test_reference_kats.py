"""The reference's own known-answer tests for the hot path (test_sparse.c, see
tests/_kats.py for line numbers), run against the oracle on CPU."""
import numpy as np

import _cases
import _kats
import _synth as S
from oracle import pyoracle as O


def test_kats_oracle():
    by_name = {c.name: c for c in _cases.all_cases()}
    _kats.check_kats(_cases.OracleBackend(), by_name)


def test_fixture_loader_kats():
    nrow, ncol, rows, cols, _ = S.fixture_sbm()
    k = _kats.KAT_READ_SBM
    assert (nrow, ncol, len(rows), rows[0], cols[0]) == (k["nrow"], k["ncol"], k["nnz"], k["rows0"], k["cols0"])
    nrow, ncol, rows, cols, vals = S.fixture_sdm()
    k = _kats.KAT_READ_SDM
    assert (nrow, ncol, len(rows)) == (k["nrow"], k["ncol"], k["nnz"])
    assert rows[1] == k["rows1"] and cols[1] == k["cols1"] and abs(vals[1] - k["vals1"]) < 1e-5
    assert rows[469] == k["rows469"] and cols[469] == k["cols469"] and abs(vals[469] - k["vals469"]) < 1e-5


def test_block_geometry_kats():
    nrow, ncol, rows, cols, _ = S.fixture_sbm()
    blk = O.coo_to_blocked(nrow, 8, rows, cols)
    k = _kats.KAT_BLOCKS
    assert blk["nblocks"] == k["nblocks"]
    assert (blk["start_row"][0], blk["start_row"][1], blk["start_row"][13]) == (0, 8, 100)
    nb, rp, cc = O.coo_to_cbcsr(2, 4, 3, np.array([0, 3, 3, 1, 2], np.int32), np.array([0, 2, 0, 2, 1], np.int32))
    assert nb == 2      # test_sparse.c:118


def test_y_is_overwritten_and_empty_rows_are_zero():
    # SURVEY N5; pyoracle pre-poisons every output with -1 (as test_sparse.c:460 does)
    r, c, v = S.synth_coo(11, 300, 100, 4, empty_frac=0.5)
    rp, cc, vv = O.coo_to_csr(300, r, c, v)
    y = O.csr_mul(300, rp, cc, vv, S.x_sin(100))
    empty = np.diff(rp) == 0
    assert empty.any() and np.all(y[empty] == 0.0) and not np.signbit(y[empty]).any()
