"""Known-answer tests held by the reference's own test file (test_sparse.c),
restated as data.  Each entry: (entry point, case name, x tag, checks) where a
check is (index, expected value, tolerance); tolerance 0 means exact ==."""
import numpy as np

# test_sparse.c:30-44 test_A_mul_B; :59-77 test_bcsr_A_mul_B; :114-135 test_cbcsr
KAT_SBM = [(0, 0.5, 0), (1, 1.9, 0), (2, -0.7, 0), (3, 2.4, 0)]
# test_sparse.c:159-173 test_At_mul_B: x over rows = (0.2, 1.3, -0.7, -0.5)
KAT_SBM_T_X = np.array([0.2, 1.3, -0.7, -0.5])
KAT_SBM_T = [(0, -0.3, 0), (1, -0.7, 0), (2, 0.8, 0)]
# test_sparse.c:46-57 test_A_mul_B2 and :137-157 test_cbcsr_A_mul_B on sbm-100-50, x = t19
KAT_FIX_SBM = [(0, 1.70095, 1e-4), (99, -0.174905, 1e-4)]
# test_sparse.c:412-423 / :425-439: sdm & csr, x = (0.5,-0.7,1.9,2.3)
KAT_SDM_Y = np.array([2.162, 2.224, 0.713, -0.378, 2.216, 0.437])
# test_sparse.c:457-468: x over rows, expected over columns
KAT_SDM_T_X = np.array([0.59, 0.37, 0.14, 0.21, 0.40, 0.81])
KAT_SDM_T_Y = np.array([0.2405, 0.6602, 0.483, 1.2102])
# test_sparse.c:441-455 test_A_mul_Bn_csr: X = [x, 10x] row-major => Y = [y, 10y]
KAT_SDM_X2 = np.array([0.5, 5.0, -0.7, -7.0, 1.9, 19.0, 2.3, 23.0]).reshape(4, 2)
KAT_SDM_Y2 = np.array([2.162, 21.62, 2.224, 22.24, 0.713, 7.13, -0.378, -3.78, 2.216, 22.16, 0.437, 4.37]).reshape(6, 2)
# test_sparse.c:185-193 / :470-482 loaders
KAT_READ_SBM = dict(nrow=100, ncol=50, nnz=504, rows0=8, cols0=0)
KAT_READ_SDM = dict(nrow=100, ncol=50, nnz=470, rows1=27, cols1=0, vals1=0.616153, rows469=40, cols469=49,
                    vals469=0.108172)
# test_sparse.c:293-345 / :484-509 blocked geometry for block size 8 on the 100-row fixtures
KAT_BLOCKS = dict(nblocks=13, start_row0=0, start_row1=8, start_row13=100)


def check_kats(be, cases_by_name):
    """Run every hot-path KAT of test_sparse.c against backend `be`."""
    c = cases_by_name["kat_sbm_4x3"]
    x = c.xs["kat"]
    for name, y in [("A_mul_B", be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, None, x)),
                    ("bcsr_A_mul_B", be.csr_mul(c.nrow, c.ncol, c.rows, c.cols, None, x)),
                    ("cbcsr_A_mul_B", be.cbcsr_mul(c.nrow, c.ncol, c.rows, c.cols, 2, x))]:
        for i, v, _ in KAT_SBM:
            assert y[i] == v, (name, i, y[i], v)
    y = be.coo_tmul(c.nrow, c.ncol, c.rows, c.cols, None, KAT_SBM_T_X)
    for i, v, _ in KAT_SBM_T:
        assert y[i] == v, ("At_mul_B", i, y[i], v)

    c = cases_by_name["fix_sbm_100x50"]
    x = c.xs["t19"]
    ycoo = be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, None, x)
    for name, y in [("A_mul_B", ycoo), ("bcsr_A_mul_B", be.csr_mul(c.nrow, c.ncol, c.rows, c.cols, None, x)),
                    ("cbcsr_A_mul_B", be.cbcsr_mul(c.nrow, c.ncol, c.rows, c.cols, 8, x))]:
        for i, v, tol in KAT_FIX_SBM:
            assert abs(y[i] - v) < tol, (name, i, y[i], v)
        assert np.max(np.abs(y - ycoo)) < 1e-4
    # test_bcsr_AA_mul_B (:79-112): A'A x against At_mul_B(A_mul_B(x))
    y2 = be.coo_tmul(c.nrow, c.ncol, c.rows, c.cols, None, ycoo)
    assert np.max(np.abs(be.aa_mul(c.nrow, c.ncol, c.rows, c.cols, x, False) - y2)) < 1e-4
    assert np.max(np.abs(be.aa_mul(c.nrow, c.ncol, c.rows, c.cols, x, True) - y2)) < 1e-4
    # test_blocked_sbm (:293-345)
    x17 = c.xs["t17"]
    y = be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, None, x17)
    yb = be.blocked_mul(c.nrow, c.ncol, c.rows, c.cols, None, 8, x17, 1, "bsbm_A_mul_B")
    assert np.linalg.norm(yb - y) < 1e-6
    i = np.arange(c.ncol)
    X = np.stack([np.sin(i * 17 + 0.2), np.sin(i * 23 + 0.7)], axis=1).copy()
    yc1 = be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, None, X[:, 1].copy())
    for Y in [be.blocked_mul(c.nrow, c.ncol, c.rows, c.cols, None, 8, X, 2, "bsbm_A_mul_B2"),
              be.csr_mul_n(c.nrow, c.ncol, c.rows, c.cols, None, X, 2, "bcsr_A_mul_B2"),
              be.csr_mul_n(c.nrow, c.ncol, c.rows, c.cols, None, X, 2, "bcsr_A_mul_Bn")]:
        assert np.max(np.abs(Y[:, 0] - y)) < 1e-6 and np.max(np.abs(Y[:, 1] - yc1)) < 1e-6

    # test_cg (:560-608): x[0], x[1] of (A'A + 5 I) x = b, residual, and the same through the 2-RHS solver
    if hasattr(be, "cg"):
        import _cases
        b1, b2 = _cases.cg_rhs(c.ncol)
        xs, _ = be.cg(c.nrow, c.ncol, c.rows, c.cols, b1, 5.0, 1e-6, False)
        assert abs(xs[0] - 0.0638578) < 1e-4 and abs(xs[1] + 0.0302702) < 1e-4
        t = be.coo_tmul(c.nrow, c.ncol, c.rows, c.cols, None, be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, None, xs))
        assert np.linalg.norm(t + 5.0 * xs - b1) < 1e-5
        X2, _ = be.cg(c.nrow, c.ncol, c.rows, c.cols, np.stack([b1, b2], axis=1).copy(), 5.0, 1e-6, True)
        assert abs(X2[0, 0] - 0.0638578) < 1e-4 and abs(X2[1, 0] + 0.0302702) < 1e-4

    c = cases_by_name["kat_sdm_6x4"]
    x = c.xs["kat"]
    assert np.max(np.abs(be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, c.vals, x) - KAT_SDM_Y)) < 1e-6
    assert np.max(np.abs(be.csr_mul(c.nrow, c.ncol, c.rows, c.cols, c.vals, x) - KAT_SDM_Y)) < 1e-6
    assert np.max(np.abs(be.coo_tmul(c.nrow, c.ncol, c.rows, c.cols, c.vals, KAT_SDM_T_X) - KAT_SDM_T_Y)) < 1e-6
    assert np.max(np.abs(be.csr_mul_n(c.nrow, c.ncol, c.rows, c.cols, c.vals, KAT_SDM_X2, 2, "csr_A_mul_Bn")
                         - KAT_SDM_Y2)) < 1e-6

    c = cases_by_name["fix_sdm_100x50"]
    y = be.coo_mul(c.nrow, c.ncol, c.rows, c.cols, c.vals, x17)
    yb = be.blocked_mul(c.nrow, c.ncol, c.rows, c.cols, c.vals, 8, x17, 1, "bsdm_A_mul_B")
    assert np.linalg.norm(yb - y) < 1e-6
