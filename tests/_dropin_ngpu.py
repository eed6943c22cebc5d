"""Runs in a child process of tests/test_gpu_parity.py with FASTSPARSE_NGPU=3 FASTSPARSE_DEVICES=0,0,0 in the environment (the
library reads them once): EVERY reference-named product entry point -- A_mul_B, At_mul_B, sdm_*, bsbm_*, bsdm_*, bcsr_*, csr_*,
cbcsr_*, bcsr_AA_mul_B, bsbm_cg / bsbm_cg2 -- through HipDropinBackend on the row-sharded path (three virtual ranks on one GPU),
against the goldens made from the real reference, with the bars of the single-GPU run (tests/test_gpu_parity.py:_check).
Modes: golden (host vectors), resident (x / y in HBM: nothing may be staged through the host)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch  # noqa: E402  (before the library, like the `hip` fixture: one HIP runtime per process, the one torch brings)

assert torch.cuda.is_available()

import _cases  # noqa: E402
import _hipbackend as H  # noqa: E402
import _kats  # noqa: E402
import _synth as S  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from libfastsparse_amd import capi  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

L = capi.lib()
L.fs_debug_dist_products.restype = C.c_long
assert os.environ.get("FASTSPARSE_NGPU") == "3"


def golden():
    if os.environ.get("FS_DIST_THREADS") == "1":       # the same cases with one issuing thread per rank (RankWorkers, fs_dist.hip)
        D = L.fs_dist_create(3, (C.c_int * 3)(0, 0, 0))
        assert D and L.fs_debug_dist_issue_threads(C.c_void_p(D)) == 3
        L.fs_dist_destroy(D)
    be = H.HipDropinBackend()
    for case in T.CASES:
        before = L.fs_debug_dist_products()
        gold = np.load(os.path.join(S.GOLDEN, case.name + ".npz"))
        out = _cases.run_case(be, case)
        assert set(k.replace("/", "|") for k in out) == set(gold.files)
        T._check(out, gold, case.name, exact=False)
        # every output came from sharded products (one or more each; the solvers many)
        assert L.fs_debug_dist_products() - before >= len(out) - 2, (case.name, L.fs_debug_dist_products() - before, len(out))
        print("golden", case.name, len(out), "outputs,", L.fs_debug_dist_products() - before, "sharded products", flush=True)
    _kats.check_kats(be, T.BY_NAME)
    L.fs_release_all()


def resident():
    """x and y in HBM: the result equals the host-vector result of the same entry point bit for bit where the sums are
    order-independent (pattern-only, integer x) and within the row-scaled bound otherwise; y pre-poisoned"""
    import torch
    be = H.HipDropinBackend()
    c = T.BY_NAME["syn_u16_2048"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).reshape(-1)).cuda()
    rp, cc, vv = O.coo_to_csr(c.nrow, c.rows, c.cols, c.vals)
    rows_sorted = np.repeat(np.arange(c.nrow, dtype=np.int32), np.diff(rp))

    def call(name, nout, A, x, *extra):
        y = torch.full((nout,), -1.0, dtype=torch.float64, device="cuda")
        xd = dev(x)
        f = getattr(L, name)
        f.restype = None
        f(C.c_void_p(y.data_ptr()), A if isinstance(A, C._Pointer) else C.byref(A), C.c_void_p(xd.data_ptr()), *extra)
        return y.cpu().numpy()

    xi, xs = c.xs["int"], c.xs["bench"]
    ui, us = c.xt("int"), c.xt("bench")
    # pattern-only, integer vectors: exact
    assert np.array_equal(call("A_mul_B", c.nrow, be.sbm(c.nrow, c.ncol, c.rows, c.cols), xi), O.csr_mul(c.nrow, rp, cc, None, xi))
    assert np.array_equal(call("At_mul_B", c.ncol, be.sbm(c.nrow, c.ncol, c.rows, c.cols), ui), O.coo_tmul(c.ncol, c.rows, c.cols, None, ui))
    B = be.bcsr(c.nrow, c.ncol, c.rows, c.cols)
    assert np.array_equal(call("bcsr_A_mul_B", c.nrow, B, xi), O.csr_mul(c.nrow, rp, cc, None, xi))
    assert np.array_equal(call("bcsr_At_mul_B", c.ncol, B, ui), O.coo_tmul(c.ncol, rows_sorted, cc, None, ui))
    assert np.array_equal(call("bcsr_AA_mul_B", c.ncol, B, xi), O.bcsr_aa_mul(c.nrow, c.ncol, rp, cc, xi))
    for k, name in ((2, "bcsr_A_mul_B2"), (8, "bcsr_A_mul_B8")):
        X = np.round(S.X_sin(c.ncol, k) * 100)
        assert np.array_equal(call(name, c.nrow * k, B, X).reshape(c.nrow, k), O.csr_mul_n(c.nrow, rp, cc, None, X, k))
    s = be.sbm(c.nrow, c.ncol, c.rows, c.cols)
    Bl = L.new_bsbm(C.byref(s), 64)
    assert np.array_equal(call("bsbm_A_mul_B", c.nrow, Bl, xi), O.csr_mul(c.nrow, rp, cc, None, xi))
    X = np.round(S.X_sin(c.ncol, 2) * 100)
    assert np.array_equal(call("bsbm_A_mul_B2", c.nrow * 2, Bl, X).reshape(c.nrow, 2), O.csr_mul_n(c.nrow, rp, cc, None, X, 2))
    K = be.cbcsr(512, c.nrow, c.ncol, c.rows, c.cols)
    assert np.array_equal(call("cbcsr_A_mul_B", c.nrow, K, xi), O.csr_mul(c.nrow, rp, cc, None, xi))
    # valued: row-scaled 1e-12
    A = be.csr(c.nrow, c.ncol, c.rows, c.cols, c.vals)
    y = call("csr_A_mul_B", c.nrow, A, xs)
    assert np.all(np.abs(y - O.csr_mul(c.nrow, rp, cc, vv, xs)) <= 1e-12 * O.csr_abs_scale(c.nrow, rp, cc, vv, xs))
    z = call("csr_At_mul_B", c.ncol, A, us)
    zr = O.coo_tmul(c.ncol, rows_sorted, cc, vv, us)
    assert np.all(np.abs(z - zr) <= 1e-12 * O.coo_tmul(c.ncol, rows_sorted, cc, np.abs(vv), np.abs(us)))
    X = S.X_sin(c.ncol, 4)
    Y = call("csr_A_mul_Bn", c.nrow * 4, A, X, C.c_int(4)).reshape(c.nrow, 4)
    assert np.all(np.abs(Y - O.csr_mul_n(c.nrow, rp, cc, vv, X, 4)) <= 1e-12 * O.csr_mul_n(c.nrow, rp, cc, np.abs(vv), np.abs(X), 4))
    sd = be.sdm(c.nrow, c.ncol, c.rows, c.cols, c.vals)
    y = call("sdm_A_mul_B", c.nrow, sd, xs)
    assert np.all(np.abs(y - O.csr_mul(c.nrow, rp, cc, vv, xs)) <= 1e-12 * O.csr_abs_scale(c.nrow, rp, cc, vv, xs))
    # the solver with b and x in HBM: the same iterates as with host vectors (fixed-order products: bit-identical)
    b1, _ = _cases.cg_rhs(c.ncol)
    st = be.sbm(c.ncol, c.nrow, c.cols, c.rows)
    Blt = L.new_bsbm(C.byref(st), 64)
    it_h, it_d = C.c_int(-1), C.c_int(-1)
    xh = np.full(c.ncol, -1.0)
    L.bsbm_cg.restype = None
    L.bsbm_cg(H._dp(xh), Bl, Blt, H._dp(b1.copy()), C.c_double(5.0), C.c_double(1e-6), C.byref(it_h))
    xd, bd = torch.full((c.ncol,), -1.0, dtype=torch.float64, device="cuda"), dev(b1)
    L.bsbm_cg(C.c_void_p(xd.data_ptr()), Bl, Blt, C.c_void_p(bd.data_ptr()), C.c_double(5.0), C.c_double(1e-6), C.byref(it_d))
    assert it_h.value == it_d.value and np.array_equal(xd.cpu().numpy(), xh), (it_h.value, it_d.value)
    # nothing was staged through the host: pageable or pinned host traffic would show as DtoH / HtoD copies; the layer has no
    # counter for that, so the check is structural -- fs_dist_spmv writes the caller's HBM vector from the unpack launch
    L.fs_release_all()


def edges():
    """shapes the shard cutter must survive: no entries at all, no rows, fewer rows than ranks, one column, one long row (every
    entry in one shard, the other ranks idle), and the same struct multiplied again after its arrays were edited (the side table's
    fingerprint holds for sharded copies too)"""
    be = H.HipDropinBackend()
    i32 = np.int32

    def mul(name, nout, A, x, *extra):
        y = np.full(max(nout, 1), -1.0)
        f = getattr(L, name)
        f.restype = None
        f(H._dp(y), A if isinstance(A, C._Pointer) else C.byref(A), H._dp(np.ascontiguousarray(x, dtype=np.float64)), *extra)
        return y[:nout]

    # nnz == 0: y all zero (overwritten), every format
    e = np.zeros(0, i32)
    assert np.array_equal(mul("A_mul_B", 5, be.sbm(5, 3, e, e), np.ones(3)), np.zeros(5))
    assert np.array_equal(mul("At_mul_B", 3, be.sbm(5, 3, e, e), np.ones(5)), np.zeros(3))
    assert np.array_equal(mul("bcsr_A_mul_B", 5, be.bcsr(5, 3, e, e), np.ones(3)), np.zeros(5))
    assert np.array_equal(mul("csr_A_mul_B", 5, be.csr(5, 3, e, e, np.zeros(0)), np.ones(3)), np.zeros(5))
    assert np.array_equal(mul("bcsr_A_mul_B2", 10, be.bcsr(5, 3, e, e), np.ones(6)), np.zeros(10))
    # nrow == 0 (blocked: no blocks), fewer rows than ranks, one column
    s0 = be.sbm(0, 7, e, e)
    B0 = L.new_bsbm(C.byref(s0), 8)
    assert B0.contents.nblocks == 0
    y = mul("bsbm_A_mul_B", 0, B0, np.ones(7))
    rows, cols = np.array([1, 0, 1], i32), np.array([0, 0, 0], i32)
    assert np.array_equal(mul("A_mul_B", 2, be.sbm(2, 1, rows, cols), np.array([3.0])), np.array([3.0, 6.0]))
    assert np.array_equal(mul("At_mul_B", 1, be.sbm(2, 1, rows, cols), np.array([2.0, 5.0])), np.array([12.0]))
    vals = np.array([0.5, 2.0, -1.0])
    assert np.array_equal(mul("sdm_A_mul_B", 2, be.sdm(2, 1, rows, cols, vals), np.array([4.0])), np.array([8.0, 2.0 - 4.0]))
    # one long row among empty ones: all entries in one shard
    n = 30_000
    rows = np.full(n, 4, i32)
    cols = (np.arange(n) % 977).astype(i32)
    x = S.x_int(3, 977)
    ref = np.zeros(9); ref[4] = x[cols].sum()
    assert np.array_equal(mul("A_mul_B", 9, be.sbm(9, 977, rows, cols), x), ref)
    zt = np.zeros(977); np.add.at(zt, cols, 2.0)
    u = np.zeros(9); u[4] = 2.0
    assert np.array_equal(mul("At_mul_B", 977, be.sbm(9, 977, rows, cols), u), zt)
    # the same struct again after an in-place edit of its arrays: the fingerprint notices, the sharded copy is rebuilt
    c = T.BY_NAME["syn_dup_1024"]
    A = be.bcsr(c.nrow, c.ncol, c.rows, c.cols)
    xi = c.xs["int"]
    L.bcsr_A_mul_B.restype = None
    y1 = np.full(c.nrow, -1.0)
    L.bcsr_A_mul_B(H._dp(y1), C.byref(A), H._dp(xi))
    rp, cc, _ = O.coo_to_csr(c.nrow, c.rows, c.cols, None)
    assert np.array_equal(y1, O.csr_mul(c.nrow, rp, cc, None, xi))
    cols_view = np.ctypeslib.as_array(A.cols, shape=(len(cc),))
    cols_view[:] = (cols_view + 1) % c.ncol
    L.bcsr_A_mul_B(H._dp(y1), C.byref(A), H._dp(xi))
    assert np.array_equal(y1, O.csr_mul(c.nrow, rp, ((cc + 1) % c.ncol).astype(i32), None, xi))
    assert L.fs_debug_dist_products() >= 10
    L.fs_release_all()


def fullsize():
    """BASELINE config 2's matrix at full size (10 M x 10 M, 16 per row, 160 M entries; valued and as a pattern) from HOST structs
    through csr_A_mul_B / csr_At_mul_B / bcsr_A_mul_B / bsbm-free entry points on three virtual ranks, vectors in HBM.  Size-independent
    checks: pattern + integer x: an exact checksum of checksums (sum of y = sum over the entries of x[col]) and
    oracle windows bit for bit; valued + sin x: oracle windows within the row-scaled 1e-12; the transposed product against
    adjointness <A x, u> = <x, A' u> in exact integer arithmetic."""
    import time
    from oracle import pysynth
    n, per = 10_000_000, 16
    rp, cc, vv = pysynth.uniform(n, n, per, 0x5EED0002)
    dev = lambda a: torch.from_numpy(a).cuda()
    out = lambda m: torch.full((m,), -1.0, dtype=torch.float64, device="cuda")

    def call(name, y, A, x):
        f = getattr(L, name)
        f.restype = None
        t0 = time.time()
        f(C.c_void_p(y.data_ptr()), C.byref(A), C.c_void_p(x.data_ptr()))
        return time.time() - t0

    def window(y, vals, x, lo, hi, exact):
        a, b = int(rp[lo]), int(rp[hi])
        lrp = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
        ref = O.csr_mul(hi - lo, lrp, cc[a:b], None if vals is None else vals[a:b], x)
        got = y[lo:hi].cpu().numpy()
        if exact:
            assert np.array_equal(got, ref), lo
        else:
            sc = O.csr_abs_scale(hi - lo, lrp, cc[a:b], vals[a:b], x)
            assert np.all(np.abs(got - ref) <= 1e-12 * sc), lo

    g = torch.Generator(device="cuda"); g.manual_seed(11)
    xi = torch.randint(-1000, 1001, (n,), device="cuda", generator=g).to(torch.float64)
    ui = torch.randint(-1000, 1001, (n,), device="cuda", generator=g).to(torch.float64)
    xi_h = xi.cpu().numpy()
    # pattern-only, integer vectors
    B = H.BCSR(n, n, n * per, H._ip(rp), H._ip(cc))
    y, z = out(n), out(n)
    before = L.fs_debug_dist_products()
    t_first = call("bcsr_A_mul_B", y, B, xi)
    t_next = call("bcsr_A_mul_B", y, B, xi)
    ccd = dev(cc)
    tot = 0
    xl = xi.to(torch.int64)
    for a in range(0, n * per, 40_000_000):
        tot += int(xl[ccd[a:a + 40_000_000].long()].sum().item())
    assert int(y.to(torch.int64).sum().item()) == tot
    for lo in (0, 3_333_333, 6_700_000, n - 2500):
        window(y, None, xi_h, lo, lo + 2500, exact=True)
    call("bcsr_At_mul_B", z, B, ui)
    # adjointness in exact integers: <A x, u> == <x, A' u>
    assert int((y.to(torch.int64) * ui.to(torch.int64)).sum().item()) == int((xi.to(torch.int64) * z.to(torch.int64)).sum().item())
    L.fs_invalidate(C.byref(B))
    # valued, sin x
    A = H.CSR(n, n, n * per, H._ip(rp), H._ip(cc), H._dp(vv))
    xs = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
    xs_h = xs.cpu().numpy()
    t_v = call("csr_A_mul_B", y, A, xs)
    for lo in (0, 5_000_000, n - 2500):
        window(y, vv, xs_h, lo, lo + 2500, exact=False)
    assert L.fs_debug_dist_products() - before >= 4
    print("config 2 across three virtual ranks from host structs: first bcsr_A_mul_B %.2f s (cut, upload, format on three shards), next %.4f s; "
          "first csr_A_mul_B %.2f s" % (t_first, t_next, t_v), flush=True)
    L.fs_release_all()


{"golden": golden, "resident": resident, "edges": edges, "fullsize": fullsize}[sys.argv[1].split("+")[0]]()
print("OK")
