"""Size-independent properties at BASELINE.json's full single-GPU sizes (configs 2, 3, 4, and one row shard of
config 5's shape), where the
oracle cannot be run on everything in seconds: exact integer checksums, adjointness, column consistency
of the multi-RHS product, and an oracle check of a row window."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _int_x(n, device, seed):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(-1000, 1001, (n,), device=device, generator=g).to(torch.float64)


def _window_check(capi, O, rp, cc, vv, x, y, lo, hi, exact):
    """rows [lo,hi) recomputed by the oracle from downloaded arrays"""
    rpw = rp[lo:hi + 1].cpu().numpy().astype(np.int64)
    a, b = int(rpw[0]), int(rpw[-1])
    ccw = cc[a:b].cpu().numpy()
    vvw = None if vv is None else vv[a:b].cpu().numpy()
    ref = O.csr_mul(hi - lo, (rpw - a).astype(np.int32), ccw, vvw, x.cpu().numpy())
    got = y[lo:hi].cpu().numpy()
    if exact:
        assert np.array_equal(got, ref)
    else:
        scale = O.csr_abs_scale(hi - lo, (rpw - a).astype(np.int32), ccw, vvw, x.cpu().numpy())
        assert np.all(np.abs(got - ref) <= 1e-12 * np.maximum(scale, 1e-300))


def test_config2_fp64_csr_10m(hip_env):
    """BASELINE config 2: CSR 10M x 10M, 16 nnz/row, A_mul_B and At_mul_B"""
    torch, capi, O = hip_env
    n, per = 10_000_000, 16
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002)
    A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
    A.build_transpose(capi.current_stream())
    st = capi.current_stream()
    y = torch.full((n,), -1.0, dtype=torch.float64, device="cuda")
    # (1) sin x, row-window vs oracle within the row-scaled 1e-12 bound
    x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
    A.spmv(y, x, st)
    for lo in (0, 4_999_000, n - 3000):
        _window_check(capi, O, rp, cc, vv, x, y, lo, lo + 3000, exact=False)
    # (1b) the same product in parts: three generations of panels = three real parts; every part's rows within the bound of
    #      the whole product (valued: the sums are not run-to-run identical), rows of later parts still poisoned
    rows = A.part_rows(4)
    assert rows[0] == 0 and rows[-1] == n and all(a <= b for a, b in zip(rows, rows[1:]))
    if A.kernel_name() == "two-pass":
        assert sum(b > a for a, b in zip(rows, rows[1:])) == 3, rows
    yp = torch.full((n,), -7.0, dtype=torch.float64, device="cuda")
    for part in range(4):
        A.spmv_part(yp, x, part, 4, st)
        assert float((yp[:rows[part + 1]] - y[:rows[part + 1]]).abs().max()) <= 1e-12 * 16.0, part
        assert rows[part + 1] == n or float(yp[rows[part + 1]:].max()) == -7.0
    # (2) linearity: A(2x) == 2 A(x) up to the order of the sums (the two-pass kernels add in arrival order)
    y2 = torch.empty_like(y)
    A.spmv(y2, 2.0 * x, st)
    assert float((y2 - 2.0 * y).abs().max()) <= 2e-12 * 16.0
    # (3) adjoint identity with the transposed product: <u, A x> == <A'u, x> to rounding
    u = torch.cos(3.0 * torch.arange(n, device="cuda", dtype=torch.float64))
    z = torch.empty_like(y)
    A.spmv(z, u, st, transposed=True)
    lhs, rhs = torch.dot(u, y).item(), torch.dot(z, x).item()
    assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(lhs))
    # (4) strict_order: storage-order sums, bit-identical to the oracle on a row window; the default
    #     (band-major) order agrees with it to rounding on every row
    capi.set_option("strict_order", 1)
    try:
        A.spmv(y2, x, st)
    finally:
        capi.set_option("strict_order", 0)
    _window_check(capi, O, rp, cc, vv, x, y2, 2_000_000, 2_003_000, exact=True)
    assert float((y2 - y).abs().max()) <= 1e-12 * 16.0
    # (5) integer data: every order gives the same bits -- pattern of the same matrix, integer x, against itself
    #     under strict order
    Ap = capi.Matrix.from_csr(n, n, rp, cc, None, borrow=True)
    xi = _int_x(n, "cuda", 5)
    Ap.spmv(y, xi, st)
    capi.set_option("strict_order", 1)
    try:
        Ap.spmv(y2, xi, st)
    finally:
        capi.set_option("strict_order", 0)
    assert torch.equal(y, y2)
    del Ap
    # (6) "reproducible": only kernels with a fixed order of additions run (the two-pass pair with its one-wave-per-panel
    #     pass 2, or the L2-tiled kernel, whichever the builder measures faster); linearity is then exact (scaling by 2
    #     commutes with every rounding) and two runs give the same bits
    del A
    capi.set_option("reproducible", 1)
    try:
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        assert A.kernel_name() in ("two-pass", "tiled")
        A.spmv(y, x, st)
        A.spmv(y2, 2.0 * x, st)
        assert torch.equal(y2, 2.0 * y)
        A.spmv(y2, x, st)
        assert torch.equal(y2, y)
        for lo in (0, n - 3000):
            _window_check(capi, O, rp, cc, vv, x, y, lo, lo + 3000, exact=False)
    finally:
        capi.set_option("reproducible", 0)


def test_config3_binary_10m_by_1m_bit_exact(hip_env):
    """BASELINE config 3: SparseBinaryMatrix 10M x 1M, 64/row, integer x: bit-exact vs CPU arithmetic"""
    torch, capi, O = hip_env
    nrow, ncol, per = 10_000_000, 1_000_000, 64
    rp, cc, _ = capi.synth_uniform(nrow, ncol, per, 0x5EED0003, valued=False)
    A = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
    x = _int_x(ncol, "cuda", 3)
    y = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
    A.spmv(y, x, capi.current_stream())
    for lo in (0, 7_000_000, nrow - 2000):
        _window_check(capi, O, rp, cc, None, x, y, lo, lo + 2000, exact=True)
    # exact checksum of checksums: sum_r y[r] == sum over entries of x[col] (integers, < 2^53)
    xi = x.to(torch.int64)
    total = 0
    step = 64_000_000
    for a in range(0, nrow * per, step):
        total += int(xi[cc[a:a + step].long()].sum().item())
    assert int(y.to(torch.int64).sum().item()) == total


def test_config4_spmm_k32_columns(hip_env):
    """BASELINE config 4 (k = 32): every column of Y equals the single-vector product of that column of X"""
    torch, capi, O = hip_env
    n, per, k = 10_000_000, 16, 32
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002)
    A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
    i = torch.arange(n, device="cuda", dtype=torch.float64)[:, None]
    c = torch.arange(k, device="cuda", dtype=torch.float64)[None, :]
    X = torch.sin(7.0 * i + 17.0 * c + 0.3).contiguous()
    Y = torch.full((n, k), -1.0, dtype=torch.float64, device="cuda")
    st = capi.current_stream()
    A.spmm(Y, X, k, st)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    capi.set_option("strict_order", 1)             # single-vector product in storage order, like the SpMM kernel
    try:
        for j in (0, 13, 31):
            xj = X[:, j].contiguous()
            A.spmv(y, xj, st)
            assert torch.equal(Y[:, j], y), j
            _window_check(capi, O, rp, cc, vv, xj, y, 123_456, 125_456, exact=True)
    finally:
        capi.set_option("strict_order", 0)


@pytest.fixture(scope="module")
def hip_env():
    import torch
    assert torch.cuda.is_available()
    from libfastsparse_amd import capi
    from oracle import pyoracle as O
    capi.lib()
    return torch, capi, O


def test_config5_one_real_shard_rank3_of_8(hip_env):
    """BASELINE config 5, ONE real shard: CSR 100 M x 100 M, power-law row lengths (mean ~32, clipped at 1e6), the rows
    rank 3 of 8 owns under the nnz-balanced cut (bench.py's own partition and generator: ~12.5 M rows, ~400 M
    non-zeros, x of 800 MB).  Row windows around the longest rows and at both ends against the oracle, the builder's
    choice against the storage-order kernel on EVERY row, and the integer checksum of checksums on the pattern."""
    torch, capi, O = hip_env
    import bench
    prov = bench.HipProvider(torch.device("cuda", torch.cuda.current_device()))
    n_global = 100_000_000
    bounds, cum, total = bench.c5_partition(prov, n_global, 8)
    assert bounds[0] == 0 and bounds[-1] == n_global and 3.0e9 < total < 3.4e9
    per = [cum[i + 1] - cum[i] for i in range(8)]
    assert max(per) - min(per) <= 2 * bench.C5_MAXLEN and max(per) <= 2**31 - 1     # equal shares, each fits an int row_ptr
    lo, hi = bounds[3], bounds[4]
    rp, cc, vv, nnz = bench.c5_shard(prov, lo, hi, n_global)
    nrow, ncol = hi - lo, n_global
    assert nnz == per[3] and 3.7e8 < nnz < 4.3e8 and 1.1e7 < nrow < 1.4e7
    A = capi.Matrix.from_csr(nrow, ncol, rp, cc, vv, borrow=True)
    st = capi.current_stream()
    x = torch.sin(7.0 * torch.arange(ncol, device="cuda", dtype=torch.float64) + 0.3)
    y = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
    A.spmv(y, x, st)
    lens = (rp[1:] - rp[:-1])
    top = torch.topk(lens, 3).indices.tolist()
    for lo_w in [0, nrow - 400] + [max(0, min(t - 100, nrow - 400)) for t in top]:
        _window_check(capi, O, rp, cc, vv, x, y, lo_w, lo_w + 400, exact=False)
    y2 = torch.empty_like(y)
    capi.set_option("strict_order", 1)
    try:
        A.spmv(y2, x, st)
    finally:
        capi.set_option("strict_order", 0)
    _window_check(capi, O, rp, cc, vv, x, y2, max(0, min(top[0] - 100, nrow - 400)), max(0, min(top[0] - 100, nrow - 400)) + 400,
                  exact=True)                     # storage order = the oracle's bits, also on the longest row
    # row-scaled bound with the row length as the scale's proxy: |x| <= 1, |v| <= 1
    bound = 1e-12 * torch.clamp(lens.to(torch.float64), min=1.0)
    assert bool(((y - y2).abs() <= bound).all()), A.kernel_name()
    kernel = A.kernel_name()
    del A, y2
    Ap = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
    xi = _int_x(ncol, "cuda", 9)
    Ap.spmv(y, xi, st)
    total_i = 0
    step = 50_000_000
    xl = xi.to(torch.int64)
    for a in range(0, nnz, step):
        total_i += int(xl[cc[a:a + step].long()].sum().item())
    assert int(y.to(torch.int64).sum().item()) == total_i, (kernel, Ap.kernel_name())
    # the product in parts (what the all-gather inside the product rides on): with cut rows the combine pass follows by ranges
    # of rows; rows below every cut are final after that part, and all parts together are the whole product bit for bit
    rows = Ap.part_rows(4)
    assert rows[0] == 0 and rows[-1] == nrow and all(a <= b for a, b in zip(rows, rows[1:]))
    if Ap.kernel_name() == "two-pass":
        assert sum(b > a for a, b in zip(rows, rows[1:])) >= 3, rows
    yp = torch.full((nrow,), -7.0, dtype=torch.float64, device="cuda")
    for part in range(4):
        Ap.spmv_part(yp, xi, part, 4, st)
        assert bool(torch.equal(yp[:rows[part + 1]], y[:rows[part + 1]])), part
    assert bool(torch.equal(yp, y))


def test_config3_through_its_own_entry_points_coo(hip_env):
    """BASELINE config 3 as the reference would run it: a SparseBinaryMatrix 10 M x 1 M with 64 entries per row given
    as COO (row-major) to A_mul_B and At_mul_B (sparse.h:58-75) -- the device COO -> CSR sort at 640 M entries and the
    transposed handle -- through (1) the device layer (fs_coo_create) and (2) the drop-in symbols with the host struct
    and host vectors.  Integer-valued x: bit-exact windows against the oracle, exact checksums, both directions."""
    import ctypes as C
    import time
    torch, capi, O = hip_env
    import _hipbackend as H
    nrow, ncol, per = 10_000_000, 1_000_000, 64
    nnz = nrow * per
    _, cols, _ = capi.synth_uniform(nrow, ncol, per, 0x5EED0003, valued=False)
    rows = torch.arange(nrow, device="cuda", dtype=torch.int32).repeat_interleave(per)
    st = capi.current_stream()
    torch.cuda.synchronize()
    t0 = time.time()
    A = capi.Matrix.from_coo(nrow, ncol, rows, cols, None)
    torch.cuda.synchronize()
    t_a = time.time() - t0
    t0 = time.time()
    At = capi.Matrix.from_coo(ncol, nrow, cols, rows, None)
    torch.cuda.synchronize()
    t_t = time.time() - t0
    print("config 3: device COO -> CSR + format, 640 M entries: A %.2f s (%s), A' %.2f s (%s)"
          % (t_a, A.kernel_name(), t_t, At.kernel_name()))
    # the device CSR is what new_bcsr would build: row-major COO => cols unchanged, row_ptr = 64 r
    rp = (torch.arange(nrow + 1, device="cuda", dtype=torch.int64) * per).to(torch.int32)
    x = _int_x(ncol, "cuda", 3)
    u = _int_x(nrow, "cuda", 4)
    y = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
    z = torch.full((ncol,), -1.0, dtype=torch.float64, device="cuda")
    A.spmv(y, x, st)
    At.spmv(z, u, st)
    for lo in (0, 6_543_210, nrow - 2000):
        _window_check(capi, O, rp, cols, None, x, y, lo, lo + 2000, exact=True)
    xl = x.to(torch.int64)
    tot = 0
    for a in range(0, nnz, 64_000_000):
        tot += int(xl[cols[a:a + 64_000_000].long()].sum().item())
    assert int(y.to(torch.int64).sum().item()) == tot
    # transposed: z[c] = sum over the entries of column c of u[row]; checksum = 64 * sum(u); column windows vs the oracle
    assert int(z.to(torch.int64).sum().item()) == per * int(u.to(torch.int64).sum().item())
    for c0 in (0, 500_000, ncol - 300):
        c1 = c0 + 300
        hit = torch.zeros(nrow * per, dtype=torch.bool, device="cuda")
        for a in range(0, nnz, 128_000_000):
            blk = cols[a:a + 128_000_000]
            hit[a:a + 128_000_000] = (blk >= c0) & (blk < c1)
        idx = hit.nonzero().squeeze(1)
        wr, wc = rows[idx].cpu().numpy(), (cols[idx] - c0).cpu().numpy().astype(np.int32)
        ref = O.coo_tmul(c1 - c0, wr, wc, None, u.cpu().numpy())
        assert np.array_equal(z[c0:c1].cpu().numpy(), ref), c0
        del hit, idx
    del A, At
    y_dev, z_dev = y.cpu().numpy(), z.cpu().numpy()
    # (2) the drop-in: host struct SparseBinaryMatrix (sparse.h:11-18), host vectors
    hr, hc = rows.cpu().numpy(), cols.cpu().numpy()
    del rows, cols, rp
    S = H.SBM(nrow, ncol, nnz, H._ip(hr), H._ip(hc))
    L = capi.lib()
    xh, uh = x.cpu().numpy(), u.cpu().numpy()
    yh, zh = np.full(nrow, -1.0), np.full(ncol, -1.0)
    for f in (L.A_mul_B, L.At_mul_B):
        f.restype = None
    t0 = time.time()
    L.A_mul_B(H._dp(yh), C.byref(S), H._dp(xh))
    t_first = time.time() - t0
    t0 = time.time()
    L.A_mul_B(H._dp(yh), C.byref(S), H._dp(xh))
    t_again = time.time() - t0
    L.At_mul_B(H._dp(zh), C.byref(S), H._dp(uh))
    print("config 3 drop-in A_mul_B with host struct + host vectors: first call %.2f s (upload + sort + format), next %.4f s"
          % (t_first, t_again))
    L.fs_invalidate(C.byref(S))
    assert np.array_equal(yh, y_dev) and np.array_equal(zh, z_dev)


def test_pattern_matrix_at_the_int32_limit(hip_env):
    """The largest matrix a struct BinaryCSR can describe: `int* row_ptr` (csr.h:19) caps nnz at 2^31 - 1.  134 217 720 rows
    x 16 entries = 2 147 483 520 non-zeros (127 short of the cap), 50 M columns, pattern-only, integer-valued x: every offset
    computed in the kernels and the format builders must survive the top of the 32-bit range.  Row windows against the
    oracle (bit-exact), the builder's choice against the storage-order kernel on every row, the integer checksum of
    checksums."""
    torch, capi, O = hip_env
    nrow, ncol, per = 134_217_720, 50_000_000, 16
    nnz = nrow * per
    assert 2**31 - 1 - nnz == 127
    rp, cc, _ = capi.synth_uniform(nrow, ncol, per, 0x5EED00FF, valued=False)
    assert int(rp[-1].item()) == nnz
    A = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
    st = capi.current_stream()
    x = _int_x(ncol, "cuda", 11)
    y = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
    A.spmv(y, x, st)
    for lo in (0, 77_777_777, nrow - 3000):
        _window_check(capi, O, rp, cc, None, x, y, lo, lo + 3000, exact=True)
    y2 = torch.empty_like(y)
    capi.set_option("strict_order", 1)
    try:
        A.spmv(y2, x, st)
    finally:
        capi.set_option("strict_order", 0)
    assert torch.equal(y, y2), A.kernel_name()
    del y2
    xl = x.to(torch.int64)
    total = 0
    for a in range(0, nnz, 100_000_000):
        total += int(xl[cc[a:a + 100_000_000].long()].sum().item())
    assert int(y.to(torch.int64).sum().item()) == total, A.kernel_name()


def test_spmm_where_the_reference_overflows_int(hip_env):
    """csr_A_mul_Bn / bcsr_A_mul_Bn compute `int col = cols[i] * ncol` (csr.h:452, :268): with k = 32 that overflows for
    column ids >= 67 108 864.  100 M columns x k = 32 (X of 25.6 GB): every column of Y against the
    single-vector product of that column of X in storage order -- bit-identical -- and an oracle window of the latter."""
    torch, capi, O = hip_env
    nrow, ncol, per, k = 1_000_000, 100_000_000, 16, 32
    rp, cc, vv = capi.synth_uniform(nrow, ncol, per, 0x5EED0032)
    assert int(cc.max().item()) * k > 2**31
    A = capi.Matrix.from_csr(nrow, ncol, rp, cc, vv, borrow=True)
    X = torch.empty(ncol, k, dtype=torch.float64, device="cuda")
    i = torch.arange(ncol, device="cuda", dtype=torch.float64)
    for j in range(k):
        X[:, j] = torch.sin(7.0 * i + 17.0 * j + 0.3)
    del i
    Y = torch.full((nrow, k), -1.0, dtype=torch.float64, device="cuda")
    st = capi.current_stream()
    A.spmm(Y, X, k, st)
    y = torch.empty(nrow, dtype=torch.float64, device="cuda")
    capi.set_option("strict_order", 1)
    try:
        for j in (0, 9, 31):
            xj = X[:, j].contiguous()
            A.spmv(y, xj, st)
            assert torch.equal(Y[:, j], y), j
        _window_check(capi, O, rp, cc, vv, xj, y, 500_000, 502_000, exact=True)
    finally:
        capi.set_option("strict_order", 0)


def test_native_multi_gpu_path_holds_more_than_2_31_entries(hip_env):
    """VERDICT r3 item 3 at size: a matrix with MORE entries than one `int row_ptr` can index (csr.h:358-366) -- 2 304 000 000 >
    2^31 - 1 -- through the native C path: three pattern-only shards of 768 M entries generated on the device, handed over as
    per-rank device arrays (fs_dist_csr_create_from_shards), A' built from them on the device
    (fs_dist_matrix_build_transpose_device), three virtual ranks on this one GPU.  Integer-valued vectors: the checksum of
    checksums of both directions must be exact, row / column windows must equal the oracle / an exact recount, and every rank
    must hold the same vectors."""
    import ctypes as C
    torch, capi, O = hip_env
    L = capi.lib()
    ranks, rows_per, per = 3, 6_000_000, 128
    nrow, ncol = ranks * rows_per, 3_000_000
    D = L.fs_dist_create(ranks, (C.c_int * ranks)(*([0] * ranks)))
    assert D
    M = None
    try:
        shards = [capi.synth_uniform(rows_per, ncol, per, 0x5EED0A, row_offset=r * rows_per, valued=False) for r in range(ranks)]
        torch.cuda.synchronize()
        srows = (C.c_int * ranks)(*([rows_per] * ranks))
        snnz = (C.c_int64 * ranks)(*([rows_per * per] * ranks))
        p_rp = (C.c_void_p * ranks)(*[s[0].data_ptr() for s in shards])
        p_cc = (C.c_void_p * ranks)(*[s[1].data_ptr() for s in shards])
        M = L.fs_dist_csr_create_from_shards(D, nrow, ncol, srows, snnz, p_rp, p_cc, None, capi.FS_DEVICE)
        assert M, L.fs_last_error()
        total = L.fs_dist_matrix_nnz(M)
        assert total == ranks * rows_per * per and total > 2**31 - 1
        counts = torch.zeros(ncol, dtype=torch.int64, device="cuda")
        for s in shards:
            for a in range(0, rows_per * per, 128_000_000):
                counts += torch.bincount(s[1][a:a + 128_000_000].long(), minlength=ncol)
        assert int(counts.sum().item()) == total
        x = _int_x(ncol, "cuda", 21)
        xh = x.cpu().numpy()
        y = np.full(nrow, -1.0)
        assert L.fs_dist_spmv(M, y.ctypes.data, xh.ctypes.data) == 0, L.fs_last_error()
        assert float(y.sum()) == float((counts.to(torch.float64) * x).sum().item())          # checksum of checksums, exact
        xd = torch.from_numpy(xh).cuda()
        for r in range(ranks):
            yd = torch.from_numpy(y[r * rows_per:(r + 1) * rows_per]).cuda()
            for lo in (0, rows_per // 2 + 17, rows_per - 300):
                _window_check(capi, O, shards[r][0], shards[r][1], None, xd, yd, lo, lo + 300, exact=True)
            g = np.empty(nrow)                          # every rank holds the whole y
            assert L.fs_copy_to_host(g.ctypes.data, L.fs_dist_y(M, r), 8 * nrow) == 0
            assert np.array_equal(g, y), r
        # A' from the device-resident shards: cut by non-zeros of the columns, every shard below 2^31 - 1
        assert L.fs_dist_matrix_build_transpose_device(M) == 0, L.fs_last_error()
        bt = (C.c_int * (ranks + 1))()
        assert L.fs_dist_matrix_bounds_t(M, bt) == 0 and bt[0] == 0 and bt[ranks] == ncol
        cs = torch.cumsum(counts, 0)
        for r in range(ranks):
            sh = L.fs_dist_matrix_shard(M, r, 1)
            want = int(cs[bt[r + 1] - 1].item()) - (int(cs[bt[r] - 1].item()) if bt[r] > 0 else 0)
            assert L.fs_matrix_nnz(sh) == want and want <= 2**31 - 1, (r, L.fs_matrix_nnz(sh), want)
            assert abs(want - total // ranks) <= int(counts.max().item())                   # cut by non-zeros
        u = _int_x(nrow, "cuda", 22)
        uh = u.cpu().numpy()
        z = np.full(ncol, -1.0)
        assert L.fs_dist_spmv_t(M, z.ctypes.data, uh.ctypes.data) == 0, L.fs_last_error()
        assert float(z.sum()) == per * float(u.sum().item())                                  # every row has `per` entries
        c0 = ncol // 2 + 5
        zref = torch.zeros(400, dtype=torch.float64, device="cuda")                          # 400 columns recounted exactly
        for r, s in enumerate(shards):
            for a in range(0, rows_per * per, 128_000_000):
                cc = s[1][a:a + 128_000_000]
                idx = torch.nonzero((cc >= c0) & (cc < c0 + 400)).squeeze(1)
                zref.index_add_(0, (cc[idx] - c0).long(), u[r * rows_per + (a + idx) // per])
        assert np.array_equal(z[c0:c0 + 400], zref.cpu().numpy())
    finally:
        if M:
            L.fs_dist_matrix_destroy(M)
        L.fs_dist_destroy(D)


def test_config2_through_the_dropin_on_three_virtual_ranks(hip_env):
    """BASELINE config 2 at full size through the reference's own entry points with FASTSPARSE_NGPU=3 (three virtual ranks on this
    GPU): host structs in, vectors in HBM, the products row-sharded; exact integer checksums, adjointness of the transposed product,
    oracle windows (tests/_dropin_ngpu.py, mode fullsize)."""
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dropin_ngpu.py")
    env = dict(os.environ, FASTSPARSE_NGPU="3", FASTSPARSE_DEVICES="0,0,0")
    p = subprocess.run([sys.executable, child, "fullsize"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and p.stdout.rstrip().endswith("OK"), p.stdout[-3000:] + p.stderr[-3000:]
    print(p.stdout[-600:])
