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
    # (6) "reproducible": the format builder keeps to kernels with a fixed order of additions; linearity is then
    #     exact (scaling by 2 commutes with every rounding) and two runs give the same bits
    del A
    capi.set_option("reproducible", 1)
    try:
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        assert A.kernel_name() != "two-pass"
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


def test_config5_like_shard_powerlaw(hip_env):
    """one row shard of BASELINE config 5's shape: 4 M rows x 100 M columns (x of 800 MB), power-law row lengths
    clipped at 1e6 (the generator of the full config, `fs_synth_powerlaw_lengths`), ~100 M non-zeros: row windows
    around the longest rows and at both ends against the oracle, the builder's choice against the storage-order
    kernel on every row, and the integer checksum of checksums on the pattern"""
    torch, capi, O = hip_env
    nrow, ncol = 4_000_000, 100_000_000
    rp, cc, vv = capi.synth_powerlaw(nrow, ncol, 2.3, 1_000_000, 0x5EED0005)
    nnz = int(rp[-1].item())
    assert 50_000_000 < nnz < 300_000_000
    A = capi.Matrix.from_csr(nrow, ncol, rp, cc, vv, borrow=True)
    st = capi.current_stream()
    x = torch.sin(7.0 * torch.arange(ncol, device="cuda", dtype=torch.float64) + 0.3)
    y = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
    A.spmv(y, x, st)
    lens = (rp[1:] - rp[:-1])
    longest = int(torch.argmax(lens).item())
    for lo in (0, max(0, min(longest - 100, nrow - 400)), nrow - 400):
        _window_check(capi, O, rp, cc, vv, x, y, lo, lo + 400, exact=False)
    y2 = torch.empty_like(y)
    capi.set_option("strict_order", 1)
    try:
        A.spmv(y2, x, st)
    finally:
        capi.set_option("strict_order", 0)
    # row-scaled bound with the row length as the scale's proxy: |x| <= 1, |v| <= 1
    bound = 1e-12 * torch.clamp(lens.to(torch.float64), min=1.0)
    assert bool(((y - y2).abs() <= bound).all()), A.kernel_name()
    del A
    Ap = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
    xi = _int_x(ncol, "cuda", 9)
    Ap.spmv(y, xi, st)
    total = 0
    step = 50_000_000
    xl = xi.to(torch.int64)
    for a in range(0, nnz, step):
        total += int(xl[cc[a:a + step].long()].sum().item())
    assert int(y.to(torch.int64).sum().item()) == total, Ap.kernel_name()
