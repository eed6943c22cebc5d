#!/usr/bin/env python3
"""Extreme shapes through the product, default kernel choice against the storage-order kernel (pattern-only matrices with
integer-valued x: every order gives the same bits) plus the integer checksum of checksums.  Not part of the test suite
(tens of GB of HBM); run on the GPU box:  python tools/stress_shapes.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402


def check(name, nrow, ncol, rp, cc, vv=None, k=0):
    st = capi.current_stream()
    t0 = time.time()
    A = capi.Matrix.from_csr(nrow, ncol, rp, cc, vv, borrow=True)
    torch.cuda.synchronize()
    tb = time.time() - t0
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    x = torch.randint(-1000, 1001, (ncol,), device="cuda", generator=g).to(torch.float64)
    y = torch.full((nrow,), -1.0, dtype=torch.float64, device="cuda")
    y2 = torch.empty_like(y)
    A.spmv(y, x, st)
    capi.set_option("strict_order", 1)
    try:
        A.spmv(y2, x, st)
    finally:
        capi.set_option("strict_order", 0)
    ok = bool(torch.equal(y, y2)) if vv is None else bool(((y - y2).abs() <= 1e-9 * (1 + y2.abs())).all())
    nnz = int(cc.numel())
    if vv is None:
        tot, xl = 0, x.to(torch.int64)
        for a in range(0, nnz, 100_000_000):
            tot += int(xl[cc[a:a + 100_000_000].long()].sum().item())
        ok = ok and int(y.to(torch.int64).sum().item()) == tot
    # the same product in parts (fs_spmv_part): rows final range by range, all parts together the whole product
    rows = A.part_rows(4)
    y3 = torch.full((nrow,), -7.0, dtype=torch.float64, device="cuda")
    for part in range(4):
        A.spmv_part(y3, x, part, 4, st)
        r1 = rows[part + 1]
        okp = bool(torch.equal(y3[:r1], y[:r1])) if vv is None else bool(((y3[:r1] - y2[:r1]).abs() <= 1e-9 * (1 + y2[:r1].abs())).all())
        ok = ok and okp
    ok = ok and rows[0] == 0 and rows[-1] == nrow
    if k:
        X = torch.stack([x + j for j in range(k)], 1).contiguous()
        Y = torch.full((nrow, k), -1.0, dtype=torch.float64, device="cuda")
        A.spmm(Y, X, k, st)
        for j in (0, k - 1):
            capi.set_option("strict_order", 1)
            try:
                A.spmv(y2, X[:, j].contiguous(), st)
            finally:
                capi.set_option("strict_order", 0)
            okj = bool(torch.equal(Y[:, j], y2)) if vv is None else bool(((Y[:, j] - y2).abs() <= 1e-9 * (1 + y2.abs())).all())
            ok = ok and okj
    print("%-55s %-10s build %.2fs  parts %s  %s" % (name, A.kernel_name(), tb, rows, "OK" if ok else "FAIL"), flush=True)
    return ok


def main():
    dev = "cuda"
    good = True
    # one row holding everything
    n = 60_000_000
    rp = torch.tensor([0, n], dtype=torch.int32, device=dev)
    cc = torch.randint(0, 5_000_000, (n,), device=dev, dtype=torch.int32)
    good &= check("1 x 5M, 60M entries in the one row", 1, 5_000_000, rp, cc)
    # one column
    rp = (torch.arange(20_000_001, device=dev, dtype=torch.int64) * 2).to(torch.int32)
    cc = torch.zeros(40_000_000, dtype=torch.int32, device=dev)
    good &= check("20M x 1, two entries per row, all in column 0", 20_000_000, 1, rp, cc, k=2)
    # empty rows in front, behind and in between; a few monster rows
    lens = torch.zeros(30_000_000, dtype=torch.int64, device=dev)
    lens[10_000_000:20_000_000:3] = 40
    lens[15_000_000] = 20_000_000
    lens[15_000_007] = 5_000_000
    rp = torch.zeros(30_000_001, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=rp[1:])
    nnz = int(rp[-1].item())
    cc = torch.randint(0, 70_000_000, (nnz,), device=dev, dtype=torch.int32)
    good &= check("30M x 70M, two thirds empty rows, rows of 20M and 5M entries (%dM)" % (nnz // 1_000_000), 30_000_000, 70_000_000,
                  rp.to(torch.int32), cc, k=4)
    # heavy tail with more long rows than accumulators (LongRows takes the longest 12 032), valued
    from libfastsparse_amd.capi import synth_powerlaw
    rp, cc, vv = synth_powerlaw(6_000_000, 50_000_000, 6.0, 400_000, 0xBEEF)
    good &= check("6M x 50M power-law (scale 6, max 400K): %dM entries, valued" % (int(cc.numel()) // 1_000_000), 6_000_000, 50_000_000,
                  rp, cc, vv, k=3)
    del rp, cc, vv
    # valued, close to the int limit
    nrow, per = 67_108_860, 31
    rp, cc, vv = capi.synth_uniform(nrow, 9_999_999, per, 77)
    good &= check("67M x 10M x 31 valued: %d entries" % (nrow * per), nrow, 9_999_999, rp, cc, vv)
    del rp, cc, vv
    # dense-ish tiles, odd column count, k = 2 on the LDS-staged class
    rp, cc, _ = capi.synth_uniform(3_000_001, 300_001, 96, 5, valued=False)
    good &= check("3M x 300001 x 96 pattern (LDS-staged class), k = 2", 3_000_001, 300_001, rp, cc, k=2)
    print("ALL OK" if good else "SOME FAILED")
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()
