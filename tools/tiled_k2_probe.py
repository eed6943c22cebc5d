#!/usr/bin/env python3
"""Would a k = 2 form of the L2-tiled kernel beat the two-pass k = 2 sweep on config 2's pattern (block CG's products)?

A lower bound without writing that kernel: the existing L2-tiled kernel gathers ONE column of a row-major two-column X (stride 2:
the 16-byte row pitch and the doubled band footprint a k = 2 kernel would have; it would add the second column's LDS work on top),
for several band widths, next to the unit-stride product on the same copy and the two-pass k = 2 sweep.

    python tools/tiled_k2_probe.py            # one JSON line per measurement
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from libfastsparse_amd import capi  # noqa: E402


def timed(f, reps=10):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n, per = 10_000_000, 16
    dev = "cuda"
    rp, cc, _ = capi.synth_uniform(n, n, per, 0x5EED0002, valued=False, device=dev)
    X = torch.sin(torch.arange(2 * n, dtype=torch.float64, device=dev) * 0.37 + 0.1).reshape(n, 2).contiguous()
    x1 = X[:, 0].contiguous()
    Y = torch.empty(n, 2, dtype=torch.float64, device=dev)
    y1 = torch.empty(n, dtype=torch.float64, device=dev)
    # the two-pass k = 2 sweep (what block CG runs today)
    capi.set_option("binning", 2)
    A = capi.Matrix.from_csr(n, n, rp, cc, None, borrow=True)
    A.prepare(2)
    rec = {"what": "two-pass", "kernel": A.kernel_name(), "k1_ms": timed(lambda: A.spmv(y1, x1)),
           "k2_ms": timed(lambda: A.spmm(Y, X, 2)), "plan_k2": A.spmm_plan(2)}
    print(json.dumps(rec), flush=True)
    Yref = Y.clone()
    A.close()
    # the L2-tiled copy alone, several band widths
    capi.set_option("binning", 0)
    capi.set_option("ldsx", 0)
    capi.set_option("tiling", 2)
    for W in (0, 262144, 131072, 65536):
        capi.set_option("tile_cols", W)
        t0 = time.perf_counter()
        A = capi.Matrix.from_csr(n, n, rp, cc, None, borrow=True)
        torch.cuda.synchronize()
        build = time.perf_counter() - t0
        rec = {"what": "L2-tiled", "tile_cols": W, "kernel": A.kernel_name(), "build_s": build,
               "k1_unit_stride_ms": timed(lambda: A.spmv(y1, x1))}
        plan = A.spmm_plan(2)
        rec["plan_k2"] = plan
        t2 = timed(lambda: A.spmm(Y, X, 2))
        rec["k2_two_strided_sweeps_ms"] = t2
        rec["one_strided_sweep_ms"] = t2 / 2
        rec["max_abs_diff_vs_two_pass"] = float((Y - Yref).abs().max())
        print(json.dumps(rec), flush=True)
        A.close()


if __name__ == "__main__":
    main()
