"""Pass 2 of the two-pass SpMV with a fixed order of additions (one wave per panel, bin_flags bit 5) against the default
sixteen-wave pass 2 and against the kernel `reproducible = 1` used before (L2-tiled): ms per product on config 2 (A x and A' u),
bit-identity run to run, agreement with the default.   python tools/ordered_reduce_ab.py [--rows N]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libfastsparse_amd import capi  # noqa: E402


def timed(f, reps=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--per-row", type=int, default=16)
    ap.add_argument("--out", default="gpurun_out/ordered_reduce_ab.jsonl")
    a = ap.parse_args()
    n, m = a.rows, a.cols or a.rows
    st = capi.current_stream()
    rp, cc, vv = capi.synth_uniform(n, m, a.per_row, 0x5EED0002)
    A = capi.Matrix.from_csr(n, m, rp, cc, vv, borrow=True)
    A.build_transpose(st)
    x = torch.sin(7.0 * torch.arange(m, device="cuda", dtype=torch.float64) + 0.3)
    u = torch.sin(11.0 * torch.arange(n, device="cuda", dtype=torch.float64) - 0.2)
    y, z = torch.empty(n, device="cuda", dtype=torch.float64), torch.empty(m, device="cuda", dtype=torch.float64)
    rec = {"what": "ordered_reduce_ab", "rows": n, "cols": m, "per_row": a.per_row, "kernel": A.kernel_name()}
    outs = {}
    for name, flags in (("default_16_waves", 0), ("ordered_1_wave", 32)):
        capi.set_option("bin_flags", flags)
        rec["ms_A_" + name] = timed(lambda: A.spmv(y, x, st))
        rec["ms_At_" + name] = timed(lambda: A.spmv(z, u, st, transposed=True))
        A.spmv(y, x, st)
        y1 = y.clone()
        same = True
        for _ in range(5):
            A.spmv(y, x, st)
            same = same and bool(torch.equal(y, y1))
        rec["bit_identical_over_6_runs_" + name] = same
        outs[name] = y1
    capi.set_option("bin_flags", 0)
    for k in (2, 4):                      # the k-column sweeps (block CG lives on k = 2)
        A.prepare(k, st)
        X = torch.sin(0.37 * torch.arange(m * k, device="cuda", dtype=torch.float64))
        Y = torch.empty(n * k, device="cuda", dtype=torch.float64)
        for name, flags in (("default_16_waves", 0), ("ordered_1_wave", 32)):
            capi.set_option("bin_flags", flags)
            rec["ms_spmm_k%d_%s" % (k, name)] = timed(lambda: A.spmm(Y, X, k, st), reps=10)
        capi.set_option("bin_flags", 0)
        del X, Y
    rec["max_abs_diff_ordered_vs_default"] = float((outs["ordered_1_wave"] - outs["default_16_waves"]).abs().max())
    print(json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    open(a.out, "a").write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
