"""Row SpMM kernel A/B on config 4's matrix (CSR fp64 10 M x 10 M x 16): 8-byte loads, one column per lane (spmm_kernel) against
16-byte loads, two columns per lane, sixteen X rows in flight (spmm_wide_kernel), for several k; results must be bit-identical.
    python tools/spmm_wide_ab.py [--rows N] [--ks 4,6,8,16,32,64]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libfastsparse_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--per-row", type=int, default=16)
    ap.add_argument("--ks", default="4,6,8,12,16,32,64")
    ap.add_argument("--pattern", action="store_true")
    ap.add_argument("--out", default="gpurun_out/spmm_wide_ab.jsonl")
    a = ap.parse_args()
    n = a.rows
    st = capi.current_stream()
    rp, cc, vv = capi.synth_uniform(n, n, a.per_row, 0x5EED0002)
    A = capi.Matrix.from_csr(n, n, rp, cc, None if a.pattern else vv, borrow=True)
    capi.set_option("spmm_kernel", 1)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    f = open(a.out, "a")
    for k in [int(v) for v in a.ks.split(",")]:
        X = torch.sin(0.37 * torch.arange(n * k, device="cuda", dtype=torch.float64))
        res = {}
        for mode, w in (("narrow", -1), ("wide", 1)):
            capi.set_option("spmm_wide", w)
            Y = torch.full((n * k,), -1.0, device="cuda", dtype=torch.float64)
            for _ in range(2):
                A.spmm(Y, X, k, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                A.spmm(Y, X, k, st)
            e1.record()
            torch.cuda.synchronize()
            res[mode] = (e0.elapsed_time(e1) / 10, Y)
        same = bool(torch.equal(res["narrow"][1], res["wide"][1]))
        alg = (4 if a.pattern else 12) * n * a.per_row + 4 * (n + 1) + 16 * k * n
        rec = {"what": "spmm_wide_ab", "rows": n, "per_row": a.per_row, "valued": not a.pattern, "k": k,
               "ms_narrow": res["narrow"][0], "ms_wide": res["wide"][0], "bit_identical": same,
               "algorithmic_TBs_wide": alg / res["wide"][0] / 1e9}
        print(json.dumps(rec), flush=True)
        f.write(json.dumps(rec) + "\n")
        del X, res
    capi.set_option("spmm_wide", 0)
    capi.set_option("spmm_kernel", 0)


if __name__ == "__main__":
    main()
