#!/usr/bin/env python3
"""Block CG's products (k = 2 right-hand sides, PATTERN matrix: config 2's pattern): the one-sweep k-column two-pass copy against one
single-vector sweep per column (option spmm_kernel 3) -- the k-column sweep reads the entries once but moves 36 B per entry, two
single-vector sweeps move 2 x 19.4 with one-byte row ids.  Also valued, and k = 4.     python tools/k2_pattern_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402

n, per = 10_000_000, 16
st = capi.current_stream()
for valued in (False, True):
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002, valued=valued)
    A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
    for k in (2, 4):
        X = torch.sin(torch.arange(n * k, dtype=torch.float64, device="cuda")).reshape(n, k)
        Y = torch.empty(n, k, dtype=torch.float64, device="cuda")
        out = {}
        ref = None
        for name, opt in (("k-column sweep", 0), ("one sweep per column", 3)):
            capi.set_option("spmm_kernel", opt)
            A.prepare(k, st)
            for _ in range(3):
                A.spmm(Y, X, k, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                A.spmm(Y, X, k, st)
            e1.record()
            torch.cuda.synchronize()
            out[name] = round(e0.elapsed_time(e1) / 20, 4)
            if ref is None:
                ref = Y.clone()
            else:
                out["max_abs_diff"] = float((Y - ref).abs().max())
        capi.set_option("spmm_kernel", 0)
        print({"valued": valued, "k": k, "kernel": A.kernel_name(), **out}, flush=True)
    A.close()
    del rp, cc, vv
