#!/usr/bin/env python3
"""What fixed-order sums cost per product on config 3 (LDS-staged kernel; A owns its panels, A' shares them between chunks):
ms per product of A x and A' u with option reproducible = 0 / 1, the same handles.  One JSON line."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libfastsparse_amd import capi

rows_ = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nrow, ncol, per = rows_, rows_ // 10, 64
_, cols, _ = capi.synth_uniform(nrow, ncol, per, 0x5EED0003, valued=False)
rows = torch.arange(nrow, device="cuda", dtype=torch.int32).repeat_interleave(per)
A = capi.Matrix.from_coo(nrow, ncol, rows, cols, None)
At = capi.Matrix.from_coo(ncol, nrow, cols, rows, None)
del rows
st = capi.current_stream()
x = torch.sin(7.0 * torch.arange(ncol, device="cuda", dtype=torch.float64) + 0.3)
u = torch.sin(11.0 * torch.arange(nrow, device="cuda", dtype=torch.float64) - 0.2)
y = torch.empty(nrow, dtype=torch.float64, device="cuda")
z = torch.empty(ncol, dtype=torch.float64, device="cuda")


def ms(f, n=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rec = {"what": "fixed_order_ab", "rows": nrow, "kernel_A": A.kernel_name(), "kernel_At": At.kernel_name()}
for rep in (0, 1, 0, 1):
    capi.set_option("reproducible", rep)
    rec.setdefault("A_ms_reproducible_%d" % rep, []).append(ms(lambda: A.spmv(y, x, st)))
    rec.setdefault("At_ms_reproducible_%d" % rep, []).append(ms(lambda: At.spmv(z, u, st)))
capi.set_option("reproducible", 0)
print(json.dumps(rec))
