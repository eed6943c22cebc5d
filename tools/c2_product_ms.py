#!/usr/bin/env python3
"""ms per product of config 2 (A x, the two-pass pair) with the library FS_LIB_PATH names: 50 products between one event pair,
three times.  No result check.     python tools/c2_product_ms.py <label> [bin_flags while building]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402

n, per = 10_000_000, 16
rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002)
build_flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # bin_flags while the copy is built (64: two-byte row ids, 0: the default)
capi.set_option("bin_flags", build_flags)
A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
capi.set_option("bin_flags", 0)
x = torch.sin(7.0 * torch.arange(n, dtype=torch.float64, device="cuda") + 0.3)
y = torch.empty(n, dtype=torch.float64, device="cuda")
st = capi.current_stream()
for _ in range(5):
    A.spmv(y, x, st)
torch.cuda.synchronize()
out = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        A.spmv(y, x, st)
    e1.record()
    torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) / 50, 4))
print(sys.argv[1] if len(sys.argv) > 1 else "", A.kernel_name(), "ms per product:", out, flush=True)
