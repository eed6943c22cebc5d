#!/usr/bin/env python3
"""Is pass 1 of the two-pass pair held back by WHERE its products go?  Its stores are 3.4-KB runs (one (band, panel) cell) scattered
over 610 panel regions; this probe times pass 1 alone (fs_debug_two_pass_run) as built and again after overwriting gdst with the
identity, so that the same kernel stores the same bytes sequentially (the products then sit in pass-1 order: wrong for pass 2, a
TIMING experiment).  Valued and pattern-only config 2.     python tools/pass1_write_order_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402

L = capi.lib()
L.fs_debug_two_pass_layout.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
L.fs_debug_two_pass_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
n, per = 10_000_000, 16
st = capi.current_stream()
x = torch.sin(7.0 * torch.arange(n, dtype=torch.float64, device="cuda") + 0.3)
y = torch.empty(n, dtype=torch.float64, device="cuda")


def timed(A, which, reps=20):
    for _ in range(3):
        capi.check(L.fs_debug_two_pass_run(A.h, 0, which, y.data_ptr(), x.data_ptr(), st), "two_pass_run")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        capi.check(L.fs_debug_two_pass_run(A.h, 0, which, y.data_ptr(), x.data_ptr(), st), "two_pass_run")
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


for valued in (True, False):
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002, valued=valued)
    A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
    assert A.kernel_name() == "two-pass"
    lay = (C.c_ulonglong * 8)()
    capi.check(L.fs_debug_two_pass_layout(A.h, 0, lay), "layout")
    groups = int(lay[5]) // 16
    rec = {"valued": valued, "entries_padded": int(lay[5]), "pass1_ms_as_built": timed(A, 1), "pass2_ms": timed(A, 2)}
    ident = np.arange(groups, dtype=np.uint32)
    capi.check(L.fs_copy_to_device(C.c_void_p(int(lay[2])), ident.ctypes.data, 4 * groups), "gdst := identity")
    torch.cuda.synchronize()
    rec["pass1_ms_sequential_stores"] = timed(A, 1)
    print(rec, flush=True)
    A.close()
    del rp, cc, vv
