#!/usr/bin/env python3
"""Does WHERE a copy lands in HBM change its speed?  The same config-2 matrix built several times in one process (identical bytes,
different allocations), each timed over the same products; then rebuilt after the earlier copies were released.

    python tools/placement_probe.py        # one JSON line per copy
"""
import ctypes as C
import json
import sys
import time
import zlib

import numpy as np

import torch

sys.path.insert(0, ".")
from libfastsparse_amd import capi  # noqa: E402


def timed(f, reps=20):
    for _ in range(3):
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
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002, valued=True, device=dev)
    x = torch.sin(torch.arange(n, dtype=torch.float64, device=dev) * 7.0 + 0.3)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    keep = []
    L = capi.lib()
    hip = C.CDLL("libamdhip64.so")
    L.fs_debug_two_pass_layout.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
    L.fs_debug_two_pass_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for i in range(8):
        t0 = time.perf_counter()
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        t = [timed(lambda: A.spmv(y, x)) for _ in range(3)]
        lay = (C.c_ulonglong * 8)()
        L.fs_debug_two_pass_layout(A.h, 0, lay)
        st = capi.current_stream()
        p1 = timed(lambda: L.fs_debug_two_pass_run(A.h, 0, 1, capi._ptr(y), capi._ptr(x), st))
        p2 = timed(lambda: L.fs_debug_two_pass_run(A.h, 0, 2, capi._ptr(y), capi._ptr(x), st))
        # are the copies the same bytes?  (the builder scatters with atomics: the order inside a run could differ)
        crc = {}
        for name, ptr, nbytes in (("gdst", lay[2], (lay[5] // 16) * 4), ("lcol", lay[0], lay[5] * 2), ("lrow", lay[3], lay[5] * 2)):
            buf = np.empty(nbytes, dtype=np.uint8)
            assert hip.hipMemcpy(C.c_void_p(buf.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), 2) == 0
            crc[name] = zlib.crc32(buf.tobytes())
        print(json.dumps({"copy": i, "build_s": round(build_s, 3), "builder_ms": A.candidate_ms(), "crc": crc, "held_copies": len(keep), "kernel": A.kernel_name(), "ms": t, "pass1_ms": p1, "pass2_ms": p2,
                          "lcol": hex(lay[0]), "vals": hex(lay[1]), "gdst": hex(lay[2]), "lrow": hex(lay[3]), "prod": hex(lay[4]),
                          "x": hex(x.data_ptr()), "y": hex(y.data_ptr()), "n": lay[5]}), flush=True)
        keep.append(A)
        if i == 3:      # release everything, let the pool hand the same blocks out again
            for B in keep:
                B.close()
            keep = []
            capi.lib().fs_release_all()
    # the same handle timed again after everything else is gone
    for B in keep:
        print(json.dumps({"again": True, "ms": [timed(lambda: B.spmv(y, x)) for _ in range(3)]}), flush=True)


if __name__ == "__main__":
    main()
