#!/usr/bin/env python3
"""Which array's placement decides the speed of the two-pass pair?  One config-2 handle; one array at a time is moved to a fresh
allocation several times (fs_debug_two_pass_realloc) and both passes are timed after every move.

    python tools/build_variants.py lab=-DFS_LAB                      # fs_debug_two_pass_realloc exists in -DFS_LAB builds only
    FS_LIB_PATH=libfastsparse_amd/build/variants/libfs_lab.so python tools/placement_probe2.py [spread | arena]
one JSON line per move; under rocprofv3 --pmc TCP_UTCL1_* the per-placement counters are summed up by tools/placement_pmc.py.
"""
import ctypes as C
import json
import sys

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
    L = capi.lib()
    L.fs_debug_two_pass_layout.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
    L.fs_debug_two_pass_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fs_debug_two_pass_realloc.argtypes = [C.c_void_p, C.c_int, C.c_int]
    capi.set_option("binning", 2)      # (under rocprofv3 --pmc the builder's own timing may prefer another copy)
    capi.set_option("tiling", 0)
    A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
    st = capi.current_stream()
    names = ["lcol", "vals", "gdst", "lrow", "prod"]

    def measure(tag):
        lay = (C.c_ulonglong * 8)()
        L.fs_debug_two_pass_layout(A.h, 0, lay)
        p1 = timed(lambda: L.fs_debug_two_pass_run(A.h, 0, 1, capi._ptr(y), capi._ptr(x), st))
        p2 = timed(lambda: L.fs_debug_two_pass_run(A.h, 0, 2, capi._ptr(y), capi._ptr(x), st))
        print(json.dumps({"moved": tag, "pass1_ms": round(p1, 4), "pass2_ms": round(p2, 4),
                          "at": {nm: hex(lay[i]) for i, nm in enumerate(names)}}), flush=True)

    measure("start")
    if len(sys.argv) > 1 and sys.argv[1] == "spread":
        # eight placements of prod held at once (the old block is not freed), each timed; then the last one timed again
        for rep in range(8):
            assert L.fs_debug_two_pass_realloc(A.h, 0, 4 | 16) == 0
            measure("prod (old kept) #%d" % rep)
        measure("again")
        for rep in range(6):
            assert L.fs_debug_two_pass_realloc(A.h, 0, 1 | 16) == 0
            measure("vals (old kept) #%d" % rep)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "arena":
        # all five arrays in ONE fresh allocation, six times (earlier blocks stay allocated: the handle must not be closed after this)
        for rep in range(6):
            assert L.fs_debug_two_pass_realloc(A.h, 0, 32 | (64 if rep % 2 else 0)) == 0
            measure("arena #%d%s" % (rep, " (vals first)" if rep % 2 else " (prod first)"))
        import os
        os._exit(0)
    for which in (4, 1, 0, 3, 2):
        for rep in range(4):
            assert L.fs_debug_two_pass_realloc(A.h, 0, which) == 0
            measure("%s #%d" % (names[which], rep))


if __name__ == "__main__":
    main()
