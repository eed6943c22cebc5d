"""Where do the phases of the LDS-staged DMA kernel spend their cycles?  Needs an instrumented build
(FS_HIPCC_EXTRA="-DFS_DMA_TRACE=1" python libfastsparse_amd/_build.py; =2 puts the middle clock in front of the wait for the slice):
wave 0 of every workgroup sums, over its phases, the shader clocks between
   the barrier -> gathers returned (lgkmcnt(0))                              [a]
   ... -> adds and next entries issued (+ TRACE=1: this wave's slice DMA has landed, vmcnt)   [bc]
   ... -> past the barrier                                                    [d]
Config 3 (binary 10 M x 1 M x 64).   python tools/dma_phase_trace.py [--rows N]"""
import argparse
import ctypes as C
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
    ap.add_argument("--cols", type=int, default=1_000_000)
    ap.add_argument("--per-row", type=int, default=64)
    ap.add_argument("--tiled-flags", type=int, default=0, help="8: the split-role kernel (DMA wave + entry waves)")
    ap.add_argument("--tile-rows", type=int, default=0, help="panel height override (the four-buffer experiment needs <= 12288)")
    ap.add_argument("--force-ldsx", action="store_true", help="keep the LDS-staged copy whatever the builder's timing says (ablation builds)")
    ap.add_argument("--out", default="gpurun_out/dma_phase_trace.jsonl")
    a = ap.parse_args()
    L = capi.lib()
    traced = hasattr(L, "fs_debug_dma_trace")       # any other build (e.g. an ablation variant): the time only
    if traced:
        L.fs_debug_dma_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    st = capi.current_stream()
    capi.set_option("tiled_flags", a.tiled_flags)
    rp, cc, _ = capi.synth_uniform(a.rows, a.cols, a.per_row, 0x5EED0003, valued=False)
    capi.set_option("tile_rows", a.tile_rows)
    if a.force_ldsx:
        for k, v in (("ldsx", 2), ("tiling", 0), ("binning", 0)):
            capi.set_option(k, v)
    A = capi.Matrix.from_csr(a.rows, a.cols, rp, cc, None, borrow=True)
    x = torch.sin(0.3 * torch.arange(a.cols, device="cuda", dtype=torch.float64))
    y = torch.empty(a.rows, device="cuda", dtype=torch.float64)
    # the kernel under test against the default one of this build on integer-valued x (bit-identical whatever the order of the adds)
    xi = ((torch.arange(a.cols, device="cuda") % 17) - 8).to(torch.float64)
    yref = torch.empty_like(y)
    capi.set_option("tiled_flags", 0)
    A.spmv(yref, xi, st)
    capi.set_option("tiled_flags", a.tiled_flags)
    A.spmv(y, xi, st)
    same = bool(torch.equal(y, yref))
    for _ in range(3):
        A.spmv(y, x, st)
    out = (C.c_ulonglong * 8)()
    if traced:
        L.fs_debug_dma_trace(out, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        A.spmv(y, x, st)
    e1.record()
    torch.cuda.synchronize()
    if traced:
        L.fs_debug_dma_trace(out, 0)
    ta, tbc, td, n, wgs = (int(out[i]) for i in range(5))
    extra = [int(out[i]) for i in range(5, 8)]
    rec = {"what": "dma_phase_trace", "build": os.path.basename(os.environ.get("FS_LIB_PATH", "product")), "tiled_flags": a.tiled_flags, "tile_rows": a.tile_rows, "equals_default_kernel_on_integer_x": same, "kernel": A.kernel_name(),
           "ms_per_product": e0.elapsed_time(e1) / reps, "instrumented": traced,
           "workgroups": wgs // reps, "phases_per_workgroup": n / max(wgs, 1),
           "clocks_per_phase": {"barrier_to_gathers_returned": ta / max(n, 1), "to_issue_done_or_slice_landed": tbc / max(n, 1),
                                "to_past_the_barrier": td / max(n, 1), "sum": (ta + tbc + td) / max(n, 1)},
           "split_kernel_entry_wave0_clocks_per_phase": {"barrier_to_gathers_returned": extra[0] / max(n, 1), "to_entries_home_incl_adds_drained": extra[1] / max(n, 1),
                                                         "to_past_the_barrier": extra[2] / max(n, 1)},
           "note": "s_memtime ticks = shader clocks here (sum x phases / time = 1.96 GHz)"}
    if not traced:
        for k in ("workgroups", "phases_per_workgroup", "clocks_per_phase", "note"):
            rec.pop(k)
    print(json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    open(a.out, "a").write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
