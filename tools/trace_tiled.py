#!/usr/bin/env python3
"""Diagnostic: per-item start times and per-panel XCD of one tiled-kernel launch on config 2 -> npz."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--tile-rows", type=int, default=0)
ap.add_argument("--tile-cols", type=int, default=0)
ap.add_argument("--out", default="gpurun_out/trace_tiled.npz")
ap.add_argument("--powerlaw", action="store_true")
ap.add_argument("--gate-kb", type=int, default=8192)
a = ap.parse_args()
capi.set_option("tile_rows", a.tile_rows)
capi.set_option("tile_cols", a.tile_cols)
n = a.rows
rp, cc, vv = (capi.synth_powerlaw(n, n, 2.3, 1_000_000, 0x5EED0005) if a.powerlaw else capi.synth_uniform(n, n, 16, 0x5EED0002))
A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
L = capi.lib()
geo = (C.c_int * 6)()
capi.check(L.fs_debug_tiled_geometry(C.c_void_p(A.h), geo))
R, W, P, J, nitems, _ = list(geo)
x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
y = torch.empty(n, dtype=torch.float64, device="cuda")
times = np.zeros(nitems, np.int64)
xcc = np.zeros(P, np.int32)
items = np.zeros((nitems, 4), np.int32)
item_ptr = np.zeros(P + 1, np.int32)
L.fs_debug_tiled_trace.argtypes = [C.c_void_p] * 7
capi.check(L.fs_debug_tiled_trace(A.h, y.data_ptr(), x.data_ptr(), times.ctypes.data, xcc.ctypes.data,
                                  items.ctypes.data, item_ptr.ctypes.data))
np.savez_compressed(a.out, times=times, xcc=xcc, items=items, item_ptr=item_ptr, geo=np.array([R, W, P, J, nitems]))
t0 = times[times > 0].min()
print("R W P J nitems", R, W, P, J, nitems, "span_us", (times.max() - t0) / 100.0)
