"""Host-side cost of ONE sharded product of the native C path (fs_dist_spmv_resident: one host thread drives every rank's launches,
events and exchange calls) as a function of the number of ranks, on virtual ranks of one GPU and a matrix small enough that the
kernels take microseconds: what is left is launch + synchronisation overhead per product.   python tools/dist_overhead_probe.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402
from oracle import pysynth  # noqa: E402

L = capi.lib()
nrow = ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
rp, cc, vv = pysynth.uniform(nrow, ncol, 16, 7)
x = np.sin(np.arange(ncol) * 0.1)
for ranks in (1, 2, 4, 8):
    D = L.fs_dist_create(ranks, (C.c_int * ranks)(*([0] * ranks)))
    M = L.fs_dist_csr_create(D, nrow, ncol, len(cc), rp.ctypes.data, cc.ctypes.data, vv.ctypes.data)
    for r in range(ranks):
        L.fs_copy_to_device(L.fs_dist_x(M, r), x.ctypes.data, 8 * ncol)
    for _ in range(5):
        L.fs_dist_spmv_resident(M)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        L.fs_dist_spmv_resident(M)
    dt = (time.perf_counter() - t0) / n
    print("ranks %d: %.1f us per resident product (%d rows x 16; conservative=%d)" % (ranks, dt * 1e6, nrow, L.fs_dist_is_conservative(D)), flush=True)
    L.fs_dist_matrix_destroy(M)
    L.fs_dist_destroy(D)
