"""Every reference-named product on ONE matrix (host structs, vectors in HBM): the FIRST call (upload + format work, and for a
new k the one-time fs_matrix_prepare the drop-in layer runs) reported apart from the steady state: ms per call and the ratio
to the plain product of the same matrix times the number of columns -- an entry point that costs far more than k plain products
is a cliff in the drop-in layer or in the kernel choice, not in the kernels.  FS_PREPARE_K=2,4,... moves the k-column work of
the listed ks into the first call on the matrix.   python tools/entrypoint_sweep.py [nrow ncol per_row]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _hipbackend as H                      # noqa: E402
from libfastsparse_amd import capi           # noqa: E402

nrow, ncol, per = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2_000_000, 200_000, 64)
rng = np.random.default_rng(3)
rows = np.repeat(np.arange(nrow, dtype=np.int32), per)
cols = rng.integers(0, ncol, nrow * per, dtype=np.int32)
vals = rng.uniform(-1, 1, nrow * per)
F = H.HostFormats()
L = F.L
sbm, sdm = F.sbm(nrow, ncol, rows, cols), F.sdm(nrow, ncol, rows, cols, vals)
bcsr, csr = F.bcsr(nrow, ncol, rows, cols), F.csr(nrow, ncol, rows, cols, vals)
bsbm, bsdm = L.new_bsbm(C.byref(sbm), 1024), L.new_bsdm(C.byref(sdm), 1024)
cb = F.cbcsr(8192, nrow, ncol, rows, cols)
dev = lambda n: torch.sin(torch.arange(n, device="cuda", dtype=torch.float64) * 0.37)
ptr = lambda t: C.c_void_p(t.data_ptr())


def timed(name, out_n, A, x_n, k=1, extra=(), reps=10):
    f = getattr(L, name)
    f.restype = None
    x, y = dev(x_n * k), torch.empty(out_n * k, device="cuda", dtype=torch.float64)
    a = A if isinstance(A, C._Pointer) else C.byref(A)
    torch.cuda.synchronize()
    t0 = time.time()
    f(ptr(y), a, ptr(x), *extra)
    torch.cuda.synchronize()
    first = (time.time() - t0) * 1e3
    f(ptr(y), a, ptr(x), *extra)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        f(ptr(y), a, ptr(x), *extra)
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e3, first


base = {}
rowsout = []
for name, out_n, A, x_n, k, extra, ref in [
        ("A_mul_B", nrow, sbm, ncol, 1, (), None), ("At_mul_B", ncol, sbm, nrow, 1, (), None),
        ("sdm_A_mul_B", nrow, sdm, ncol, 1, (), None), ("sdm_At_mul_B", ncol, sdm, nrow, 1, (), None),
        ("bcsr_A_mul_B", nrow, bcsr, ncol, 1, (), "A_mul_B"), ("csr_A_mul_B", nrow, csr, ncol, 1, (), "sdm_A_mul_B"),
        ("bsbm_A_mul_B", nrow, bsbm, ncol, 1, (), "A_mul_B"), ("bsdm_A_mul_B", nrow, bsdm, ncol, 1, (), "sdm_A_mul_B"),
        ("cbcsr_A_mul_B", nrow, cb, ncol, 1, (), "A_mul_B"),
        ("bsbm_A_mul_B2", nrow, bsbm, ncol, 2, (), "A_mul_B"), ("bsbm_A_mul_B4", nrow, bsbm, ncol, 4, (), "A_mul_B"),
        ("bsbm_A_mul_Bn", nrow, bsbm, ncol, 3, (C.c_int(3),), "A_mul_B"),
        ("bcsr_A_mul_B2", nrow, bcsr, ncol, 2, (), "A_mul_B"), ("bcsr_A_mul_B4", nrow, bcsr, ncol, 4, (), "A_mul_B"),
        ("bcsr_A_mul_B8", nrow, bcsr, ncol, 8, (), "A_mul_B"), ("bcsr_A_mul_B8_auto", nrow, bcsr, ncol, 8, (), "A_mul_B"),
        ("bcsr_A_mul_Bn", nrow, bcsr, ncol, 5, (C.c_int(5),), "A_mul_B"),
        ("bcsr_A_mul_B32n", nrow, bcsr, ncol, 32, (C.c_int(32),), "A_mul_B"),
        ("csr_A_mul_Bn", nrow, csr, ncol, 2, (C.c_int(2),), "sdm_A_mul_B"), ("csr_A_mul_Bn", nrow, csr, ncol, 8, (C.c_int(8),), "sdm_A_mul_B"),
        ("bcsr_AA_mul_B", ncol, bcsr, ncol, 1, (), None)]:
    ms, first = timed(name, out_n, A, x_n, k, extra)
    if ref is None:
        base[name] = ms
    ratio = ms / (base[ref] * k) if ref else 1.0
    print("%-22s k=%-2d %8.3f ms   x%.2f of k plain products   first call %9.2f ms%s"
          % (name, k, ms, ratio, first, "   <-- look at this" if ratio > 2.5 else ""), flush=True)
L.fs_release_all()
