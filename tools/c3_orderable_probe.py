import sys, ctypes as C, time
sys.path.insert(0, '/root/repo')
import torch
from libfastsparse_amd import capi
L = capi.lib()
L.fs_debug_ldsx_orderable.argtypes = [C.c_void_p, C.c_int]
nrow, ncol, per = 10_000_000, 1_000_000, 64
_, cols, _ = capi.synth_uniform(nrow, ncol, per, 0x5EED0003, valued=False)
rows = torch.arange(nrow, device="cuda", dtype=torch.int32).repeat_interleave(per)
torch.cuda.synchronize(); t0 = time.time()
A = capi.Matrix.from_coo(nrow, ncol, rows, cols, None)
torch.cuda.synchronize(); t1 = time.time()
At = capi.Matrix.from_coo(ncol, nrow, cols, rows, None)
torch.cuda.synchronize(); t2 = time.time()
print("build s", t1 - t0, t2 - t1, "kernels", A.kernel_name(), At.kernel_name(), "orderable", L.fs_debug_ldsx_orderable(A.h, 0), L.fs_debug_ldsx_orderable(At.h, 0))
print("build_ms A ", A.build_ms())
print("build_ms At", At.build_ms())
print("held", A.device_bytes(), At.device_bytes())
