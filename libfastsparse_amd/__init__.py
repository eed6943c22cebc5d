"""libfastsparse_amd -- MI355X (gfx950) implementation of libfastsparse's A_mul_B / At_mul_B path.

The product is the C-ABI shared library `libfastsparse_hip.so` (sources in csrc/, headers in
../include).  This package only builds it and binds it with ctypes for tests and bench.py;
PyTorch is used by callers as a device-memory / stream / torch.distributed provider.
There is no CPU implementation in here: without the library or without a GPU, calls fail.
"""
from . import capi  # noqa: F401
from ._build import build  # noqa: F401

__all__ = ["capi", "build"]
