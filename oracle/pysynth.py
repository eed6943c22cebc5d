"""numpy front-end of the CPU synthetic generators (oracle/fs_synth.c) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import pyoracle

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def _lib():
    L = pyoracle.load()
    L.fso_synth_uniform.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, _i32p, _i32p, C.c_void_p]
    L.fso_synth_powerlaw_lengths.argtypes = [C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_int64, _i32p]
    L.fso_synth_fill.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64, _i32p, _i32p, C.c_void_p]
    for f in ("fso_synth_uniform", "fso_synth_powerlaw_lengths", "fso_synth_fill"):
        getattr(L, f).restype = None
    return L


def uniform(nrow, ncol, per_row, seed, row_offset=0, valued=True):
    nnz = nrow * per_row
    rp = np.empty(nrow + 1, np.int32)
    cc = np.empty(nnz, np.int32)
    vv = np.empty(nnz, np.float64) if valued else None
    _lib().fso_synth_uniform(nrow, ncol, per_row, seed, row_offset, rp, cc, None if vv is None else vv.ctypes.data)
    return rp, cc, vv


def powerlaw(nrow, ncol, scale, max_len, seed, row_offset=0, valued=True):
    lens = np.empty(nrow, np.int32)
    _lib().fso_synth_powerlaw_lengths(nrow, float(scale), max_len, seed, row_offset, lens)
    rp = np.zeros(nrow + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    assert rp[-1] <= 2**31 - 1
    rp = rp.astype(np.int32)
    cc = np.empty(int(rp[-1]), np.int32)
    vv = np.empty(int(rp[-1]), np.float64) if valued else None
    _lib().fso_synth_fill(nrow, ncol, seed, row_offset, rp, cc, None if vv is None else vv.ctypes.data)
    return rp, cc, vv
