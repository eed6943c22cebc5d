"""Seeded synthetic inputs for the parity tests (numpy only, bit-reproducible).

The generator is a counter-based splitmix64, so a matrix is a pure function of
(seed, shape) and can be regenerated anywhere; golden files only store outputs.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(z):
    z = (np.asarray(z, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _u01(h):
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def read_coo_file(path, valued):
    """Julia-style dump read by read_sbm (sparse.h:112-139) / read_sdm
    (dsparse.h:64-93): 3 x int64 (nrow, ncol, nnz), int32 rows, int32 cols,
    [float64 vals]; indices are 1-based in the file."""
    raw = open(path, "rb").read()
    nrow, ncol, nnz = (int(v) for v in np.frombuffer(raw, np.int64, 3))
    rows = np.frombuffer(raw, np.int32, nnz, 24) - 1
    cols = np.frombuffer(raw, np.int32, nnz, 24 + 4 * nnz) - 1
    vals = np.frombuffer(raw, np.float64, nnz, 24 + 8 * nnz).copy() if valued else None
    return nrow, ncol, rows.astype(np.int32), cols.astype(np.int32), vals


def fixture_sbm():
    return read_coo_file(os.path.join(GOLDEN, "sbm-100-50.data"), False)


def fixture_sdm():
    return read_coo_file(os.path.join(GOLDEN, "sdm-100-50.data"), True)


def synth_coo(seed, nrow, ncol, per_row, *, empty_frac=0.0, dup_frac=0.0, long_row=None,
              shuffle=True, valued=True):
    """COO with `per_row` entries per non-empty row, uniform columns.

    empty_frac : fraction of rows left empty (chosen by hash)
    dup_frac   : fraction of entries that repeat the previous entry's column
    long_row   : (row, length) -> that row gets `length` entries instead
    shuffle    : entries are emitted in a hashed permutation (COO order != row order)
    """
    r = np.arange(nrow, dtype=np.uint64)
    keep = _u01(splitmix64(r * np.uint64(7919) + np.uint64(seed))) >= empty_frac
    lens = np.where(keep, per_row, 0).astype(np.int64)
    if long_row is not None:
        lens[long_row[0]] = long_row[1]
    rows = np.repeat(np.arange(nrow, dtype=np.int32), lens)
    nnz = len(rows)
    k = np.arange(nnz, dtype=np.uint64)
    h = splitmix64(k + np.uint64(seed) * np.uint64(0x100000001B3))
    cols = (h % np.uint64(ncol)).astype(np.int32)
    if dup_frac > 0 and nnz > 1:
        dup = _u01(splitmix64(h)) < dup_frac
        dup[0] = False
        same_row = np.concatenate([[False], rows[1:] == rows[:-1]])
        idx = np.nonzero(dup & same_row)[0]
        cols[idx] = cols[idx - 1]
    vals = (2.0 * _u01(splitmix64(h ^ np.uint64(0xABCDEF))) - 1.0) if valued else None
    if shuffle and nnz:
        perm = np.argsort(splitmix64(k ^ np.uint64(seed + 17)), kind="stable")
        rows, cols = rows[perm], cols[perm]
        if valued:
            vals = vals[perm]
    return (np.ascontiguousarray(rows), np.ascontiguousarray(cols),
            None if vals is None else np.ascontiguousarray(vals))


def x_sin(n, a=7.0, b=0.3):
    """x[i] = sin(a*i + b) (bench_a_mul_b.c:142 uses a=7, b=0.3)"""
    return np.sin(a * np.arange(n, dtype=np.float64) + b)


def x_int(seed, n, lo=-1000, hi=1000):
    """integer-valued doubles in [lo, hi]: any summation order is exact"""
    h = splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(seed * 1315423911))
    return (h % np.uint64(hi - lo + 1)).astype(np.float64) + lo


def X_sin(n, k):
    """X[i,c] = sin(7 i + 17 c + 0.3), row-major (bench_a_mul_b.c:149)"""
    i = np.arange(n, dtype=np.float64)[:, None]
    c = np.arange(k, dtype=np.float64)[None, :]
    return np.ascontiguousarray(np.sin(7.0 * i + 17.0 * c + 0.3))
