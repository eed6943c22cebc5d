"""Writes a synthetic SparseBinaryMatrix file in the reference's format (three longs, then 1-based int32 rows and columns:
sparse.h read_sbm) for libfastsparse_amd/bench_a_mul_b:  python tools/make_sbm.py <file> <nrow> <ncol> <per_row> [seed]"""
import sys

import numpy as np

path, nrow, ncol, per = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(int(sys.argv[5]) if len(sys.argv) > 5 else 1)
nnz = nrow * per
with open(path, "wb") as f:
    np.array([nrow, ncol, nnz], dtype=np.int64).tofile(f)
    (np.repeat(np.arange(nrow, dtype=np.int32), per) + 1).tofile(f)
    (rng.integers(0, ncol, nnz, dtype=np.int32) + 1).tofile(f)
print(path, nrow, ncol, nnz)
