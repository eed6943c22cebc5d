"""Writes a synthetic SparseBinaryMatrix file in the reference's format (three longs, then 1-based int32 rows and columns:
sparse.h read_sbm; with --values also fp64 values behind them: dsparse.h read_sdm, which makes bench_a_mul_b run its
csr-f64 section) for libfastsparse_amd/bench_a_mul_b:  python tools/make_sbm.py <file> <nrow> <ncol> <per_row> [seed] [--values]"""
import sys

import numpy as np

values = "--values" in sys.argv
argv = [a for a in sys.argv if a != "--values"]
path, nrow, ncol, per = argv[1], int(argv[2]), int(argv[3]), int(argv[4])
rng = np.random.default_rng(int(argv[5]) if len(argv) > 5 else 1)
nnz = nrow * per
with open(path, "wb") as f:
    np.array([nrow, ncol, nnz], dtype=np.int64).tofile(f)
    (np.repeat(np.arange(nrow, dtype=np.int32), per) + 1).tofile(f)
    (rng.integers(0, ncol, nnz, dtype=np.int32) + 1).tofile(f)
    if values:
        rng.uniform(-1.0, 1.0, nnz).tofile(f)
print(path, nrow, ncol, nnz)
