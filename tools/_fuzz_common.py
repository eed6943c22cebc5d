"""What tools/fuzz_parity.py and tools/fuzz_dist.py share: the random matrix generator and the bar."""
import numpy as np

TOL = 1e-12


def make(rng, max_nnz=6_000_000):
    nrow = int(rng.choice([1, 7, 300, 5_000, 60_000, 400_000]))
    ncol = int(rng.choice([1, 3, 64, 2_049, 40_000, 300_001]))
    mean = float(rng.choice([0.3, 2, 9, 40]))
    kind = rng.integers(0, 4)
    if kind == 0:
        lens = rng.poisson(mean, nrow)
    elif kind == 1:
        lens = np.minimum((mean / np.maximum(rng.uniform(size=nrow), 1e-6)).astype(np.int64), 50_000)     # heavy tail
    elif kind == 2:
        lens = np.where(rng.uniform(size=nrow) < 0.7, 0, rng.poisson(3 * mean, nrow))                        # mostly empty
    else:
        lens = np.full(nrow, int(mean) + 1)
    lens = lens.astype(np.int64)
    while lens.sum() > max_nnz:
        lens //= 2
    rp = np.zeros(nrow + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    nnz = int(rp[-1])
    band = rng.integers(0, 3)
    if band == 0 or ncol < 16:
        cc = rng.integers(0, ncol, nnz)
    elif band == 1:                                                # banded: columns near the diagonal
        rows = np.repeat(np.arange(nrow), lens)
        cc = (rows * ncol // max(nrow, 1) + rng.integers(-8, 9, nnz)) % ncol
    else:                                                          # few hot columns (duplicates galore)
        cc = rng.integers(0, min(ncol, 5), nnz)
    return nrow, ncol, rp.astype(np.int32), cc.astype(np.int32), rng.uniform(-1, 1, nnz)


def check(got, ref, scale, exact, what, terms=None):
    """terms: number of addends of every output element.  1e-12 * sum |a||x| is the bar -- up to ~4 500 terms: beyond,
    two different orders of the same sum may differ by 2 n 2^-53 sum |a||x| (the a-priori bound of each, reached when all terms
    have one sign -- ncol = 1 makes such rows), and only strict_order (the reference's own order) can promise more."""
    if exact:
        assert np.array_equal(got, ref), what
    else:
        tol = TOL if terms is None else np.maximum(TOL, 2.0 * terms.reshape(terms.shape + (1,) * (scale.ndim - 1)) * 2.0 ** -53)
        bad = np.abs(got - ref) - tol * scale
        assert np.all(bad <= 0), (what, float(bad.max()), None if terms is None else int(terms.reshape(-1)[int(np.argmax(bad.reshape(bad.shape[0], -1).max(1)))]))
