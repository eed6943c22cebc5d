#!/usr/bin/env python3
"""Randomised differential run of the REFERENCE-NAMED entry points (host structs, host vectors -- what an unmodified C caller
does): random COO triples in random entry order with duplicates, through every product family of sparse.h / dsparse.h / csr.h /
cbcsr.h (A_mul_B, At_mul_B, sdm_*, bsbm_A_mul_B/_B2/_B4/_Bn, bsdm_A_mul_B, csr_A_mul_B/_Bn, csr_At_mul_B, bcsr_A_mul_B/_B2/_B4/_B8/
_B8_auto/_Bn/_B32n, bcsr_At_mul_B, bcsr_AA_mul_B, parallel_bcsr_AA_mul_B, cbcsr_A_mul_B), each against the oracle's restatement of
the same reference function (tests/_cases.py: the interface the golden tests use).  Pattern-only with integer x: bit for bit;
otherwise the row-scaled bound of tools/_fuzz_common.py.  FASTSPARSE_NGPU=3 FASTSPARSE_DEVICES=0,0,0 in the environment runs the
same calls across three virtual ranks.     python tools/fuzz_dropin.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import psutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402,F401  (before the library: it brings the HIP runtime the library binds to)

import _cases  # noqa: E402
import _hipbackend as H  # noqa: E402
from _fuzz_common import check, make  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed)
ORA = _cases.OracleBackend()
# FS_FUZZ_REF=1: the expected results come from the REAL reference library (oracle/_ref/libfsref.so: travels to the GPU box as a built file)
# instead of the oracle's restatement (the CPU suite pins the two to each other bit for bit)
REF = None
if os.environ.get("FS_FUZZ_REF") == "1":
    import _refbind
    if not _refbind.available():
        raise SystemExit("FS_FUZZ_REF=1 but oracle/_ref/libfsref.so is missing")
    REF = _cases.RefBackend()
STRICT = os.environ.get("FS_STRICT_ORDER") == "1"      # storage-order sums: the oracle's bits for ARBITRARY x and values, every family

FAMILIES = ["coo", "coo_t", "csr", "csr_t", "csr_n", "bin_n", "aa", "aa_parallel", "blocked", "blocked_n", "blocked_valued", "cbcsr"]
BIN_N = [("bcsr_A_mul_B2", 2), ("bcsr_A_mul_B4", 4), ("bcsr_A_mul_B8", 8), ("bcsr_A_mul_B8_auto", 8), ("bcsr_A_mul_Bn", 3), ("bcsr_A_mul_Bn", 8),
         ("bcsr_A_mul_B32n", 5), ("bcsr_A_mul_B32n", 32)]
BLK_N = [("bsbm_A_mul_B2", 2), ("bsbm_A_mul_B4", 4), ("bsbm_A_mul_Bn", 3), ("bsbm_A_mul_Bn", 7)]

class DeviceVectors(H.HipDropinBackend):
    """the same calls with x and y in HBM: the entry points take device pointers in place (hipPointerGetAttributes decides per vector)"""

    def _call(self, name, nout, A, x, *extra):
        import ctypes as C
        xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).reshape(-1)).cuda()
        yd = torch.full((max(nout, 1),), -1.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        f = getattr(self.L, name)
        f.restype = None
        f(C.c_void_p(yd.data_ptr()), A if isinstance(A, C._Pointer) else C.byref(A), C.c_void_p(xd.data_ptr()), *extra)
        self.L.fs_invalidate(A if isinstance(A, C._Pointer) else C.byref(A))
        return yd.cpu().numpy()[:nout]


t_end = time.time() + budget
cases = 0
seen = {}
while time.time() < t_end:
    fam = FAMILIES[rng.integers(0, len(FAMILIES))]
    small = fam.startswith("blocked") or fam == "cbcsr"             # their structs are never freed (the reference has no free_bsbm)
    nrow, ncol, rp, cc, vv = make(rng, 300_000 if small else 3_000_000)
    nnz = len(cc)
    rows = np.repeat(np.arange(nrow, dtype=np.int32), np.diff(rp))
    order = rng.permutation(nnz)                                      # COO entry order is the caller's: any
    rows, cols, vals = np.ascontiguousarray(rows[order]), np.ascontiguousarray(cc[order]), np.ascontiguousarray(vv[order])
    integer = bool(rng.integers(0, 2))
    in_hbm = bool(rng.integers(0, 3) == 0)
    be = DeviceVectors() if in_hbm else H.HipDropinBackend()
    what = dict(hbm=in_hbm, family=fam, nrow=nrow, ncol=ncol, nnz=nnz, integer=integer, seed=seed, case=cases, ngpu=os.environ.get("FASTSPARSE_NGPU", "1"))

    def vec(n, k=1):
        if integer:
            return rng.integers(-100, 101, (n, k) if k > 1 else n).astype(np.float64)
        a = np.sin(7.0 * np.arange(n * k, dtype=np.float64) + 0.3)
        return a.reshape(n, k) if k > 1 else a

    def both(method, *args, valued, out_terms):
        got = getattr(be, method)(*args)
        ref = getattr(REF or ORA, method)(*args)
        # the scale: the same call on |values|, |x| through the oracle
        a = list(args)
        for i, v in enumerate(a):
            if isinstance(v, np.ndarray) and v.dtype == np.float64:
                a[i] = np.abs(v)
        scale = getattr(ORA, method)(*a)
        exact = (integer and not valued) or STRICT
        check(np.asarray(got), np.asarray(ref), np.asarray(scale), exact, what, out_terms)

    lens = np.bincount(rows, minlength=nrow).astype(np.float64)
    clen = np.bincount(cols, minlength=ncol).astype(np.float64)
    valued = bool(rng.integers(0, 2))
    v = vals if valued else None
    if fam == "coo":
        both("coo_mul", nrow, ncol, rows, cols, v, vec(ncol), valued=valued, out_terms=lens)
    elif fam == "coo_t":
        both("coo_tmul", nrow, ncol, rows, cols, v, vec(nrow), valued=valued, out_terms=clen)
    elif fam == "csr":
        both("csr_mul", nrow, ncol, rows, cols, v, vec(ncol), valued=valued, out_terms=lens)
    elif fam == "csr_t":
        got = be.transposed_csr_mul(nrow, ncol, rows, cols, v, u := vec(nrow))
        # csr_At_mul_B adds a column's terms in CSR order = ascending row, within a row the caller's order: new_csr is stable
        o2 = np.argsort(rows, kind="stable")
        ref = ORA.coo_tmul(nrow, ncol, rows[o2], cols[o2], None if v is None else v[o2], u)
        sc = ORA.coo_tmul(nrow, ncol, rows[o2], cols[o2], None if v is None else np.abs(v[o2]), np.abs(u))
        check(got, ref, sc, (integer and not valued) or STRICT, what, clen)
    elif fam == "csr_n":
        k = int(rng.choice([2, 3, 4, 8, 17, 32]))
        both("csr_mul_n", nrow, ncol, rows, cols, vals, vec(ncol, k), k, "csr_A_mul_Bn", valued=True, out_terms=lens)
    elif fam == "bin_n":
        name, k = BIN_N[rng.integers(0, len(BIN_N))]
        what["name"] = name
        both("csr_mul_n", nrow, ncol, rows, cols, None, vec(ncol, k), k, name, valued=False, out_terms=lens)
    elif fam in ("aa", "aa_parallel"):
        x = vec(ncol)
        got = be.aa_mul(nrow, ncol, rows, cols, x, fam == "aa_parallel")
        ref = (REF or ORA).aa_mul(nrow, ncol, rows, cols, x, False)
        sc = ORA.aa_mul(nrow, ncol, rows, cols, np.abs(x), False)
        if integer or STRICT:
            assert np.array_equal(got, ref), what
        else:
            tol = np.maximum(1e-12, 2.0 * (clen + lens.max(initial=0.0)) * 2.0 ** -53) * 2.0
            assert np.all(np.abs(got - ref) <= tol * sc), (what, float(np.max(np.abs(got - ref) - tol * sc)))
    elif fam == "blocked":
        bs = int(rng.choice([1, 8, 64, 1024]))
        what["bs"] = bs
        both("blocked_mul", nrow, ncol, rows, cols, None, bs, vec(ncol), 1, "bsbm_A_mul_B", valued=False, out_terms=lens)
    elif fam == "blocked_n":
        bs = int(rng.choice([1, 8, 64, 1024]))
        name, k = BLK_N[rng.integers(0, len(BLK_N))]
        what["bs"], what["name"] = bs, name
        both("blocked_mul", nrow, ncol, rows, cols, None, bs, vec(ncol, k), k, name, valued=False, out_terms=lens)
    elif fam == "blocked_valued":
        bs = int(rng.choice([1, 8, 64, 1024]))
        what["bs"] = bs
        both("blocked_mul", nrow, ncol, rows, cols, vals, bs, vec(ncol), 1, "bsdm_A_mul_B", valued=True, out_terms=lens)
    else:
        cbs = int(rng.choice([1, 16, 1000, 100_000]))
        while -(-ncol // cbs) * (nrow + 1) > 20_000_000:             # the format holds a row_ptr of nrow + 1 ints PER column block (cbcsr.h:16-65)
            cbs *= 4
        what["cbs"] = cbs
        both("cbcsr_mul", nrow, ncol, rows, cols, cbs, vec(ncol), valued=False, out_terms=lens)
    seen[fam] = seen.get(fam, 0) + 1
    cases += 1
    if cases % 25 == 0:
        print("%d cases ok" % cases, flush=True)
    if cases % 200 == 0:
        be.L.fs_release_all()
    if REF:
        REF.R._keep.clear()
    if psutil.Process().memory_info().rss > 64 << 30:                 # (never near the box's cap again)
        raise SystemExit("fuzz_dropin: more than 64 GiB resident after %r: stopping" % (what,))
print("fuzz_dropin: %d cases, all within the bars (seed %d, FASTSPARSE_NGPU=%s, expected results from %s): %s"
      % (cases, seed, os.environ.get("FASTSPARSE_NGPU", "1"), "the REAL reference library" if REF else "the oracle", ", ".join("%s %d" % kv for kv in sorted(seen.items()))))
