"""Multi-rank path on CPU: world_size-2 gloo processes run the same sharding / all-gather / transpose-exchange
code the GPU job runs over RCCL (libfastsparse_amd/dist.py), with the oracle injected as the local product."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _synth as S
        from libfastsparse_amd import dist as fsd
        from oracle import pyoracle as O
        n = 3000
        if mode == "powerlaw":
            from oracle import pysynth
            rp, cc, vv = pysynth.powerlaw(n, n, 2.3, 2000, 99)
        else:
            r, c, v = S.synth_coo(1234, n, n, 9, empty_frac=0.1, dup_frac=0.05)
            rp, cc, vv = O.coo_to_csr(n, r, c, v)
        x = S.x_sin(n)
        ref = O.csr_mul(n, rp, cc, vv, x)
        bounds = fsd.nnz_balanced_partition(rp, world) if mode == "powerlaw" else fsd.even_row_partition(n, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        a, b = int(rp[lo]), int(rp[hi])
        lrp = (rp[lo:hi + 1] - a).astype(np.int32)
        lcc, lvv = cc[a:b].copy(), vv[a:b].copy()

        def local_spmv(y_local, x_full):
            y_local.copy_(torch.from_numpy(O.csr_mul(hi - lo, lrp, lcc, lvv, x_full.numpy())))

        op = fsd.ShardedOperator(local_spmv, bounds)
        y = torch.full((n,), -1.0, dtype=torch.float64)
        op.apply(y, torch.from_numpy(x))
        ok = np.array_equal(y.numpy(), ref)                       # rows are independent: bit-exact
        if mode == "powerlaw":
            nnz_per = [int(rp[bounds[i + 1]] - rp[bounds[i]]) for i in range(world)]
            ok = ok and max(nnz_per) - min(nnz_per) <= 2 * int(np.diff(rp).max())
        # transposed direction: exchange entries, build the local shard of A', same operator
        rows_g = np.repeat(np.arange(lo, hi, dtype=np.int32), np.diff(lrp))
        cb = fsd.even_row_partition(n, world)
        tr, tc, tv = fsd.exchange_transpose_entries(torch.from_numpy(rows_g), torch.from_numpy(lcc),
                                                    torch.from_numpy(lvv), cb)
        tlo, thi = cb[rank], cb[rank + 1]
        trp, tcc, tvv = O.coo_to_csr(thi - tlo, tr.numpy().astype(np.int32), tc.numpy().astype(np.int32), tv.numpy())

        def local_tspmv(y_local, x_full):
            y_local.copy_(torch.from_numpy(O.csr_mul(thi - tlo, trp, tcc, tvv, x_full.numpy())))

        opt = fsd.ShardedOperator(local_tspmv, cb)
        z = torch.full((n,), -1.0, dtype=torch.float64)
        opt.apply(z, torch.from_numpy(x))
        rows_all = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
        zref = O.coo_tmul(n, rows_all, cc, vv, x)                  # column sums in ascending-row order
        ok = ok and np.array_equal(z.numpy(), zref)
        # transposed direction, second scheme (bench.py): local A_r' u_r, partial results summed by all-reduce
        rows_l = np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(lrp))

        def local_tspmv2(z_full, u_local):
            z_full.copy_(torch.from_numpy(O.coo_tmul(n, rows_l, lcc, lvv, u_local.numpy())))

        opr = fsd.TransposedShardedOperator(local_tspmv2, bounds)
        z2 = torch.full((n,), -1.0, dtype=torch.float64)
        opr.apply(z2, torch.from_numpy(x))
        scale = O.coo_tmul(n, rows_all, cc, np.abs(vv), np.abs(x))
        ok = ok and bool(np.all(np.abs(z2.numpy() - zref) <= 1e-12 * np.maximum(scale, 1e-300)))
        # asynchronous form used by bench.py: start the exchange, do other work, wait
        y3 = torch.full((n,), -1.0, dtype=torch.float64)
        h = op.gather_async(y3, op.local(y3, torch.from_numpy(x)))
        z3 = torch.full((n,), -1.0, dtype=torch.float64)
        opr.apply_local(z3, torch.from_numpy(x))
        h.wait()
        h2 = opr.reduce_async(z3)
        h2.wait()
        ok = ok and np.array_equal(y3.numpy(), ref) and np.array_equal(z3.numpy(), z2.numpy())
        # the config-5 scheme for A'u (bench.py --workload c5): shard of A' straight from the local CSR, product with
        # the replicated u, all-gather of the z slices -- bit-identical to the column sums in ascending-row order
        br, bc, bv = fsd.build_transposed_shard(torch.from_numpy(lrp), torch.from_numpy(lcc), torch.from_numpy(lvv), lo, cb)
        ok = ok and torch.equal(br, tr) and torch.equal(bc, tc) and torch.equal(bv, tv)
        opg = fsd.TransposedGatherOperator(local_tspmv, cb)
        z4 = torch.full((n,), -1.0, dtype=torch.float64)
        hz = opg.gather_async(z4, opg.local(z4, torch.from_numpy(x)))
        hz.wait()
        ok = ok and np.array_equal(z4.numpy(), zref)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["uniform", "powerlaw"])
def test_two_rank_sharded_product_and_transpose(mode):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _cg_worker(rank, world, port, ret):
    """overlapped exchange + the iterating consumers: ShardedOperator.apply_overlapped (parts of unequal size on every rank,
    empty parts included) and ShardedCG, both schemes, one and two right-hand sides, against the oracle's bsbm_cg / bsbm_cg2
    restatement (oracle/fs_oracle_cg.c; the reference's own KAT size is 100 x 50, test_sparse.c:560-608) on the fixture
    matrix and on a larger power-law one"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _synth as S
        from libfastsparse_amd import dist as fsd
        from oracle import pyoracle as O
        from oracle import pysynth
        ok = True
        why = []
        cases = []
        _, _, rows, cols, _ = S.fixture_sbm()
        cases.append(("sbm-100-50", 100, 50, rows, cols))
        rp, cc, _ = pysynth.powerlaw(2500, 700, 2.3, 400, 7, valued=False)
        cases.append(("powerlaw", 2500, 700, np.repeat(np.arange(2500, dtype=np.int32), np.diff(rp)), cc))
        for name, N, F, rows, cols in cases:
            a_rp, a_cc, _ = O.coo_to_csr(N, rows, cols, None)
            t_rp, t_cc, _ = O.coo_to_csr(F, cols, rows, None)
            rb = fsd.nnz_balanced_partition(a_rp, world)          # unequal row shards (some may be empty at 8 ranks)
            cb = fsd.even_row_partition(F, world)
            lo, hi = rb[rank], rb[rank + 1]
            tlo, thi = cb[rank], cb[rank + 1]
            l_rp = (a_rp[lo:hi + 1] - a_rp[lo]).astype(np.int32)
            l_cc = a_cc[a_rp[lo]:a_rp[hi]].copy()
            lt_rp = (t_rp[tlo:thi + 1] - t_rp[tlo]).astype(np.int32)
            lt_cc = t_cc[t_rp[tlo]:t_rp[thi]].copy()
            l_rows = np.repeat(np.arange(hi - lo, dtype=np.int32), np.diff(l_rp))
            for k in (1, 2):
                def mul(nr, rp_, cc_, out, xin):
                    X = xin.numpy().reshape(-1, k)
                    Y = O.csr_mul_n(nr, rp_, cc_, None, np.ascontiguousarray(X), k) if k > 1 else O.csr_mul(nr, rp_, cc_, None, X[:, 0])
                    out.copy_(torch.from_numpy(np.ascontiguousarray(Y).reshape(-1)))

                def a_local(y_local, x_full):
                    mul(hi - lo, l_rp, l_cc, y_local, x_full)

                def t_local(z_local, u_full):
                    mul(thi - tlo, lt_rp, lt_cc, z_local, u_full)

                def t_partial(z_full, u_local):            # A_r' u_r: the transpose of this rank's rows, full length
                    U = u_local.numpy().reshape(-1, k)
                    Z = np.stack([O.coo_tmul(F, l_rows, l_cc, None, np.ascontiguousarray(U[:, j])) for j in range(k)], 1)
                    z_full.copy_(torch.from_numpy(np.ascontiguousarray(Z).reshape(-1)))

                op_a = fsd.ShardedOperator(a_local, rb, parts=fsd.EvenParts(a_local, hi - lo), k=k)
                op_d = fsd.ShardedOperator(a_local, rb, parts=fsd.EvenParts(a_local, hi - lo), k=k, exchange="direct")
                # (1) the exchange inside the product: 1, 3 and 5 parts give the product with the whole matrix, bit for bit --
                # by padded all-gathers and by direct point-to-point sends of every part's rows
                X = np.ascontiguousarray(np.stack([S.x_int(5 + j, F) for j in range(k)], 1)).reshape(-1)
                ref = (O.csr_mul_n(N, a_rp, a_cc, None, X.reshape(F, k), k) if k > 1 else O.csr_mul(N, a_rp, a_cc, None, X)).reshape(-1)
                for op_x, how in ((op_a, "allgather"), (op_d, "direct")):
                    for nparts in (1, 3, 5):
                        y = torch.full((N * k,), -1.0, dtype=torch.float64)
                        op_x.apply_overlapped(y, torch.from_numpy(X), nparts)
                        if not np.array_equal(y.numpy(), ref):
                            ok = False
                            why.append((name, k, "overlapped", how, nparts))
                # (1b) first contact with a fabric: verify_overlap compares the exchange inside the product with the plain one on every
                # rank (sound here: stays overlapped); an operator that one rank finds unsound goes conservative on ALL ranks --
                # provoked by a local kernel that corrupts a part on the last rank only -- and then still gives the product
                y = torch.full((N * k,), -1.0, dtype=torch.float64)
                chk = op_a.verify_overlap(y, torch.from_numpy(X), 3)
                if not (chk["mode"] == "overlapped" and chk["checked"] and chk["max_rel_diff"] == 0.0 and np.array_equal(y.numpy(), ref)):
                    ok = False
                    why.append((name, k, "verify_overlap", chk))

                class _BadParts(fsd.EvenParts):
                    def run(self, y_local, x_full, part, nparts):
                        super().run(y_local, x_full, part, nparts)
                        if rank == world - 1 and part == nparts - 1 and nparts > 1 and y_local.numel():
                            y_local[-1] += 12345.0          # what a missing stream dependency would look like: a stale row
                op_b = fsd.ShardedOperator(a_local, rb, parts=_BadParts(a_local, hi - lo), k=k)
                chk = op_b.verify_overlap(y, torch.from_numpy(X), 3)
                op_b.apply_overlapped(y, torch.from_numpy(X), 3)
                if not (chk["mode"] == "conservative" and op_b.conservative and np.array_equal(y.numpy(), ref)):
                    ok = False
                    why.append((name, k, "verify_overlap did not fall back", chk))
                # (2) the solvers
                B = np.ascontiguousarray(np.stack([np.sin(0.37 * np.arange(F) + 1.0 + j) for j in range(k)], 1)).reshape(-1)
                xref, itref = O.cg_normal(N, F, rows, cols, B.reshape(F, k) if k > 1 else B, 0.5, 1e-8, two=(k == 2))
                for scheme in ("gather", "reduce"):
                    op_t = fsd.ShardedOperator(t_local, cb, k=k) if scheme == "gather" else \
                        fsd.TransposedShardedOperator(t_partial, rb)
                    cg = fsd.ShardedCG(op_a, op_t, scheme=scheme, nparts=3)
                    xs, it = cg.solve(torch.from_numpy(B), 0.5, 1e-8)
                    err = float(np.max(np.abs(xs.numpy() - np.asarray(xref).reshape(-1))))
                    # both stop at ||r|| <= 1e-8 ||b||; the dots are summed in another order (slices, then ranks), so a long
                    # solve may stop a couple of iterations apart (the reference's own KAT bar is 1e-4, test_sparse.c:598)
                    if not (abs(it - itref) <= max(2, itref // 20) and err <= 1e-7 * max(1.0, float(np.max(np.abs(xref))))):
                        ok = False
                        why.append((name, k, scheme, it, itref, err))
        ret[rank] = (bool(ok), why)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_overlapped_exchange_and_row_sharded_cg(world):
    """the all-gather overlapped INSIDE one product and the iterating consumer on top of it"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cg_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(v[0] for v in dict(ret).values()) and len(ret) == world, dict(ret)


def test_partitions():
    sys.path.insert(0, ROOT)
    from libfastsparse_amd import dist as fsd
    assert fsd.even_row_partition(10, 4) == [0, 3, 6, 8, 10]
    rp = np.array([0, 100, 101, 102, 103, 200], np.int64)
    b = fsd.nnz_balanced_partition(rp, 2)
    assert b[0] == 0 and b[-1] == 5 and 1 <= b[1] <= 4


# ---- bench.py --workload c5 rehearsed on CPU: 8 gloo ranks, the oracle behind bench.py's provider interface -----------
class _Ev:
    def record(self):
        import time
        self.t = time.perf_counter()


class OracleProvider:
    """bench.HipProvider's interface on CPU tensors with the oracle as the local product (test infrastructure)"""
    name, dev = "oracle", "cpu"

    def __init__(self):
        import bench
        from libfastsparse_amd import dist as fsd
        from oracle import pyoracle, pysynth
        self.B, self.O, self.S, self.L, self.fsd = bench, pyoracle, pysynth, pysynth._lib(), fsd
        self.copy_segments = fsd.copy_segments_torch

    def stream(self):
        return None

    def synchronize(self):
        pass

    def event(self):
        return _Ev()

    def elapsed_ms(self, e0, e1):
        return max((e1.t - e0.t) * 1e3, 1e-6)

    def empty(self, n, dtype=None):
        return torch.full((n,), -1.0, dtype=dtype or torch.float64)

    def sin_vector(self, n, a, b):
        return torch.sin(a * torch.arange(n, dtype=torch.float64) + b)

    def uniform(self, nrow, ncol, per, seed, row_offset=0, valued=True):
        rp, cc, vv = self.S.uniform(nrow, ncol, per, seed, row_offset=row_offset, valued=valued)
        return torch.from_numpy(rp), torch.from_numpy(cc), (torch.from_numpy(vv) if valued else None)

    def powerlaw_lengths(self, nrow, row_offset):
        lens = np.empty(nrow, np.int32)
        self.L.fso_synth_powerlaw_lengths(nrow, self.B.C5_SCALE, self.B.C5_MAXLEN, self.B.SEED_C5, row_offset, lens)
        return torch.from_numpy(lens)

    def fill(self, row_ptr, ncol, row_offset, valued=True):
        rp = row_ptr.numpy()
        cc = np.empty(int(rp[-1]), np.int32)
        vv = np.empty(int(rp[-1]), np.float64)
        self.L.fso_synth_fill(len(rp) - 1, ncol, self.B.SEED_C5, row_offset, rp, cc, vv.ctypes.data)
        return torch.from_numpy(cc), torch.from_numpy(vv)

    class _M:
        def __init__(self, nrow, ncol, rp, cc, vv):
            self.nrow, self.ncol, self.rp, self.cc, self.vv, self.nnz = nrow, ncol, rp, cc, vv, len(cc)
            self.rows = np.repeat(np.arange(nrow, dtype=np.int32), np.diff(rp))

    def csr(self, nrow, ncol, rp, cc, vv, transpose=False):
        return self._M(nrow, ncol, rp.numpy(), cc.numpy(), None if vv is None else vv.numpy())

    def coo(self, nrow, ncol, rows, cols, vals):
        rp, cc, vv = self.O.coo_to_csr(nrow, rows.numpy(), cols.numpy(), None if vals is None else vals.numpy())
        return self._M(nrow, ncol, rp, cc, vv)

    def spmv(self, A, y, x, transposed=False):
        if transposed:      # column sums in ascending-row order = what the cached CSR of A' gives
            y.copy_(torch.from_numpy(self.O.coo_tmul(A.ncol, A.rows, A.cc, A.vv, x.numpy())))
        else:
            y.copy_(torch.from_numpy(self.O.csr_mul(A.nrow, A.rp, A.cc, A.vv, x.numpy())))

    spmv_strict = spmv

    def parts(self, A, transposed=False):
        return self.fsd.EvenParts(lambda yl, xf: self.spmv(A, yl, xf, transposed), A.ncol if transposed else A.nrow)

    def kernel_name(self, A, transposed=False):
        return "oracle"

    def candidate_ms(self, A, transposed=False):
        return None

    def release(self):
        pass


def _c5_worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import argparse
        import bench
        from oracle import pyoracle as O
        from oracle import pysynth
        prov = OracleProvider()
        args = argparse.Namespace(c5_rows=n, steps=2, warmup=1, transpose=True, parts=3, no_cpu_baseline=True, cpu_sample_rows=0)
        out = {}
        rec = bench.run_c5(args, prov, world, rank, False, out=out)
        # the whole matrix on every rank (it is small here) with the same generator: the gathered y must be the
        # oracle's product bit for bit (rows are independent), z the column sums in ascending-row order
        rp, cc, vv = pysynth.powerlaw(n, n, bench.C5_SCALE, bench.C5_MAXLEN, bench.SEED_C5)
        x = prov.sin_vector(n, 7.0, 0.3).numpy()
        u = prov.sin_vector(n, 11.0, -0.2).numpy()
        ok = np.array_equal(out["y"].numpy(), O.csr_mul(n, rp, cc, vv, x))
        rows_all = np.repeat(np.arange(n, dtype=np.int32), np.diff(rp))
        ok = ok and out["z"] is not None and np.array_equal(out["z"].numpy(), O.coo_tmul(n, rows_all, cc, vv, u))
        b = out["bounds"]
        per = [int(rp[b[i + 1]] - rp[b[i]]) for i in range(world)]
        ok = ok and b[0] == 0 and b[-1] == n and max(per) - min(per) <= 2 * int(np.diff(rp).max())
        ok = ok and len(set(np.diff(b))) > 1            # the shards really are unequal: the padded async gather ran
        if rank == 0:
            ok = ok and rec is not None and rec["n_gpus"] == world and rec["value"] > 0 and rec["config"]["self_check"]["ok"] \
                and rec["config"]["transpose_error"] is None and rec["config"]["total_nnz"] == int(rp[-1]) \
                and rec["config"]["with_transpose"]["value"] > 0 and rec["config"]["ms_per_step_without_exchanges"] > 0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_bench_c5_workload_rehearsed_on_eight_gloo_ranks():
    """`bench.py --workload c5 --gpus 8 --transpose` with small rows per rank: nnz-balanced cut of a power-law matrix,
    local products, asynchronous all-gather of UNEQUAL y shards, A'u by exchanged row shards of A' + all-gather"""
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_c5_worker, args=(world, _free_port(), 60_000, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _c2_worker(rank, world, port, z_scheme, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import argparse
        import bench
        from oracle import pyoracle as O
        from oracle import pysynth
        prov = OracleProvider()
        n, per = 2500, 16
        args = argparse.Namespace(rows=n, per_row=per, c5_rows=9000, steps=2, warmup=1, transpose=False, strong=False, parts=3,
                                  z_scheme=z_scheme, no_cpu_baseline=True, no_reproducible_cost=True, cpu_sample_rows=0,
                                  spmm_kernel=0, exchange="direct" if z_scheme == "reduce" else "allgather")   # both ways of shipping a part
        out = {}
        rec = bench.run_c2(args, prov, world, rank, False, out=out)
        # weak scaling: the global matrix is (world * n) x n with the same generator row by row; y and z must be the oracle's
        rp, cc, vv = pysynth.uniform(world * n, n, per, bench.SEED_C2)
        x = prov.sin_vector(n, 7.0, 0.3).numpy()
        u = prov.sin_vector(world * n, 11.0, -0.2).numpy()
        ok = np.array_equal(out["y"].numpy(), O.csr_mul(world * n, rp, cc, vv, x))
        rows_all = np.repeat(np.arange(world * n, dtype=np.int32), per)
        zref = O.coo_tmul(n, rows_all, cc, vv, u)
        if out["z_scheme"] == "gather":                     # rows of A' in ascending A-row order: the oracle's bits
            ok = ok and np.array_equal(out["z"].numpy(), zref)
        else:                                               # partial sums per rank, then the all-reduce: rounding apart
            ok = ok and bool(np.all(np.abs(out["z"].numpy() - zref) <= 1e-12 * (1.0 + np.abs(zref))))
        ok = ok and out["z_scheme"] == z_scheme
        also = bench.run_also(args, prov, world, rank, False)
        if rank == 0:
            ex = rec["config"]["exchange"]
            ok = ok and rec["n_gpus"] == world and rec["scaling"] == "weak" and rec["config"]["self_check"]["ok"] \
                and ex["z_scheme"] == z_scheme and rec["config"]["ms_per_step_without_exchanges"] > 0 \
                and ex["bytes_received_per_rank_per_step"]["z_all_reduce_ring"] == 2 * ex["bytes_received_per_rank_per_step"]["z_row_shards_of_At_plus_all_gather"]
            names = [r.get("config", {}).get("workload", r.get("workload", "")) for r in also]
            ok = ok and len(also) == 3 and "strong scaling" in names[0] and "config 5" in names[1] and "ShardedCG" in names[2] \
                and all("error" not in r and r["config"]["self_check"]["ok"] and r["roofline"]["achieved"] > 0 for r in also) \
                and also[0]["scaling"] == "strong" and also[1]["config"]["with_transpose"]["value"] > 0
            ret["names"] = names
            # the line proves its world and names its exchange
            rc = rec["rccl"]
            ok = ok and rc["ranks_counted_by_all_reduce_of_ones"] == world and len(rc["ranks"]) == world \
                and sorted(r["rank"] for r in rc["ranks"]) == list(range(world)) and len(rec["per_rank"]) == world \
                and ex["parts"] == 3 and ex["mode"].startswith("overlapped") and "exchange_fault" not in rec \
                and "WEAK scaling" in rec["config"]["workload"] and rec["aggregate_frac_of_hbm_peak"] > 0 \
                and (z_scheme != "gather" or rec["config"]["ms_per_step_products_independent"] > 0)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _fault_worker(rank, world, port, strict, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FS_DIST_INJECT_MISMATCH="1:2")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import argparse
        import bench
        prov = OracleProvider()
        args = argparse.Namespace(rows=2500, per_row=16, c5_rows=9000, steps=2, warmup=1, transpose=False, strong=False, parts=3,
                                  z_scheme="gather", no_cpu_baseline=True, no_reproducible_cost=True, cpu_sample_rows=0,
                                  spmm_kernel=0, exchange="allgather", strict_exchange=strict)
        try:
            rec = bench.run_c2(args, prov, world, rank, False)
            info = None
        except bench.ExchangeFault as ex:
            rec, info = None, ex.info
        if strict:          # every rank raises (the verdict of verify_overlap is collective); main() turns it into the error line
            seg = info["vectors"]["y"]["first_bad_segment_per_rank"][rank]
            ret[rank] = bool(rec is None and seg["rows_owned_by_rank"] == 1 and seg["part"] == 2 and seg["row_range"][1] > seg["row_range"][0])
        elif rank == 0:     # default: the fault is in the line, the numbers are the conservative exchange's
            f = rec["exchange_fault"]
            seg = f["vectors"]["y"]["first_bad_segment_per_rank"][0]
            ret[rank] = bool(rec["value"] > 0 and rec["config"]["self_check"]["ok"] and seg["rows_owned_by_rank"] == 1 and seg["part"] == 2
                             and rec["config"]["exchange"]["mode"].startswith("conservative") and f["parts"] == 3)
        else:
            ret[rank] = rec is None
    finally:
        dist.destroy_process_group()


def test_bench_error_line_shape():
    """the line main() prints when the headline workload of an N > 1 run raises: "error", the rank, the fault's segment, the exchange
    that was selected -- json-serialisable, no value"""
    import argparse
    import json
    sys.path.insert(0, ROOT)
    import bench
    args = argparse.Namespace(parts=4, exchange="allgather")
    fault = {"vectors": {"y": {"first_bad_segment_per_rank": [{"rows_owned_by_rank": 1, "part": 2, "row_range": [10, 20]}]}}, "parts": 4}
    try:
        raise bench.ExchangeFault(fault)
    except bench.ExchangeFault as ex:
        rec = json.loads(json.dumps(bench.error_record(ex, args, 8, 5)))
    assert rec["value"] is None and rec["n_gpus"] == 8 and rec["error_rank"] == 5 and rec["error_kind"] == "exchange_fault"
    assert rec["exchange_fault"]["vectors"]["y"]["first_bad_segment_per_rank"][0]["part"] == 2
    assert rec["exchange"] == {"parts": 4, "mode": "overlapped", "how": "allgather"} and "--parts 1" in rec["hint"]
    try:
        raise RuntimeError("NCCL error: unhandled system error")
    except RuntimeError as ex:
        rec = bench.error_record(ex, argparse.Namespace(parts=1, exchange="direct"), 2, 0)
    assert rec["error_kind"].startswith("exception") and rec["exchange"]["mode"] == "conservative" and "NCCL error" in rec["error"]


@pytest.mark.parametrize("strict", [False, True])
def test_bench_reports_a_faulty_overlapped_exchange(strict):
    """the first contact with RCCL must be cheap to diagnose.  An overlapped exchange whose part 2 of rank 1
    arrives wrong (injected: FS_DIST_INJECT_MISMATCH) is found by verify_overlap BEFORE the timed loops; by default the line says
    which part / rank / row range (exchange_fault) and carries the conservative exchange's numbers; under --strict-exchange every
    rank raises ExchangeFault, which main() prints as ONE JSON line with "error" and exits 3 (the other shape of the line:
    test_bench_error_line_shape)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_fault_worker, args=(world, _free_port(), strict, ret), nprocs=world, join=True)
    assert ret.get(0) is True and ret.get(1) is True, dict(ret)


@pytest.mark.parametrize("z_scheme", ["gather", "reduce"])
def test_bench_default_workload_and_its_sub_records_rehearsed_on_two_gloo_ranks(z_scheme):
    """`bench.py --gpus 2` = the config-2 line (weak scaling, the exchange inside the products, z by row
    shards of A' + all-gather or by all-reduce) plus the "also" sub-records a multi-GPU run appends: config 2 under strong
    scaling and config 5 across the ranks with and without the all-gather and with A'u -- every vector against the oracle"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_c2_worker, args=(world, _free_port(), z_scheme, ret), nprocs=world, join=True)
    assert ret.get(0) is True and ret.get(1) is True, dict(ret)


def test_bench_refuses_a_world_that_is_not_the_one_asked_for():
    """--gpus 8 under a launcher that started one rank must fail loudly (it used to run one rank and print n_gpus 1);
    without a launcher bench.py starts the ranks itself (exercised on the GPU box)"""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check: on a GPU box the ranks would really run")
def test_bench_starts_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 3` without a launcher: the parent starts 3 rank processes itself (each with RANK /
    WORLD_SIZE / MASTER_* set) and returns their failure; here, without a GPU, every rank stops at the "needs a GPU" check"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # every rank stops at the check -- unless the first one to fail got the others terminated before they reached it
    # (spawn_ranks ends the survivors of a failed rank: they would wait in a collective for ever)
    assert 1 <= (p.stderr + p.stdout).count("bench.py needs a GPU") <= 3
