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


def test_partitions():
    sys.path.insert(0, ROOT)
    from libfastsparse_amd import dist as fsd
    assert fsd.even_row_partition(10, 4) == [0, 3, 6, 8, 10]
    rp = np.array([0, 100, 101, 102, 103, 200], np.int64)
    b = fsd.nnz_balanced_partition(rp, 2)
    assert b[0] == 0 and b[-1] == 5 and 1 <= b[1] <= 4
