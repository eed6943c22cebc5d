"""The multi-rank HIP path on ONE GPU: two gloo ranks share the card (no RCCL there), each owns a row shard as an fs_matrix_t,
the local products run on the HIP kernels IN PARTS (fs_spmv_part), the unpack is fs_copy_segments, and on top of that
ShardedOperator.apply_overlapped and ShardedCG (both schemes, one and two right-hand sides) -- checked against the oracle's
product / bsbm_cg / bsbm_cg2 restatement with the whole matrix.  ("the shared-GPU gloo rehearsal for the
HIP path".  With gloo the device tensors of a collective travel through the host; RCCL moves them directly.)"""
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


def _worker(rank, world, port, ret, backend="gloo"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if backend == "nccl":       # ONE rank (RCCL refuses two ranks on one card): the multi-rank code path is forced instead
        os.environ["FS_DIST_FORCE_COLLECTIVES"] = "1"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from libfastsparse_amd import capi
        from libfastsparse_amd import dist as fsd
        from oracle import pyoracle as O
        from oracle import pysynth
        dev = torch.device("cuda", 0)
        prov = bench.HipProvider(dev)
        N, F, per = 2_000_000, 200_000, 4
        rp, cc, _ = pysynth.uniform(N, F, per, 0xC6, valued=False)
        rows = np.repeat(np.arange(N, dtype=np.int32), per)
        rb = fsd.even_row_partition(N, world)
        cb = fsd.even_row_partition(F, world)
        lo, hi = rb[rank], rb[rank + 1]
        tlo, thi = cb[rank], cb[rank + 1]
        l_rp = torch.from_numpy((rp[lo:hi + 1] - rp[lo]).astype(np.int32)).to(dev)
        l_cc = torch.from_numpy(cc[rp[lo]:rp[hi]].copy()).to(dev)
        # rows of A' this rank owns: entries with column in [tlo, thi), in ascending A-row order
        sel = (cc >= tlo) & (cc < thi)
        t_rows = torch.from_numpy((cc[sel] - tlo).astype(np.int32)).to(dev)
        t_cols = torch.from_numpy(rows[sel].copy()).to(dev)
        capi.set_option("binning", 2)      # keep the two-pass copy on these small shards ...
        capi.set_option("bin_rows", 2048)  # ... with many small panels: more than one generation, so the parts really are cut

        def build():
            A = capi.Matrix.from_csr(hi - lo, F, l_rp, l_cc, None, borrow=True)
            A.build_transpose(capi.current_stream())
            At = capi.Matrix.from_coo(thi - tlo, N, t_rows, t_cols, None)
            return A, At
        A, At = bench.in_turns(build, prov, world, rank, False)     # ranks sharing a card build one after the other

        def prepare_two_columns():                                   # ... and so do the k-column copies (device sorts inside)
            A.prepare(2, capi.current_stream())
            A.prepare(2, capi.current_stream(), transposed=True)
            At.prepare(2, capi.current_stream())
        bench.in_turns(prepare_two_columns, prov, world, rank, False)
        ok, why = True, []
        assert A.kernel_name() == "two-pass", A.kernel_name()
        cuts = A.part_rows(3)
        if sum(b > a for a, b in zip(cuts, cuts[1:])) < 2:
            ok = False
            why.append(("not cut", cuts))
        st = capi.current_stream
        for k in (1, 2):
            def a_local(yl, xf):
                (A.spmv(yl, xf, st()) if k == 1 else A.spmm(yl, xf, k, st()))

            def t_local(zl, uf):
                (At.spmv(zl, uf, st()) if k == 1 else At.spmm(zl, uf, k, st()))

            def t_partial(zf, ul):
                (A.spmv(zf, ul, st(), transposed=True) if k == 1 else A.spmm(zf, ul, k, st(), transposed=True))

            parts = fsd.HipParts(A, k=k)                          # k = 2: the two-column sweep in parts (fs_spmm_part)
            op_a = fsd.ShardedOperator(a_local, rb, parts=parts, k=k, copy_segments=prov.copy_segments)
            X = np.ascontiguousarray(np.stack([((np.arange(F) * (3 + j)) % 17 - 8).astype(np.float64) for j in range(k)], 1))
            ref = O.csr_mul_n(N, rp, cc, None, X, k) if k > 1 else O.csr_mul(N, rp, cc, None, X[:, 0])
            for nparts in (1, 3):
                y = torch.full((N * k,), -1.0, dtype=torch.float64, device=dev)
                op_a.apply_overlapped(y, torch.from_numpy(X.reshape(-1)).to(dev), nparts)
                if not np.array_equal(y.cpu().numpy(), np.asarray(ref).reshape(-1)):
                    ok = False
                    why.append((k, "overlapped", nparts))
            B = np.ascontiguousarray(np.stack([np.sin(0.37 * np.arange(F) + 1.0 + j) for j in range(k)], 1))
            xref, itref = O.cg_normal(N, F, rows, cc, B if k > 1 else B[:, 0], 2.0, 1e-8, two=(k == 2))
            for scheme in ("gather", "reduce"):
                op_t = fsd.ShardedOperator(t_local, cb, k=k, copy_segments=prov.copy_segments) if scheme == "gather" else \
                    fsd.TransposedShardedOperator(t_partial, rb)
                xs, it = fsd.ShardedCG(op_a, op_t, scheme=scheme, nparts=3).solve(torch.from_numpy(B.reshape(-1)).to(dev), 2.0, 1e-8)
                err = float(np.max(np.abs(xs.cpu().numpy() - np.asarray(xref).reshape(-1))))
                if not (abs(it - itref) <= max(2, itref // 20) and err <= 1e-7 * max(1.0, float(np.max(np.abs(xref))))):
                    ok = False
                    why.append((k, scheme, it, itref, err))
        if backend == "nccl":
            # the remaining collectives of the path on device tensors: the all-reduce of z partials (scheme "reduce" of config 2),
            # the all-to-all that builds the row shard of A', the plain asynchronous gather
            u = torch.from_numpy(np.cos(0.11 * np.arange(N))).to(dev)
            zt = torch.empty(F, dtype=torch.float64, device=dev)
            op_r = fsd.TransposedShardedOperator(lambda zf, ul: A.spmv(zf, ul, st(), transposed=True), rb)
            op_r.apply_local(zt, u)
            op_r.reduce_async(zt).wait()
            zref = np.asarray(O.coo_tmul(F, rows, cc, None, u.cpu().numpy()))
            zg = torch.empty(F, dtype=torch.float64, device=dev)
            fsd.ShardedOperator(lambda zl, uf: At.spmv(zl, uf, st()), cb).apply(zg, u)
            if not torch.allclose(zt, zg, rtol=0, atol=1e-9) or not np.allclose(zt.cpu().numpy(), zref, rtol=0, atol=1e-9):
                ok = False
                why.append(("all-reduce of z", float((zt - zg).abs().max())))
            r_at, c_at, _ = fsd.build_transposed_shard(l_rp.to(torch.int64), l_cc, None, lo, cb)
            if not (torch.equal(r_at.to(torch.int32), t_rows) and torch.equal(c_at.to(torch.int32), t_cols)):
                ok = False
                why.append(("all-to-all transpose",))
        ret[rank] = (ok, why)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_gloo_ranks_on_one_gpu_overlapped_exchange_and_cg_on_the_hip_kernels():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world and all(v[0] for v in ret.values()), dict(ret)


@pytest.mark.gpu
def test_one_rccl_rank_runs_the_multi_rank_code_path_on_device_tensors():
    """RCCL itself under the same code: a one-rank "nccl" group with FS_DIST_FORCE_COLLECTIVES=1, so that the asynchronous
    all-gathers of the parts (on RCCL's stream, ordered against the product's stream by the work handles), the unpack, the
    all-reduces of the dots and of z and the all-to-all of the transpose build all run on device tensors.  What one rank cannot
    show is the transport between GPUs."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(1, _free_port(), ret, "nccl"), nprocs=1, join=True)
    assert len(ret) == 1 and all(v[0] for v in ret.values()), dict(ret)


@pytest.mark.gpu
def test_bench_runs_its_multi_rank_plan_on_one_rccl_rank():
    """`bench.py` with FS_BENCH_FORCE_MULTI=1: the N > 1 plan (config 2 weak + strong, config 5 across the ranks, the row-sharded
    CG) on a ONE-rank "nccl" group -- init_process_group with device_id, the device-side timing all-reduce and barrier, the
    all-gathers inside the products, every sub-record's self-check.  Reduced sizes; full size: profiles/r03_bench_n_gt_1_plan_on_one_rccl_rank.json"""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, FS_BENCH_FORCE_MULTI="1", FS_DIST_FORCE_COLLECTIVES="1", FS_BENCH_WATCHDOG="300")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--rows", "2000000", "--c5-rows", "3000000"], env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    recs = [rec] + rec["also"]
    assert len(recs) >= 4, [r["config"]["workload"][:40] for r in recs]
    native = [r for r in recs if "native_one_process_path" in r]
    for r in recs:
        if "native_one_process_path" not in r:
            assert r["config"]["self_check"]["ok"], r["config"]
    assert "all-gather" in rec["config"]["exchange"]["y"]
    # the line proves its world -- the rank count from an all-reduce of ones, every rank's device
    rc = rec["rccl"]
    assert rc["ranks_counted_by_all_reduce_of_ones"] == 1 and rc["world_size"] == 1 and rc["backend"].startswith("nccl") \
        and rc["ranks"][0]["rank"] == 0 and "pci_bus_id" in rc["ranks"][0] and rc["rccl_version"], rc
    assert len(rec["per_rank"]) == 1 and rec["aggregate_frac_of_hbm_peak"] > 0 and rec["config"]["exchange"]["parts"] == 4
    # and the same GPU through the ONE-PROCESS C path (fs_dist_*, ncclCommInitAll on a one-device group here), both exchanges
    assert len(native) == 1, [r.get("workload") for r in recs]
    nat = native[0]["native_one_process_path"]
    for tag in ("overlapped_4_parts", "conservative_1_part"):
        assert "error" not in nat[tag] and nat[tag]["uses_rccl"] and nat[tag]["self_check"]["ok"] and nat[tag]["value"] > 0, nat[tag]
    assert nat["conservative_1_part"]["conservative_exchange"] and not nat["overlapped_4_parts"]["conservative_exchange"]
    assert nat["both_exchanges_agree"]
