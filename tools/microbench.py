#!/usr/bin/env python3
"""GPU micro-benchmarks used while tuning (not part of the product): times the SpMV kernel variants on
BASELINE config 2/3/4 shapes and a few pure-bandwidth probes.  Writes one JSON line per measurement."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[0], ts[len(ts) // 2], sum(ts) / len(ts)


def report(out, name, bytes_, t):
    rec = {"name": name, "ms_min": t[0], "ms_med": t[1], "ms_mean": t[2], "GBs_med": bytes_ / t[1] / 1e6,
           "frac_of_8TBs": bytes_ / t[1] / 1e6 / 8000}
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n")
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/microbench.jsonl")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--what", default="probes,c2,c3,c4")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--tile-cols", type=int, default=0)
    ap.add_argument("--tiled-flags", type=int, default=0)
    ap.add_argument("--tiling", type=int, default=1)
    ap.add_argument("--ncols", type=int, nargs="*", default=None)
    ap.add_argument("--gate-kb", type=int, default=8192)
    ap.add_argument("--bin-rows", type=int, default=0)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    out = open(args.out, "a")
    st = capi.current_stream()
    what = args.what.split(",")
    capi.set_option("tile_rows", args.tile_rows)
    capi.set_option("tile_cols", args.tile_cols)
    capi.set_option("tiled_flags", args.tiled_flags)
    n = args.rows
    if "probes" in what:
        a = torch.empty(240_000_000, dtype=torch.float64, device="cuda").normal_()
        b = torch.empty_like(a)
        report(out, "copy_1.92GB(read+write)", 2 * a.numel() * 8, timeit(lambda: b.copy_(a)))
        report(out, "sum_1.92GB(read only)", a.numel() * 8, timeit(lambda: a.sum()))
        del b
        for tab_mb in (2, 8, 80, 640):
            tn = tab_mb * 1_000_000 // 8
            table = a[:tn]
            idx = torch.randint(0, tn, (160_000_000,), device="cuda", dtype=torch.int32)
            report(out, f"torch_gather_160M_from_{tab_mb}MB(8B useful per gather)", 160_000_000 * (8 + 4 + 8),
                   timeit(lambda: torch.index_select(table, 0, idx), iters=5, warm=1))
            del idx
        del a
    if "hybrid" in what:
        # does the gather-bound tiled kernel overlap with the HBM-bound two-pass pair?  rows split in two handles,
        # one forced tiled, one forced two-pass, products launched on two streams
        frac = args.bin_rows / 100.0 if args.bin_rows else 0.5
        capi.set_option("bin_rows", 0)
        n1 = int(n * frac)
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        rp2 = (rp[n1:] - rp[n1]).contiguous()
        capi.set_option("tiling", 2)
        A1 = capi.Matrix.from_csr(n1, n, rp[:n1 + 1].contiguous(), cc[:n1 * 16], vv[:n1 * 16], borrow=True)
        capi.set_option("tiling", 1)
        capi.set_option("binning", 2)
        A2 = capi.Matrix.from_csr(n - n1, n, rp2, cc[n1 * 16:], vv[n1 * 16:], borrow=True)
        capi.set_option("binning", 1)
        print("kernels", A1.kernel_name(), A2.kernel_name(), flush=True)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        B = 12 * n * 16 + 4 * (n + 1) + 16 * n

        def both():
            ev = torch.cuda.Event()
            ev.record()
            s1.wait_event(ev)
            s2.wait_event(ev)
            A1.spmv(y[:n1], x, s1.cuda_stream)
            A2.spmv(y[n1:], x, s2.cuda_stream)
            torch.cuda.current_stream().wait_stream(s1)
            torch.cuda.current_stream().wait_stream(s2)
        report(out, f"c2_hybrid_tiled{frac:.2f}_two_pass_concurrent", B, timeit(both))
        report(out, f"c2_hybrid_tiled_part_alone", B, timeit(lambda: A1.spmv(y[:n1], x, st)))
        report(out, f"c2_hybrid_two_pass_part_alone", B, timeit(lambda: A2.spmv(y[n1:], x, st)))
        del A1, A2
    if "hybridmask" in what:
        # Two different bottlenecks at once?  The tiled kernel is bound by L2 gather requests, the two-pass pair by HBM bytes.
        # Rows split between a tiled handle and a two-pass handle, each on a stream masked to its own set of CUs
        # (hipExtStreamCreateWithCUMask), so that the two really run side by side (unmasked, two 1024-thread workgroups
        # with > 100 KB of LDS each cannot share a CU and the launches take turns).
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        B = 12 * n * 16 + 4 * (n + 1) + 16 * n
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        words = (ncu + 31) // 32

        def masked_stream(pred):
            m = (C.c_uint32 * words)()
            for cu in range(ncu):
                if pred(cu):
                    m[cu // 32] |= 1 << (cu % 32)
            st_ = C.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(st_), words, m)
            assert rc == 0, rc
            return st_

        for frac, cu_share, pattern in ((0.45, 0.5, "alt"), (0.45, 0.5, "half"), (0.35, 0.375, "alt"), (0.55, 0.625, "alt"), (0.5, 0.5, "alt")):
            n1 = int(n * frac) // 1024 * 1024
            capi.set_option("tiling", 2)
            A1 = capi.Matrix.from_csr(n1, n, rp[:n1 + 1].contiguous(), cc[:n1 * 16], vv[:n1 * 16], borrow=True)
            capi.set_option("tiling", 1)
            capi.set_option("binning", 2)
            A2 = capi.Matrix.from_csr(n - n1, n, (rp[n1:] - rp[n1]).contiguous(), cc[n1 * 16:], vv[n1 * 16:], borrow=True)
            capi.set_option("binning", 1)
            k8 = int(round(cu_share * 8))
            if pattern == "alt":      # cu % 8 < k8 -> tiled: the same share of every XCD under either CU numbering
                s1 = masked_stream(lambda cu: cu % 8 < k8)
                s2 = masked_stream(lambda cu: cu % 8 >= k8)
            else:                     # contiguous halves
                s1 = masked_stream(lambda cu: cu < int(ncu * cu_share))
                s2 = masked_stream(lambda cu: cu >= int(ncu * cu_share))
            torch.cuda.synchronize()

            def both():
                A1.spmv(y[:n1], x, s1.value)
                A2.spmv(y[n1:], x, s2.value)

            def wall(fn, iters=20):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / iters * 1e3
            t_both = wall(both)
            t1 = wall(lambda: A1.spmv(y[:n1], x, s1.value))
            t2 = wall(lambda: A2.spmv(y[n1:], x, s2.value))
            t1f = wall(lambda: A1.spmv(y[:n1], x, st))
            t2f = wall(lambda: A2.spmv(y[n1:], x, st))
            rec = {"name": f"c2_hybrid_cu_masked_rows{frac}_cus{cu_share}_{pattern}", "ms_both_concurrent": t_both,
                   "ms_tiled_part_on_its_cus": t1, "ms_two_pass_part_on_its_cus": t2, "ms_tiled_part_all_cus": t1f,
                   "ms_two_pass_part_all_cus": t2f, "kernels": [A1.kernel_name(), A2.kernel_name()],
                   "GBs_both": B / t_both / 1e6, "frac_of_8TBs": B / t_both / 1e6 / 8000}
            print(json.dumps(rec), flush=True)
            out.write(json.dumps(rec) + "\n")
            del A1, A2
        del rp, cc, vv
    if "wgs" in what:
        # pass-1 time of the two-pass pair against the number of persistent workgroups (= the share size): HBM channel camping?
        capi.set_option("binning", 2)
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        for w in (256, 255, 254, 253, 252, 250, 248, 247, 240, 232, 224, 256):
            capi.set_option("bin_wgs", w)
            report(out, f"c2_two_pass_pass1_wgs{w}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
        capi.set_option("bin_wgs", 0)
        capi.set_option("binning", 1)
        del A
    if "sortcols" in what:
        # does the order of a row's entries matter to the two-pass kernels?  config 2 with each row's columns sorted
        capi.set_option("binning", 2)
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        report(out, "c2_two_pass_random_order_in_rows", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
        del A
        cs, _ = torch.sort(cc.view(n, 16), dim=1)
        cs = cs.contiguous().view(-1)
        A = capi.Matrix.from_csr(n, n, rp, cs, vv, borrow=True)
        report(out, "c2_two_pass_sorted_columns_in_rows", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
        del A
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        report(out, "c2_two_pass_random_order_in_rows_again", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
        del A
        A = capi.Matrix.from_csr(n, n, rp, cs, vv, borrow=True)
        report(out, "c2_two_pass_sorted_columns_in_rows_again", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
        del A
        capi.set_option("binning", 1)
    if "tiledab" in what:
        # the tiled kernel alone on the config-2 shape, valued and pattern-only (A/B runs of kernel changes)
        capi.set_option("tiling", 2)
        for valued in (True, False):
            rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002, valued=valued)
            A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
            x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            for fl in (0, 1, 0):
                capi.set_option("tiled_flags", fl)
                report(out, f"c2_{'f64' if valued else 'pattern'}_{A.kernel_name()}_flags{fl}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
            capi.set_option("tiled_flags", 0)
            del A, rp, cc, vv
        capi.set_option("tiling", args.tiling)
    if "bin" in what:
        # two-pass kernels against the tiled kernel on the config-2 shape (valued and pattern-only), k = 1, 2, A'
        capi.set_option("bin_rows", args.bin_rows)
        capi.set_option("tiling", 2)
        capi.set_option("binning", 2)
        ncol = args.ncols[0] if args.ncols else n
        for valued in (True, False):
            rp, cc, vv = capi.synth_uniform(n, ncol, 16, 0x5EED0002, valued=valued)
            t0 = time.time()
            A = capi.Matrix.from_csr(n, ncol, rp, cc, vv, borrow=True)
            torch.cuda.synchronize()
            print("build s", time.time() - t0, "kernel", A.kernel_name(), flush=True)
            x = torch.sin(7.0 * torch.arange(ncol, device="cuda", dtype=torch.float64) + 0.3)
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            tag = "f64" if valued else "pattern"
            for kern, label, fl in ((7, "two_pass", 0), (6, "tiled", 0), (7, "two_pass_p2_nt_loads", 4), (7, "two_pass_p1_plain_stores", 8),
                                    (7, "two_pass_p1_nt_loads", 16), (7, "two_pass", 0)):
                capi.set_option("spmv_kernel", kern)
                capi.set_option("bin_flags", fl)
                report(out, f"c2_{tag}_ncol{ncol}_{label}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
            capi.set_option("bin_flags", 0)
            capi.set_option("spmv_kernel", 0)
            if valued:
                X = torch.sin(torch.arange(ncol * 2, device="cuda", dtype=torch.float64)).reshape(ncol, 2)
                Y = torch.empty(n, 2, dtype=torch.float64, device="cuda")
                report(out, f"c2_{tag}_spmm_k2_auto", A.algorithmic_bytes(2), timeit(lambda: A.spmm(Y, X, 2, st), iters=5, warm=1))
                del X, Y
            del A, rp, cc, vv
        capi.set_option("tiling", args.tiling)
        capi.set_option("binning", 1)
    if "c2" in what:
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        B = A.algorithmic_bytes()
        for kern, label in ((0, "auto:" + A.kernel_name()), (1, "stream_nt"), (2, "vector")):
            capi.set_option("spmv_kernel", kern)
            report(out, f"c2_csr_f64_{label}", B, timeit(lambda: A.spmv(y, x, st)))
        capi.set_option("spmv_kernel", 0)
        t0 = time.time()
        A.build_transpose(st)
        torch.cuda.synchronize()
        print("transpose build s", time.time() - t0, flush=True)
        report(out, "c2_csr_f64_At_mul_B", B, timeit(lambda: A.spmv(y, x, st, transposed=True)))
        # locality probe: same matrix shape but columns within a 1M-wide band around the diagonal block
        if "c4" in what:
            k = 32
            X = torch.sin(torch.arange(n * k, device="cuda", dtype=torch.float64)).reshape(n, k)
            Y = torch.empty(n, k, dtype=torch.float64, device="cuda")
            report(out, "c4_spmm_k32", A.algorithmic_bytes(k), timeit(lambda: A.spmm(Y, X, k, st), iters=5, warm=1))
            del X, Y
            for k in (2, 4, 8):
                X = torch.sin(torch.arange(n * k, device="cuda", dtype=torch.float64)).reshape(n, k)
                Y = torch.empty(n, k, dtype=torch.float64, device="cuda")
                report(out, f"spmm_k{k}", A.algorithmic_bytes(k), timeit(lambda: A.spmm(Y, X, k, st), iters=5, warm=1))
                del X, Y
        del A, rp, cc, vv
    if "c3" in what:
        nrow, ncol = n, max(n // 10, 1)
        rp, cc, _ = capi.synth_uniform(nrow, ncol, 64, 0x5EED0003, valued=False)
        A = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
        x = torch.randint(-1000, 1001, (ncol,), device="cuda").to(torch.float64)
        y = torch.empty(nrow, dtype=torch.float64, device="cuda")
        for kern, label in ((0, "auto:" + A.kernel_name()), (1, "stream_nt")):
            capi.set_option("spmv_kernel", kern)
            report(out, f"c3_bcsr_{label}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
        capi.set_option("spmv_kernel", 0)
        t0 = time.time()
        A.build_transpose(st)
        torch.cuda.synchronize()
        print("c3 transpose build s", time.time() - t0, "kernel", A.kernel_name(True), "builder timed", A.candidate_ms(True),
              "forward", A.candidate_ms(), flush=True)
        u = torch.randint(-1000, 1001, (nrow,), device="cuda").to(torch.float64)
        z = torch.empty(ncol, dtype=torch.float64, device="cuda")
        report(out, f"c3_bcsr_At_mul_B_auto:{A.kernel_name(True)}", A.algorithmic_bytes(), timeit(lambda: A.spmv(z, u, st, transposed=True)))
        del A
        for opt, label in (("ldsx", "lds_staged"), ("tiling", "tiled"), ("binning", "two_pass")):
            capi.set_option(opt, 2)
            A = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
            capi.set_option(opt, 1)
            report(out, f"c3_bcsr_forced_{label}:{A.kernel_name()}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
            del A
        capi.set_option("spmv_kernel", 0)
    if "hostvec" in what:
        # products with HOST vectors (fs_spmv_host): config 3's shape, A (y 80 MB down) and A' (x 80 MB up), wall time per call
        import numpy as np
        nrow, ncol = n, max(n // 10, 1)
        rp, cc, _ = capi.synth_uniform(nrow, ncol, 64, 0x5EED0003, valued=False)
        A = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
        A.build_transpose(st)
        torch.cuda.synchronize()
        xh = np.round(1000 * np.sin(np.arange(ncol) * 0.7))
        uh = np.round(1000 * np.cos(np.arange(nrow) * 0.3))
        yh, zh = np.empty(nrow), np.empty(ncol)
        for tr, (o, i) in ((False, (yh, xh)), (True, (zh, uh))):
            A.spmv_host(o, i, transposed=tr)
            ts = []
            for _ in range(10):
                t0 = time.time()
                A.spmv_host(o, i, transposed=tr)
                ts.append((time.time() - t0) * 1e3)
            ts.sort()
            od = torch.empty(o.size, dtype=torch.float64, device="cuda")
            A.spmv(od, torch.from_numpy(i).cuda(), st, transposed=tr)
            rec = {"name": "c3_%s_host_vectors:%s" % ("At_mul_B" if tr else "A_mul_B", A.kernel_name(tr)), "ms_min": ts[0], "ms_med": ts[5],
                   "path": capi.lib().fs_debug_last_host_path(), "host_chunks": os.environ.get("FS_HOST_CHUNKS", "8"),
                   "equal_to_device_vector_product": bool(np.array_equal(od.cpu().numpy(), o))}
            print(json.dumps(rec), flush=True)
            out.write(json.dumps(rec) + "\n")
        del A
    if "cols" in what:
        # how the product behaves as x shrinks (column count), rows and non-zeros as config 2
        capi.set_option("tiling", args.tiling)
        for ncol in (args.ncols or (131072, 262144, 393216, 524288, 1048576, 4194304)):
            rp, cc, vv = capi.synth_uniform(n, ncol, 16, 0x5EED0002)
            A = capi.Matrix.from_csr(n, ncol, rp, cc, vv, borrow=True)
            x = torch.sin(7.0 * torch.arange(ncol, device="cuda", dtype=torch.float64) + 0.3)
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            print("ncol", ncol, "builder timed", A.candidate_ms(), flush=True)
            for kern, label in ((0, "auto:" + A.kernel_name()),):
                capi.set_option("spmv_kernel", kern)
                report(out, f"c2rows_ncol{ncol}_{label}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st), iters=10))
            capi.set_option("spmv_kernel", 0)
            del A, rp, cc, vv
    if "colsspmm" in what:
        # multi-column products as x shrinks: one sweep per column (LDS-staged copy) against the row kernel
        for ncol in (args.ncols or (131072, 262144, 524288, 786432)):
            rp, cc, vv = capi.synth_uniform(n, ncol, 16, 0x5EED0002)
            A = capi.Matrix.from_csr(n, ncol, rp, cc, vv, borrow=True)
            for k in (2, 4, 8):
                X = torch.sin(torch.arange(ncol * k, device="cuda", dtype=torch.float64)).reshape(ncol, k)
                Y = torch.empty(n, k, dtype=torch.float64, device="cuda")
                for sk, label in ((0, "auto"), (1, "row_kernel")):
                    capi.set_option("spmm_kernel", sk)
                    report(out, f"c2rows_ncol{ncol}_{A.kernel_name()}_spmm_k{k}_{label}", A.algorithmic_bytes(k),
                           timeit(lambda: A.spmm(Y, X, k, st), iters=5, warm=1))
                capi.set_option("spmm_kernel", 0)
                del X, Y
            del A, rp, cc, vv
    if "banded" in what:
        # structured columns (FEM / stencil like): 16 per row within +-w of the diagonal; which kernel does the builder keep, how fast
        for w in (1000, 100000):
            rows = torch.arange(n, device="cuda", dtype=torch.int64).repeat_interleave(16)
            cc = (rows + torch.randint(-w, w + 1, (n * 16,), device="cuda")).clamp_(0, n - 1)
            cc = cc.view(n, 16).sort(dim=1).values.reshape(-1).to(torch.int32)
            rp = (torch.arange(n + 1, device="cuda", dtype=torch.int64) * 16).to(torch.int32)
            vv = torch.rand(n * 16, device="cuda", dtype=torch.float64)
            del rows
            A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
            x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            print("banded", w, "builder timed", A.candidate_ms(), flush=True)
            report(out, f"banded_w{w}_auto:{A.kernel_name()}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st), iters=10))
            del A, rp, cc, vv
    if "spmm" in what:
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        for k in (2, 3, 4, 8, 32):
            X = torch.sin(torch.arange(n * k, device="cuda", dtype=torch.float64)).reshape(n, k)
            Y = torch.empty(n, k, dtype=torch.float64, device="cuda")
            report(out, f"c2_spmm_k{k}", A.algorithmic_bytes(k), timeit(lambda: A.spmm(Y, X, k, st), iters=5, warm=1))
            del X, Y
        del A, rp, cc, vv
    if "c3spmm" in what:
        # multi-column products on config 3's shape (bsbm_A_mul_B2 / _B4 / _Bn on a tall binary matrix, bsbm_cg2's products)
        nrow, ncol = n, max(n // 10, 1)
        rp, cc, _ = capi.synth_uniform(nrow, ncol, 64, 0x5EED0003, valued=False)
        A = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
        for k in (1, 2, 4, 8, 16, 32):
            X = torch.sin(torch.arange(ncol * k, device="cuda", dtype=torch.float64)).reshape(ncol, k)
            Y = torch.empty(nrow, k, dtype=torch.float64, device="cuda")
            report(out, f"c3_spmm_k{k}:{A.kernel_name()}", A.algorithmic_bytes(k), timeit(lambda: A.spmm(Y, X, k, st), iters=5, warm=1))
            if k >= 8:
                capi.set_option("spmm_kernel", 1)
                report(out, f"c3_spmm_k{k}:row_kernel", A.algorithmic_bytes(k), timeit(lambda: A.spmm(Y, X, k, st), iters=3, warm=1))
                capi.set_option("spmm_kernel", 0)
            del X, Y
        del A, rp, cc
    if "build" in what:
        # new_bcsr on config 3's 640 M COO entries, host loop against the device build (upload + stable
        # sort + download), host arrays in and out either way
        import ctypes as C
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _hipbackend as H
        nrow, ncol, per = n, max(n // 10, 1), 64
        _, cc, _ = capi.synth_uniform(nrow, ncol, per, 0x5EED0003, valued=False)
        hc = cc.cpu().numpy()
        del cc
        hr = np.repeat(np.arange(nrow, dtype=np.int32), per)
        L = capi.lib()
        for mode, label in ((2, "device"), (0, "host_loop"), (2, "device_again")):
            capi.set_option("device_build", mode)
            A = H.BCSR()
            t0 = time.time()
            L.new_bcsr(C.byref(A), C.c_long(len(hr)), nrow, ncol, H._ip(hr), H._ip(hc))
            dt = time.time() - t0
            rec = {"name": f"new_bcsr_{nrow}x{ncol}x{per}_{label}", "seconds": dt, "entries": int(len(hr))}
            print(json.dumps(rec), flush=True)
            out.write(json.dumps(rec) + "\n")
            L.free_bcsr(C.byref(A))
        capi.set_option("device_build", -1)
    if "ata" in what:
        # y = A'A x (bcsr_AA_mul_B): two products (A, cached A') against the fused single kernel, config 3's shape
        nrow, ncol = n, max(n // 10, 1)
        rp, cc, _ = capi.synth_uniform(nrow, ncol, 64, 0x5EED0003, valued=False)
        A = capi.Matrix.from_csr(nrow, ncol, rp, cc, None, borrow=True)
        A.build_transpose(st)
        x = torch.randint(-1000, 1001, (ncol,), device="cuda").to(torch.float64)
        y = torch.empty(ncol, dtype=torch.float64, device="cuda")
        y2 = torch.empty(ncol, dtype=torch.float64, device="cuda")
        tmp = torch.empty(nrow, dtype=torch.float64, device="cuda")
        B = 2 * 4 * nrow * 64 + 8 * (nrow + 1) + 16 * ncol
        report(out, f"c3_ata_two_products_{A.kernel_name()}+{A.kernel_name(True)}", B, timeit(lambda: A.ata(y, x, tmp, st), iters=10))
        capi.set_option("ata_kernel", 2)
        report(out, "c3_ata_fused_single_kernel", B, timeit(lambda: A.ata(y2, x, tmp, st), iters=10))
        capi.set_option("ata_kernel", 0)
        print("fused == two products (integer x):", bool(torch.equal(y, y2)), flush=True)
        del A
    if "mall" in what:
        # do row super-blocks whose products stay in the 256 MB Infinity Cache (with x, 80 MB, also
        # resident) beat one sweep?  The config-2 matrix as S handles of n/S rows, each forced onto the two-pass kernels and
        # run back to back (expand_s, reduce_s, expand_s+1, ...): the intermediate of one block is 1.31 GB / S
        capi.set_option("binning", 2)
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        x = torch.sin(7.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.3)
        y = torch.empty(n, dtype=torch.float64, device="cuda")
        B = 12 * n * 16 + 4 * (n + 1) + 16 * n
        for S_ in (1, 2, 4, 8, 16, 32):
            hs = []
            step = n // S_
            for k in range(S_):
                lo, hi = k * step, (n if k == S_ - 1 else (k + 1) * step)
                rpk = (rp[lo:hi + 1] - rp[lo]).contiguous()
                hs.append((capi.Matrix.from_csr(hi - lo, n, rpk, cc[lo * 16:hi * 16], vv[lo * 16:hi * 16], borrow=True), lo, hi))

            def run():
                for h, lo, hi in hs:
                    h.spmv(y[lo:hi], x, st)
            report(out, f"c2_two_pass_in_{S_}_row_superblocks_intermediate_{1311 // S_}MB", B, timeit(run))
            del hs
        capi.set_option("binning", 1)
        del rp, cc, vv
    if "spmmab" in what:
        # multi-column products on the config-2 matrix: the k-column two-pass sweep against one sweep per column, the
        # row kernel and (k = 32) the matrix-core experiment
        rp, cc, vv = capi.synth_uniform(n, n, 16, 0x5EED0002)
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        for k in (2, 3, 4, 8, 32):
            X = torch.sin(torch.arange(n * k, device="cuda", dtype=torch.float64)).reshape(n, k)
            Y = torch.empty(n, k, dtype=torch.float64, device="cuda")
            modes = ((0, "auto"), (3, "per_column_sweeps"), (1, "row_kernel")) if k <= 4 else ((1, "row_kernel"), (4, "mfma_f64_16x16x4"))
            for mode, label in modes:
                capi.set_option("spmm_kernel", mode)
                t0 = time.time()
                A.spmm(Y, X, k, st)
                torch.cuda.synchronize()
                first = time.time() - t0
                t = timeit(lambda: A.spmm(Y, X, k, st), iters=5, warm=1)
                print("first call s", first, flush=True)
                report(out, f"c2_spmm_k{k}_{label}", A.algorithmic_bytes(k), t)
            capi.set_option("spmm_kernel", 0)
            del X, Y
        del A, rp, cc, vv
    if "cbcsr" in what:
        # column-blocked binary CSR (cbcsr.h): 2 M x 1 M, 64 per row, 4 column blocks of 262144
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _hipbackend as H
        from oracle import pysynth
        nr, nc, per, cbs = 2_000_000, 1_000_000, 64, 262144
        rp, cc, _ = pysynth.uniform(nr, nc, per, 0x5EED0003, valued=False)
        rows = np.repeat(np.arange(nr, dtype=np.int32), per)
        F = H.HostFormats()
        t0 = time.time()
        K = F.cbcsr(cbs, nr, nc, rows, cc)
        print("host new_cbcsr s", time.time() - t0, flush=True)
        crp = F.arr(K.row_ptr, K.nblocks * nr + 1, np.int32)
        ccc = F.arr(K.cols, len(cc), np.int32)
        m = capi.ColBlockMatrix(nr, nc, K.nblocks, cbs, torch.from_numpy(crp).cuda(), torch.from_numpy(ccc).cuda())
        x = torch.randint(-1000, 1001, (nc,), device="cuda").to(torch.float64)
        y = torch.empty(nr, dtype=torch.float64, device="cuda")
        B = 4 * len(cc) + 4 * (K.nblocks * nr + 1) + 8 * nr + 8 * nc
        for mode, label in ((0, "general_path"), (9, "cells_streamed"), (5, "thread_per_row"), (4, "thread_per_row_lds_x")):
            capi.set_option("spmv_kernel", mode)
            report(out, f"cbcsr_2Mx1M_64_{label}", B, timeit(lambda: m.spmv(y, x, st), iters=5, warm=1))
        capi.set_option("spmv_kernel", 0)
    if "cg" in what:
        # the consumer of the path (cg.h): (A'A + 5 I) x = b on the config-2 pattern, vectors resident in HBM
        import ctypes as C
        rp, cc, _ = capi.synth_uniform(n, n, 16, 0x5EED0002, valued=False)
        A = capi.Matrix.from_csr(n, n, rp, cc, None, borrow=True)
        A.build_transpose(st)
        # A' as its own handle (the reference passes B and Bt): rows of A' = columns of A
        rows = torch.arange(n, device="cuda", dtype=torch.int32).repeat_interleave(16)
        At = capi.Matrix.from_coo(n, n, cc, rows, None)
        del rows
        b = torch.sin(19.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.4)
        x = torch.empty(n, dtype=torch.float64, device="cuda")
        L = capi.lib()
        for two in (False, True):
            bb = torch.stack([b, torch.cos(23.0 * torch.arange(n, device="cuda", dtype=torch.float64) + 0.7)], 1).contiguous() if two else b
            xx = torch.empty_like(bb)
            it = C.c_int(0)
            f = L.fs_cg2 if two else L.fs_cg
            capi.check(f(A.h, At.h, xx.data_ptr(), bb.data_ptr(), 5.0, 1e-8, C.byref(it), st))   # warm
            torch.cuda.synchronize()
            t0 = time.time()
            capi.check(f(A.h, At.h, xx.data_ptr(), bb.data_ptr(), 5.0, 1e-8, C.byref(it), st))
            torch.cuda.synchronize()
            dt = time.time() - t0
            iters = it.value + 1
            rec = {"name": "cg2_config2_pattern" if two else "cg_config2_pattern", "iterations": iters, "total_ms": dt * 1e3,
                   "ms_per_iteration": dt * 1e3 / iters,
                   "spmv_GBs_equiv": (2 if two else 1) * 2 * A.algorithmic_bytes() * iters / dt / 1e9}
            print(json.dumps(rec), flush=True)
            out.write(json.dumps(rec) + "\n")
        del A, At
    if "dropin" in what:
        # the reference-named entry point with HOST vectors and a host struct (what an unmodified C caller does):
        # per call = fingerprint of the host arrays + 80 MB up + kernel + 80 MB down
        import ctypes as C
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _hipbackend as H
        from oracle import pysynth
        rp, cc, vv = pysynth.uniform(n, n, 16, 0x5EED0002)
        A = H.CSR(n, n, len(cc), rp.ctypes.data_as(H.ip), cc.ctypes.data_as(H.ip), vv.ctypes.data_as(H.dp))
        xh = np.sin(7.0 * np.arange(n) + 0.3)
        yh = np.empty(n)
        L = capi.lib()
        L.csr_A_mul_B.restype = None
        t0 = time.time()
        L.csr_A_mul_B(yh.ctypes.data_as(H.dp), C.byref(A), xh.ctypes.data_as(H.dp))
        first = time.time() - t0
        ts = []
        for _ in range(10):
            t0 = time.time()
            L.csr_A_mul_B(yh.ctypes.data_as(H.dp), C.byref(A), xh.ctypes.data_as(H.dp))
            ts.append((time.time() - t0) * 1e3)
        ts.sort()
        B = 12 * len(cc) + 4 * (n + 1) + 16 * n
        rec = {"name": "dropin_csr_A_mul_B_host_vectors", "first_call_s_incl_upload": first, "ms_min": ts[0],
               "ms_med": ts[5], "GBs_med": B / ts[5] / 1e6}
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
        xd = torch.from_numpy(xh).cuda()
        yd = torch.empty(n, dtype=torch.float64, device="cuda")
        ts = []
        for _ in range(10):
            t0 = time.time()
            L.csr_A_mul_B(C.c_void_p(yd.data_ptr()), C.byref(A), C.c_void_p(xd.data_ptr()))
            ts.append((time.time() - t0) * 1e3)
        ts.sort()
        rec = {"name": "dropin_csr_A_mul_B_device_vectors", "ms_min": ts[0], "ms_med": ts[5], "GBs_med": B / ts[5] / 1e6,
               "host_chunks": os.environ.get("FS_HOST_CHUNKS", "8"),
               "max_abs_diff_host_vs_device_vectors": float(np.abs(yd.cpu().numpy() - yh).max())}
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
    if "c5" in what:
        nrow = n
        c5cols = (args.ncols[0] if args.ncols else nrow)
        rp, cc, vv = capi.synth_powerlaw(nrow, c5cols, 2.3, 1_000_000, 0x5EED0005)
        A = capi.Matrix.from_csr(nrow, c5cols, rp, cc, vv, borrow=True)
        x = torch.sin(7.0 * torch.arange(c5cols, device="cuda", dtype=torch.float64) + 0.3)
        y = torch.empty(nrow, dtype=torch.float64, device="cuda")
        for kern, label in ((0, "auto:" + A.kernel_name()), (1, "stream_nt")):
            capi.set_option("spmv_kernel", kern)
            report(out, f"c5shard_powerlaw_ncol{c5cols}_{label}_nnz{A.nnz}", A.algorithmic_bytes(), timeit(lambda: A.spmv(y, x, st)))
        capi.set_option("spmv_kernel", 0)


if __name__ == "__main__":
    main()
