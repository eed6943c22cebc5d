"""Where do the 5 % between the kernels' own durations and the event-timed product go (perf item 3a)?
Config 2 (CSR 10 M x 10 M x 16, fp64), A x and A' u alternating like bench.py's step, timed four ways:
  per-product HIP events (what bench.py r2 did), ONE event pair around the whole loop, wall clock around the loop, and -- under
  `rocprofv3 --kernel-trace` -- the kernels' own start / end stamps (tools/kernel_gaps.py reads the trace).
(Round 3 also ran a fused single launch of both passes here -- 1.13 ms against 0.857 -- before it was withdrawn:
profiles/r03_gap_probe.jsonl keeps those records.)   python tools/gap_probe.py [--rows N]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libfastsparse_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--per-row", type=int, default=16)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--out", default="gpurun_out/gap_probe.jsonl")
    a = ap.parse_args()
    n, per = a.rows, a.per_row
    st = capi.current_stream()
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002)
    A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
    A.build_transpose(st)
    i = torch.arange(n, device="cuda", dtype=torch.float64)
    x, u = torch.sin(7.0 * i + 0.3), torch.sin(11.0 * i - 0.2)
    y, z = torch.empty_like(x), torch.empty_like(x)
    nbytes = A.algorithmic_bytes()
    recs = []
    ref = {}
    for mode, flags in (("two launches", 0), ("two launches", 0)):
        capi.set_option("bin_flags", flags)
        for _ in range(3):
            A.spmv(y, x, st)
            A.spmv(z, u, st, transposed=True)
        torch.cuda.synchronize()
        # (1) per-product events
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.reps)]
        t0 = time.perf_counter()
        for e in evs:
            e[0].record(); A.spmv(y, x, st); e[1].record()
            e[2].record(); A.spmv(z, u, st, transposed=True); e[3].record()
        torch.cuda.synchronize()
        wall_ev = (time.perf_counter() - t0) / (2 * a.reps) * 1e3
        per_product = sum(e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]) for e in evs) / (2 * a.reps)
        # (2) one event pair, (3) wall clock, no events inside
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.reps):
            A.spmv(y, x, st)
            A.spmv(z, u, st, transposed=True)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / (2 * a.reps) * 1e3
        whole = e0.elapsed_time(e1) / (2 * a.reps)
        if mode not in ref:
            ref[mode] = (y.clone(), z.clone())
        d = None
        if mode == "fused":
            d = max(float((y - ref["two launches"][0]).abs().max()), float((z - ref["two launches"][1]).abs().max()))
        rec = {"what": "gap_probe", "mode": mode, "rows": n, "per_row": per, "kernel": A.kernel_name(),
               "ms_per_product_events_around_each": per_product, "ms_per_product_one_event_pair": whole,
               "ms_per_product_wall_no_events": wall, "ms_per_product_wall_with_events": wall_ev,
               "TBs_one_event_pair": nbytes / (whole * 1e-3) / 1e12, "max_abs_diff_vs_two_launches": d}
        print(json.dumps(rec), flush=True)
        recs.append(rec)
    capi.set_option("bin_flags", 0)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "a") as f:
        for r in recs:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
