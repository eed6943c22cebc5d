#!/usr/bin/env python3
"""Does the Infinity Cache carry the products of the two-pass pair?  Config 2 cut into S row slabs, every slab its own handle with
its own two-pass copy (binning forced); one product = the S slabs' pass 1 + pass 2 one after the other, so that a slab's products
(1.28 GB / S) are read back right after they were written.  Prints ms per product against the one-handle product.
    python tools/slab_mall_probe.py [S ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libfastsparse_amd import capi  # noqa: E402

SEED = 0x5EED0002
n, ncol, per = 10_000_000, 10_000_000, 16
slabs = [int(a) for a in sys.argv[1:]] or [1, 8, 16, 32]
st = capi.current_stream()
x = torch.sin(7.0 * torch.arange(ncol, dtype=torch.float64, device="cuda") + 0.3)
ref = None
capi.set_option("binning", 2)
capi.set_option("ldsx", 0)
capi.set_option("tiling", 0)
for S in slabs:
    rows = [(n * s // S, n * (s + 1) // S) for s in range(S)]
    hs = []
    for a, b in rows:
        m = b - a
        rp, cc, vv = capi.synth_uniform(m, ncol, per, SEED, a)
        torch.cuda.synchronize()
        hs.append(capi.Matrix.from_csr(m, ncol, rp, cc, vv))
        del rp, cc, vv
    y = torch.empty(n, dtype=torch.float64, device="cuda")

    def product():
        for (a, b), h in zip(rows, hs):
            h.spmv(y[a:b], x, st)

    for _ in range(3):
        product()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        product()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    # the same launches from a captured graph: no host gaps between the 2 S kernels
    g = torch.cuda.CUDAGraph()
    gms = None
    try:
        with torch.cuda.graph(g):
            for (a, b), h in zip(rows, hs):
                h.spmv(y[a:b], x, capi.current_stream())
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        gms = e0.elapsed_time(e1) / reps
    except Exception as e:  # noqa: BLE001
        gms = "graph capture failed: %s" % str(e)[:80]
    if ref is None:
        ref = y.clone()
    diff = float((y - ref).abs().max())
    print({"slabs": S, "kernel": hs[0].kernel_name(), "ms_per_product": ms, "ms_from_a_graph": gms, "max_abs_diff_vs_first": diff,
           "products_per_slab_MB": 8.0 * n * per / S / 1e6}, flush=True)
    for h in hs:
        h.close()
    del hs, y
    torch.cuda.empty_cache()
