#!/usr/bin/env python3
"""Are some regions of this GPU's HBM slower than others?  40 buffers of 1.25 GB held at once; a plain read (sum) and a plain
write (fill) of each, timed.  Then four two-pass copies of config 2 to see whether this box is one where identical copies differ.

    python tools/region_probe.py
"""
import json
import sys

import torch

sys.path.insert(0, ".")
from libfastsparse_amd import capi  # noqa: E402


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda"
    nb = 163_519_472          # doubles: the product stream of config 2
    bufs = [torch.empty(nb, dtype=torch.float64, device=dev) for _ in range(40)]
    for b in bufs:
        b.fill_(1.0)
    rd, wr = [], []
    for b in bufs:
        rd.append(nb * 8 / timed(lambda: b.sum()) / 1e6)
        wr.append(nb * 8 / timed(lambda: b.fill_(2.0)) / 1e6)
    print(json.dumps({"read_GBs": [round(v) for v in rd], "write_GBs": [round(v) for v in wr],
                      "at": [hex(b.data_ptr()) for b in bufs]}), flush=True)
    del bufs
    torch.cuda.empty_cache()
    n, per = 10_000_000, 16
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002, valued=True, device=dev)
    x = torch.sin(torch.arange(n, dtype=torch.float64, device=dev) * 7.0 + 0.3)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    capi.set_option("placement_trials", 0)
    keep = []
    for i in range(6):
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        keep.append(A)
        print(json.dumps({"copy": i, "ms": round(timed(lambda: A.spmv(y, x), 20), 4)}), flush=True)


if __name__ == "__main__":
    main()
