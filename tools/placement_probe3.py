#!/usr/bin/env python3
"""Is the placement effect a resonance between the spacing of pass 1's 256 equal shares and the address mapping?  Six identical
copies of config 2; each timed with several numbers of pass-1 workgroups (option bin_wgs: other share spacings).

    python tools/placement_probe3.py
"""
import json
import sys

import torch

sys.path.insert(0, ".")
from libfastsparse_amd import capi  # noqa: E402


def timed(f, reps=20):
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
    n, per = 10_000_000, 16
    dev = "cuda"
    rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002, valued=True, device=dev)
    x = torch.sin(torch.arange(n, dtype=torch.float64, device=dev) * 7.0 + 0.3)
    y = torch.empty(n, dtype=torch.float64, device=dev)
    keep = []
    for i in range(6):
        A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
        keep.append(A)
        rec = {"copy": i}
        for wgs in (0, 255, 253, 251, 247, 240, 224, 192):
            capi.set_option("bin_wgs", wgs)
            rec["wgs%d" % wgs] = round(timed(lambda: A.spmv(y, x)), 4)
        capi.set_option("bin_wgs", 0)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
