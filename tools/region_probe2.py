#!/usr/bin/env python3
"""Reads and writes together: device-to-device copies between pairs of 1.25 GB regions (plain reads and plain fills of the same
regions are uniform, tools/region_probe.py).  One JSON line: GB/s (read + written) per pair.   python tools/region_probe2.py"""
import json

import torch


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


nb = 163_519_472
bufs = [torch.empty(nb, dtype=torch.float64, device="cuda") for _ in range(24)]
for b in bufs:
    b.fill_(1.0)
out = {}
for i in range(0, 24, 2):
    out["%d->%d" % (i, i + 1)] = round(2 * nb * 8 / timed(lambda: bufs[i + 1].copy_(bufs[i])) / 1e6)
for i in range(0, 12):
    out["%d->%d" % (i, 23 - i)] = round(2 * nb * 8 / timed(lambda: bufs[23 - i].copy_(bufs[i])) / 1e6)
print(json.dumps(out))
