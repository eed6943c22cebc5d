"""How fast does this chip move pass 1's read : write mix when BOTH streams are sequential?  (pass 1 of the two-pass pair reads
10.25 B and writes 8 B per entry, the writes as 128-byte lines scattered over the panels.)  torch elementwise kernels on 164 M
entries: p = v * c (8 B read + 8 B written), p = v * a (a: int16 -> 10 B read + 8 B written), and a plain copy."""
import sys, os, time
import torch
n = 164_000_000
v = torch.rand(n, dtype=torch.float64, device="cuda")
a = torch.ones(n, dtype=torch.int16, device="cuda")
p = torch.empty_like(v)
def t(f, reps=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, f, b in (("copy 8R+8W", lambda: p.copy_(v), 16), ("mul scalar 8R+8W", lambda: torch.mul(v, 2.0, out=p), 16),
                   ("mul int16 10R+8W", lambda: torch.mul(v, a, out=p), 18), ("read only (sum) 8R", lambda: v.sum(), 8)):
    ms = t(f)
    print("%-22s %.3f ms  %.2f TB/s" % (name, ms, n * b / ms / 1e9))
