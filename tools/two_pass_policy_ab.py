import sys, json, torch
sys.path.insert(0, '.')
from libfastsparse_amd import capi
n, per = 10_000_000, 16
st = capi.current_stream()
rp, cc, vv = capi.synth_uniform(n, n, per, 0x5EED0002)
A = capi.Matrix.from_csr(n, n, rp, cc, vv, borrow=True)
i = torch.arange(n, device="cuda", dtype=torch.float64)
x = torch.sin(7.0 * i + 0.3); y = torch.empty_like(x)
print(A.kernel_name(), A.candidate_ms())
def t(flags, reps=40):
    capi.set_option("bin_flags", flags)
    for _ in range(5): A.spmv(y, x, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): A.spmv(y, x, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rnd in range(2):
    for flags, name in ((0, "default: plain loads, nt stores"), (8, "plain stores"), (4, "pass-2 nt loads"), (16, "pass-1 nt loads"), (24, "pass-1 nt loads + plain stores"), (1, "unroll 8"), (2, "unroll 2")):
        print(json.dumps({"what": "two-pass cache-policy A/B, config 2 A x", "bin_flags": flags, "name": name, "ms": t(flags)}), flush=True)
capi.set_option("bin_flags", 0)
