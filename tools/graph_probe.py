import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libfastsparse_amd import capi
torch.cuda.init()
for name, (n, m, per) in {"two-pass-ish": (2_000_000, 2_000_000, 16), "lds-staged-ish": (2_000_000, 200_000, 64), "small": (100_000, 100_000, 16)}.items():
    rp, cc, vv = capi.synth_uniform(n, m, per, 7)
    A = capi.Matrix.from_csr(n, m, rp, cc, vv, borrow=True)
    A.build_transpose(capi.current_stream())
    x = torch.sin(torch.arange(m, device="cuda", dtype=torch.float64))
    y = torch.empty(n, device="cuda", dtype=torch.float64); z = torch.empty(m, device="cuda", dtype=torch.float64)
    yref = torch.empty_like(y); zref = torch.empty_like(z)
    A.spmv(yref, x, capi.current_stream()); A.spmv(zref, yref, capi.current_stream(), transposed=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        A.spmv(y, x, capi.current_stream()); A.spmv(z, y, capi.current_stream(), transposed=True)   # warm on this stream
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            A.spmv(y, x, capi.current_stream())
            A.spmv(z, y, capi.current_stream(), transposed=True)
    y.fill_(-1); z.fill_(-1)
    g.replay(); torch.cuda.synchronize()
    ok = torch.allclose(y, yref, rtol=0, atol=1e-9) and torch.allclose(z, zref, rtol=0, atol=1e-6)
    # timing: 200 x (two products) eager vs graph
    t0 = time.perf_counter()
    for _ in range(200):
        A.spmv(y, x, capi.current_stream()); A.spmv(z, y, capi.current_stream(), transposed=True)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 200
    print(name, A.kernel_name(), A.kernel_name(True), "captured+replayed ok" if ok else "MISMATCH", "eager %.1f us, graph %.1f us per pair" % (te * 1e6, tg * 1e6), flush=True)
