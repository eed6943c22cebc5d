"""Builds libfastsparse_hip.so in-tree with hipcc for gfx950 (no GPU needed to build)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libfastsparse_hip.so")
ARCH = "gfx950"

HIP_SOURCES = ["fs_probes.hip", "fs_kernels.hip", "fs_kernels_tiled.hip", "fs_kernels_twopass.hip", "fs_format.hip", "fs_abi.hip", "fs_dropin.hip", "fs_cg.hip", "fs_dist.hip"]
C_SOURCES = ["fs_host.c", "fs_sort.c"]
HEADERS = [os.path.join(CSRC, "fs_common.h"), os.path.join(CSRC, "fs_kernel_util.h")] + [os.path.join(ROOT, "include", h) for h in
                                                 ("fastsparse_hip.h", "sparse.h", "dsparse.h", "csr.h", "cbcsr.h", "cg.h", "linalg.h", "hilbert.h", "quickSort.h", "quickSortD.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libfastsparse_hip.so."""
    os.makedirs(OBJ, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    extra = os.environ.get("FS_HIPCC_EXTRA", "").split()      # experiments (e.g. -DFS_EXP_VALU=8); implies a full rebuild
    force = force or bool(extra)
    objs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        if force or _stale(o, [s] + HEADERS):
            # -ffp-contract=off: products and sums round separately, like the strict CPU loops
            _run([_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                  "-Wall", "-Wno-unused-result"] + extra + inc + ["-c", s, "-o", o], verbose)
        objs.append(o)
    for src in C_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        if force or _stale(o, [s] + HEADERS):
            _run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-Wall"] + inc + ["-c", s, "-o", o], verbose)
        objs.append(o)
    if force or _stale(LIB, objs):
        _run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs +
             ["-o", LIB, "-lm", "-ldl", "-pthread", "-Wl,-rpath,/opt/rocm/lib"], verbose)
    return LIB


def build_c_bench(verbose=False):
    """Label-compatible C driver of the reference's bench (csrc/bench_a_mul_b.c), linked against the library."""
    src = os.path.join(CSRC, "bench_a_mul_b.c")
    out = os.path.join(HERE, "bench_a_mul_b")
    if not os.path.exists(src):
        return None
    if _stale(out, [src, LIB] + HEADERS):
        _run(["gcc", "-std=gnu99", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
              "-L" + HERE, "-lfastsparse_hip", "-lm", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"], verbose)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
