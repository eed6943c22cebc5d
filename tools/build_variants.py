"""Experiment builds of the library next to the product build: the kernel translation units (and fs_abi.hip, for the debug exports) compiled with
extra -D flags, linked with the product build's other objects into libfastsparse_amd/build/variants/libfs_<name>.so; a tool then runs
with FS_LIB_PATH=<that file>.  Several variants fit into ONE gpurun call, i.e. are timed on the same box.
    python tools/build_variants.py name=-DFS_DMA_ABL=3 other="-DFS_DMA_ABL=4 -DFS_DMA_SETS=9" ...
A variant whose flags name one of the laboratory switches (FS_DMA_ABL, FS_DMA_TRACE, FS_DMA_EXTRA_*, FS_DMA_ADDS_LAST) is built with
-DFS_LAB, which swaps the instrumented copy of the LDS-DMA kernel (csrc/experiments/ldsx_dma_lab.inc) in for the product kernel;
the product build never sees that file."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libfastsparse_amd import _build  # noqa: E402


def one(spec):
    name, flags = spec.split("=", 1)
    if any(k in flags for k in ("FS_DMA_ABL", "FS_DMA_TRACE", "FS_DMA_EXTRA", "FS_DMA_ADDS_LAST")) and "-DFS_LAB" not in flags:
        flags = "-DFS_LAB " + flags
    out_dir = os.path.join(_build.OBJ, "variants")
    os.makedirs(out_dir, exist_ok=True)
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + _build.CSRC]
    objs = []
    for src in _build.HIP_SOURCES:
        if src in ("fs_kernels.hip", "fs_kernels_tiled.hip", "fs_kernels_twopass.hip", "fs_abi.hip"):
            o = os.path.join(out_dir, "%s_%s.o" % (name, src))
            subprocess.check_call([_build._hipcc(), "--offload-arch=" + _build.ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                   "-Wall", "-Wno-unused-result"] + flags.split() + inc + ["-c", os.path.join(_build.CSRC, src), "-o", o])
        else:
            o = os.path.join(_build.OBJ, src + ".o")
        objs.append(o)
    objs += [os.path.join(_build.OBJ, src + ".o") for src in _build.C_SOURCES]
    lib = os.path.join(out_dir, "libfs_%s.so" % name)
    subprocess.check_call([_build._hipcc(), "--offload-arch=" + _build.ARCH, "-shared", "-fPIC"] + objs +
                          ["-o", lib, "-lm", "-ldl", "-pthread", "-Wl,-rpath,/opt/rocm/lib"])
    return lib


if __name__ == "__main__":
    _build.build()
    with ThreadPoolExecutor(max_workers=3) as ex:
        for lib in ex.map(one, sys.argv[1:]):
            print(lib)
