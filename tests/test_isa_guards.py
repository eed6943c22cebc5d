"""The LDS-DMA kernel (spmv_ldsx_dma_kernel) places its vmcnt waits by hand: the count it waits for is the number of vector-memory
operations issued after the DMA that has to have landed.  That count is a property of the COMPILED code (one 8-byte load for a
pair of entries, one 16-byte load for a pair of values, one DMA per phase, in that order), so this test compiles the kernels'
translation unit to gfx950 assembly and checks every phase of every instantiation: if a compiler ever splits or reorders those
loads the wait would be too weak, and this fails instead of the results going quietly wrong."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libfastsparse_amd", "csrc")


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_dma_kernel_waits_match_the_compiled_memory_operations(tmp_path):
    out = tmp_path / "fs_kernels_tiled.s"
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S",
                    os.path.join(CSRC, "fs_kernels_tiled.hip"), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    found = 0
    for m in re.finditer(r"^(_ZN2fs20spmv_ldsx_dma_kernelILb([01])ELb[01]ELi(\d+)ELb([01])EEE[^:\n]*):", text, re.M):
        entries = 2 if m.group(2) == "1" else 1          # loads per pair of entries: packed words (+ values)
        nsets = int(m.group(3))                          # phases per trip of the unrolled loop
        body = text[m.end():text.index("s_endpgm", m.end())]
        ev = []                                          # D = slice DMA, L = entry load, W<n> = vmcnt wait, in program order
        for l in body.split("\n"):
            l = l.split(";")[0].strip()
            if l.startswith("global_load_lds_dwordx4"):
                ev.append("D")
            elif re.match(r"global_load_dword(x\d)?\b", l):
                ev.append("L")
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                ev.append("W%d" % int(re.search(r"vmcnt\((\d+)\)", l).group(1)))
        # after the loop: everything has landed before the y slice is read (what follows that wait is the slice leaving LDS:
        # stores or atomics, and for fixed-order sums the ticket of chunks that share a panel)
        assert "W0" in ev, (m.group(1), ev)
        ev = ev[:ev.index("W0")]
        # the loop (the compiler rotates it: the text may start in the middle of a phase): nsets phases of
        # "DMA, the entry loads behind it, wait for all but the 2 * entries + 1 youngest operations", and nothing else --
        # in particular no stronger wait added by the compiler
        phase = ["D"] + ["L"] * entries + ["W%d" % (2 * entries + 1)]
        assert sorted(ev) == sorted(phase * nsets), (m.group(1), ev)
        k = ev.index("D")
        assert ev[k:] + ev[:k] == phase * nsets, (m.group(1), ev)
        found += 1
    assert found == 8      # pattern / valued x plain / non-temporal entry loads x arrival-order / fixed-order sums
