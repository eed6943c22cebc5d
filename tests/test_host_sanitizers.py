"""SURVEY.md 5 ("host: -fsanitize=address,undefined test build"): the host half of the library -- fs_host.c
(the reference's constructors, loaders, (de)serialisation, ownership rules) and fs_sort.c (Hilbert / quick sorts, the sort_*
functions) -- compiled with AddressSanitizer + UndefinedBehaviorSanitizer and run under the host tests: every constructor
against the oracle, the fixture loaders, block geometry, the reference's known-answer tests of the helpers.  CPU only (the
GPU pool refuses sanitizer runs); the device layer's four entry points the host code calls are stubbed to "no device build"."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libfastsparse_amd", "csrc")

STUBS = r'''
#include <stdint.h>
/* the device layer, as the host constructors see it: never asked to build on the device, nothing cached */
int fs_device_build_wanted(int64_t nnz) { (void)nnz; return 0; }
int fs_bucket_coo(int kind, int param, int nrow, int ncol, int64_t nbuckets, int64_t nnz, const int *rows, const int *cols,
                  const double *vals, int *offsets, int *rows_out, int *cols_out, double *vals_out)
{ (void)kind; (void)param; (void)nrow; (void)ncol; (void)nbuckets; (void)nnz; (void)rows; (void)cols; (void)vals; (void)offsets;
  (void)rows_out; (void)cols_out; (void)vals_out; return -1; }
void fs_invalidate(const void *p) { (void)p; }
const char *fs_last_error(void) { return ""; }
'''


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None or _runtime("libasan.so") is None, reason="gcc with libasan not available")
def test_host_half_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    stubs = tmp_path / "device_stubs.c"
    stubs.write_text(STUBS)
    lib = tmp_path / "libfastsparse_host_san.so"
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    "-I" + CSRC, os.path.join(CSRC, "fs_host.c"), os.path.join(CSRC, "fs_sort.c"), str(stubs), "-o", str(lib), "-lm"],
                   check=True, capture_output=True)
    preload = ":".join(p for p in (_runtime("libasan.so"), _runtime("libubsan.so")) if p)
    env = dict(os.environ, LD_PRELOAD=preload, FS_HOST_ONLY_LIB=str(lib),
               # python itself leaks by design; everything else is fatal
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = ["tests/test_abi_host.py::test_host_constructors_match_oracle", "tests/test_abi_host.py::test_host_loaders_and_containers",
             "tests/test_reference_kats.py::test_fixture_loader_kats", "tests/test_reference_kats.py::test_block_geometry_kats",
             "tests/test_oracle_vs_ref.py", "-k", "not product and not mul"]
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    out = p.stdout[-3000:] + p.stderr[-3000:]
    assert p.returncode == 0 and "passed" in p.stdout, out
    assert "AddressSanitizer" not in out and "runtime error" not in out, out
