"""A short, seeded stretch of the randomised differential runs (tools/fuzz_parity.py: device layer; tools/fuzz_dist.py: the
multi-GPU layer on virtual ranks; tools/fuzz_dropin.py: the reference-named entry points; tools/fuzz_cg.py: the solvers) -- the open-ended runs are recorded in
profiles/r05_fuzz_parity.txt; here a few hundred cases each keep them from rotting.  Every result is compared with the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,env", [("fuzz_parity.py", {}), ("fuzz_dist.py", {}), ("fuzz_dist.py", {"FS_DIST_PARTS": "7", "FS_DIST_THREADS": "1"}),
                                      ("fuzz_dropin.py", {}), ("fuzz_dropin.py", {"FASTSPARSE_NGPU": "3", "FASTSPARSE_DEVICES": "0,0,0"}),
                                      # storage-order sums: every family BIT FOR BIT against the oracle for arbitrary x and values, one GPU and three ranks
                                      ("fuzz_dropin.py", {"FS_STRICT_ORDER": "1"}),
                                      ("fuzz_dropin.py", {"FS_STRICT_ORDER": "1", "FASTSPARSE_NGPU": "3", "FASTSPARSE_DEVICES": "0,0,0"}),
                                      # the solvers (bsbm_cg / bsbm_cg2 and fs_cg / fs_cg2): true residuals, iteration counts, two solves bit for bit
                                      ("fuzz_cg.py", {}), ("fuzz_cg.py", {"FASTSPARSE_NGPU": "3", "FASTSPARSE_DEVICES": "0,0,0"}),
                                      # expected results from the REAL reference library instead of the oracle's restatement
                                      ("fuzz_dropin.py", {"FS_FUZZ_REF": "1", "FS_STRICT_ORDER": "1"}), ("fuzz_cg.py", {"FS_FUZZ_REF": "1"})])
def test_randomised_differential_stretch(tool, env):
    if env.get("FS_FUZZ_REF") and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libfsref.so")):
        pytest.skip("oracle/_ref/libfsref.so (the real reference, built where /root/reference exists) did not travel")
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "6", "20251005"], cwd=ROOT, env=e, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "all within the bars" in out.stdout, out.stdout[-2000:]
