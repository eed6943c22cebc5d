#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference.

Run in the build container only (needs /root/reference):
    make -C oracle ref && python tests/golden/make_golden.py
Each file holds the outputs of every hot-path entry point of the reference
(strict-IEEE build of oracle/_ref/libfsref.so, see oracle/Makefile) for one case
of tests/_cases.py.  Inputs are not stored: they are regenerated from the seeds
in _cases.py / the two bundled .data fixtures.  `spread.json` records how far the
reference's own -O3 -march=native -ffast-math build (reference Makefile:2) moves
from the strict build on the same inputs (SURVEY.md note N2).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _cases  # noqa: E402


def main():
    """python tests/golden/make_golden.py [case name ...]  -- no names: every case; names: only those (spread.json is
    updated, not rewritten)"""
    strict = _cases.RefBackend(fast=False)
    fast = _cases.RefBackend(fast=True)
    only = set(sys.argv[1:])
    spread = {}
    if only and os.path.exists(os.path.join(HERE, "spread.json")):
        spread = json.load(open(os.path.join(HERE, "spread.json")))
    for case in _cases.all_cases():
        if only and case.name not in only:
            continue
        out = _cases.run_case(strict, case)
        np.savez_compressed(os.path.join(HERE, case.name + ".npz"), **{k.replace("/", "|"): v for k, v in out.items()})
        outf = _cases.run_case(fast, case)
        worst = 0.0
        for k in out:
            den = max(np.max(np.abs(out[k])), 1e-300)
            worst = max(worst, float(np.max(np.abs(out[k] - outf[k])) / den))
        spread[case.name] = {"outputs": len(out), "max_abs_diff_over_max_abs_y__fast_vs_strict": worst}
        print(case.name, len(out), "outputs; fast-math spread", worst)
    json.dump(spread, open(os.path.join(HERE, "spread.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
