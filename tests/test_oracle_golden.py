"""Pins the oracle: oracle/fs_oracle.c must reproduce, bit for bit, the outputs the
real reference produced for every hot-path entry point (tests/golden/*.npz,
written by tests/golden/make_golden.py from oracle/_ref/libfsref.so)."""
import os

import numpy as np
import pytest

import _cases
import _synth as S

CASES = _cases.all_cases()


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_oracle_reproduces_reference_golden(case):
    gold = np.load(os.path.join(S.GOLDEN, case.name + ".npz"))
    out = _cases.run_case(_cases.OracleBackend(), case)
    assert set(k.replace("/", "|") for k in out) == set(gold.files)
    for k, v in out.items():
        g = gold[k.replace("/", "|")]
        assert v.shape == g.shape, k
        assert np.array_equal(v.view(np.int64), g.view(np.int64)), f"{case.name}:{k} differs from the reference golden"


def test_golden_covers_every_entry_point():
    """one golden output at least for each A_mul_B-family symbol of SURVEY 8a"""
    names = set()
    for c in CASES:
        gold = np.load(os.path.join(S.GOLDEN, c.name + ".npz"))
        names |= {f.split("|")[0] for f in gold.files}
    expected = {"A_mul_B", "At_mul_B", "sdm_A_mul_B", "sdm_At_mul_B", "csr_A_mul_B", "csr_A_mul_Bn",
                "bcsr_A_mul_B", "bcsr_A_mul_B2", "bcsr_A_mul_B4", "bcsr_A_mul_B8", "bcsr_A_mul_B8_auto",
                "bcsr_A_mul_Bn", "bcsr_A_mul_B32n", "bcsr_AA_mul_B", "parallel_bcsr_AA_mul_B",
                "bsbm_A_mul_B", "bsbm_A_mul_B2", "bsbm_A_mul_B4", "bsbm_A_mul_Bn", "bsdm_A_mul_B",
                "cbcsr_A_mul_B"}
    assert expected <= names, expected - names


def test_integer_x_is_order_independent():
    """SURVEY note N1: with integer-valued x every binary kernel agrees exactly,
    whatever its summation order (COO scatter, CSR rows, row blocks, column blocks)."""
    for c in CASES:
        if "int" not in c.xs:
            continue
        out = _cases.run_case(_cases.OracleBackend(), c, tags={"int"}, light=True)
        base = out["A_mul_B/int"]
        for k, v in out.items():
            if k.split("/")[0] in ("bcsr_A_mul_B", "bsbm_A_mul_B", "cbcsr_A_mul_B"):
                assert np.array_equal(v, base), (c.name, k)
        assert np.array_equal(out["bcsr_AA_mul_B/int"], out["parallel_bcsr_AA_mul_B/int"])
