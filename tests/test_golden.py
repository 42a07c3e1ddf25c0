"""Oracle-generated regression vectors (tests/golden/make_golden.py): the oracle must still
reproduce them on the CPU, and the HIP library must reproduce them on the GPU."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_golden  # noqa: E402
from oracle_py import OracleMapper  # noqa: E402


def _check(name, make):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    rows, last = make_golden.run(make_golden.SCENARIOS[name], make)
    assert np.array_equal(rows, g["rows"]), "per-frame checksums / wave statistics differ"
    assert np.array_equal(last["type"], g["type"])
    assert np.array_equal(last["dist_sq"], g["dist_sq"])
    assert np.array_equal(last["coc"], g["coc"])
    assert np.allclose(last["edt"], g["edt"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("name", sorted(make_golden.SCENARIOS))
def test_oracle_reproduces_golden(oracle_lib, name):
    _check(name, OracleMapper)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.SCENARIOS))
def test_hip_reproduces_golden(name):
    import gie
    _check(name, gie.Mapper)
