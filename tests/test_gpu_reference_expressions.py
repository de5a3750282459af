"""The DEVICE path against the reference's own energy expression strings, without the oracle in between: the cases of
tests/test_reference_expressions.py (every `*_FORCE_TYPE` branch of model.py's pair and external terms, evaluated from the
reference's text and summed as OpenMM sums them, NoCutoff) computed through the C ABI on the GPU.  fp32 device arithmetic:
energies within 4e-6 relative + 1e-3 kJ/mol, forces against minus the NUMERICAL gradient of the reference's expression sum."""
import os
import sys

import numpy as np
import pytest

from multimm_amd.engine import engine_for

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_reference_expressions import expression_cases  # noqa: E402

pytestmark = pytest.mark.gpu
CASES = expression_cases()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_device_energy_equals_the_reference_expression(case):
    name, s, want, term, energy_of = case
    with engine_for(s) as eng:
        et, F = eng.compute()
    assert np.isfinite(F).all()
    assert abs(et[term] - want) <= 4e-6 * max(abs(want), 1.0) + 1e-3, (name, et[term], want)
    others = np.delete(et, term)
    assert np.abs(others).max() == 0.0, (name, et)       # every other term is switched off
    # forces: minus the gradient of the reference's expression sum, by central differences in fp64 (h = 1e-6 nm); the device
    # saw the positions rounded to fp32, hence the slope term of the tolerance
    x32 = s.positions.astype(np.float32).astype(np.float64)
    h = 1e-6
    G = np.zeros_like(x32)
    for i in range(s.n_beads):
        for c in range(3):
            xp, xm = x32.copy(), x32.copy()
            xp[i, c] += h
            xm[i, c] -= h
            G[i, c] = (energy_of(xp) - energy_of(xm)) / (2 * h)
    fmax = max(np.abs(G).max(), 1e-3)
    assert np.abs(F + G).max() <= 2e-4 * fmax + 2e-3, (name, np.abs(F + G).max(), fmax)
