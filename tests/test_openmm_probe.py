"""The opportunistic OpenMM check of SURVEY.md section 8c(4).  Where ``import openmm`` works, the oracle (CPU tests) and
the HIP path (GPU test) are compared with OpenMM's Reference platform on a System built by this repository's own host
code (oracle/openmm_probe.py) -- that comparison is what would lift the oracle's "parity unpinned" label.  On this
image OpenMM is absent: the comparisons skip, and the probe's report of WHY is what bench.py prints."""
import numpy as np
import pytest

from multimm_amd import synthetic_system
from oracle.openmm_probe import probe

ALL_ON = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
              IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)


def test_probe_reports_what_it_found():
    mm, why = probe()
    assert isinstance(why, str) and why
    if mm is None:
        assert "openmm" in why.lower()   # e.g. "ModuleNotFoundError: No module named 'openmm'"


@pytest.mark.parametrize("cutoff", [0.0, 0.6])
def test_oracle_against_openmm_reference(cutoff):
    pytest.importorskip("openmm")
    from oracle.openmm_probe import openmm_eval
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=1500, jitter=0.03, seed=3, NB_CUTOFF=cutoff, **ALL_ON)
    et, F = openmm_eval(s, "Reference")
    et_o, F_o = Oracle(s).eval()
    assert np.allclose(et, et_o, rtol=1e-9, atol=1e-6 * np.abs(et_o).sum())
    assert np.abs(F - F_o).max() <= 1e-8 * np.abs(F_o).max() + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("cutoff", [0.0, 0.6])
def test_engine_against_openmm_reference(cutoff):
    """north_star: "per-step forces and total energy match OpenMM's Reference/CPU platform on identical inputs within a
    stated fp32 tolerance" -- the tolerances of tests/test_gpu_parity.py."""
    pytest.importorskip("openmm")
    from multimm_amd.engine import engine_for
    from oracle.openmm_probe import openmm_eval
    s = synthetic_system("gw_200k", n_beads=4000, jitter=0.03, seed=3, NB_CUTOFF=cutoff, **ALL_ON)
    et_ref, F_ref = openmm_eval(s, "Reference")
    with engine_for(s) as eng:
        et, F = eng.compute()
    assert np.all(np.abs(et - et_ref) <= 2e-6 * np.abs(et_ref).sum() + 1e-3)
    assert np.abs(F - F_ref).max() <= 4e-6 * np.abs(F_ref).max() + 2e-3   # F_RTOL, F_ATOL of tests/test_gpu_parity.py
