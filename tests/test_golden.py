"""Committed fixtures of tests/golden/ (generator: tests/golden/make_golden.py -- read its header: they are frozen
outputs of this repository's fp64 restatements, NOT of the reference, which ships none; parity stays "unpinned").

CPU tests: the C oracle, the numpy oracle, its L-BFGS and MD restatements reproduce the frozen numbers.
GPU tests: the HIP path, through the C ABI, reproduces them at the fp32 tolerances of test_gpu_parity.py.
"""
import json
import os

import numpy as np
import pytest

from multimm_amd.system import ChromatinSystem, ForceFieldParams

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META = json.load(open(os.path.join(HERE, "golden.json")))
CASES = sorted(META["cases"])
E_RTOL, E_ATOL, F_RTOL, F_ATOL = 2e-6, 1e-3, 4e-6, 2e-3


def load_case(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    c = META["cases"][name]
    s = ChromatinSystem(n_beads=c["n_beads"], positions=z["positions"].astype(np.float64), chr_ends=z["chr_ends"],
                        labels=z["labels"], loop_m=z["loop_m"], loop_n=z["loop_n"], loop_r0=z["loop_r0"],
                        ff=ForceFieldParams(**c["ff"]),
                        chrom_strength=z["chrom_strength"] if "chrom_strength" in z.files else None, name=name)
    return s, z, c


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_fixture(name):
    from oracle import oracle_np
    from oracle.oracle import Oracle
    s, z, c = load_case(name)
    et, F = Oracle(s).eval()
    assert np.allclose(et, z["energy_terms"], rtol=1e-11, atol=1e-9)
    assert np.abs(F - z["forces"]).max() <= 1e-9 * np.abs(z["forces"]).max()
    assert float(et.sum()) == pytest.approx(c["energy_total"], rel=1e-11)
    if all(v == getattr(ForceFieldParams(), k) for k, v in c["ff"].items() if k.endswith("_FORCE_TYPE")):
        en = oracle_np.energy_terms(s)  # the independent numpy restatement (fp64 constants: 1e-6 agreement)
        for i, k in enumerate(META["terms"]):
            assert en[k] == pytest.approx(z["energy_terms"][i], rel=2e-6, abs=1e-6), k


@pytest.mark.parametrize("name", [n for n in CASES if "minimize" in META["cases"][n]])
def test_oracle_minimizer_reproduces_fixture(name):
    from oracle.oracle import Oracle
    s, z, c = load_case(name)
    m = c["minimize"]
    x, st = Oracle(s).minimize(tolerance=0.0, max_iters=m["max_iters"])
    assert (st.iterations, st.evaluations, st.status) == (m["iterations"], m["evaluations"], m["status"])
    assert st.e_final == pytest.approx(m["e_final"], rel=1e-10) and st.e_initial == pytest.approx(m["e_initial"], rel=1e-11)
    assert np.abs(x - z["minimized_positions"]).max() < 1e-8


def test_oracle_md_reproduces_fixture():
    from oracle.oracle import Oracle, md_velocities
    s, z, c = load_case("gw_512_cutoff")
    m = c["md"]
    v0 = md_velocities(s.n_beads, m["temperature"], m["mass"], m["velocity_seed"])
    x, v, st = Oracle(s).md_step(s.positions, v0, m["n_steps"], kind=m["kind"], dt=m["dt"], temperature=m["temperature"],
                                 friction=m["friction"], mass=m["mass"], seed=m["seed"])
    assert np.abs(x - z["md_positions"]).max() < 1e-12 and np.abs(v - z["md_velocities"]).max() < 1e-12
    assert st.potential == pytest.approx(m["potential"], rel=1e-11) and st.kinetic == pytest.approx(m["kinetic"], rel=1e-11)


def test_oracle_amd_reproduces_fixture():
    from oracle.oracle import Oracle, md_velocities
    s, z, c = load_case("gw_512_cutoff")
    m = c["amd"]
    v0 = md_velocities(s.n_beads, 310.0, m["mass"], m["velocity_seed"])
    x, v, st = Oracle(s).md_step(s.positions, v0, m["n_steps"], kind="amd", dt=m["dt"], mass=m["mass"],
                                 amd_alpha=m["alpha"], amd_e=m["e"])
    assert np.abs(x - z["amd_positions"]).max() < 1e-12 and np.abs(v - z["amd_velocities"]).max() < 1e-12
    assert st.potential == pytest.approx(m["potential"], rel=1e-10) and st.kinetic == pytest.approx(m["kinetic"], rel=1e-10)
    assert np.abs(z["amd_positions"] - z["md_positions"]).max() > 1e-6   # a different trajectory than the Langevin one


def test_closed_forms_and_philox_vectors():
    """Hand-derivable values and the published Random123 vectors: independent of any oracle code."""
    from oracle.oracle import Oracle, philox4x32_10
    cf = META["closed_forms"]
    off = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False, NB_CUTOFF=0.0)
    two = ChromatinSystem(2, np.array([[0, 0, 0], [0.1, 0, 0.0]]), np.array([7]), np.zeros(2, np.int8),
                          ff=ForceFieldParams(**off))
    et, F = Oracle(two, as_float32_inputs=False).eval()
    assert et[0] == pytest.approx(cf["ev_two_beads_r0.1"]["energy"], rel=1e-12)
    assert abs(F[0, 0]) == pytest.approx(cf["ev_two_beads_r0.1"]["force"], rel=1e-12)
    aa = ChromatinSystem(2, np.array([[0, 0, 0], [0.15, 0, 0.0]]), np.array([7]), np.array([1, 2], np.int8),
                         ff=ForceFieldParams(EV_USE_EXCLUDED_VOLUME=False, COB_USE_COMPARTMENT_BLOCKS=True, **off))
    assert Oracle(aa, as_float32_inputs=False).eval()[0][1] == pytest.approx(cf["cob_AA_r0.15"]["energy"], rel=1e-12)
    for v in META["philox4x32_10"]:
        assert [int(t) for t in philox4x32_10(v["counter"], v["key"])] == v["out"]


# ---- GPU: the HIP path against the committed vectors ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_fixture(name):
    from multimm_amd.engine import engine_for
    s, z, c = load_case(name)
    with engine_for(s) as eng:
        et, F = eng.compute()
    ref_e, ref_f = z["energy_terms"], z["forces"]
    scale = np.abs(ref_e).sum()
    assert np.all(np.abs(et - ref_e) <= E_RTOL * scale + E_ATOL), (et, ref_e)
    assert np.abs(F - ref_f).max() <= F_RTOL * np.abs(ref_f).max() + F_ATOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in CASES if "minimize" in META["cases"][n]])
def test_gpu_minimizer_against_fixture(name):
    """Same number of L-BFGS iterations from the same start: fp32 and fp64 trajectories separate slowly, so the
    energy reached must agree to a fraction of the decrease (5 %), and the device's own energy bookkeeping must
    agree with the oracle at the returned point (2e-5)."""
    from multimm_amd.engine import engine_for
    from oracle.oracle import Oracle
    s, z, c = load_case(name)
    m = c["minimize"]
    with engine_for(s) as eng:
        st = eng.minimize(tolerance=0.0, max_iters=m["max_iters"])
        x = eng.get_positions()
    assert st.iterations == m["iterations"] and st.status == m["status"]
    tol = E_RTOL * np.abs(z["energy_terms"]).sum() + E_ATOL  # the total is a sum of large cancelling terms
    assert abs(st.e_initial - m["e_initial"]) <= tol
    drop = m["e_initial"] - m["e_final"]
    assert abs(st.e_final - m["e_final"]) <= 0.05 * drop
    assert abs(Oracle(s).energy(x) - st.e_final) <= tol


@pytest.mark.gpu
def test_gpu_md_against_fixture():
    from multimm_amd.engine import engine_for
    from oracle.oracle import md_velocities
    s, z, c = load_case("gw_512_cutoff")
    m = c["md"]
    with engine_for(s) as eng:
        eng.md_configure(m["kind"], dt_ps=m["dt"], temperature_K=m["temperature"], friction_per_ps=m["friction"],
                         mass_amu=m["mass"], seed=m["seed"])
        eng.set_velocities_to_temperature(m["temperature"], seed=m["velocity_seed"])
        st = eng.md_step(m["n_steps"])
        x, v = eng.get_positions().astype(np.float64), eng.get_velocities().astype(np.float64)
    travelled = np.abs(z["md_positions"] - s.positions).max()
    assert np.abs(x - z["md_positions"]).max() <= 2e-6 + 1e-3 * travelled
    assert np.abs(v - z["md_velocities"]).max() <= 1e-3 * np.abs(z["md_velocities"]).max()
    assert st.kinetic == pytest.approx(m["kinetic"], rel=1e-4)
    assert st.potential == pytest.approx(m["potential"], rel=2e-5, abs=1e-2 + 2e-5 * np.abs(z["energy_terms"]).sum())


@pytest.mark.gpu
def test_gpu_amd_against_fixture():
    from multimm_amd.engine import engine_for
    from oracle.oracle import md_velocities
    s, z, c = load_case("gw_512_cutoff")
    m = c["amd"]
    with engine_for(s) as eng:
        eng.md_configure("amd", dt_ps=m["dt"], mass_amu=m["mass"], amd_alpha=m["alpha"], amd_e=m["e"])
        eng.set_velocities(md_velocities(s.n_beads, 310.0, m["mass"], m["velocity_seed"]).astype(np.float32))
        st = eng.md_step(m["n_steps"])
        x, v = eng.get_positions().astype(np.float64), eng.get_velocities().astype(np.float64)
    travelled = np.abs(z["amd_positions"] - s.positions).max()
    assert np.abs(x - z["amd_positions"]).max() <= 2e-6 + 1e-3 * travelled
    assert np.abs(v - z["amd_velocities"]).max() <= 1e-3 * np.abs(z["amd_velocities"]).max()
    assert st.kinetic == pytest.approx(m["kinetic"], rel=1e-4)
    assert st.potential == pytest.approx(m["potential"], rel=2e-5, abs=1e-2 + 2e-5 * np.abs(z["energy_terms"]).sum())
