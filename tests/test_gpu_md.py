"""GPU parity of the MD integrators against the fp64 restatement in oracle/, through the C ABI.

Reference call sites: integrators model.py:768-808, setVelocitiesToTemperature model.py:878, simulation.step /
getState model.py:931-937.  Both sides draw their noise from Philox4x32-10 indexed by (seed, bead, step), so a
short trajectory can be compared bead by bead; OpenMM's own generator is not reproduced (statistical tests
below cover what must hold for any generator).

Tolerances: positions 2e-6 nm + 1e-3 of the distance travelled, velocities 1e-3 of the thermal velocity;
energies 2e-5 * sum|E_t| + 1e-3 kJ/mol (as in test_gpu_parity.py), kinetic energy 1e-4 relative.
"""
import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd.engine import BEAD_MASS_AMU, MMXError, engine_for

pytestmark = pytest.mark.gpu

KB = 0.008314462618
SIGMA_V = float(np.sqrt(KB * 310.0 / BEAD_MASS_AMU))  # thermal velocity per component, nm/ps


def _relaxed(n=4000, iters=300, **ff):
    """A relaxed start (MD after minimization, as MultiMM.run does: model.py:1228-1234)."""
    s = synthetic_system("gw_200k", n_beads=n, seed=2, **ff)
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=iters)
        x = eng.get_positions()
    from dataclasses import replace
    return replace(s, positions=x.astype(np.float64))  # the confinement centre follows (mass centre of the start)


def test_initial_velocities_match_restatement():
    from oracle.oracle import md_velocities
    s = synthetic_system("chr1_50k", n_beads=50000)
    with engine_for(s) as eng:
        eng.md_configure("langevin", seed=11)
        eng.set_velocities_to_temperature(310.0, seed=5)
        v = eng.get_velocities().astype(np.float64)
    ref = md_velocities(s.n_beads, 310.0, BEAD_MASS_AMU, 5)
    assert np.abs(v - ref).max() <= 2e-6 * SIGMA_V * 6.0
    # Maxwell-Boltzmann: zero mean, variance kT/m per component (150k samples: 4-sigma bands)
    assert abs(v.mean()) < 4 * SIGMA_V / np.sqrt(v.size)
    assert abs(v.var() / SIGMA_V ** 2 - 1.0) < 4 * np.sqrt(2.0 / v.size)


@pytest.mark.parametrize("kind,dt,friction", [("verlet", 0.01, 0.0), ("langevin", 0.01, 0.5), ("langevin", 0.001, 0.5),
                                               ("brownian", 0.002, 50.0)])
def test_short_trajectory_matches_restatement(kind, dt, friction):
    from oracle.oracle import Oracle, md_velocities
    s = _relaxed()
    n_steps = 60
    v0 = md_velocities(s.n_beads, 310.0, BEAD_MASS_AMU, 3).astype(np.float32)
    with engine_for(s) as eng:
        eng.md_configure(kind, dt_ps=dt, temperature_K=310.0, friction_per_ps=friction, seed=17)
        eng.set_velocities(v0)
        x0 = eng.get_positions().astype(np.float64)
        st = eng.md_step(n_steps)
        x = eng.get_positions().astype(np.float64)
        v = eng.get_velocities().astype(np.float64)
    assert np.array_equal(x0, s.positions)
    orc = Oracle(s)
    xr, vr, sr = orc.md_step(x0, v0.astype(np.float64), n_steps, kind=kind, dt=dt, temperature=310.0,
                             friction=friction, mass=BEAD_MASS_AMU, seed=17)
    travelled = np.abs(xr - x0).max()
    assert travelled > 50 * 2e-6, "the test must move the beads by much more than the tolerance"
    assert np.abs(x - xr).max() <= 2e-6 + 1e-3 * travelled
    vscale = max(SIGMA_V, np.abs(vr).max())
    assert np.abs(v - vr).max() <= 1e-3 * vscale
    assert st.step_count == n_steps and st.n_steps == n_steps
    scale_e = np.abs(np.array(sr.eterms[:])).sum()
    assert abs(st.potential - sr.potential) <= 2e-5 * scale_e + 1e-3
    assert abs(st.kinetic - sr.kinetic) <= 1e-4 * sr.kinetic + 1e-6
    assert abs(st.temperature - sr.temperature) <= 1e-4 * sr.temperature + 1e-6


@pytest.mark.parametrize("regime", ["boosted", "above_threshold"])
def test_amd_trajectory_matches_restatement(regime):
    """SIM_INTEGRATOR_TYPE = amd (model.py:794-800): the boost factor is recomputed from the potential energy of the
    current positions at every step, on the device.  "boosted": E above U for the whole run; "above_threshold": E
    below U, where aMD must reproduce the Verlet trajectory of the same engine bit for bit."""
    from oracle.oracle import Oracle, md_velocities
    s = _relaxed()
    n_steps, dt = 60, 0.01
    v0 = md_velocities(s.n_beads, 310.0, BEAD_MASS_AMU, 3).astype(np.float32)
    with engine_for(s) as eng:
        eng.set_option("deterministic", 1)   # the Verlet comparison below is bit for bit
        et, _ = eng.compute()
        u0 = float(np.sum(et))
        alpha, e = (2000.0, u0 + 3000.0) if regime == "boosted" else (100.0, u0 - 1e9)
        eng.md_configure("amd", dt_ps=dt, amd_alpha=alpha, amd_e=e)
        eng.set_velocities(v0)
        x0 = eng.get_positions().astype(np.float64)
        st = eng.md_step(n_steps)
        x, v = eng.get_positions().astype(np.float64), eng.get_velocities().astype(np.float64)
        if regime == "above_threshold":
            eng.set_positions(x0)
            eng.md_configure("verlet", dt_ps=dt)
            eng.set_velocities(v0)
            eng.md_step(n_steps)
            assert np.array_equal(eng.get_positions().astype(np.float64), x)
            assert np.array_equal(eng.get_velocities().astype(np.float64), v)
    xr, vr, sr = Oracle(s).md_step(x0, v0.astype(np.float64), n_steps, kind="amd", dt=dt, mass=BEAD_MASS_AMU,
                                   amd_alpha=alpha, amd_e=e)
    if regime == "boosted":   # the two trajectories must differ from plain Verlet, i.e. the boost was applied
        xv, _, _ = Oracle(s).md_step(x0, v0.astype(np.float64), n_steps, kind="verlet", dt=dt, mass=BEAD_MASS_AMU)
        assert np.abs(xv - xr).max() > 100 * 2e-6
    travelled = np.abs(xr - x0).max()
    assert np.abs(x - xr).max() <= 2e-6 + 1e-3 * travelled
    assert np.abs(v - vr).max() <= 1e-3 * max(SIGMA_V, np.abs(vr).max())
    scale_e = np.abs(np.array(sr.eterms[:])).sum()
    assert abs(st.potential - sr.potential) <= 2e-5 * scale_e + 1e-3
    assert abs(st.kinetic - sr.kinetic) <= 1e-4 * sr.kinetic + 1e-6   # unshifted m v^2 / 2 on both sides


def test_md_with_exact_all_pairs_forces():
    """NB_CUTOFF <= 0 (the reference's NoCutoff semantics): MD on the all-pairs kernel, against the restatement."""
    from oracle.oracle import Oracle, md_velocities
    s = _relaxed(n=1500, iters=150, NB_CUTOFF=0.0)
    v0 = md_velocities(s.n_beads, 310.0, BEAD_MASS_AMU, 3).astype(np.float32)
    with engine_for(s) as eng:
        eng.md_configure("langevin", dt_ps=0.01, seed=5)
        eng.set_velocities(v0)
        st = eng.md_step(40)
        x = eng.get_positions().astype(np.float64)
    xr, vr, sr = Oracle(s).md_step(s.positions, v0.astype(np.float64), 40, kind="langevin", dt=0.01, seed=5,
                                   mass=BEAD_MASS_AMU)
    travelled = np.abs(xr - s.positions).max()
    assert np.abs(x - xr).max() <= 2e-6 + 1e-3 * travelled
    assert abs(st.kinetic - sr.kinetic) <= 1e-4 * sr.kinetic


def test_amd_step_calls_compose_bitwise():
    """aMD reads the potential energy of the current positions in the first step of a call; the forces (and that energy)
    are cached between calls: step(30) + step(30) must be step(60), also in the boosted regime where the energy
    decides the force scale (run_md() calls md_step in chunks of the sampling interval, model.py:907-995)."""
    s = _relaxed(n=3000, iters=100)
    out = []
    for chunks in ((60,), (30, 30), (1, 59)):
        with engine_for(s) as eng:
            eng.set_option("deterministic", 1)
            et, _ = eng.compute()
            eng.md_configure("amd", dt_ps=0.01, amd_alpha=2000.0, amd_e=float(np.sum(et)) + 3000.0)
            eng.set_velocities_to_temperature(310.0, seed=5)
            for c in chunks:
                st = eng.md_step(c)
            out.append((eng.get_positions(), eng.get_velocities(), st.potential, st.step_count))
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1]) and o[2:] == out[0][2:]


def test_step_calls_compose_bitwise():
    """step(20) == step(7) + step(13): the step counter indexes the noise, forces are cached between calls."""
    s = _relaxed(n=3000, iters=100)
    out = []
    for chunks in ((20,), (7, 13)):
        with engine_for(s) as eng:
            eng.set_option("deterministic", 1)
            eng.md_configure("langevin", dt_ps=0.005, seed=9)
            eng.set_velocities_to_temperature(310.0, seed=9)
            for c in chunks:
                st = eng.md_step(c)
            out.append((eng.get_positions(), eng.get_velocities(), st.potential, st.kinetic, st.step_count))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert out[0][2:] == out[1][2:]


def test_verlet_conserves_energy():
    s = _relaxed(n=20000, iters=400)
    with engine_for(s) as eng:
        eng.md_configure("verlet", dt_ps=0.005)
        eng.set_velocities_to_temperature(310.0, seed=1)
        e = []
        for _ in range(6):
            st = eng.md_step(200)
            e.append((st.potential + st.kinetic, st.kinetic))
    tot = np.array([a for a, _ in e])
    ke = e[0][1]
    assert np.abs(tot - tot[0]).max() < 0.01 * ke, (tot, ke)


def test_langevin_thermostat_reaches_temperature():
    """From rest, a strongly coupled Langevin bath (friction 20/ps) brings 3N kinetic degrees of freedom to
    T within statistical error (60 000 dof: sigma_T/T = 0.6 %) -- holds for any correct noise generator."""
    s = _relaxed(n=20000, iters=400)
    with engine_for(s) as eng:
        eng.md_configure("langevin", dt_ps=0.005, temperature_K=310.0, friction_per_ps=20.0, seed=4)
        eng.set_velocities(np.zeros((s.n_beads, 3), np.float32))
        st0 = eng.md_step(0)
        assert st0.kinetic < 1e-3 * 1.5 * s.n_beads * KB * 310.0 + 50.0
        eng.md_step(400)  # 2 ps = 40 relaxation times
        temps = [eng.md_step(40).temperature for _ in range(10)]
    assert abs(np.mean(temps) / 310.0 - 1.0) < 0.03, temps


def test_md_errors():
    s = synthetic_system("region_5k", n_beads=500)
    with engine_for(s) as eng:
        with pytest.raises(MMXError):
            eng.md_step(1)  # not configured
        with pytest.raises(MMXError):
            eng.md_configure("variable_langevin")
        with pytest.raises(MMXError):
            eng.md_configure("amd", amd_alpha=0.0)
        with pytest.raises(MMXError):
            eng.md_configure("brownian", friction_per_ps=0.0)
        with pytest.raises(MMXError):
            eng.md_configure("langevin", dt_ps=0.0)
        # a far too large step blows up on the Hilbert lattice start: reported, not hidden
        eng.md_configure("verlet", dt_ps=50.0)
        eng.set_velocities_to_temperature(310.0, seed=0)
        with pytest.raises(MMXError):
            for _ in range(20):
                eng.md_step(50)


def test_md_full_size_200k_runs():
    s = synthetic_system("gw_200k")
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=100)
        eng.md_configure("langevin", seed=0)
        eng.set_velocities_to_temperature(310.0, seed=0)
        st = eng.md_step(100)
        assert st.step_count == 100 and np.isfinite(st.potential)
        assert 0.5 < st.temperature / 310.0 < 3.0
        print(f"200k MD: {100 / st.seconds:.0f} steps/s")


def test_multimm_run_with_md_writes_reference_outputs(tmp_path):
    """REGION-like run with SIM_RUN_MD: the files run_md() leaves behind (model.py:907-995) and md_history."""
    import os
    from multimm_amd import cif
    from multimm_amd.config import load_config
    from multimm_amd.dcd import read_dcd
    from multimm_amd.model import MultiMM
    out = tmp_path / "out"
    cfg = load_config({"PLATFORM": "MI355X", "N_BEADS": 2000, "OUT_PATH": str(out), "SIM_RUN_MD": "True",
                       "SIM_N_STEPS": 200, "SIM_SAMPLING_STEP": 50, "TRJ_FRAMES": 20, "MIN_MAX_ITERATIONS": 200,
                       "SIM_INTEGRATOR_STEP": "5 femtosecond"})
    m = MultiMM(cfg)
    m.run()
    assert m.md_history["step"] == [50, 100, 150, 200]
    assert m.md_history["temperature"] == [310.0] * 4          # langevin: integrator.getTemperature()
    tot = np.array(m.md_history["potential"]) + np.array(m.md_history["kinetic"])
    assert np.allclose(tot, m.md_history["total"]) and np.all(np.isfinite(tot))
    for i in range(1, 5):
        assert os.path.exists(out / "md_frames" / f"frame_{i}.cif")
    after = cif.read_positions(str(out / "model" / "MultiMM_afterMD.cif"))
    assert after.shape == (2000, 3) and np.abs(after - m.state_positions).max() <= 5.1e-5
    d = read_dcd(str(out / "metadata" / "MultiMM_annealing.dcd"))
    assert (d["n_frames"], d["n_atoms"], d["interval"], d["first_step"], d["last_step"]) == (20, 2000, 10, 10, 210)
    assert np.abs(d["frames_nm"][-1] - m.state_positions).max() < 1e-5
    params = open(out / "metadata" / "parameters.txt").read()   # save_args_to_txt, model.py:1246
    assert "SIM_N_STEPS = 200\n" in params and "EV_POWER = 6.0\n" in params and "LOC_START = \n" in params
    minimized = cif.read_positions(str(out / "model" / "MultiMM_minimized.cif"))
    moved = np.abs(after - minimized).max()
    assert 1e-4 < moved < 0.5, moved
