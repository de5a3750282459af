"""CPU tests of the oracle itself: hand-derivable known answers, finite differences
of its own energy, C vs numpy restatement, reference index quirks, Hilbert start, L-BFGS restatement.
The reference holds no golden vectors for this path ("parity unpinned"): these tests are what pins it."""
import dataclasses

import numpy as np
import pytest

from multimm_amd import synthetic_system, hilbert_points, backbone_flags
from multimm_amd.system import ChromatinSystem, ForceFieldParams, chrom_strength_per_bead, gw_chr_ends, set_radiuses

OFF = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False,
           EV_USE_EXCLUDED_VOLUME=False, NB_CUTOFF=0.0)


def _sys(pos, labels=None, chr_ends=None, loops=None, **ff):
    pos = np.asarray(pos, dtype=np.float64)
    n = len(pos)
    kw = dict(OFF)
    kw.update(ff)
    m, nn, r0 = loops if loops else ([], [], [])
    return ChromatinSystem(n, pos, np.array(chr_ends if chr_ends is not None else [n + 5], np.int32),
                           np.zeros(n, np.int8) if labels is None else np.asarray(labels, np.int8),
                           loop_m=m, loop_n=nn, loop_r0=r0, ff=ForceFieldParams(**kw))


def _eval(s, **kw):
    from oracle.oracle import Oracle
    return Oracle(s, as_float32_inputs=False, **kw).eval()


def test_kat_excluded_volume_two_beads(oracle_lib):
    et, F = _eval(_sys([[0, 0, 0], [0.1, 0, 0]], EV_USE_EXCLUDED_VOLUME=True))
    e = 100.0 * (0.1 / 0.15) ** 6
    assert et[0] == pytest.approx(e, rel=1e-12) and e == pytest.approx(8.77915, rel=1e-5)
    assert F[1, 0] == pytest.approx(6 * e / 0.15, rel=1e-12) and F[0, 0] == pytest.approx(-351.166, rel=1e-5)


def test_kat_compartment_gaussian(oracle_lib):
    pos = [[0, 0, 0], [0.15, 0, 0]]
    for la, lb, want in ((1, 2, -np.exp(-0.5)), (-1, -2, -2 * np.exp(-0.5)), (1, -1, 0.0), (0, 1, 0.0)):
        et, _ = _eval(_sys(pos, labels=[la, lb], COB_USE_COMPARTMENT_BLOCKS=True))
        assert et[1] == pytest.approx(want, abs=1e-12)
    # SCB: only identical labels interact, amplitudes Ea1(2), Ea2(1), Eb1(-1), Eb2(-2) = 1, 1.33, 1.66, 2
    for lab, amp in ((2, 1.0), (1, 1.33), (-1, 1.66), (-2, 2.0)):
        et, _ = _eval(_sys(pos, labels=[lab, lab], SCB_USE_SUBCOMPARTMENT_BLOCKS=True))
        assert et[1] == pytest.approx(-amp * np.exp(-0.5), rel=1e-12)
    et, _ = _eval(_sys(pos, labels=[1, 2], SCB_USE_SUBCOMPARTMENT_BLOCKS=True))
    assert et[1] == 0.0


def test_kat_angle_and_bond(oracle_lib):
    straight = [[0, 0, 0], [0.1, 0, 0], [0.2, 0, 0]]
    right = [[0, 0, 0], [0.1, 0, 0], [0.1, 0.1, 0]]
    et, F = _eval(_sys(straight, POL_USE_HARMONIC_ANGLE=True))
    assert et[3] == pytest.approx(0.0, abs=1e-20) and np.abs(F).max() == 0.0
    et, F = _eval(_sys(right, POL_USE_HARMONIC_ANGLE=True))
    assert et[3] == pytest.approx(0.5 * 100 * (np.pi / 2) ** 2, rel=1e-12)  # 123.370
    # the angle wants to open: end beads are pushed away from each other, |F| = k*(pi/2)/0.1
    assert F[0, 1] == pytest.approx(-100 * (np.pi / 2) / 0.1, rel=1e-9) and F[2, 0] == pytest.approx(1570.796, rel=1e-6)
    assert np.abs(F.sum(0)).max() < 1e-9
    et, F = _eval(_sys([[0, 0, 0], [0.12, 0, 0]], POL_USE_HARMONIC_BOND=True))
    assert et[2] == pytest.approx(0.5 * 3e5 * 0.02 ** 2) and F[0, 0] == pytest.approx(3e5 * 0.02)


def test_kat_container_lamina_central(oracle_lib):
    n = 1001
    R1, R2, _ = set_radiuses(n, 0.1)
    # the other 1000 beads: antipodal pairs on the mid-shell sphere (force-free annulus, mean exactly 0)
    half = np.random.default_rng(0).normal(size=(500, 3))
    half *= (0.5 * (R1 + R2)) / np.linalg.norm(half, axis=1, keepdims=True)
    rest = np.concatenate([half, -half])

    def at_r(r, label, **ff):
        p = np.zeros((n, 3))
        p[1:] = rest
        p[0, 0] = r * n / (n - 1)              # centre = p0/n  =>  |p0 - centre| = r
        s = _sys(p, labels=[label] + [0] * (n - 1), **ff)
        assert np.linalg.norm(s.positions[0] - s.centre) == pytest.approx(r, rel=1e-9)
        return s

    sc = dict(SC_USE_SPHERICAL_CONTAINER=True)
    et_all, _ = _eval(at_r(R2 + 0.1, 0, **sc))
    et_in, _ = _eval(at_r(0.5 * (R1 + R2), 0, **sc))
    assert et_all[5] - et_in[5] == pytest.approx(1000 * 0.01, rel=1e-6)       # C*(r-R2)^2 = 10
    ibl = dict(IBL_USE_B_LAMINA_INTERACTION=True)
    assert _eval(at_r(0.5 * (R1 + R2), -1, **ibl))[0][6] == pytest.approx(0.0, abs=1e-9)   # sin^8 = 1
    assert _eval(at_r(R1, -2, **ibl))[0][6] == pytest.approx(-400.0, rel=1e-9)
    assert _eval(at_r(R1, 2, **ibl))[0][6] == 0.0


def test_finite_difference_gradient_all_terms(oracle_lib):
    from oracle import oracle_np
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=300, jitter=0.02, NB_CUTOFF=0.0, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
                         CF_USE_CENTRAL_FORCE=True, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_DE=0.5)
    et, F = Oracle(s, as_float32_inputs=False).eval()
    en = oracle_np.energy_terms(s)
    for i, k in enumerate(("ev", "gauss", "bond", "angle", "loop", "container", "lamina", "central", "chb")):
        assert et[i] == pytest.approx(en[k], rel=1e-11, abs=1e-9), k
        assert et[i] != 0.0, k
    beads = [0, 1, 2, 5, 150, 298, 299] + list(np.unique(np.r_[s.loop_m, s.loop_n])[:3])
    fd = oracle_np.fd_forces(s, beads=beads)
    assert np.abs(fd - F[beads]).max() < 1e-3 * np.abs(F).max() * 1e-3 + 1e-3


def test_cutoff_cells_match_bruteforce(oracle_lib):
    from oracle import oracle_np
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=700, jitter=0.03, NB_CUTOFF=0.6)
    et, F = Oracle(s, as_float32_inputs=False).eval()
    en = oracle_np.energy_terms(s, cutoff=0.6)
    assert et[0] == pytest.approx(en["ev"], rel=1e-11) and et[1] == pytest.approx(en["gauss"], rel=1e-11)
    et0, _ = Oracle(s, cutoff=0.0, as_float32_inputs=False).eval()
    assert et0[0] > et[0]  # the truncated tail is repulsive energy


def test_backbone_quirks_follow_reference(oracle_lib):
    """bond (i,i+1) missing iff i in chr_ends; angle missing iff i in chr_ends or chr_ends-1 (model.py:629,712)."""
    from oracle.oracle import backbone_flags_c
    n = 40
    ce = np.array([0, 11, 25, n])
    f = backbone_flags(n, ce)
    assert np.array_equal(f, backbone_flags_c(n, ce))
    bonds = [i for i in range(n - 1) if i not in ce]
    angles = [i for i in range(n - 2) if (i not in ce) and (i not in ce - 1)]
    assert [i for i in range(n) if f[i] & 1] == bonds and [i for i in range(n) if f[i] & 2] == angles
    assert not (f[0] & 1) and (f[10] & 1) and not (f[11] & 1) and (f[12] & 1)   # boundary shifted by one bead
    assert not (f[10] & 2) and not (f[11] & 2) and (f[9] & 2)


def test_hilbert_curve_properties(oracle_lib):
    from oracle.oracle import hilbert_points_c
    p = hilbert_points(5000)
    assert p[:10].tolist() == [[0, 0, 0], [0, 1, 0], [1, 1, 0], [1, 0, 0], [1, 0, 1], [1, 1, 1], [0, 1, 1], [0, 0, 1],
                               [0, 0, 2], [0, 0, 3]]
    assert np.array_equal(p, hilbert_points_c(5000))
    assert np.all(np.abs(np.diff(p, axis=0)).sum(1) == 1)          # consecutive beads are lattice neighbours
    assert len({tuple(r) for r in p}) == len(p)
    for k in (1, 2, 3, 4):                                           # first 8^k points tile a 2^k cube
        assert p[:8 ** k].max() == 2 ** k - 1
    # the documented first-order curves of the hilbertcurve package (its README example for p = 1, n = 2, and the
    # same Gray-code walk in three dimensions): points_from_distances(range(2^n))
    assert hilbert_points(4, p=1, n=2).tolist() == [[0, 0], [0, 1], [1, 1], [1, 0]]
    assert hilbert_points(8, p=1, n=3).tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 1], [0, 1, 0], [1, 1, 0], [1, 1, 1],
                                                    [1, 0, 1], [1, 0, 0]]


def test_gw_chr_ends_and_strength():
    ce = gw_chr_ends(200000)
    assert len(ce) == 23 and ce[0] == 0 and ce[-1] == 200000 and np.all(np.diff(ce) > 0)
    w = chrom_strength_per_bead(ce, 200000)
    assert w.min() >= 0 and w.max() <= 1 and w[0] == pytest.approx(0.0)


def test_lbfgs_restatement_converges(oracle_lib):
    from oracle.oracle import Oracle
    s = synthetic_system("region_5k", n_beads=400, jitter=0.01, NB_CUTOFF=0.6)
    orc = Oracle(s, as_float32_inputs=False)
    x, st = orc.minimize(tolerance=10.0, max_iters=0)
    assert st.status == 0 and st.iterations > 5 and st.e_final < st.e_initial
    _, F = orc.eval(x)
    # OpenMM stop rule: |g| / max(1,|x|) <= tol / max(1, rms|x_i|)
    xr = np.asarray(s.positions)
    eps = 10.0 / max(1.0, np.sqrt((xr * xr).sum() / len(xr)))
    assert np.linalg.norm(F) / max(1.0, np.linalg.norm(x)) <= eps * (1 + 1e-9)
    # fixed iteration budget is honoured exactly
    _, st2 = orc.minimize(tolerance=0.0, max_iters=7)
    assert st2.iterations == 7 and st2.status == 1 and st2.evaluations >= 8


# ---- MD restatement -------------------------------------------------------------------
def test_philox_known_answers():
    """Philox4x32-10 against the three known-answer vectors published with Random123 (kat_vectors):
    zero counter/key, all-ones, and the digits-of-pi vector."""
    from oracle.oracle import philox4x32_10
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert [int(v) for v in philox4x32_10(ctr, key)] == want


def test_md_normals_and_initial_velocities():
    from oracle.oracle import md_velocities, normal3
    z = np.array([normal3(i, 3, 0, 99) for i in range(20000)])
    assert abs(z.mean()) < 4 / np.sqrt(z.size) and abs(z.var() - 1.0) < 4 * np.sqrt(2.0 / z.size)
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03 and abs(np.corrcoef(z[:-1, 2], z[1:, 2])[0, 1]) < 0.03
    assert not np.array_equal(normal3(5, 3, 0, 99), normal3(5, 4, 0, 99))
    v = md_velocities(30000, 310.0, 16427.889, 1)
    kT_over_m = 0.008314462618 * 310.0 / 16427.889
    assert abs(v.var() / kT_over_m - 1.0) < 4 * np.sqrt(2.0 / v.size)


def test_md_integrators_basic_properties():
    """Verlet: energy conserved and time-reversible; Langevin with zero friction and temperature == Verlet;
    Brownian at T = 0 is steepest descent (energy decreases)."""
    from oracle.oracle import Oracle, md_velocities
    s = synthetic_system("region_5k", n_beads=300, jitter=0.01, seed=1)
    orc = Oracle(s)
    x0, _ = orc.minimize(tolerance=0.0, max_iters=200)
    v0 = md_velocities(s.n_beads, 310.0, 16427.889, 2)
    kw = dict(dt=0.005, mass=16427.889)
    x1, v1, st1 = orc.md_step(x0, v0, 200, kind="verlet", **kw)
    _, _, st0 = orc.md_step(x0, v0, 0, kind="verlet", **kw)
    e0, e1 = st0.potential + st0.kinetic, st1.potential + st1.kinetic
    assert abs(e1 - e0) < 2e-3 * st0.kinetic, (e0, e1, st0.kinetic)
    assert st1.step_count == 200
    # leap-frog reversibility: the stored velocity is v_{n-1/2}; restarting from (x_n, -v_{n+1/2}) retraces
    # the positions exactly (to rounding), so 200 steps later the start is recovered
    orc64 = Oracle(s, as_float32_inputs=False)
    xa, va, _ = orc64.md_step(x0, v0, 200, kind="verlet", **kw)
    _, Fa = orc64.eval(xa)
    xb, _, _ = orc64.md_step(xa, -(va + kw["dt"] * Fa / kw["mass"]), 200, kind="verlet", **kw)
    assert np.abs(xb - x0).max() < 1e-9
    xl, vl, _ = orc.md_step(x0, v0, 50, kind="langevin", temperature=0.0, friction=0.0, **kw)
    xv, vv, _ = orc.md_step(x0, v0, 50, kind="verlet", **kw)
    assert np.allclose(xl, xv, atol=1e-12) and np.allclose(vl, vv, atol=1e-12)
    _, _, sb0 = orc.md_step(s.positions, 0 * v0, 0, kind="brownian", temperature=0.0, friction=50.0, **kw)
    _, _, sb1 = orc.md_step(s.positions, 0 * v0, 20, kind="brownian", temperature=0.0, friction=50.0, dt=1e-4,
                            mass=16427.889)
    assert sb1.potential < sb0.potential


def test_amd_integrator_known_answers():
    """mm.amd.AMDIntegrator(dt, alpha, E), model.py:794-800: f' = f (alpha/(alpha+E-U))^2 while U <= E, f otherwise.
    E far below U: exactly the Verlet trajectory.  E above U: the first step is the Verlet step with the force scaled by
    the hand-computed boost; the kinetic energy carries no half-step shift (CustomIntegrator default m v^2 / 2)."""
    from oracle.oracle import Oracle, md_velocities
    s = synthetic_system("region_5k", n_beads=300, jitter=0.01, seed=1)
    orc = Oracle(s, as_float32_inputs=False)   # eval() and md_step() must see the very same positions
    x0, _ = orc.minimize(tolerance=0.0, max_iters=200)
    v0 = md_velocities(s.n_beads, 310.0, 16427.889, 2)
    kw = dict(dt=0.005, mass=16427.889)
    et, F = orc.eval(x0)
    u0 = float(np.sum(et))
    xv, vv, _ = orc.md_step(x0, v0, 40, kind="verlet", **kw)
    xa, va, _ = orc.md_step(x0, v0, 40, kind="amd", amd_alpha=100.0, amd_e=u0 - 1e9, **kw)
    assert np.array_equal(xa, xv) and np.array_equal(va, vv)
    alpha, e = 250.0, u0 + 500.0
    boost = (alpha / (alpha + e - u0)) ** 2
    assert boost == pytest.approx((250.0 / 750.0) ** 2)
    x1, v1, st1 = orc.md_step(x0, v0, 1, kind="amd", amd_alpha=alpha, amd_e=e, **kw)
    v_ref = v0 + kw["dt"] * boost * F / kw["mass"]
    assert np.allclose(v1, v_ref, rtol=0, atol=1e-15) and np.allclose(x1, x0 + kw["dt"] * v_ref, rtol=0, atol=1e-15)
    assert st1.kinetic == pytest.approx(0.5 * kw["mass"] * np.sum(v1 * v1), rel=1e-12)
    # the boost flattens the landscape: with E above U the boosted run gains less kinetic energy from the same
    # forces than the plain one over the first steps
    _, _, sb = orc.md_step(x0, 0 * v0, 5, kind="amd", amd_alpha=alpha, amd_e=e, **kw)
    _, _, sp = orc.md_step(x0, 0 * v0, 5, kind="amd", amd_alpha=alpha, amd_e=u0 - 1e9, **kw)
    assert sb.kinetic < sp.kinetic


# ---- alternative functional forms (config.py:269-312) ------------------------------------
def test_kat_alternative_pair_forms(oracle_lib):
    """Hand-derivable values of the non-default pair forms (model.py:205-209, 262-288, 340-377)."""
    two = lambda r: [[0, 0, 0], [r, 0, 0]]
    ev = dict(EV_USE_EXCLUDED_VOLUME=True, EV_FORCE_TYPE="gaussian_core")
    et, F = _eval(_sys(two(0.1), **ev))
    assert et[0] == pytest.approx(100.0 * np.exp(-0.5), rel=1e-12)            # eps*exp(-r^2/(2 sigma^2)), r = sigma
    assert F[0, 0] == pytest.approx(-et[0] * 0.1 / 0.01, rel=1e-12)           # repulsive: -dE/dr = E r/sigma^2
    cob = dict(COB_USE_COMPARTMENT_BLOCKS=True)
    et, F = _eval(_sys(two(0.15), labels=[-1, -2], COB_FORCE_TYPE="yukawa", **cob))
    assert et[1] == pytest.approx(-2.0 * np.exp(-1.0) / 0.15, rel=1e-12)      # -Eb exp(-r/l)/r at r = l = r_comp
    assert F[0, 0] == pytest.approx(2.0 * np.exp(-1.0) / 0.15 * (1 / 0.15 + 1 / 0.15), rel=1e-12)  # attractive
    # the COB yukawa expression reads s1 twice (model.py:266-267): the LOWER index decides the amplitude
    assert _eval(_sys(two(0.15), labels=[1, -1], COB_FORCE_TYPE="yukawa", **cob))[0][1] == \
        pytest.approx(-1.0 * np.exp(-1.0) / 0.15, rel=1e-12)
    assert _eval(_sys(two(0.15), labels=[-1, 1], COB_FORCE_TYPE="yukawa", **cob))[0][1] == \
        pytest.approx(-2.0 * np.exp(-1.0) / 0.15, rel=1e-12)
    assert _eval(_sys(two(0.15), labels=[0, 1], COB_FORCE_TYPE="yukawa", **cob))[0][1] == 0.0
    assert _eval(_sys(two(0.15), labels=[1, -1], COB_FORCE_TYPE="gaussian", **cob))[0][1] == 0.0   # symmetric form
    et, F = _eval(_sys(two(0.1), labels=[1, 2], COB_FORCE_TYPE="theta", **cob))
    assert et[1] == -1.0 and np.all(F == 0.0)                                 # -Ea step(r_comp - r), no force
    assert _eval(_sys(two(0.15), labels=[1, 2], COB_FORCE_TYPE="theta", **cob))[0][1] == -1.0   # step(0) = 1
    assert _eval(_sys(two(0.151), labels=[1, 2], COB_FORCE_TYPE="theta", **cob))[0][1] == 0.0
    scb = dict(SCB_USE_SUBCOMPARTMENT_BLOCKS=True)
    assert _eval(_sys(two(0.3), labels=[1, 1], SCB_FORCE_TYPE="yukawa", **scb))[0][1] == \
        pytest.approx(-1.33 * np.exp(-2.0) / 0.3, rel=1e-12)
    assert _eval(_sys(two(0.3), labels=[1, 2], SCB_FORCE_TYPE="yukawa", **scb))[0][1] == 0.0
    # COB and SCB with different forms at once: two Force objects in the reference, their energies add
    both = _eval(_sys(two(0.1), labels=[2, 2], COB_FORCE_TYPE="theta", SCB_FORCE_TYPE="gaussian", **cob, **scb))[0][1]
    assert both == pytest.approx(-1.0 - 1.0 * np.exp(-0.01 / (2 * 0.0225)), rel=1e-12)


def test_kat_alternative_single_bead_and_bond_forms(oracle_lib):
    """Loops (model.py:664-701), chromosomal blocks (:424-443), lamina shells (:508-539), central (:588-612)."""
    two = lambda r: [[0, 0, 0], [r, 0, 0]]
    lo = dict(LE_USE_HARMONIC_BOND=True)
    k = 30000.0
    e = _eval(_sys(two(0.3), loops=([0], [1], [0.15]), LE_LOOP_FORCE_TYPE="fene_soft", **lo))[0][4]
    assert e == pytest.approx(k * 0.15 ** 2 / 2.0, rel=1e-12)                 # u = r0: k u^2/(1 + 1)
    e = _eval(_sys(two(0.3), loops=([0], [1], [0.2]), LE_LOOP_FORCE_TYPE="gaussian_tether", **lo))[0][4]
    assert e == pytest.approx(k * (1 - np.exp(-1.0)), rel=1e-12)              # u = sigma = r0/2
    e = _eval(_sys(two(0.3), loops=([0], [1], [0.2]), **lo))[0][4]
    assert e == pytest.approx(0.5 * k * 0.01, rel=1e-12)                      # default keeps the 1/2
    chb = dict(CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_KC=0.3, CHB_DE=0.5)
    s = _sys(two(2.0), chr_ends=[0, 2], CHB_FORCE_TYPE="gaussian", **chb)
    assert _eval(s)[0][8] == pytest.approx(-0.5 * np.exp(-0.3 * 4.0), rel=1e-12)
    s = _sys(two(2.0), chr_ends=[0, 2], CHB_FORCE_TYPE="saturating", **chb)
    assert _eval(s)[0][8] == pytest.approx(-0.5 / (1 + 0.3 * 4.0), rel=1e-12)
    n = 1001
    R1, R2, _ = set_radiuses(n, 0.1)
    half = np.random.default_rng(0).normal(size=(500, 3))
    half *= (0.5 * (R1 + R2)) / np.linalg.norm(half, axis=1, keepdims=True)

    def at_r(r, label, **ff):
        p = np.zeros((n, 3))
        p[1:] = np.concatenate([half, -half])
        p[0, 0] = r * n / (n - 1)
        return _sys(p, labels=[label] + [0] * (n - 1), chr_ends=[0, n], **ff)

    ibl = dict(IBL_USE_B_LAMINA_INTERACTION=True)
    sg = 0.1 * (R2 - R1)
    e = _eval(at_r(R1, -1, BLAMINA_FORCE_TYPE="gaussian_shell", **ibl))[0][6]
    assert e == pytest.approx(-400.0 * (1.0 + np.exp(-(R2 - R1) ** 2 / (2 * sg * sg))), rel=1e-9)
    e = _eval(at_r(R2, -2, BLAMINA_FORCE_TYPE="harmonic_shell", **ibl))[0][6]
    assert e == pytest.approx(400.0 * (0.5 * (R2 - R1)) ** 2, rel=1e-9)
    e = _eval(at_r(0.5 * (R1 + R2), -1, BLAMINA_FORCE_TYPE="logistic_shell", **ibl))[0][6]
    assert e == pytest.approx(-400.0 * 2.0 / (1.0 + np.exp(-10.0)), rel=1e-9)  # both walls 10 lambda away
    assert _eval(at_r(R1, 1, BLAMINA_FORCE_TYPE="harmonic_shell", **ibl))[0][6] == 0.0
    cf = dict(CF_USE_CENTRAL_FORCE=True)
    w = chrom_strength_per_bead(np.array([0, n]), n)
    s = at_r(0.5 * R1, 0, CENTRAL_FORCE_TYPE="gaussian", **cf)
    s.chrom_strength = w
    rr = np.linalg.norm(s.positions - s.centre, axis=1)
    assert _eval(s)[0][7] == pytest.approx(np.sum(-20.0 * w * np.exp(-rr ** 2 / (2 * (0.5 * R1) ** 2))), rel=1e-9)
    s = at_r(R1, 0, CENTRAL_FORCE_TYPE="logistic", **cf)
    s.chrom_strength = w
    rr = np.linalg.norm(s.positions - s.centre, axis=1)
    assert _eval(s)[0][7] == pytest.approx(np.sum(-20.0 * w / (1 + np.exp((rr - R1) / (0.2 * R1)))), rel=1e-9)


@pytest.mark.parametrize("forms", [
    dict(EV_FORCE_TYPE="gaussian_core", COB_FORCE_TYPE="yukawa", SCB_FORCE_TYPE="yukawa", CHB_FORCE_TYPE="gaussian",
         BLAMINA_FORCE_TYPE="gaussian_shell", CENTRAL_FORCE_TYPE="gaussian", LE_LOOP_FORCE_TYPE="fene_soft"),
    dict(COB_FORCE_TYPE="theta", SCB_FORCE_TYPE="yukawa", CHB_FORCE_TYPE="saturating",
         BLAMINA_FORCE_TYPE="harmonic_shell", CENTRAL_FORCE_TYPE="logistic", LE_LOOP_FORCE_TYPE="gaussian_tether"),
    dict(COB_FORCE_TYPE="gaussian", SCB_FORCE_TYPE="theta", BLAMINA_FORCE_TYPE="logistic_shell"),
])
@pytest.mark.parametrize("cutoff", [0.0, 0.6])
def test_finite_difference_gradient_alternative_forms(oracle_lib, forms, cutoff):
    """Analytic forces of every non-default form against central differences of the oracle's own energy."""
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=240, jitter=0.03, seed=5, NB_CUTOFF=cutoff, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
                         COB_USE_COMPARTMENT_BLOCKS=True, CF_USE_CENTRAL_FORCE=True, CHB_USE_CHROMOSOMAL_BLOCKS=True,
                         CHB_DE=0.5, **forms)
    orc = Oracle(s, as_float32_inputs=False)
    et, F = orc.eval()
    assert all(et[t] != 0.0 for t in (0, 1, 4, 6, 7, 8))
    x = s.positions.copy()
    beads = [0, 1, 7, 120, 239] + list(np.unique(np.r_[s.loop_m, s.loop_n])[:2])
    h = 1e-6
    for b in beads:
        for k in range(3):
            xp, xm = x.copy(), x.copy()
            xp[b, k] += h
            xm[b, k] -= h
            fd = -(orc.energy(xp) - orc.energy(xm)) / (2 * h)
            # theta steps / the cutoff make the energy piecewise: a bead pair within h of a discontinuity is
            # astronomically unlikely with jittered coordinates
            assert fd == pytest.approx(F[b, k], rel=2e-5, abs=2e-4 * max(1.0, np.abs(F[b]).max())), (b, k)


def test_form_name_tables_agree_and_unknown_names_raise():
    from multimm_amd.config import load_config
    from multimm_amd.system import FORM_NAMES, form_index
    from oracle.oracle import FORM_NAMES as ORC_NAMES
    assert FORM_NAMES == ORC_NAMES
    assert form_index("BLAMINA_FORCE_TYPE", "logistic_shell") == 3
    with pytest.raises(ValueError, match="Unknown EV_FORCE_TYPE"):
        load_config({"EV_FORCE_TYPE": "soft_lj"})
    assert load_config({"COB_FORCE_TYPE": "yukawa"}).ff.COB_FORCE_TYPE == "yukawa"


# ---- invariances (size-independent properties of the path) ----------------------------------------------
def test_oracle_invariances_translation_rotation_newton3(oracle_lib):
    """Internal terms (pair, backbone, loops, chromosomal blocks) are invariant under rigid motions and their
    forces sum to zero with zero net torque; the external terms move with the centre, which the reference ties
    to the mass centre of the START structure (model.py:759) -- so the whole energy is translation invariant
    when the system is rebuilt from the moved start."""
    from dataclasses import replace
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=400, jitter=0.03, seed=9, NB_CUTOFF=0.0, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
                         COB_USE_COMPARTMENT_BLOCKS=True, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_DE=0.5,
                         SC_USE_SPHERICAL_CONTAINER=False, IBL_USE_B_LAMINA_INTERACTION=False)
    et, F = Oracle(s, as_float32_inputs=False).eval()
    scale = np.abs(F).max()
    assert np.abs(F.sum(0)).max() < 1e-9 * scale * len(F)                        # Newton's third law
    x = s.positions - s.positions.mean(0)
    assert np.abs(np.cross(x, F).sum(0)).max() < 1e-8 * scale * len(F)          # no net torque
    rng = np.random.default_rng(1)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    moved = replace(s, positions=s.positions @ q.T + np.array([3.0, -2.0, 0.5]))
    et2, F2 = Oracle(moved, as_float32_inputs=False).eval()
    assert np.allclose(et2, et, rtol=1e-10, atol=1e-8)
    assert np.allclose(F2, F @ q.T, rtol=1e-8, atol=1e-8 * scale)
    # with the external terms on: translation of the start moves the centre with it
    full = synthetic_system("gw_200k", n_beads=400, jitter=0.03, seed=9, NB_CUTOFF=0.6, CF_USE_CENTRAL_FORCE=True)
    shifted = replace(full, positions=full.positions + np.array([1.0, 2.0, -3.0]))
    e_a, F_a = Oracle(full, as_float32_inputs=False).eval()
    e_b, F_b = Oracle(shifted, as_float32_inputs=False).eval()
    assert np.allclose(e_a, e_b, rtol=1e-9, atol=1e-7) and np.allclose(F_a, F_b, rtol=1e-7, atol=1e-7 * np.abs(F_a).max())


def test_oracle_cutoff_is_plain_truncation(oracle_lib):
    """CutoffNonPeriodic semantics: enlarging the cutoff only ADDS pairs; at a cutoff beyond the system size the
    cell-list path equals the all-pairs path exactly."""
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=300, jitter=0.02, seed=2)
    e = [Oracle(s, cutoff=rc, as_float32_inputs=False).eval()[0] for rc in (0.3, 0.45, 0.6, 0.9)]
    ev = [t[0] for t in e]
    assert ev[0] < ev[1] < ev[2] < ev[3]                     # EV is repulsive: every added pair adds energy
    gauss = [t[1] for t in e]
    assert gauss[0] > gauss[1] > gauss[2] >= gauss[3]        # the Gaussians are attractive
    big = np.linalg.norm(s.positions.max(0) - s.positions.min(0)) + 1.0
    e_all = Oracle(s, cutoff=0.0, as_float32_inputs=False).eval()[0]
    e_big = Oracle(s, cutoff=big, as_float32_inputs=False).eval()[0]
    assert np.allclose(e_all, e_big, rtol=1e-12, atol=1e-10)


# ---- the tuned CPU baseline (oracle/mmx_cpu_fast.c): bench.py's second cpu_baseline, itself checked against the restatement
def test_fast_cpu_baseline_equals_the_fp64_restatement(oracle_lib):
    """fp32 / SIMD / OpenMP evaluation of the default force field against the fp64 restatement on the same (fp32-rounded)
    inputs: every term within 2e-6 of the summed energies, forces within 4e-6 of the largest component -- the tolerances of
    the GPU parity tests -- on the lattice-dense jittered start, a relaxed state and a sparse one (cells nearly empty)."""
    from oracle.oracle import Oracle
    full = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
                IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)
    for n, jitter, scale in ((6000, 0.03, 1.0), (3000, 0.02, 2.5), (500, 0.05, 9.0)):
        s = synthetic_system("gw_200k", n_beads=n, jitter=jitter, seed=3, NB_CUTOFF=0.6, **full)
        s = dataclasses.replace(s, positions=s.positions * scale)      # thinned out: bonds stretched, few pairs inside the cutoff
        o = Oracle(s)
        et0, F0 = o.eval()
        et, F, swept = o.fast_eval()
        assert np.all(np.abs(et - et0) <= 2e-6 * np.abs(et0).sum() + 1e-3), (n, et, et0)
        assert np.abs(F - F0).max() <= 4e-6 * np.abs(F0).max() + 2e-3, n
        assert swept >= n
    # what it does not cover is refused, not approximated
    with pytest.raises(NotImplementedError):
        Oracle(synthetic_system("gw_200k", n_beads=300, NB_CUTOFF=0.0)).fast_eval()
    with pytest.raises(NotImplementedError):
        Oracle(synthetic_system("gw_200k", n_beads=300, NB_CUTOFF=0.6, EV_FORCE_TYPE="gaussian_core")).fast_eval()


def test_fast_cpu_baseline_minimizes_like_the_restatement(oracle_lib):
    """The same liblbfgs control flow on the tuned evaluation: identical iteration / evaluation counts on a jittered start
    over 25 iterations and energies that agree to fp32 trajectory noise."""
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=4000, jitter=0.03, seed=5, NB_CUTOFF=0.6)
    o = Oracle(s)
    x0, st0 = o.minimize(10.0, 25)
    x1, st1, swept = o.fast_minimize(10.0, 25)
    assert (st1.iterations, st1.status) == (st0.iterations, st0.status) == (25, 1)
    assert st1.e_initial == pytest.approx(st0.e_initial, rel=2e-6)
    assert abs(st1.e_final - st0.e_final) <= 2e-2 * abs(st0.e_initial - st0.e_final)
    assert abs(o.energy(x1) - st1.e_final) <= 2e-6 * abs(st1.e_final) + 1e-2     # the energy it reports is the energy of its point
    assert swept > 0
