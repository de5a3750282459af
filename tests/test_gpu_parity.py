"""GPU parity: libmmx.so (HIP, through the C ABI) against the fp64 CPU oracle on identical fp32 inputs.

Tolerances (stated per north_star: "within a stated fp32 tolerance"):
  energies  : |E_gpu - E_ref| <= 2e-6 * sum_terms|E_ref| + 1e-3 kJ/mol   (3e-5 on the lattice start with a cutoff,
              where thousands of pairs sit exactly on the discontinuity of the truncated potential)
  forces    : max_i |F_gpu - F_ref|_inf <= 4e-6 * max_i |F_ref|_inf + 2e-3 kJ/mol/nm   (1e-5 on that lattice start)
The absolute force floor covers fp32 round-off of bond lengths: k_bond * ulp(r) ~ 3e5 * 1e-8 nm.  (E_RTOL, E_ATOL, F_RTOL,
F_ATOL below; round 1 stood at 2e-5 / 5e-3 while the pair kernels worked in scaled length units.)
"""
import dataclasses

import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for, TERM_NAMES

pytestmark = pytest.mark.gpu

# fp32 tolerances of the parity claim: per-term energies within 2e-6 of sum_t |E_t| (+1e-3 kJ/mol), forces within 4e-6
# of the largest force component (+2e-3 kJ/mol/nm).  Measured (run with -s: every check prints an ACC line): <= 3e-7
# and <= 3.9e-6, typically 1e-6, on every kernel path (the pair kernels read the positions unscaled, bit for bit:
# round 1's scaled units needed 2e-5 here).
E_RTOL, E_ATOL = 2e-6, 1e-3
F_RTOL, F_ATOL = 4e-6, 2e-3
# The Hilbert LATTICE start puts thousands of pairs at exactly the cutoff distance (6 lattice steps = 0.6 nm), where
# the truncated potential jumps by E_ev(r_c) = 1.3e-3 kJ/mol: whether such a pair counts is decided by the last bit of
# r^2, in fp32 here and in fp64 in the oracle.  Only that test gets the wider energy band -- and a wider force band:
# the Gaussian's force does not vanish at the cutoff either (measured 5.1e-6 of max |F|, the same on both kernels).
E_RTOL_AT_CUTOFF = 3e-5
F_RTOL_AT_CUTOFF = 1e-5


def _check(system, cutoff, label, e_rtol=E_RTOL, e_atol=E_ATOL, f_rtol=F_RTOL):
    from oracle.oracle import Oracle
    s = system.with_ff(NB_CUTOFF=cutoff)
    et_ref, F_ref = Oracle(s).eval()
    scale_e = np.abs(et_ref).sum()
    fmax = np.abs(F_ref).max()
    # both pair kernels of the cell-list path, each forced (by default the engine picks per state, see use_n3 in
    # mmx_engine.hpp): the half-shell kernel (Newton's third law, atomics; nb_variant bit 4096) and the full-shell kernel
    # with a fixed summation order (what option deterministic selects); without a cutoff there is one all-pairs kernel
    for kernel, variant, det in ((("half-shell", 4096, 0), ("full-shell", 0, 1)) if cutoff > 0 else (("all-pairs", 0, 0),)):
        with engine_for(s) as eng:
            eng.set_option("deterministic", det)
            eng.set_option("nb_variant", variant)
            et, F = eng.compute()
        for t in range(len(TERM_NAMES)):
            assert abs(et[t] - et_ref[t]) <= e_rtol * scale_e + e_atol, (
                f"{label} {kernel}: term {TERM_NAMES[t]} gpu={et[t]!r} ref={et_ref[t]!r}")
        ferr = np.abs(F.astype(np.float64) - F_ref).max()
        l2 = np.sqrt(((F - F_ref) ** 2).sum() / max((F_ref ** 2).sum(), 1e-300))
        print(f"ACC {label} {kernel}: max err / max|F| = {ferr / max(fmax, 1e-300):.2e} (abs {ferr:.2e}), rel L2 = {l2:.2e}")
        assert ferr <= f_rtol * fmax + F_ATOL, f"{label} {kernel}: force err {ferr} (max |F| {fmax})"
    return et, F


ALL_ON = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
              IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)


@pytest.mark.parametrize("n", [64, 512, 4096])
@pytest.mark.parametrize("cutoff", [0.0, 0.6])
def test_all_terms_lattice(n, cutoff):
    """Hilbert lattice start (every angle exactly pi or pi/2, every bond exactly r0)."""
    _check(synthetic_system("gw_200k", n_beads=n, **ALL_ON), cutoff, f"lattice n={n} rc={cutoff}",
           e_rtol=E_RTOL_AT_CUTOFF if cutoff > 0 else E_RTOL, f_rtol=F_RTOL_AT_CUTOFF if cutoff > 0 else F_RTOL)


@pytest.mark.parametrize("n", [64, 512, 4096, 20000])
@pytest.mark.parametrize("cutoff", [0.0, 0.6])
def test_all_terms_jittered(n, cutoff):
    if n == 20000 and cutoff == 0.0:
        pytest.skip("all-pairs oracle at 20k is minutes of CPU")
    _check(synthetic_system("gw_200k", n_beads=n, jitter=0.03, seed=3, **ALL_ON), cutoff, f"jitter n={n} rc={cutoff}")


@pytest.mark.parametrize("long_items", [0, 1])
def test_half_shell_work_item_lengths(long_items):
    """The half-shell kernel's work items come in two lengths (runs of 16 clusters below 150 000 beads, of 24 -- dense
    cells: 30 -- above; option n3_long_items forces either).  Both against the fp64 oracle where the other one is the
    default, over the densities a run passes through: the lattice (216 beads per cell: dense cells cut by themselves,
    windows in two passes), a jittered lattice, the state after 40 iterations, and a dilute gas (runs that span many cells)."""
    from oracle.oracle import Oracle
    cases = [("lattice", synthetic_system("gw_200k", n_beads=30000, **ALL_ON), E_RTOL_AT_CUTOFF, F_RTOL_AT_CUTOFF),
             ("jitter", synthetic_system("gw_200k", n_beads=30000, jitter=0.03, seed=11, **ALL_ON), E_RTOL, F_RTOL)]
    s40 = synthetic_system("gw_200k", n_beads=30000, **ALL_ON)
    with engine_for(s40) as eng:
        eng.minimize(tolerance=0.0, max_iters=40)
        x = eng.get_positions()
    cases.append(("after 40 iterations", dataclasses.replace(s40, positions=x.astype(np.float64)), E_RTOL, F_RTOL))
    gas = synthetic_system("gw_200k", n_beads=30000, **ALL_ON)
    rng = np.random.default_rng(5)
    cases.append(("gas", dataclasses.replace(gas, positions=rng.uniform(-9.0, 9.0, size=(30000, 3))), E_RTOL, F_RTOL))
    for label, s, e_rtol, f_rtol in cases:
        et_ref, F_ref = Oracle(s).eval()
        with engine_for(s) as eng:
            eng.set_option("nb_variant", 4096)
            eng.set_option("n3_long_items", long_items)
            et, F = eng.compute()
            assert eng.get_option("n3_launches") > 0 and eng.get_option("n3_items") > 0
        assert np.all(np.abs(et - et_ref) <= e_rtol * np.abs(et_ref).sum() + E_ATOL), (label, et, et_ref)
        ferr = np.abs(F - F_ref).max()
        print(f"ACC items {'long' if long_items else 'short'}, {label}: max err / max|F| = {ferr / np.abs(F_ref).max():.2e}")
        assert ferr <= f_rtol * np.abs(F_ref).max() + F_ATOL, (label, ferr)


@pytest.mark.parametrize("iters", [0, 30])
def test_half_shell_kernel_repeats_itself(iters):
    """The half-shell kernel's unit pipeline is a protocol between 16 waves (windows handed over, flushed and restaged
    without a workgroup barrier): a rare race would show as an occasional wrong force, not as a wrong one every time.  Forty
    evaluations of one state with each work-item length against the full-shell kernel's forces (scripts/n3_repeat_check.py
    is the long version; it caught a work-stealing experiment that every single-shot parity test had passed)."""
    s = synthetic_system("gw_200k", n_beads=30000, **ALL_ON)
    with engine_for(s) as eng:
        if iters:
            eng.minimize(tolerance=0.0, max_iters=iters)
        eng.set_option("nb_variant", 8192)
        _, F0 = eng.compute()
        fmax = np.abs(F0).max()
        eng.set_option("nb_variant", 4096)
        for long_items in (0, 1):
            eng.set_option("n3_long_items", long_items)
            worst = max(np.abs(eng.compute()[1] - F0).max() for _ in range(40))
            assert worst <= 2e-5 * fmax, (long_items, worst / fmax)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7])
def test_tiny_systems(n):
    """Edge of the input range: a chain of one to seven beads, no loops (the reference builds its forces from whatever the
    structure holds: `model.py:625-720` loops over empty lists).  Forces against the oracle with and without cutoff, and a
    minimization that ends with finite positions.  (Energies to 2e-2 kJ/mol: the angle term near theta = pi is where fp32
    `acos` loses digits -- the reason the kernel clamps its argument --, and three beads have nothing to average it out.)"""
    from oracle.oracle import Oracle
    base = synthetic_system("region_5k", n_beads=8, start="circle")
    for cutoff in (0.0, 0.6):
        s = dataclasses.replace(base, n_beads=n, positions=base.positions[:n].copy(), chr_ends=np.array([0, n], np.int32),
                                labels=base.labels[:n].copy(), loop_m=np.zeros(0, np.int32), loop_n=np.zeros(0, np.int32),
                                loop_r0=np.zeros(0)).with_ff(NB_CUTOFF=cutoff)
        et_ref, F_ref = Oracle(s).eval()
        with engine_for(s) as eng:
            et, F = eng.compute()
            st = eng.minimize(tolerance=10.0, max_iters=50)
            x = eng.get_positions()
        assert np.abs(et - et_ref).max() <= 2e-2 + 1e-5 * np.abs(et_ref).sum(), (n, cutoff, et, et_ref)
        assert np.abs(F - F_ref).max() <= F_RTOL * max(np.abs(F_ref).max(), 1.0) + F_ATOL, (n, cutoff)
        assert st.status in (0, 1) and np.isfinite(x).all() and x.shape == (n, 3)


def test_region_preset_circle_start():
    """BASELINE config 1: EV + bonds + angles + loops, circle start (config_specific_region.ini)."""
    for n in (500, 5000):
        _check(synthetic_system("region_5k", n_beads=n, start="circle"), 0.0, f"circle n={n}")
        _check(synthetic_system("region_5k", n_beads=n, start="circle"), 0.6, f"circle n={n}")


def test_single_terms():
    base = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False,
                EV_USE_EXCLUDED_VOLUME=False)
    for on in ("POL_USE_HARMONIC_BOND", "POL_USE_HARMONIC_ANGLE", "LE_USE_HARMONIC_BOND", "EV_USE_EXCLUDED_VOLUME",
               "SC_USE_SPHERICAL_CONTAINER", "COB_USE_COMPARTMENT_BLOCKS", "SCB_USE_SUBCOMPARTMENT_BLOCKS",
               "IBL_USE_B_LAMINA_INTERACTION", "CF_USE_CENTRAL_FORCE"):
        kw = dict(base)
        kw[on] = True
        s = synthetic_system("chr1_50k", n_beads=3000, jitter=0.02, **kw)
        if on == "CF_USE_CENTRAL_FORCE":
            from multimm_amd.system import chrom_strength_per_bead, gw_chr_ends
            s.chrom_strength = chrom_strength_per_bead(gw_chr_ends(3000), 3000)
        et, _ = _check(s, 0.6, on)
        assert np.count_nonzero(et) == 1, (on, et)


def test_chromosomal_blocks_all_pairs_per_chromosome():
    """CHB polynomial term (model.py:416-419, enabled by examples/config_gw.ini): exact all-pairs inside every
    chromosome; strong dE so that the term is well above fp32 noise, plus the shipped force set of config_gw.ini."""
    for n in (3000, 20000):
        s = synthetic_system("gw_200k", n_beads=n, jitter=0.03, seed=7, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_DE=0.05)
        et, _ = _check(s, 0.6, f"CHB n={n}")
        assert et[8] != 0.0
    shipped = dict(SC_USE_SPHERICAL_CONTAINER=True, CHB_USE_CHROMOSOMAL_BLOCKS=True, COB_USE_COMPARTMENT_BLOCKS=False,
                   SCB_USE_SUBCOMPARTMENT_BLOCKS=True, IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)
    _check(synthetic_system("gw_200k", n_beads=8000, jitter=0.03, **shipped), 0.6, "config_gw.ini force set")


def test_half_shell_kernel_dense_cells_and_overlapping_beads():
    """The half-shell pair kernel's corner cases.  (1) A collapsed globule: thousands of beads in a few cells, so that an
    item's candidate set is far larger than the LDS window and is swept in several passes.  (2) Beads almost on top of
    each other: batch sums beyond the fixed-point range of the LDS accumulation take the global-atomic bypass.
    (3) A sparse gas: most cells hold one bead, items span many cells of a row."""
    from multimm_amd.system import ChromatinSystem, ForceFieldParams
    rng = np.random.default_rng(11)
    ff = ForceFieldParams(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False,
                          COB_USE_COMPARTMENT_BLOCKS=True, NB_CUTOFF=0.6)
    n = 6000
    labels = rng.choice(np.array([-2, -1, 0, 1, 2], np.int8), n)
    blob = rng.normal(0.0, 0.35, (n, 3))                       # ~ 6000 beads within a 1 nm ball
    _check(ChromatinSystem(n, blob, np.array([0, n]), labels, ff=ff), 0.6, "collapsed globule")
    close = rng.uniform(0.0, 1.0, (2000, 3))
    u = rng.normal(size=(1000, 3))
    close[1::2] = close[0::2] + 0.02 * u / np.linalg.norm(u, axis=1)[:, None]  # pairs 0.02 nm apart: 7e4 kJ/mol/nm each
    _check(ChromatinSystem(2000, close, np.array([0, 2000]), labels[:2000], ff=ff), 0.6, "overlapping beads")
    # (no two beads closer than 0.3 nm: next to a contact the pair energy changes by 1e4 kJ/mol per nm, and an fp32
    # coordinate of 12 nm carries a rounding of 5e-7 nm of its own)
    g = np.stack(np.meshgrid(*[np.arange(15)] * 3, indexing="ij"), -1).reshape(-1, 3)[rng.permutation(3375)[:3000]]
    gas = 0.8 * g + rng.uniform(-0.25, 0.25, (3000, 3))
    # The whole system has 3 kJ/mol of pair energy.  The cell-list kernels sweep every bead's r = 0 self pair with the
    # rest (6400 kJ/mol each) and take it out again in fp32: what stays behind is bounded by N * 6400 * 2^-24 = 1.1
    # kJ/mol and measured at 2e-3 (half-shell) / 5e-2 (full-shell) -- invisible next to the 1e7 kJ/mol of a chromatin
    # system, but above the 1e-3 floor of the other tests.
    _check(ChromatinSystem(3000, gas, np.array([0, 3000]), labels[:3000], ff=ff), 0.6, "sparse gas", e_atol=0.1)


def test_half_shell_window_passes_are_records_of_their_own():
    """A run of i-clusters whose candidates exceed the LDS window (424 clusters) is emitted as one record of the item list per
    window pass, slices of equal length (N3Item::w0; mmx_nonbonded_n3.hpp), instead of one work item whose passes a single
    workgroup walks: more records, the same forces.  A collapsed globule (every run needs several passes) and the lattice start
    (two passes), against the full-shell kernel and against the one-record-per-run layout (option n3_pass_records = 0)."""
    from multimm_amd.system import ChromatinSystem, ForceFieldParams
    rng = np.random.default_rng(12)
    ff = ForceFieldParams(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False,
                          COB_USE_COMPARTMENT_BLOCKS=True, NB_CUTOFF=0.6)
    n = 8000
    labels = rng.choice(np.array([-2, -1, 0, 1, 2], np.int8), n)
    blob = ChromatinSystem(n, rng.normal(0.0, 0.4, (n, 3)), np.array([0, n]), labels, ff=ff)
    for label, s in (("globule", blob), ("lattice", synthetic_system("gw_200k", n_beads=30000, **ALL_ON))):
        with engine_for(s) as eng:
            eng.set_option("nb_variant", 8192)
            e0, F0 = eng.compute()
            fmax = np.abs(F0).max()
            eng.set_option("nb_variant", 4096)
            out = {}
            for rec in (0, 1):
                eng.set_option("n3_pass_records", rec)
                e, F = eng.compute()
                out[rec] = (int(eng.get_option("n3_items")), e, F)
                assert np.abs(F - F0).max() <= 2e-5 * fmax, (label, rec, np.abs(F - F0).max() / fmax)
                assert np.all(np.abs(e - e0) <= 2e-6 * np.abs(e0).sum() + 1e-2), (label, rec, e, e0)
            assert out[1][0] > out[0][0], (label, out[0][0], out[1][0])      # the passes became records
            assert np.abs(out[1][2] - out[0][2]).max() <= 2e-5 * fmax


def test_pair_kernel_choice_follows_size_and_cell_occupancy():
    """Default options (use_n3 in mmx_engine.hpp): systems below 55 000 beads (70 000 without the compartment Gaussians) always take
    the full-shell kernel, larger ones the half-shell kernel as long as the last poll saw >= 20 beads per grid cell.  Whatever is picked, a minimization ends
    where a run pinned to either kernel ends."""
    with engine_for(synthetic_system("gw_200k", n_beads=30000, **ALL_ON)) as eng:
        eng.minimize(tolerance=0.0, max_iters=20)
        assert eng.get_option("n3_launches") == 0
    s = synthetic_system("gw_200k", n_beads=100000, **ALL_ON)
    ends = {}
    for name, variant in (("auto", 0), ("half-shell", 4096), ("full-shell", 8192)):
        with engine_for(s) as eng:
            eng.set_option("nb_variant", variant)
            st = eng.minimize(tolerance=0.0, max_iters=100)
            ends[name] = (st.e_initial, st.e_final, eng.get_option("n3_launches"), st.evaluations)
    e0, ef, n3, evals = ends["auto"]
    assert n3 >= evals                                           # crowded cells throughout: half shell
    assert ends["half-shell"][2] >= ends["half-shell"][3] and ends["full-shell"][2] == 0
    for name in ("half-shell", "full-shell"):
        assert abs(ends[name][0] - e0) <= 2e-6 * abs(e0)
        assert abs(ends[name][1] - ef) <= 2e-2 * abs(e0 - ef)   # 100 iterations from the lattice: chaotic, see DESIGN.md 9
    # the same beads spread over 27 times the volume (~5 per cell): after the first poll the full-shell kernel takes over
    import dataclasses
    centre = s.positions.mean(axis=0)
    thin = dataclasses.replace(s, positions=centre + 3.0 * (s.positions - centre))
    with engine_for(thin.with_ff(SC_USE_SPHERICAL_CONTAINER=False, IBL_USE_B_LAMINA_INTERACTION=False)) as eng:
        eng.minimize(tolerance=0.0, max_iters=3)
        n3 = eng.get_option("n3_launches")
        st = eng.minimize(tolerance=0.0, max_iters=20)
        assert st.evaluations >= 20 and eng.get_option("n3_launches") == n3


def test_graph_replay_equals_direct_launches_bitwise():
    """Option use_graph: the trial evaluations of the minimizer replayed from a hipGraph (pairs of evaluations, or 8 at
    a time) instead of launch by launch -- same kernels, same arguments, same order: the same bits."""
    s = synthetic_system("gw_200k", n_beads=20000, jitter=0.03, seed=4, **ALL_ON)
    res = []
    for use_graph, per_graph in ((0, 2), (1, 2), (1, 8)):
        with engine_for(s) as eng:
            eng.set_option("deterministic", 1)
            eng.set_option("use_graph", use_graph)
            eng.set_option("graph_evals", per_graph)
            st = eng.minimize(tolerance=0.0, max_iters=70)
            res.append((st.iterations, st.evaluations, st.e_final, eng.get_positions()))
    for r in res[1:]:
        assert r[:3] == res[0][:3] and np.array_equal(r[3], res[0][3])


def test_generic_ev_power():
    for p in (3.0, 4.5):
        _check(synthetic_system("chr1_50k", n_beads=2000, jitter=0.02, EV_POWER=p), 0.6, f"EV_POWER={p}")
        _check(synthetic_system("chr1_50k", n_beads=2000, jitter=0.02, EV_POWER=p), 0.0, f"EV_POWER={p}")


def test_known_answers_two_and_three_beads():
    """Hand-derivable values."""
    from multimm_amd.system import ChromatinSystem, ForceFieldParams
    ff = ForceFieldParams(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False,
                          NB_CUTOFF=0.0)
    s = ChromatinSystem(2, np.array([[0, 0, 0], [0.1, 0, 0.0]]), np.array([0, 2]), np.zeros(2, np.int8), ff=ff)
    with engine_for(s) as eng:
        et, F = eng.compute()
    assert et[0] == pytest.approx(100 * (0.1 / 0.15) ** 6, rel=1e-5)
    assert abs(F[0, 0]) == pytest.approx(6 * 100 * (0.1 / 0.15) ** 6 / 0.15, rel=1e-5)
    assert F[0, 0] < 0 < F[1, 0]


def test_deterministic_bitwise():
    s = synthetic_system("gw_200k", n_beads=30000, jitter=0.03, **ALL_ON)
    outs = []
    for _ in range(2):
        with engine_for(s) as eng:
            eng.set_option("deterministic", 1)   # the full-shell pair kernel: fixed summation order
            et, F = eng.compute()
            outs.append((et.copy(), F.copy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    # a whole minimization is reproducible too, and no cell ever outgrew the in-LDS sort (the guarantee's condition)
    runs = []
    for _ in range(2):
        with engine_for(s) as eng:
            eng.set_option("deterministic", 1)
            st = eng.minimize(tolerance=0.0, max_iters=80)
            assert eng.get_option("order_fallbacks") == 0
            runs.append((st.e_final, st.evaluations, eng.get_positions()))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and np.array_equal(runs[0][2], runs[1][2])


def test_minimize_small_matches_oracle_quality():
    """Trajectories are not expected to match step for step (fp32 device vs fp64 host L-BFGS); the end
    state must satisfy the same stop rule and reach a comparable energy."""
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=2000, jitter=0.01, **ALL_ON)
    xr, st_ref = Oracle(s).minimize(tolerance=10.0, max_iters=0)
    with engine_for(s) as eng:
        st = eng.minimize(tolerance=10.0, max_iters=0)
        x = eng.get_positions()
        et, F = eng.compute()
    assert st.status in (0, -5, -3), st.status
    assert st.rms_force < 20.0
    assert st.e_final < st.e_initial
    # energy reported by the minimizer is the energy of the returned positions
    assert et.sum() == pytest.approx(st.e_final, rel=1e-5, abs=1e-2)
    e_ref_at_gpu_x = Oracle(s).energy(x)
    assert e_ref_at_gpu_x == pytest.approx(st.e_final, rel=2e-5, abs=1e-2)
    # comparable minimum: within 2 % of the span the oracle covered
    span = st_ref.e_initial - st_ref.e_final
    assert st.e_final - st_ref.e_final < 0.02 * span


def test_minimize_fixed_iterations_and_errors():
    s = synthetic_system("chr1_50k", n_beads=5000)
    with engine_for(s) as eng:
        st = eng.minimize(tolerance=0.0, max_iters=25)
        assert st.iterations == 25 and st.status == 1 and st.evaluations >= 26
        assert st.e_final < st.e_initial
    from multimm_amd.engine import MMXError
    with Engine(10) as eng:
        with pytest.raises(MMXError):
            eng.compute()  # positions not set
        with pytest.raises(MMXError):
            eng.set_loops([0], [0], [0.1], 1.0)  # degenerate loop
        with pytest.raises(MMXError):
            eng.set_labels(np.full(10, 3, np.int8))


def test_domain_decomposition_owned_ranges():
    """Multi-GPU ownership logic on one GPU: `world` handles in one process, each evaluating its own bead
    slice against the whole system's positions (no communicator needed for a single evaluation).  Forces
    concatenate to the single-domain result and the per-rank energy shares add up to the total."""
    from multimm_amd.parallel import slice_of
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", n_beads=20000, jitter=0.03, seed=5, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_DE=0.05,
                         **ALL_ON)
    et_ref, F_ref = Oracle(s).eval()
    for world in (2, 3, 8):
        et_sum = np.zeros(len(TERM_NAMES))
        parts = []
        for r in range(world):
            with engine_for(s, rank=r, world=world) as eng:
                assert (eng.own_lo, eng.own_lo + eng.n_own) == slice_of(s.n_beads, r, world)   # whole 62-bead segments
                et, f = eng.compute()
                assert f.shape == (eng.n_own, 3)
                et_sum += et
                parts.append(f)
        F = np.concatenate(parts)
        assert F.shape == F_ref.shape
        scale_e = np.abs(et_ref).sum()
        assert np.all(np.abs(et_sum - et_ref) <= E_RTOL * scale_e + E_ATOL), (world, et_sum, et_ref)
        ferr = np.abs(F.astype(np.float64) - F_ref).max()
        assert ferr <= F_RTOL * np.abs(F_ref).max() + F_ATOL, (world, ferr)


def test_rccl_path_single_rank():
    """The collective plumbing (in-place ncclAllGather of pos4, fp64 ncclAllReduce of the slot sums and Gram
    rows, split controller kernels) with a one-rank communicator: same arithmetic, same result."""
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=2, **ALL_ON)
    with engine_for(s) as eng:
        eng.set_option("deterministic", 1)
        st0 = eng.minimize(tolerance=0.0, max_iters=30)
        x0 = eng.get_positions()
    with engine_for(s) as eng:
        eng.set_option("deterministic", 1)
        eng.comm_init(Engine.comm_unique_id())
        et, F = eng.compute()
        st1 = eng.minimize(tolerance=0.0, max_iters=30)
        x1 = eng.get_positions()
    assert st1.iterations == st0.iterations == 30 and st1.evaluations == st0.evaluations
    assert st1.e_final == st0.e_final
    assert np.array_equal(x0, x1)


def test_full_size_properties_200k():
    """BASELINE-size checks that do not need the O(minutes) oracle: three independent pair kernels agree (half shell -- the
    default at this size --, full shell, round 1's cell kernel), internal forces sum to zero (Newton's third law), energy is
    translation and rotation invariant, the minimizer decreases the energy monotonically in accepted iterations."""
    s = synthetic_system("gw_200k", jitter=0.02, seed=1)
    internal = s.with_ff(SC_USE_SPHERICAL_CONTAINER=False, IBL_USE_B_LAMINA_INTERACTION=False)
    with engine_for(internal) as eng:
        et0, F0 = eng.compute()
        assert eng.get_option("n3_launches") >= 1  # the half-shell kernel ran
        fmax = np.abs(F0).max()
        for variant in (8192, 1):                # full-shell kernel; v1 cell kernel: different code, same physics
            eng.set_option("nb_variant", variant)
            et1, F1 = eng.compute()
            assert np.abs(F0 - F1).max() <= 2 * F_RTOL * fmax + F_ATOL, variant
            assert np.all(np.abs(et0 - et1) <= 2 * E_RTOL * np.abs(et0).sum() + E_ATOL), variant
        eng.set_option("nb_variant", 0)
        # sum of internal forces vanishes; fp32 accumulation noise ~ sqrt(N) * eps * |F|
        assert np.abs(F0.astype(np.float64).sum(0)).max() <= 1e-3 * fmax
        shifted = internal.positions + np.array([3.0, -2.0, 1.0])
        eng.set_positions(shifted)
        et2, F2 = eng.compute()
        assert np.all(np.abs(et2 - et0) <= 5e-5 * np.abs(et0).sum() + 1e-2)
        # rigid rotation: energies unchanged, forces co-rotate (different cells, clusters and summation order)
        q, _ = np.linalg.qr(np.random.default_rng(3).normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        eng.set_positions(internal.positions @ q.T)
        et3, F3 = eng.compute()
        assert np.all(np.abs(et3 - et0) <= 5e-5 * np.abs(et0).sum() + 1e-2)
        assert np.abs(F3 - F0 @ q.T.astype(np.float32)).max() <= 5e-4 * fmax + 5e-2
    with engine_for(s) as eng:
        e_prev = None
        for _ in range(4):
            st = eng.minimize(tolerance=0.0, max_iters=25)
            assert st.iterations == 25 and np.isfinite(st.e_final)
            assert st.e_final < st.e_initial
            if e_prev is not None:
                assert st.e_initial == pytest.approx(e_prev, rel=1e-6)   # restart resumes from the same point
            e_prev = st.e_final


def test_full_size_oracle_parity_200k():
    """BASELINE config 3 size against the fp64 oracle itself (its OpenMP evaluation of 200 000 beads takes a second): every
    term on, both cell-list pair kernels, at the lattice start (at-cutoff bands, see E_RTOL_AT_CUTOFF) and on the state the
    default path reaches after 60 iterations -- the collapse phase, where the half-shell kernel's work items take two window
    passes and its tail shares are in play."""
    import dataclasses
    from oracle.oracle import Oracle
    s = synthetic_system("gw_200k", **ALL_ON)
    x = s.positions
    for relaxed in (False, True):
        if relaxed:
            with engine_for(s) as eng:
                eng.minimize(tolerance=0.0, max_iters=60)
                x = eng.get_positions().astype(np.float64)
        s2 = dataclasses.replace(s, positions=x)   # (the mass centre of the confinement terms belongs to the system)
        et_ref, F_ref = Oracle(s2).eval()
        scale_e, fmax = np.abs(et_ref).sum(), np.abs(F_ref).max()
        with engine_for(s2) as eng:
            for variant in (4096, 8192):
                eng.set_option("nb_variant", variant)
                et, F = eng.compute()
                e_rtol, f_rtol = (E_RTOL, F_RTOL) if relaxed else (E_RTOL_AT_CUTOFF, F_RTOL_AT_CUTOFF)
                assert np.all(np.abs(et - et_ref) <= e_rtol * scale_e + E_ATOL), (relaxed, variant, et, et_ref)
                assert np.abs(F - F_ref).max() <= f_rtol * fmax + F_ATOL, (relaxed, variant)


def test_full_size_oracle_parity_1m():
    """BASELINE config 5 size against the fp64 oracle itself (1 000 000 beads: a few seconds of its OpenMP evaluation on the
    box's host cores): the state the default path reaches after 40 iterations, both cell-list pair kernels, every term on."""
    import dataclasses
    from oracle.oracle import Oracle
    s = synthetic_system("gw_1m", **ALL_ON)
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=40)
        x = eng.get_positions().astype(np.float64)
    s2 = dataclasses.replace(s, positions=x)
    et_ref, F_ref = Oracle(s2).eval()
    scale_e, fmax = np.abs(et_ref).sum(), np.abs(F_ref).max()
    with engine_for(s2) as eng:
        for variant in (4096, 8192):
            eng.set_option("nb_variant", variant)
            et, F = eng.compute()
            print(f"ACC gw_1m variant {variant}: max err / max|F| = {np.abs(F - F_ref).max() / fmax:.2e}")
            assert np.all(np.abs(et - et_ref) <= E_RTOL * scale_e + E_ATOL), (variant, et, et_ref)
            assert np.abs(F - F_ref).max() <= F_RTOL * fmax + F_ATOL, variant


def test_one_million_beads_runs():
    """BASELINE config 5 size on one GPU: allocation, cell build and kernels at N = 1e6."""
    s = synthetic_system("gw_1m")
    with engine_for(s) as eng:
        et, F = eng.compute()
        assert np.all(np.isfinite(et)) and np.all(np.isfinite(F))
        # Hilbert lattice: every bond is exactly r0, so the bond energy is ~0 (fp32 lattice round-off)
        assert abs(et[2]) < 1e-2 * s.n_beads * 1e-3
        # the two pair kernels (half shell is the default here) agree; at the lattice thousands of pairs sit exactly at the
        # cutoff and either kernel may count one or not (see E_RTOL_AT_CUTOFF): the bands of test_all_terms_lattice, twice
        eng.set_option("nb_variant", 8192)
        et1, F1 = eng.compute()
        eng.set_option("nb_variant", 0)
        assert np.abs(F - F1).max() <= 2 * F_RTOL_AT_CUTOFF * np.abs(F).max() + F_ATOL
        assert np.all(np.abs(et - et1) <= 2 * E_RTOL_AT_CUTOFF * np.abs(et).sum() + E_ATOL)
        st = eng.minimize(tolerance=0.0, max_iters=10)
        assert st.iterations == 10 and st.e_final < st.e_initial


def test_multimm_run_end_to_end(tmp_path):
    """The reference's own smoke-test shape (tests/test_simulations.py: run, then assert the output files) on
    PLATFORM = MI355X, plus what the reference never checks: the written structure is the minimized one."""
    import os
    from multimm_amd import cif
    from multimm_amd.config import load_config
    from multimm_amd.model import MultiMM
    ini = tmp_path / "config.ini"
    ini.write_text("[Main]\nPLATFORM = MI355X\nN_BEADS = 3000\nOUT_PATH = %s\nSC_USE_SPHERICAL_CONTAINER = True\n"
                   "COB_USE_COMPARTMENT_BLOCKS = True\nIBL_USE_B_LAMINA_INTERACTION = True\nMIN_MAX_ITERATIONS = 300\n"
                   % (tmp_path / "out"))
    m = MultiMM(load_config(str(ini)))
    st = m.run()
    out = tmp_path / "out"
    assert os.path.exists(out / "metadata" / "MultiMM_init.cif")
    assert os.path.exists(out / "model" / "MultiMM_minimized.cif")
    assert os.path.exists(out / "model" / "chromosomes" / "MultiMM_minimized_chr1.cif")
    assert st.iterations > 0 and st.e_final < st.e_initial
    x = cif.read_positions(str(out / "model" / "MultiMM_minimized.cif"))
    assert x.shape == (3000, 3) and np.abs(x - m.state_positions).max() <= 5.1e-5
    bonds = np.linalg.norm(np.diff(x, axis=0), axis=1)[1:]
    assert 0.05 < np.median(bonds) < 0.2


@pytest.mark.parametrize("kind", ["rw", "confined_rw", "knot", "helix", "spiral", "sphere", "circle", "self_avoiding_rw"])
def test_every_initial_structure_type_minimizes(tmp_path, kind):
    """INITIAL_STRUCTURE_TYPE (config.py:138-141, model.py:745): every start the reference can build goes through the
    mmCIF round trip and minimizes on the device; the energy reported is the oracle's at the returned structure."""
    from oracle.oracle import Oracle
    from multimm_amd.config import load_config
    from multimm_amd.model import MultiMM
    n = 300 if kind == "self_avoiding_rw" else 1200
    m = MultiMM(load_config(dict(PLATFORM="MI355X", N_BEADS=n, CHROM="chr1", OUT_PATH=str(tmp_path / kind),
                                 INITIAL_STRUCTURE_TYPE=kind, MIN_MAX_ITERATIONS=150, SHUFFLING_SEED=5)))
    st = m.run()
    m.engine.close()
    assert st.iterations > 0 and st.e_final < st.e_initial and np.all(np.isfinite(m.state_positions))
    e_ref = Oracle(m.system).energy(m.state_positions)
    assert abs(e_ref - st.e_final) <= 2e-5 * abs(e_ref) + 1e-2


def test_config2_converges_to_openmm_tolerance():
    """BASELINE config 2 (chr1, 50k beads, EV + backbone + loops) minimized until OpenMM's default stop rule holds;
    the oracle confirms energy and stop rule at the returned point."""
    from oracle.oracle import Oracle
    s = synthetic_system("chr1_50k")
    with engine_for(s) as eng:
        st = eng.minimize(tolerance=10.0, max_iters=0)
        x = eng.get_positions().astype(np.float64)
    assert st.status == 0 and 200 < st.iterations < 20000
    assert st.e_final < 0.2 * st.e_initial
    et, F = Oracle(s).eval(x)
    assert et.sum() == pytest.approx(st.e_final, rel=2e-5)
    x0 = s.positions
    eps = 10.0 / max(1.0, np.sqrt((x0 * x0).sum() / len(x0)))
    assert np.linalg.norm(F) / max(1.0, np.linalg.norm(x)) <= eps * 1.02
    bonds = np.linalg.norm(np.diff(x, axis=0), axis=1)[1:]
    assert abs(np.median(bonds) - 0.1) < 0.01


def test_edge_cases_tiny_coincident_and_crowded():
    """Edge cases: 1-3 beads, no loops, coincident beads (r = 0: energy counted, zero force), every bead in ONE
    cell (5000 beads: beyond the in-LDS sort, 625 clusters in one cell), beads far apart (one bead per cell)."""
    from multimm_amd.system import ChromatinSystem, ForceFieldParams
    from oracle.oracle import Oracle
    ff = ForceFieldParams(LE_USE_HARMONIC_BOND=False, NB_CUTOFF=0.6)
    for n in (1, 2, 3, 7, 9):
        pos = np.arange(3 * n, dtype=np.float64).reshape(n, 3) * 0.037
        s = ChromatinSystem(n, pos, np.array([0, n]), np.zeros(n, np.int8), ff=ff)
        _check(s, 0.6, f"tiny n={n}")
        _check(s, 0.0, f"tiny n={n} all-pairs")
    # two beads at the same point + a third one: E_ev(0) = 100*(0.1/0.05)^6 = 6400 for the coincident pair
    pos = np.array([[0.2, 0.2, 0.2], [0.2, 0.2, 0.2], [0.5, 0.2, 0.2]])
    s = ChromatinSystem(3, pos, np.array([0, 3]), np.zeros(3, np.int8),
                        ff=ForceFieldParams(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False,
                                            LE_USE_HARMONIC_BOND=False, NB_CUTOFF=0.6))
    for rc in (0.6, 0.0):
        et, F = _check(s, rc, f"coincident rc={rc}")
        assert et[0] > 6400.0 and np.all(np.isfinite(F))
    rng = np.random.default_rng(0)
    n = 5000
    crowded = ChromatinSystem(n, rng.random((n, 3)) * 0.4, np.array([0, n]), rng.integers(-2, 3, n).astype(np.int8),
                              ff=ForceFieldParams(LE_USE_HARMONIC_BOND=False, COB_USE_COMPARTMENT_BLOCKS=True, NB_CUTOFF=0.6))
    _check(crowded, 0.6, "crowded: one cell")
    with engine_for(crowded) as eng:   # 5000 beads in one cell: beyond the in-LDS sort, reported as such
        eng.compute()
        assert eng.get_option("order_fallbacks") >= 1
    sparse = ChromatinSystem(300, rng.random((300, 3)) * 40.0, np.array([0, 300]), np.zeros(300, np.int8),
                             ff=ForceFieldParams(LE_USE_HARMONIC_BOND=False, NB_CUTOFF=0.6))
    _check(sparse, 0.6, "sparse: one bead per cell")


FORM_SETS = [
    dict(EV_FORCE_TYPE="gaussian_core", COB_FORCE_TYPE="yukawa", SCB_FORCE_TYPE="yukawa", CHB_FORCE_TYPE="gaussian",
         BLAMINA_FORCE_TYPE="gaussian_shell", CENTRAL_FORCE_TYPE="gaussian", LE_LOOP_FORCE_TYPE="fene_soft"),
    dict(COB_FORCE_TYPE="theta", SCB_FORCE_TYPE="yukawa", CHB_FORCE_TYPE="saturating",
         BLAMINA_FORCE_TYPE="harmonic_shell", CENTRAL_FORCE_TYPE="logistic", LE_LOOP_FORCE_TYPE="gaussian_tether"),
    dict(COB_FORCE_TYPE="gaussian", SCB_FORCE_TYPE="theta", BLAMINA_FORCE_TYPE="logistic_shell"),
    dict(EV_FORCE_TYPE="gaussian_core"),
    dict(COB_FORCE_TYPE="yukawa", SCB_USE_SUBCOMPARTMENT_BLOCKS=False),
]


@pytest.mark.parametrize("forms", FORM_SETS)
@pytest.mark.parametrize("n,cutoff", [(2048, 0.0), (2048, 0.6), (20000, 0.6)])
def test_alternative_functional_forms(forms, n, cutoff):
    """Every *_FORCE_TYPE alternative of config.py:269-312, all-pairs and cell-list kernels."""
    kw = dict(ALL_ON, CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_DE=0.5)
    kw.update(forms)
    et, _ = _check(synthetic_system("gw_200k", n_beads=n, jitter=0.03, seed=4, **kw), cutoff,
                   f"forms {forms} n={n} rc={cutoff}")
    assert et[0] != 0.0 and et[1] != 0.0


def test_alternative_forms_minimize():
    """Bounded forms minimize like the defaults.  (gaussian_core + yukawa is unbounded below -- -E/r against a
    finite core -- so that combination of the reference's options is only checked for parity above.)"""
    forms = dict(EV_FORCE_TYPE="gaussian_core", COB_FORCE_TYPE="theta", SCB_FORCE_TYPE="gaussian",
                 CHB_FORCE_TYPE="saturating", BLAMINA_FORCE_TYPE="gaussian_shell", CENTRAL_FORCE_TYPE="logistic",
                 LE_LOOP_FORCE_TYPE="gaussian_tether", CHB_USE_CHROMOSOMAL_BLOCKS=True)
    s = synthetic_system("gw_200k", n_beads=6000, **dict(ALL_ON, **forms))
    with engine_for(s) as eng:
        st = eng.minimize(tolerance=0.0, max_iters=150)
        assert st.iterations == 150 and np.isfinite(st.e_final) and st.e_final < st.e_initial
        x = eng.get_positions()
    from oracle.oracle import Oracle
    e_ref = Oracle(s).energy(x)
    assert abs(e_ref - st.e_final) <= 2e-5 * abs(e_ref) + 1e-2


def test_shipped_genome_wide_example_settings_run(tmp_path):
    """The force set of the reference's examples/config_gw.ini (500 000 beads; CHB + SCB + IBL + CF + SC on, COB off)
    -- what SURVEY 8 f1 was for -- through the MultiMM mirror: runs, energy decreases, every enabled term is live.
    The data files the example points to are not available; inputs are synthetic with the same tensor contracts."""
    from multimm_amd.config import load_config
    from multimm_amd.model import MultiMM
    ini = tmp_path / "gw.ini"
    ini.write_text("[Main]\nplatform = MI355X\ninitial_structure_type = hilbert\nout_path = %s\nn_beads = 500000\n"
                   "modelling_level = \nsc_use_spherical_container = True\nchb_use_chromosomal_blocks = True\n"
                   "cob_use_compartment_blocks = False\nscb_use_subcompartment_blocks = True\n"
                   "ibl_use_b_lamina_interaction = True\ncf_use_central_force = True\nsim_run_md = False\n"
                   "loc_start = \nloc_end = \nchrom = \nmin_max_iterations = 15\n" % (tmp_path / "out"))
    cfg = load_config(str(ini))
    assert cfg.N_BEADS == 500000 and cfg.ff.CHB_USE_CHROMOSOMAL_BLOCKS and not cfg.ff.COB_USE_COMPARTMENT_BLOCKS
    m = MultiMM(cfg)
    assert len(m.chr_ends) == 23 and m.chrom_strength.max() == 1.0   # genome-wide layout: CHB and CF see chromosomes
    st = m.run()
    assert st.iterations == 15 and st.e_final < st.e_initial
    et = np.array(st.energy_terms[:])
    for t in (0, 1, 2, 5, 6, 7, 8):  # ev, scb gaussians, bonds, container, lamina, central, chb
        assert et[t] != 0.0, TERM_NAMES[t]
    print(f"config_gw-like 500k: {st.iterations} iterations in {st.seconds:.2f} s")


def test_fused_bonded_kernel_equals_separate_kernels_bitwise():
    """Default = backbone + loops + confinement in one kernel; option fused_bonded = 0 runs the three kernels one
    after the other.  Same per-bead arithmetic in the same order: forces and per-term energies are bit-identical,
    and so is a whole minimization."""
    s = synthetic_system("gw_200k", n_beads=20000, jitter=0.03, seed=6, CHB_USE_CHROMOSOMAL_BLOCKS=False, **ALL_ON)
    res = []
    for fused in (1, 0):
        with engine_for(s) as eng:
            eng.set_option("fused_bonded", fused)
            eng.set_option("deterministic", 1)   # bit-for-bit comparison: the pair kernel with a fixed summation order
            et, F = eng.compute()
            st = eng.minimize(tolerance=0.0, max_iters=60)
            res.append((et.copy(), F.copy(), st.e_final, st.evaluations, eng.get_positions()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] and np.array_equal(res[0][4], res[1][4])


def test_ensemble_loop_like_the_reference(tmp_path):
    """run.py:471-485: replicas with SHUFFLING_SEED = i in <name>/run_<i>, archived as .tar.gz; different seeds give
    different inputs (loops / labels) and therefore different structures."""
    import os
    import tarfile
    from multimm_amd.ensemble import run_ensemble
    cfg = {"PLATFORM": "MI355X", "N_BEADS": 1500, "OUT_PATH": str(tmp_path / "ens"), "N_ENSEMBLE": 3,
           "MIN_MAX_ITERATIONS": 60, "DETERMINISTIC_FORCES": True}   # replicas are compared bit for bit below
    res = run_ensemble(cfg)
    assert [i for i, _, _ in res] == [0, 1, 2]
    for i, path, st in res:
        assert path.endswith(f"run_{i}.tar.gz") and os.path.getsize(path) > 0 and not os.path.exists(path[:-7])
        names = tarfile.open(path).getnames()
        assert f"run_{i}/model/MultiMM_minimized.cif" in names and f"run_{i}/metadata/parameters.txt" in names
        assert st.iterations == 60
    assert len({round(st.e_final, 3) for _, _, st in res}) == 3
    # rank 1 of 2 takes replica 1 only
    res1 = run_ensemble(dict(cfg, OUT_PATH=str(tmp_path / "ens2")), rank=1, world=2, archive=False, device=0)
    assert [i for i, _, _ in res1] == [1]
    # run.py:469-491: GENERATE_ENSEMBLE switches between the loop and a single run
    from multimm_amd.ensemble import run
    res4 = run(dict(cfg, OUT_PATH=str(tmp_path / "ens4"), GENERATE_ENSEMBLE=True, N_ENSEMBLE=2))
    assert [i for i, _, _ in res4] == [0, 1] and res4[1][2].e_final == res[1][2].e_final
    single = run(dict(cfg, OUT_PATH=str(tmp_path / "one")))
    assert single.e_final == res[0][2].e_final and os.path.exists(tmp_path / "one" / "model" / "MultiMM_minimized.cif")
    # three replicas in flight on the one GPU (own handle, stream and host thread each): same results, same order
    res3 = run_ensemble(dict(cfg, OUT_PATH=str(tmp_path / "ens3")), concurrent=3)
    assert [i for i, _, _ in res3] == [0, 1, 2]
    for (_, _, a), (_, _, b) in zip(res, res3):
        assert (a.iterations, a.evaluations, a.e_initial, a.e_final) == (b.iterations, b.evaluations, b.e_initial, b.e_final)


def test_kept_cell_structure_is_exact_and_falls_back_when_it_goes_stale():
    """Option cell_reuse (default on): once the structure has thinned out, a single-domain minimization keeps the cell
    structure of a full build -- membership, clusters, work items -- for several evaluations and only refreshes cluster
    positions and boxes; k_pack checks every evaluation that no bead has moved more than half the skin (cell edge -
    cutoff) from where it was binned.  Exactness: the energy the minimizer reports for its last point, found on a kept
    structure, is the energy a fresh build -- and the oracle -- give for that point.  A structure that does
    go stale (inject_fault bit 3: a skin of nothing) voids its evaluation, which is repeated after a full build: same
    iteration count, an end point of the same depth.  Runs with the same options repeat bit for bit."""
    from oracle.oracle import Oracle
    s = synthetic_system("chr1_50k", n_beads=30000)
    runs = {}
    for name, opts in (("rebuild", dict(cell_reuse=0)), ("keep", dict()), ("keep again", dict()), ("stale", dict(inject_fault=8))):
        with engine_for(s) as eng:
            eng.set_option("deterministic", 1)
            for k, v in opts.items():
                eng.set_option(k, v)
            st = eng.minimize(tolerance=0.0, max_iters=1500)
            stats = {k: eng.get_option(k) for k in ("cell_builds", "cell_reuses", "cell_stale_halts")}
            x = eng.get_positions()
            et, F = eng.compute()
            runs[name] = (st.iterations, st.status, st.e_initial, st.e_final, stats, x, et, F)
    it, status, e0, e1, stats, x, et, F = runs["keep"]
    assert runs["rebuild"][4]["cell_reuses"] == 0
    assert stats["cell_reuses"] > 300 and stats["cell_stale_halts"] <= 3, stats      # most late evaluations ran on a kept structure
    assert abs(et.sum() - e1) <= 2e-6 * abs(e1)                                      # ... and found every pair
    et_ref, _ = Oracle(s).eval(x)                                                    # ... as the oracle counts them
    assert np.all(np.abs(et - et_ref) <= E_RTOL * np.abs(et_ref).sum() + E_ATOL)
    assert runs["keep again"][:4] == runs["keep"][:4] and np.array_equal(runs["keep again"][5], x)     # reproducible
    for other in ("rebuild", "stale"):
        o_it, o_status, o_e0, o_e1 = runs[other][:4]
        assert (o_it, o_status) == (it, status) == (1500, 1) and o_e0 == e0
        assert abs(o_e1 - e1) <= 2e-2 * abs(e0 - e1), (other, o_e1, e1)          # other summation orders: chaotic drift only
    assert runs["stale"][4]["cell_stale_halts"] >= 5, runs["stale"][4]
    assert abs(runs["stale"][6].sum() - runs["stale"][3]) <= 2e-6 * abs(runs["stale"][3])


def test_extreme_compartment_radius():
    """A tiny r_comp (here POL_HARMONIC_BOND_R0 = 1e-4 nm => r_comp = 1.5e-4 nm): the Gaussian's exponent constant
    -log2(e) / (2 r_comp^2) = -3.2e7 nm^-2 underflows every pair to zero without producing anything non-finite."""
    s = synthetic_system("gw_200k", n_beads=1500, jitter=0.03, seed=8, COB_USE_COMPARTMENT_BLOCKS=True,
                         POL_HARMONIC_BOND_R0=1e-4, POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False,
                         SC_USE_SPHERICAL_CONTAINER=False, IBL_USE_B_LAMINA_INTERACTION=False)
    assert s.radii[2] == pytest.approx(1.5e-4)
    _check(s, 0.6, "tiny r_comp")


def test_zero_strength_terms_are_left_out():
    """EV_EPSILON = 0 must behave as "no excluded volume" (the pair kernels factor eps out of the pair loop and would
    otherwise divide by it), and a zero compartment amplitude as no attraction."""
    s0 = synthetic_system("gw_200k", n_beads=2000, jitter=0.03, seed=1, COB_USE_COMPARTMENT_BLOCKS=True, EV_EPSILON=0.0)
    et, F = _check(s0, 0.6, "eps = 0")
    assert et[0] == 0.0 and et[1] != 0.0 and np.all(np.isfinite(F))
    s1 = synthetic_system("gw_200k", n_beads=2000, jitter=0.03, seed=1, COB_USE_COMPARTMENT_BLOCKS=True, COB_EA=0.0, COB_EB=0.0)
    et, F = _check(s1, 0.6, "Ea = Eb = 0")
    assert et[1] == 0.0 and et[0] != 0.0
    from multimm_amd.engine import MMXError
    with engine_for(s1) as eng:
        with pytest.raises(MMXError):
            eng.set_excluded_volume(100.0, 0.1, 0.05, 0.0, 0.6)   # power must be positive


@pytest.mark.parametrize("which", ["bonded_only", "pairs_only", "separate_bonded", "serial_bonded", "with_chb"])
def test_minimize_under_every_launch_shape(which):
    """The bonded pass is the first writer of the gradient and normally rides in the cell scan's launch; the pair
    kernels add to it.  Every other shape of an evaluation must minimize to the same quality: no pair term at all
    (the bonded pass writes the whole gradient), no bonded term (it writes zeros), the three bonded kernels launched
    separately after a memset, the fused pass as its own launch, chromosomal blocks in between.  Checked: the energy
    the minimizer reports is the oracle's energy at the returned positions, and the same number of iterations from the
    same start lands within 1e-3 of the decrease of the default shape."""
    from oracle.oracle import Oracle
    ff = {}
    opts = {}
    if which == "bonded_only":
        ff = dict(EV_USE_EXCLUDED_VOLUME=False, COB_USE_COMPARTMENT_BLOCKS=False)
    elif which == "pairs_only":
        ff = dict(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False,
                  SC_USE_SPHERICAL_CONTAINER=False, IBL_USE_B_LAMINA_INTERACTION=False)
    elif which == "separate_bonded":
        opts = {"fused_bonded": 0}
    elif which == "serial_bonded":
        opts = {"overlap_bonded": 0}
    elif which == "with_chb":
        ff = dict(CHB_USE_CHROMOSOMAL_BLOCKS=True, CHB_DE=0.05)
    s = synthetic_system("gw_200k", n_beads=3000, jitter=0.02, seed=6, **ff)
    # launch shapes are compared on the pair kernel with a fixed summation order: 80 L-BFGS iterations amplify the
    # last-bit differences of the default (half-shell, atomics) kernel to ~0.2 % of the energy drop in the bond-free case
    with engine_for(s) as eng:
        eng.set_option("deterministic", 1)
        st0 = eng.minimize(tolerance=0.0, max_iters=80)
    with engine_for(s) as eng:
        eng.set_option("deterministic", 1)
        for k, v in opts.items():
            eng.set_option(k, v)
        st = eng.minimize(tolerance=0.0, max_iters=80)
        x = eng.get_positions()
        et, _ = eng.compute()
    assert st.iterations == 80 and st.e_final < st.e_initial
    scale = np.abs(et).sum()
    assert abs(Oracle(s).energy(x) - st.e_final) <= E_RTOL * scale + E_ATOL
    assert abs(et.sum() - st.e_final) <= E_RTOL * scale + E_ATOL
    assert abs(st.e_final - st0.e_final) <= 1e-3 * abs(st0.e_initial - st0.e_final)
