"""Domain decomposition (BASELINE config 5) executed with several ranks on ONE GPU.

The ranks are handles of this process driven by one thread each and joined by the loopback communicator
(mmx_comm_init_local: host barrier + HIP events + device copies in rank order).  What runs is exactly the
multi-GPU control flow of the library -- owned slices, ghost lists rebuilt at re-decomposition, the per-evaluation halo
exchange of the listed beads only, ONE fp64 all-reduce per evaluation (energies, Gram rows, stale flag), identical
line-search decisions on every rank, the halt-and-repeat protocol when a list goes stale -- with RCCL's transport
(ncclSend/ncclRecv groups, all-gathers, all-reduce) replaced.  (The RCCL calls themselves: test_rccl_path_single_rank.)
"""
import threading

import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for

pytestmark = pytest.mark.gpu

ALL_ON = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
              IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)


def run_ranks(system, world, fn, **options):
    """fn(engine) on every rank concurrently; returns the per-rank results (raises the first error)."""
    engines = [engine_for(system, rank=r, world=world) for r in range(world)]
    for e in engines:
        for k, v in options.items():
            e.set_option(k, v)
    Engine.comm_init_local(engines)
    out, err = [None] * world, []

    def work(r):
        try:
            out[r] = fn(engines[r])
        except Exception as e:  # noqa: BLE001
            err.append((r, e))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    for e in engines:
        e.close()
    if err:
        raise err[0][1]
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_compute_all_reduced_energies_and_owned_forces(world):
    s = synthetic_system("gw_200k", n_beads=5000, jitter=0.02, seed=2, **ALL_ON)
    with engine_for(s) as eng:
        et0, F0 = eng.compute()
    res = run_ranks(s, world, lambda e: (e.compute(), e.own_lo, e.n_own))
    F = np.zeros_like(F0)
    for (et, f), lo, no in res:
        assert np.allclose(et, et0, rtol=2e-6, atol=1e-6)          # all-reduced totals (fp32 pair sums: order differs)
        assert np.array_equal(et, res[0][0][0])                     # ... bit-identical on every rank
        F[lo:lo + no] = f
    assert np.abs(F - F0).max() <= 1e-5 * np.abs(F0).max()


@pytest.mark.parametrize("world", [2, 4])
def test_minimize_decomposed_matches_single_domain(world):
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=2, **ALL_ON)
    with engine_for(s) as eng:
        st0 = eng.minimize(tolerance=0.0, max_iters=40)
        x0 = eng.get_positions()

    def job(e):
        st = e.minimize(tolerance=0.0, max_iters=40)
        return (st.iterations, st.evaluations, st.status, st.e_initial, st.e_final), e.get_positions()

    res = run_ranks(s, world, job)
    for stats, x in res:
        assert stats == res[0][0]                       # identical decisions and energies on every rank
        assert np.array_equal(x, res[0][1])             # every rank sees the same final structure
    it, ev, status, e_i, e_f = res[0][0]
    assert (it, status) == (st0.iterations, st0.status)
    scale = abs(st0.e_initial) + abs(st0.e_final)
    assert abs(e_i - st0.e_initial) <= 2e-6 * scale
    # same algorithm, different fp64/fp32 summation order: trajectories stay together over 40 iterations
    assert abs(e_f - st0.e_final) <= 1e-3 * abs(st0.e_initial - st0.e_final)
    assert np.abs(res[0][1] - x0).max() < 5e-2  # lattice start: soft directions amplify rounding differences


def test_decomposed_minimization_converges():
    s = synthetic_system("chr1_50k", n_beads=4000)
    with engine_for(s) as eng:
        st0 = eng.minimize(tolerance=10.0)
    res = run_ranks(s, 2, lambda e: (lambda st: (st.status, st.iterations, st.e_final, st.rms_force))(e.minimize(tolerance=10.0)))
    assert res[0] == res[1]
    assert res[0][0] == 0 == st0.status
    # both runs stop at the OpenMM criterion; they end in neighbouring local minima of a rugged landscape
    assert abs(res[0][2] - st0.e_final) <= 2e-2 * abs(st0.e_final)


def test_md_decomposed_matches_single_domain():
    """The noise is indexed by the GLOBAL bead id, the integrator is per-bead: a decomposed trajectory is the
    single-domain trajectory up to fp32 summation order in the forces."""
    s = synthetic_system("gw_200k", n_beads=5000, **ALL_ON)
    with engine_for(s) as eng:  # one common relaxed start (minimizer trajectories of different decompositions
        eng.minimize(tolerance=0.0, max_iters=60)  # separate slowly; the MD comparison should not inherit that)
        x_start = eng.get_positions()

    def job(e):
        e.set_positions(x_start)
        e.md_configure("langevin", dt_ps=0.005, seed=3)
        e.set_velocities_to_temperature(310.0, seed=3)
        st = e.md_step(40)
        return (st.potential, st.kinetic, st.step_count), e.get_positions()

    with engine_for(s) as eng:
        ref = job(eng)
    res = run_ranks(s, 2, job)
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1])
    assert abs(res[0][0][1] - ref[0][1]) <= 1e-4 * ref[0][1]
    assert np.abs(res[0][1] - ref[1]).max() < 1e-4


@pytest.mark.parametrize("world", [2, 4, 8])
def test_halo_exchange_equals_the_all_gather_of_every_position(world):
    """The boundary-only ghost exchange (default) against the all-gather of every position (dd_halo = 0).  Same physics;
    not always the same bits: the all-gather path also bins the foreign beads of the grid's outermost, partly empty cell
    layer (beyond the cutoff of every owned bead: zero force), which changes how a cell's beads fall into clusters of 8
    and with it the order of fp32 sums.  Forces agree to rounding, a minimization stays together; a rank receives only
    its halo."""
    s = synthetic_system("gw_200k", n_beads=12000, jitter=0.02, seed=5, **ALL_ON)

    def job(e):
        et, f = e.compute()
        ghosts_at_start = e.get_option("dd_ghosts")     # lists built with the initial skin of 0.1 nm
        st = e.minimize(tolerance=0.0, max_iters=40)
        stats = {k: e.get_option(k) for k in ("dd_ghosts", "dd_redecompositions", "dd_exchanges", "dd_bytes_sent")}
        stats["ghosts_at_start"] = ghosts_at_start
        return et, f, (st.iterations, st.status, st.e_initial, st.e_final), e.get_positions(), stats, e.n_own

    halo = run_ranks(s, world, job, dd_halo=1)
    full = run_ranks(s, world, job, dd_halo=0)
    for a, b in zip(halo, full):
        assert np.allclose(a[0], b[0], rtol=2e-6, atol=1e-6)
        assert np.abs(a[1] - b[1]).max() <= 1e-5 * np.abs(b[1]).max()
        assert a[2][:2] == b[2][:2] and abs(a[2][2] - b[2][2]) <= 2e-6 * abs(b[2][2])
        assert abs(a[2][3] - b[2][3]) <= 3e-3 * abs(b[2][2] - b[2][3])   # of the energy drop (the lattice start is chaotic)
        assert np.abs(a[3] - b[3]).max() < 0.1
    for r, (_, _, _, _, st, n_own) in enumerate(halo):
        assert 0 < st["dd_ghosts"] <= s.n_beads - n_own
        assert st["dd_exchanges"] > 40 and 2 <= st["dd_redecompositions"] < st["dd_exchanges"] / 2
        assert st["dd_bytes_sent"] / st["dd_exchanges"] <= 16 * n_own * (world - 1)   # never more than the all-gather moves
    # a halo, not everybody else -- at the start, with the initial skin: 12 000 beads are a box of ~2.5 nm, and the skin the
    # collapse phase grows to (up to 1.6 nm) reaches across all of it
    if world < 8:
        assert min(h[4]["ghosts_at_start"] / (s.n_beads - h[5]) for h in halo) < 1.0
    assert all(f[4]["dd_exchanges"] == 0 for f in full)


def test_stale_ghost_lists_halt_and_repeat():
    """A skin so thin that nearly every trial move outruns it: the evaluation that notices decides nothing (PH_HALT on
    every rank, through the all-reduced flag), the host rebuilds the lists at the trial point and repeats it.  The
    minimization must come out exactly as with a comfortable skin."""
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=2, **ALL_ON)

    def job(e):
        st = e.minimize(tolerance=0.0, max_iters=30)
        return (st.iterations, st.evaluations, st.status, st.e_final), e.get_positions(), e.get_option("dd_redecompositions")

    ref = run_ranks(s, 3, job, dd_skin=1.0)
    thin = run_ranks(s, 3, job, dd_skin=1e-5)   # doubles on every early halt: 1e-5 ... 0.02 nm over 30 iterations
    for a, b in zip(ref, thin):   # (a wider skin lists more ghosts: cluster composition and fp32 sum order may differ)
        assert a[0][0] == b[0][0] and a[0][2] == b[0][2] and abs(a[0][3] - b[0][3]) <= 3e-3 * abs(b[0][3])
        assert np.abs(a[1] - b[1]).max() < 0.1
    assert all(t[0] == thin[0][0] and np.array_equal(t[1], thin[0][1]) for t in thin)   # ranks agree bit for bit
    assert thin[0][2] > ref[0][2] + 10      # it did halt and repeat, many times


def test_local_communicator_argument_checks():
    s = synthetic_system("region_5k", n_beads=600)
    a, b = engine_for(s, rank=0, world=2), engine_for(s, rank=1, world=2)
    from multimm_amd.engine import MMXError
    with pytest.raises(MMXError):
        Engine.comm_init_local([b, a])      # ranks out of order
    Engine.comm_init_local([a, b])
    with pytest.raises(MMXError):
        Engine.comm_init_local([a, b])      # already initialised
    a.close()
    b.close()
