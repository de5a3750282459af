"""Domain decomposition (BASELINE config 5) executed with several ranks on ONE GPU.

The ranks are handles of this process driven by one thread each and joined by the loopback communicator
(mmx_comm_init_local: host barrier + HIP events + device copies in rank order).  What runs is exactly the
multi-GPU control flow of the library -- owned slices, ghost lists rebuilt on the stream from the peers' need-maps, the
per-evaluation halo exchange of the listed beads only (messages of host-known capacity), ONE fp64 all-reduce per
evaluation (energies, Gram rows, flags), identical line-search decisions on every rank, the halt-and-repeat protocol
when a list goes stale or outgrows its message -- with RCCL's transport (ncclSend/ncclRecv groups, all-gathers,
all-reduce) replaced.  (The RCCL calls themselves: test_rccl_path_single_rank.)
"""
import threading

import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd.engine import Engine, engine_for

pytestmark = pytest.mark.gpu

ALL_ON = dict(SC_USE_SPHERICAL_CONTAINER=True, COB_USE_COMPARTMENT_BLOCKS=True, SCB_USE_SUBCOMPARTMENT_BLOCKS=True,
              IBL_USE_B_LAMINA_INTERACTION=True, CF_USE_CENTRAL_FORCE=True)


def run_ranks(system, world, fn, timeout=300, **options):
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

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout)
    stuck = [r for r, t in enumerate(th) if t.is_alive()]
    if stuck:  # an engine that is still being driven must not be closed under its thread: fail, leave it to the process end
        pytest.fail(f"ranks {stuck} did not come back within {timeout} s (first error so far: {err[:1]})", pytrace=False)
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
    res = run_ranks(s, world, lambda e: (e.compute(), e.owned_beads()))
    F = np.zeros_like(F0)
    for (et, f), ids in res:
        assert np.allclose(et, et0, rtol=2e-6, atol=1e-6)          # all-reduced totals (fp32 pair sums: order differs)
        assert np.array_equal(et, res[0][0][0])                     # ... bit-identical on every rank
        F[ids] = f
    assert sorted(np.concatenate([ids for _, ids in res]).tolist()) == list(range(s.n_beads))   # the ranks partition the beads
    assert np.abs(F - F0).max() <= 1e-5 * np.abs(F0).max()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_half_shell_pair_kernel_on_decomposed_ranks(world):
    """The half-shell kernel (every pair once, reaction through the LDS window) on ranks that hold ghosts: pairs of an owned
    bead and a ghost are evaluated once, by whichever cluster has the lower id, the force on the ghost is dropped and
    half the pair's energy booked; pairs of two ghosts are culled.  nb_variant 4096 forces the kernel at this size (it
    is the default from 100 000 local beads: test_config5_size_gw_1m_on_8_ranks).  Against the single-domain forces of
    the full-shell kernel and against the full-shell kernel on the same ranks."""
    s = synthetic_system("gw_200k", n_beads=12000, jitter=0.02, seed=7, **ALL_ON)
    with engine_for(s) as eng:
        eng.set_option("nb_variant", 8192)
        et0, F0 = eng.compute()

    def job(e):
        (et, f), n3, ids = e.compute(), e.get_option("n3_launches"), e.owned_beads()
        st = e.minimize(tolerance=0.0, max_iters=12)
        return et, f, ids, None, n3, (st.iterations, st.status, st.e_initial, st.e_final)

    half = run_ranks(s, world, job, nb_variant=4096)
    full = run_ranks(s, world, job, nb_variant=8192)
    F = np.zeros_like(F0)
    for (et, f, ids, _, n3, mini), ref in zip(half, full):
        assert n3 >= 1 and ref[4] == 0
        assert np.allclose(et, et0, rtol=2e-6, atol=1e-3), (et, et0)
        assert np.array_equal(et, half[0][0])
        F[ids] = f
        assert mini[:2] == ref[5][:2]
        assert abs(mini[3] - ref[5][3]) <= 3e-3 * abs(ref[5][2] - ref[5][3])
    assert np.abs(F - F0).max() <= 4e-6 * np.abs(F0).max() + 2e-3


@pytest.mark.parametrize("world", [2, 4])
def test_halo_beside_the_owned_build(world):
    """Option dd_overlap (off by default: DESIGN.md section 8): list kernels, both collectives of the halo and the ghost count on the
    handle's second stream while the evaluation's own stream sorts the owned beads into their clusters; the ghosts' clusters, the
    work items and the bonded pass follow in a second launch once the ghosts are in place.  The cluster list is the one-launch
    build's (owned beads sort before ghosts, the kinds never share a cluster): same decisions, same energies to summation order.
    (Without chromosomal blocks: that term has no cutoff and keeps the all-gather of every position.  20 000 beads per rank: the
    slot table of the direct build does not fit systems much smaller.)"""
    s = synthetic_system("gw_200k", n_beads=20000 * world, jitter=0.02, seed=7)

    def job(e):
        e.minimize(tolerance=0.0, max_iters=15)   # (the slot table of the direct build is sized by the first call's polls)
        st = e.minimize(tolerance=0.0, max_iters=25)
        et, f = e.compute()
        return ((st.iterations, st.status, st.e_initial, st.e_final), et, f, e.owned_beads(), e.get_option("dd_overlapped"),
                e.get_option("direct_builds"), e.get_option("n3_launches"), e.get_positions())

    one = run_ranks(s, world, job, nb_variant=4096, dd_overlap=0)
    two = run_ranks(s, world, job, nb_variant=4096, dd_overlap=1)
    for a, b in zip(one, two):
        assert a[4] == 0 and b[4] >= 20, (a[4], b[4])          # every evaluation after the first synchronous rebuild overlapped
        assert a[5] > 0 and b[5] > 0 and b[6] > 0               # direct builds, half-shell kernel
        assert a[0][:2] == b[0][:2]
        # (loose on purpose: the half-shell kernel's float atomics make two runs of ONE build differ by several per cent of the
        #  descent after 40 iterations of collapse; what pins the overlapped build is the pair of exact checks below)
        assert abs(a[0][3] - b[0][3]) <= 0.4 * abs(a[0][2] - a[0][3])
        for r in (a, b):
            # the energy the minimizer's last (overlapped) evaluation accepted = a fresh evaluation of the final positions, which
            # rebuilds its lists synchronously and takes the one-launch build
            assert abs(r[1].sum() - r[0][3]) <= 2e-6 * np.abs(r[1]).sum() + 1e-2, (r[1].sum(), r[0][3])
    # ... and that fresh evaluation's forces against one domain
    x = two[0][7]
    with engine_for(s) as eng:
        eng.set_positions(x)
        et0, F0 = eng.compute()
    F = np.zeros_like(F0)
    for r in two:
        assert np.allclose(r[1], et0, rtol=2e-6, atol=1e-3), (r[1], et0)
        F[r[3]] = r[2]
    assert np.abs(F - F0).max() <= 4e-6 * np.abs(F0).max() + 2e-3


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


def test_md_steps_on_stale_ghost_lists_are_taken_back():
    """A ghost list that outgrows its message during MD is only seen at a poll, after positions and velocities have been
    integrated with forces that lacked ghosts.  mmx_md_step then returns MMX_ERR_STATE ("void ... call again") -- and the
    steps really are void: x, v, xlo and the step counter are put back to the last poll that found the lists in order
    (round 3 left them where the contaminated steps had taken them).  Messages without any slack (inject_fault bit 2)
    make that happen every few steps (and would make it happen in EVERY call of more than one step: the first step of a call
    that has no valid lists builds them synchronously with fresh capacities, so the retries go step by step); the
    trajectory that comes out of the retries must be the single-domain one."""
    from multimm_amd.engine import MMXError
    s = synthetic_system("gw_200k", n_beads=5000, **ALL_ON)
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=60)
        x_start = eng.get_positions()
    total = 48

    def job(e, tight):
        chunk = 1 if tight else 4
        e.set_positions(x_start)
        e.md_configure("langevin", dt_ps=0.005, seed=3)
        e.set_velocities_to_temperature(310.0, seed=3)
        done, failures = 0, 0
        while done < total:
            try:
                st = e.md_step(chunk)
                done += chunk
                assert st.step_count == done
            except MMXError as exc:
                assert tight and exc.code == -5 and "taken back" in str(exc), exc
                assert e.get_option("md_step") == done          # the step counter went back with the state
                failures += 1
                assert failures < 200
        return (st.potential, st.kinetic, st.step_count), e.get_positions(), failures

    with engine_for(s) as eng:
        ref = job(eng, False)
    res = run_ranks(s, 3, lambda e: job(e, True), inject_fault=4)
    assert all(r[0] == res[0][0] and np.array_equal(r[1], res[0][1]) and r[2] == res[0][2] for r in res)
    assert res[0][2] >= 1, "no list outgrew its message: the test did not exercise the roll-back"
    assert res[0][0][2] == ref[0][2] == total
    assert abs(res[0][0][1] - ref[0][1]) <= 1e-4 * ref[0][1]
    assert np.abs(res[0][1] - ref[1]).max() < 1e-4


@pytest.mark.parametrize("world", [2, 4, 8])
def test_halo_exchange_equals_the_all_gather_of_every_position(world):
    """The boundary-only ghost exchange (default) against the all-gather of every position (dd_halo = 0).  Same physics;
    not always the same bits: the all-gather path also bins the foreign beads of the grid's outermost, partly empty cell
    layer (beyond the cutoff of every owned bead: zero force), which changes how a cell's beads fall into clusters of 8
    and with it the order of fp32 sums.  Forces agree to rounding, a minimization stays together; a rank receives only
    its halo.  (15 iterations: the two paths put different ghosts into the cells, cluster composition and fp32 summation
    order differ, and from the lattice start 40 iterations amplify that to a per cent of the energy drop.)"""
    s = synthetic_system("gw_200k", n_beads=12000, jitter=0.02, seed=5, **ALL_ON)

    def job(e):
        et, f = e.compute()
        ghosts_at_start = e.get_option("dd_ghosts")     # lists built with the initial skin of 0.1 nm
        st = e.minimize(tolerance=0.0, max_iters=15)
        stats = {k: e.get_option(k) for k in ("dd_ghosts", "dd_redecompositions", "dd_exchanges", "dd_bytes_sent",
                                              "dd_sync_rebuilds", "dd_halts")}
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
        # lists rebuilt on the stream before every evaluation; the host only synchronises at the start of a call
        assert st["dd_exchanges"] > 15 and st["dd_redecompositions"] >= st["dd_exchanges"] - 2
        assert st["dd_sync_rebuilds"] <= 2 + st["dd_halts"]
        assert st["dd_bytes_sent"] / st["dd_exchanges"] <= 16 * n_own * (world - 1)   # never more than the all-gather moves
    # a halo, not everybody else (12 000 beads are a box of ~2.5 nm: with 8 ranks a slice is all surface)
    if world < 8:
        assert min(h[4]["ghosts_at_start"] / (s.n_beads - h[5]) for h in halo) < 1.0
    assert all(f[4]["dd_exchanges"] == 0 for f in full)


def _halting_job(e):
    st = e.minimize(tolerance=0.0, max_iters=30)
    return (st.iterations, st.evaluations, st.status, st.e_final), e.get_positions(), e.get_option("dd_halts")


def _same_minimization(ref, other):
    for a, b in zip(ref, other):   # (other lists: cluster composition and fp32 sum order may differ)
        assert a[0][0] == b[0][0] and a[0][2] == b[0][2] and abs(a[0][3] - b[0][3]) <= 3e-3 * abs(b[0][3])
        assert np.abs(a[1] - b[1]).max() < 0.1
    assert all(t[0] == other[0][0] and np.array_equal(t[1], other[0][1]) for t in other)   # ranks agree bit for bit


def test_stale_ghost_lists_halt_and_repeat():
    """Lists rebuilt every 4th evaluation under a skin so thin that nearly every trial move outruns it: the evaluation
    that notices decides nothing (PH_HALT on every rank, through the all-reduced flag), the host rebuilds the lists at the
    trial point and repeats it.  The minimization must come out as with lists rebuilt before every evaluation."""
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=2, **ALL_ON)
    ref = run_ranks(s, 3, _halting_job, dd_rebuild_every=1)
    assert ref[0][2] == 0                                       # exact lists, comfortable messages: nothing to repeat
    thin = run_ranks(s, 3, _halting_job, dd_rebuild_every=4, dd_skin=1e-5)   # doubles on every halt: 1e-5 ... 0.01 nm
    _same_minimization(ref, thin)
    assert thin[0][2] > 10      # it did halt and repeat, many times
    wide = run_ranks(s, 3, _halting_job, dd_rebuild_every=4, dd_skin=0.8)    # a skin that outlasts four evaluations
    _same_minimization(ref, wide)
    assert wide[0][2] <= 2


def test_ghost_list_outgrowing_its_message_halts_and_repeats():
    """Messages sized without any slack (inject_fault bit 2): whenever a ghost list grows between two polls of the host it
    no longer fits, the evaluation is void on every rank and is repeated with fresh capacities."""
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=2, **ALL_ON)
    ref = run_ranks(s, 3, _halting_job, dd_rebuild_every=1)
    tight = run_ranks(s, 3, _halting_job, dd_rebuild_every=1, inject_fault=4)
    _same_minimization(ref, tight)
    assert tight[0][2] >= 1


def test_chromosomal_blocks_fall_back_to_the_all_gather():
    """The chromosomal-block term has no cutoff: every bead of a chromosome acts on every other one, across slices.  A
    decomposed run with it must not use the halo (round 2 did, and computed the term from stale positions): energies
    and forces equal the single-domain ones, also after the beads have moved."""
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=4, CHB_USE_CHROMOSOMAL_BLOCKS=True, **ALL_ON)
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=15)
        et0, F0 = eng.compute()
        x1 = eng.get_positions()

    def job(e):
        e.minimize(tolerance=0.0, max_iters=3)      # leaves positions of an earlier state in every rank's pos4
        e.set_positions(x1)
        return e.compute(), e.owned_beads(), e.get_option("dd_exchanges")

    for opts in (dict(), dict(dd_halo=0)):
        res = run_ranks(s, 3, job, **opts)
        F = np.zeros_like(F0)
        for (et, f), ids, nx in res:
            assert nx == 0                                          # no halo exchange took place
            assert np.allclose(et, et0, rtol=2e-6, atol=1e-6)
            F[ids] = f
        assert np.abs(F - F0).max() <= 1e-5 * np.abs(F0).max()


def test_segments_are_reassigned_spatially_and_the_minimization_carries_on():
    """north_star / SURVEY 8e: "the particle domain is spatially decomposed ... re-decompose".  Ownership is a set of
    62-bead segments per rank; while the structure deforms mmx_minimize re-assigns the segments by recursive bisection of
    their centroids and the per-bead vectors of L-BFGS migrate with them (csrc/mmx_engine.hpp:dd_reassign).  The
    optimisation must not notice: same iteration count, energies that differ by rounding only, forces equal to a
    single-domain engine's at the same positions, every rank the same structure -- and afterwards the ranks still partition
    the beads, in other pieces than before.  MD steps on the re-assigned handles follow the single-domain trajectory."""
    s = synthetic_system("gw_200k", n_beads=30000, jitter=0.02, seed=3, **ALL_ON)
    world, iters = 4, 120

    def job(e):
        ids0 = e.owned_beads()
        st = e.minimize(tolerance=0.0, max_iters=iters)
        x = e.get_positions()
        et, f = e.compute()
        ids1 = e.owned_beads()
        e.md_configure("verlet", dt_ps=0.001, seed=5)
        e.set_velocities_to_temperature(310.0, seed=5)
        md = e.md_step(6)
        x_md = e.get_positions()
        stats = {k: e.get_option(k) for k in ("dd_reassignments", "dd_reassign_attempts", "dd_segments_moved", "dd_halts")}
        return (st.iterations, st.status, st.e_initial, st.e_final), x, et, f, ids0, ids1, stats, x_md, md.potential

    fixed = run_ranks(s, world, job, dd_spatial=0)
    moved = run_ranks(s, world, job, dd_reassign_first=8, dd_reassign_max=32)
    assert all(o[6]["dd_reassignments"] == 0 for o in fixed)
    assert all(o[6]["dd_reassignments"] >= 2 and o[6]["dd_segments_moved"] > 0 for o in moved)
    assert len({tuple(sorted(o[6].items())) for o in moved}) == 1            # every rank took the same decisions
    for res in (fixed, moved):
        assert all(o[0] == res[0][0] and np.array_equal(o[1], res[0][1]) for o in res)
        assert sorted(np.concatenate([o[5] for o in res]).tolist()) == list(range(s.n_beads))
    assert all(np.array_equal(a[4], b[4]) for a, b in zip(fixed, moved))      # the same initial index ranges
    assert any(not np.array_equal(o[4], o[5]) for o in moved)                  # ... and other pieces afterwards
    assert all(np.array_equal(o[4], o[5]) for o in fixed)
    assert all(len(o[5]) % 62 == 0 or r == max(range(world), key=lambda q: moved[q][5][-1]) for r, o in enumerate(moved))
    (it_f, st_f, e0_f, e1_f), (it_m, st_m, e0_m, e1_m) = fixed[0][0], moved[0][0]
    assert (it_f, st_f) == (it_m, st_m) == (iters, 1) and e0_f == e0_m
    assert abs(e1_m - e1_f) <= 2e-2 * abs(e0_f - e1_f)       # same algorithm, other summation order
    # forces of the re-assigned ranks against one domain at the same positions
    x = moved[0][1]
    with engine_for(s) as ref:
        ref.set_positions(x)
        et0, F0 = ref.compute()
        ref.md_configure("verlet", dt_ps=0.001, seed=5)
        ref.set_velocities_to_temperature(310.0, seed=5)
        md0 = ref.md_step(6)
        x_md0 = ref.get_positions()
    F = np.zeros_like(F0)
    for o in moved:
        assert np.allclose(o[2], et0, rtol=2e-6, atol=1e-3)
        F[o[5]] = o[3]
    assert np.abs(F - F0).max() <= 1e-5 * np.abs(F0).max()
    assert np.abs(moved[0][7] - x_md0).max() <= 2e-6 and abs(moved[0][8] - md0.potential) <= 2e-6 * abs(md0.potential)
    assert all(np.array_equal(o[7], moved[0][7]) for o in moved)


def test_a_cell_too_large_for_the_in_cell_sort_voids_the_evaluation_on_a_decomposed_rank():
    """More than 4096 beads in one grid cell: the in-cell sort keeps arrival order there.  A single domain lives with that
    (results to rounding); on a decomposed rank the cell's owned beads and ghosts would no longer form separate clusters
    and the half-shell kernel's per-cluster ownership -- energy weights, ghost-ghost cull -- would be wrong, silently.  The
    cell build flags it and the call ends in MMX_ERR_STATE instead (round-3 advisor finding)."""
    from multimm_amd.engine import MMXError
    from multimm_amd.system import ChromatinSystem, ForceFieldParams
    rng = np.random.default_rng(4)
    n = 12400
    ff = ForceFieldParams(POL_USE_HARMONIC_BOND=False, POL_USE_HARMONIC_ANGLE=False, LE_USE_HARMONIC_BOND=False,
                          COB_USE_COMPARTMENT_BLOCKS=True, NB_CUTOFF=0.6)
    x = rng.uniform(0.0, 6.0, (n, 3))
    x[::2] = rng.uniform(2.7, 2.95, (n // 2, 3))          # 6 200 beads inside one 0.6 nm cell, owned by both ranks alike
    s = ChromatinSystem(n, x, np.array([0, n]), rng.choice(np.array([-2, -1, 0, 1, 2], np.int8), n), ff=ff)
    with engine_for(s) as eng:
        et0, _ = eng.compute()                             # one domain: fine
        assert eng.get_option("order_fallbacks") >= 1

    def job(e):
        try:
            e.compute()
        except MMXError as exc:
            return exc.code, str(exc)
        return 0, ""

    for code, text in run_ranks(s, 2, job):
        assert code == -5 and "4096 beads" in text, (code, text)


def test_a_rank_must_own_a_bead():
    """Slices are whole 62-bead segments, ceil(ceil(N / 62) / world) of them: 200 beads (4 segments) on 3 ranks would leave
    the third rank without any (and its launches without a grid).  mmx_create_dd refuses such a decomposition instead of
    failing later."""
    from multimm_amd.engine import MMXError
    with pytest.raises(MMXError) as exc:
        Engine(200, rank=2, world=3)
    assert exc.value.code == -1 and "too many ranks" in str(exc.value)
    Engine(200, rank=1, world=2).close()


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


def test_config5_size_gw_1m_on_8_ranks():
    """BASELINE config 5 at its own size: gw_1m (1 000 000 beads) over 8 ranks.  Owned force slices against the
    single-domain forces at the parity tolerance of tests/test_gpu_parity.py, all-reduced energies, and asserted bounds
    on what the halo costs: ghosts per owned bead and bytes on the wire per evaluation -- at the lattice start (1 000
    beads / nm^3: the densest state a run passes through) and after the first iterations."""
    s = synthetic_system("gw_1m", jitter=0.02, seed=1)   # (the bare lattice has millions of pairs at exactly r_c, where the last
    world = 8                                            # bit of r^2 decides: tests/test_gpu_parity.py's documented exception)
    with engine_for(s) as eng:
        et0, F0 = eng.compute()
        st0 = eng.minimize(tolerance=0.0, max_iters=10)

    def job(e):
        et, f = e.compute()
        ids = e.owned_beads()
        g0, slots0 = e.get_option("dd_ghosts"), e.get_option("dd_ghost_slots")
        st = e.minimize(tolerance=0.0, max_iters=10)
        stats = {k: e.get_option(k) for k in ("dd_ghosts", "dd_ghost_slots", "dd_exchanges", "dd_bytes_sent",
                                              "dd_sync_rebuilds", "dd_halts", "n3_launches")}
        return et, f, ids, len(ids), g0, slots0, (st.iterations, st.status, st.e_initial, st.e_final), stats

    res = run_ranks(s, world, job, timeout=900)
    F = np.zeros_like(F0)
    for et, f, ids, no, *_ in res:
        assert np.allclose(et, et0, rtol=2e-6, atol=1e-3)
        assert np.array_equal(et, res[0][0])
        F[ids] = f
    err = np.abs(F - F0).max()
    print(f"gw_1m on 8 ranks: max force difference {err:.3g} of max |F| {np.abs(F0).max():.4g}")
    assert err <= 4e-6 * np.abs(F0).max() + 2e-3       # F_RTOL, F_ATOL of tests/test_gpu_parity.py
    for r, (_, _, lo, no, g0, slots0, mini, st) in enumerate(res):
        print(f"  rank {r}: owned {no}, ghosts at the start {g0:.0f} (slots {slots0:.0f}), after 10 iterations "
              f"{st['dd_ghosts']:.0f}; {st['dd_bytes_sent'] / max(st['dd_exchanges'], 1) / 1e6:.2f} MB sent per evaluation; "
              f"halts {st['dd_halts']:.0f}, synchronous rebuilds {st['dd_sync_rebuilds']:.0f}")
        assert mini == res[0][6]
        # lattice start: a slice of 125 000 beads is a 6.4 x 6.4 x 3.2 nm brick of 1 000 beads / nm^3; the shell within
        # 0.6 nm of it holds ~115 000 beads, the need-map (cells of 0.3 nm grown by two) reaches 0.6-0.9 nm
        assert 0 < g0 <= 1.6 * no
        assert st["dd_ghosts"] <= 1.6 * no
        assert st["dd_bytes_sent"] / st["dd_exchanges"] <= 16 * 1.6 * no * 1.2
        assert st["dd_sync_rebuilds"] <= 3 + st["dd_halts"] and st["dd_halts"] <= 4   # (the first iterations from the lattice add ghosts fast)
        assert st["n3_launches"] >= 10          # 125 000 owned beads + ghosts: the half-shell kernel's DD instance ran
    it, status, e_i, e_f = res[0][6]
    assert (it, status) == (st0.iterations, st0.status)
    assert abs(e_i - st0.e_initial) <= 2e-6 * (abs(st0.e_initial) + abs(st0.e_final))
    assert abs(e_f - st0.e_final) <= 5e-3 * abs(st0.e_initial - st0.e_final)


def test_randomised_decompositions():
    """scripts/dd_stress.py: ten random combinations of size, rank count, rebuild interval, skin, message slack and pair
    kernel; each minimizes, runs a few MD steps and ends with forces that must equal a single-domain engine's at the same
    positions (1e-5 of max |F|), identical on every rank."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "dd_stress.py"), "10", "7"], cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "bad cases: 0" in r.stdout


def test_adaptive_lifetime_of_the_ghost_lists():
    """Option dd_adaptive: the polls choose how many evaluations (1 .. dd_rebuild_every) a set of ghost lists serves from the
    largest trial move they read back (all-reduced: every rank arrives at the same number) -- 1, exact lists and no skin, while
    the structure collapses; more once it has settled.  The minimization must come out as with lists rebuilt before every
    evaluation, and on a relaxed structure the lists must really be kept."""
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=2, **ALL_ON)
    ref = run_ranks(s, 3, _halting_job, dd_rebuild_every=1)
    ada = run_ranks(s, 3, _halting_job, dd_rebuild_every=4, dd_adaptive=1, dd_skin=0.15)
    _same_minimization(ref, ada)
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=400)
        s.positions = eng.get_positions().astype(np.float64)

    def job(e):
        st = e.minimize(tolerance=0.0, max_iters=150)
        return (st.iterations, st.evaluations, st.status, st.e_final), e.get_positions(), e.get_option("dd_halts"), \
            e.get_option("dd_lists_serve"), e.get_option("dd_redecompositions"), e.get_option("dd_move_seen")

    ref2 = run_ranks(s, 3, job, dd_rebuild_every=1)
    ada2 = run_ranks(s, 3, job, dd_rebuild_every=4, dd_adaptive=1, dd_skin=0.15)
    for a, b in zip(ref2, ada2):
        assert a[0][0] == b[0][0] and abs(a[0][3] - b[0][3]) <= 3e-3 * abs(a[0][3])
    assert all(b[3] == ada2[0][3] for b in ada2)                 # every rank the same lifetime
    assert ada2[0][3] > 1 and ada2[0][4] < 0.7 * ref2[0][4], (ada2[0][3:], ref2[0][4])     # lists kept over several evaluations: fewer rebuilds
