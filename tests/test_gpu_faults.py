"""A force kernel that cannot do its work must end the call with an error, not with partial forces.

The half-shell pair kernel (k_nb_n3) has two such exits: a wave of its work-unit pipeline gives up waiting (a protocol
bug, bounded spin-wait) and a cell build that needs more work items than the list holds.  Neither can be provoked by
data, so the option "inject_fault" provokes them: bit 0 makes every wait time out at once, bit 1 shrinks the item list
to one entry.  Partial force sums are finite -- without the error word the minimizer would carry on with a wrong
gradient and return MMX_OK (round 2 did)."""
import numpy as np
import pytest

from multimm_amd import synthetic_system
from multimm_amd.engine import MMXError, engine_for

pytestmark = pytest.mark.gpu
MMX_ERR_STATE = -5


@pytest.mark.parametrize("fault", [1, 2])
def test_pair_kernel_failure_surfaces_from_every_entry_point(fault):
    s = synthetic_system("gw_200k", n_beads=20000, jitter=0.02, seed=3)
    with engine_for(s) as eng:
        eng.set_option("nb_variant", 4096)          # the half-shell kernel, whatever the size
        et0, f0 = eng.compute()
        assert np.isfinite(et0).all() and eng.get_option("n3_launches") > 0
        eng.set_option("inject_fault", fault)
        with pytest.raises(MMXError) as exc:
            eng.compute()
        assert exc.value.code == MMX_ERR_STATE and "k_nb_n3" in str(exc.value)
        with pytest.raises(MMXError) as exc:
            eng.minimize(tolerance=0.0, max_iters=5)
        assert exc.value.code == MMX_ERR_STATE
        eng.md_configure("verlet", dt_ps=0.001)
        with pytest.raises(MMXError) as exc:
            eng.md_step(3)
        assert exc.value.code == MMX_ERR_STATE
        # the handle survives: without the fault the same calls work again and give the same answer
        eng.set_option("inject_fault", 0)
        eng.set_positions(s.positions)              # (the void MD steps moved the beads with partial forces)
        et1, f1 = eng.compute()
        assert np.allclose(et1, et0, rtol=1e-6, atol=1e-3)
        assert np.abs(f1 - f0).max() <= 1e-5 * np.abs(f0).max()
        st = eng.minimize(tolerance=0.0, max_iters=5)
        assert st.iterations == 5 and st.status == 1


@pytest.mark.parametrize("name, n", [("chr1_50k", 30000), ("gw_200k", 100000)])
def test_slot_table_rows_that_overflow_void_the_evaluation_and_are_enlarged(name, n):
    """Trial moves write their sort keys into per-cell rows of a slot table (option cell_slots) cut from the last poll's
    fullest cell; a cell that outgrows its row voids the evaluation (PH_HALT, reason 8), the host cuts longer rows and repeats
    it.  inject_fault bit 4 cuts rows of 64 slots at every poll, so the crowded cells of a collapse from the lattice overflow
    again and again: the minimization must end where the counting sort's fill path ends, with the halts counted -- and without
    the cell counters of the skipped cells leaking into the repeat (they once did: counts doubled, offsets ran past the
    arrays).  Without overflows the slot path is bitwise the fill path (`deterministic`); a repeated evaluation bins on the
    grid its void twin laid out, not on its predecessor's, so clusters -- and roundings -- differ from there on: 1e-6."""
    s = synthetic_system(name, n_beads=n)
    ends = {}
    for label, slots, fault in (("fill", 0, 0), ("slots", 1, 0), ("rows of 64", 1, 16)):
        with engine_for(s) as eng:
            eng.set_option("deterministic", 1)
            eng.set_option("cell_slots", slots)
            eng.set_option("inject_fault", fault)
            st = eng.minimize(tolerance=0.0, max_iters=150)
            ends[label] = (st.e_final, eng.get_positions(), eng.get_option("cell_slot_halts"), st.iterations)
    assert ends["rows of 64"][2] > 0 and ends["fill"][2] == 0
    assert ends["slots"][3] == ends["rows of 64"][3] == ends["fill"][3] == 150
    assert ends["slots"][0] == ends["fill"][0] and np.array_equal(ends["slots"][1], ends["fill"][1])
    assert abs(ends["rows of 64"][0] - ends["fill"][0]) <= 1e-6 * abs(ends["fill"][0])
    assert np.abs(ends["rows of 64"][1] - ends["fill"][1]).max() < 0.05   # nm, after 150 iterations


# ---- every way an evaluation can be declared void, one by one ------------------------------------------------------------
# PH_HALT reasons (MinState::halt_reason): 1 a ghost list went stale, 2 a ghost list outgrew its message, 4 the kept cell
# structure went stale, 8 a cell outgrew its row of the slot table, 16 the grid is beyond the direct build.  What a void
# evaluation leaves behind -- cell counters, row totals, slot rows, fsort slots of skipped cells, the work-item count, the
# queue head, ghost-list lengths -- is listed with the place it is reset in DESIGN.md ("void evaluations"); this test
# provokes each reason and demands the minimization the un-faulted run gives.

def _min_job(eng, iters=120):
    st = eng.minimize(tolerance=0.0, max_iters=iters)
    return st, eng.get_positions()


@pytest.mark.parametrize("name, n, half_shell", [("chr1_50k", 30000, False), ("gw_200k", 100000, True)])
def test_every_single_domain_halt_reason_is_survived(name, n, half_shell):
    s = synthetic_system(name, n_beads=n)

    def variants(cases, e_tol):
        runs = {}
        for label, fault, counter in cases:
            with engine_for(s) as eng:
                if not half_shell:
                    eng.set_option("deterministic", 1)
                eng.set_option("inject_fault", fault)
                st, x = _min_job(eng)
                runs[label] = (st, x, eng.get_option(counter) if counter else None, eng.get_option("direct_builds"),
                               eng.get_option("kernel_error"), eng.get_option("cell_reuses"))
                # the energy the minimizer reports for its last point is the energy of that point (a fresh evaluation)
                eng.set_option("inject_fault", 0)
                et, _ = eng.compute()
                assert abs(et.sum() - st.e_final) <= 2e-6 * np.abs(et).sum() + 1e-3, label
        st0, x0 = runs["clean"][0], runs["clean"][1]
        for label, (st, x, _, _, kerr, _) in runs.items():
            assert kerr == 0, label
            assert st.iterations == st0.iterations == 120 and st.status == 1, label
            # (a repeated evaluation bins on another grid / another structure: other clusters, other roundings, from there on)
            assert abs(st.e_final - st0.e_final) <= e_tol * abs(st0.e_final), label
            assert half_shell or np.abs(x - x0).max() < 0.05, label
        return runs

    # the collapse from the lattice: crowded cells (rows of 64 slots overflow again and again), a build per evaluation
    dense = variants((("clean", 0, None), ("slot rows of 64 (8)", 16, "cell_slot_halts"),
                      ("grid beyond the direct build (16)", 64, None), ("both", 16 | 64, "cell_slot_halts")),
                     5e-2 if half_shell else 1e-6)   # (half shell: float atomics, the collapse amplifies their roundings -- two clean
                                                      #  runs part by as much; what holds a run to the physics is the fresh evaluation above)
    assert dense["clean"][3] > 100                                 # the direct build is what ran
    assert dense["slot rows of 64 (8)"][2] > 0 and dense["both"][2] > 0
    # (one batch of direct builds is enqueued before the poll that sees the first of them voided; the scan-based build from there on)
    assert dense["grid beyond the direct build (16)"][3] <= 40 < dense["clean"][3]
    # a relaxed state: cell structures are kept over several evaluations (inject_fault bit 3: always found stale)
    with engine_for(s) as eng:
        eng.minimize(tolerance=0.0, max_iters=1200)
        s.positions = eng.get_positions().astype(np.float64)
    relaxed = variants((("clean", 0, None), ("stale structure (4)", 8, "cell_stale_halts"),
                        ("stale structure + grid beyond the direct build", 8 | 64, "cell_stale_halts")),
                       5e-2 if half_shell else 3e-4)   # (120 more iterations of a relaxed state: roundings decide line searches)
    assert relaxed["clean"][3] > 5
    # (whether a structure is kept within these 120 iterations depends on the displacements the first polls see: where the clean
    #  run kept structures for ten evaluations or more, the injected run must have found some stale)
    assert relaxed["clean"][5] < 10 or relaxed["stale structure (4)"][2] > 0


def test_decomposed_halt_reasons_on_two_loopback_ranks():
    from test_gpu_dd import ALL_ON, _halting_job, _same_minimization, run_ranks
    s = synthetic_system("gw_200k", n_beads=6000, jitter=0.02, seed=2, **ALL_ON)
    ref = run_ranks(s, 2, _halting_job, dd_rebuild_every=1)
    assert ref[0][2] == 0
    stale = run_ranks(s, 2, _halting_job, dd_rebuild_every=4, dd_skin=1e-5)     # reason 1
    _same_minimization(ref, stale)
    assert stale[0][2] > 5
    tight = run_ranks(s, 2, _halting_job, dd_rebuild_every=1, inject_fault=4)    # reason 2
    _same_minimization(ref, tight)
    assert tight[0][2] >= 1


def test_a_send_list_that_outgrows_its_message_under_the_direct_build():
    """Decomposed ranks at a size where the one-launch build runs (20 000 beads per rank), collapsing from the lattice (cells of
    more than 256 beads: the workgroup-wide sort by counting), messages without slack (inject_fault bit 2): a list that outgrows its
    message keeps stale ids behind its last entry, the peer receives some ghosts TWICE -- equal keys in a cell.  The evaluation is
    void (flag through the all-reduce) and must be repeated, not end the call: equal keys may not cost the sort a place
    (block_rank_sort ranks them by their place in the input).  Round 5 regression: the RCCL twin of this failed once with
    KERR_BOUNDS before that."""
    from test_gpu_dd import run_ranks
    s = synthetic_system("gw_200k", n_beads=60000, jitter=0.02, seed=7)

    def job(e):
        e.minimize(tolerance=0.0, max_iters=15)   # (sizes the slot table: the direct build runs from the second call on)
        st = e.minimize(tolerance=0.0, max_iters=60)
        return st.iterations, st.status, st.e_final, e.get_option("dd_halts"), e.get_option("direct_builds"), e.get_positions()

    ref = run_ranks(s, 3, job, nb_variant=4096)
    tight = run_ranks(s, 3, job, nb_variant=4096, inject_fault=4)
    for a, b in zip(ref, tight):
        assert a[:2] == b[:2] and b[4] > 0
        assert b[3] >= 1                                        # lists did outgrow their messages
    assert all(np.array_equal(t[5], tight[0][5]) for t in tight)   # ranks agree bit for bit


def test_a_fold_that_never_gets_its_partials_is_an_error_code():
    """k_tail's folding workgroup polls the other workgroups' tagged partial sums with a bounded spin (inject_fault bit 5: it
    gives up at once): the evaluation is void, the call ends in MMX_ERR_STATE, the handle works again afterwards."""
    s = synthetic_system("gw_200k", n_beads=20000, jitter=0.02, seed=3)
    with engine_for(s) as eng:
        st0 = eng.minimize(tolerance=0.0, max_iters=5)
        eng.set_positions(s.positions)
        eng.set_option("inject_fault", 32)
        with pytest.raises(MMXError) as exc:
            eng.minimize(tolerance=0.0, max_iters=5)
        assert exc.value.code == MMX_ERR_STATE and "k_tail" in str(exc.value)
        eng.set_option("inject_fault", 0)
        eng.set_positions(s.positions)
        st1 = eng.minimize(tolerance=0.0, max_iters=5)
        assert st1.iterations == 5 and abs(st1.e_final - st0.e_final) <= 1e-5 * abs(st0.e_final)


def test_no_halo_after_a_reassignment_is_refused():
    """A minimization with the halo re-assigns the 62-bead segments to the ranks; the paths without a halo (dd_halo = 0,
    chromosomal blocks) assume contiguous slices.  They must refuse (MMX_ERR_STATE), not compute on scattered beads as if
    they were a slice."""
    from test_gpu_dd import ALL_ON, run_ranks
    s = synthetic_system("gw_200k", n_beads=9000, jitter=0.02, seed=5, **ALL_ON)

    def job(e):
        e.minimize(tolerance=0.0, max_iters=60)
        moved = e.get_option("dd_reassignments")
        e.set_option("dd_halo", 0)
        try:
            e.compute()
            return moved, None
        except MMXError as exc:
            return moved, (exc.code, str(exc))

    res = run_ranks(s, 3, job, dd_reassign_first=8, dd_reassign_max=16)
    assert all(m >= 1 for m, _ in res), "no re-assignment took place: the test did not exercise the refusal"
    assert all(err is not None and err[0] == MMX_ERR_STATE and "re-assigned" in err[1] for _, err in res)
