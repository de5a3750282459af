/*
 * mmx.h -- C ABI of libmmx.so: MI355X (gfx950) chromatin force field + L-BFGS minimizer.
 *
 * This is the drop-in boundary for the one path MultiMM obtains from OpenMM in
 * MultiMM.min_energy() (reference: src/multimm/model.py:859-897) for the force terms that
 * MultiMM.add_forcefield() installs (model.py:812-857).  Every entry point names the reference
 * call sites it replaces.  Plain pointers and sizes only; no C++/torch types cross the ABI.
 *
 * Conventions
 *   - Units: nm, kJ/mol, rad (OpenMM's defaults, i.e. what the reference's bare floats mean).
 *   - Every function returns int: MMX_OK (0) or a negative MMX_ERR_* class; nothing throws or
 *     aborts across the ABI.  mmx_last_error(h) gives the message of the last failure.  On the device side every offset
 *     derived from counters is checked against its array before it is used as an address (KERR_BOUNDS), every wait between
 *     workgroups is bounded (KERR_*_WAIT / KERR_N3_SPIN): a corrupt or stale state ends in MMX_ERR_STATE, not in a fault.
 *   - The caller owns all input buffers (host memory); the library copies before returning.
 *     Output buffers are caller-allocated host memory.
 *   - A handle is bound to one GPU and one HIP stream; it is not thread-safe.  Distinct handles
 *     (same or different GPUs) may be driven from distinct host threads/processes.
 *   - There is no CPU fallback: if no gfx950 device is usable mmx_create fails with MMX_ERR_HIP.
 */
#ifndef MMX_H
#define MMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMX_OK 0
#define MMX_ERR_BAD_ARG (-1)
#define MMX_ERR_HIP (-2)
#define MMX_ERR_RCCL (-3)
#define MMX_ERR_NAN (-4)   /* non-finite energy met by the minimizer / compute */
#define MMX_ERR_STATE (-5) /* call made in the wrong order (e.g. compute before set_positions) */
#define MMX_ERR_NOMEM (-6) /* host allocation failed inside the library */

/* Energy-term slots of mmx_compute()/mmx_stats (order of model.py:812-857 collapsed on kernels). */
enum {
    MMX_T_EV = 0,        /* add_evforce               model.py:164-217 */
    MMX_T_GAUSS = 1,     /* add_compartment_blocks + add_subcompartment_blocks  model.py:219-384 */
    MMX_T_BOND = 2,      /* add_harmonic_bonds        model.py:625-636 */
    MMX_T_ANGLE = 3,     /* add_stiffness             model.py:708-720 */
    MMX_T_LOOP = 4,      /* add_loops (harmonic)      model.py:638-659 */
    MMX_T_CONTAINER = 5, /* add_spherical_container   model.py:453-466 */
    MMX_T_LAMINA = 6,    /* add_Blamina_interaction   model.py:468-507 */
    MMX_T_CENTRAL = 7,   /* add_central_force         model.py:552-623 */
    MMX_T_CHB = 8,       /* add_chromosomal_blocks    model.py:386-451 */
    MMX_N_TERMS = 9
};

/* Kernel slots of mmx_stats.kernel_* and mmx_time_kernel(). */
enum {
    MMX_K_CELL_BUILD = 0, /* K1: pack (+ trial move, direction) + cell hash + count + scan (+ bonded pass) + fill + order */
    MMX_K_NONBONDED = 1,  /* K2: cell-list pair kernel (or K2x all-pairs when cutoff <= 0) */
    MMX_K_BACKBONE = 2,   /* K3: bonds + angles */
    MMX_K_LOOPS = 3,      /* K4: loop restraints */
    MMX_K_CONFINE = 4,    /* K5: container + lamina + central */
    MMX_K_LBFGS = 5,      /* K6: history pass of an evaluation ((s,y) candidate, Gram rows, g.d, x.x) */
    MMX_K_REDUCE = 6,     /* fold of the partials + line-search controller (+ direction coefficients) */
    MMX_K_CHB = 7,        /* chromosomal blocks: all pairs inside each chromosome */
    MMX_K_DD_LISTS = 101, /* mmx_time_kernel only, decomposed handles: what a rank computes per evaluation for its halo besides the force
                             kernels -- occupancy map, dilation, send lists, message pack and unpack -- on the maps and messages
                             last received, no collective */
    MMX_K_FORCES = 100,   /* mmx_time_kernel only: every launch of ONE force evaluation as the minimizer enqueues them (pack,
                             cell build with the bonded pass riding in the scan launch, pair kernel, unsort), no fold */
    MMX_N_KERNELS = 8
};

/* mmx_set_compartments() modes */
#define MMX_COMP_COB 0 /* E[2] = {Ea, Eb}:            model.py:246-253 */
#define MMX_COMP_SCB 1 /* E[4] = {Ea1, Ea2, Eb1, Eb2}: model.py:322-333 */

/* mmx_stats.status */
#define MMX_MIN_CONVERGED 0
#define MMX_MIN_MAX_ITERATIONS 1
#define MMX_MIN_LS_INCREASE_GRADIENT (-2) /* liblbfgs LBFGSERR_INCREASEGRADIENT */
#define MMX_MIN_LS_MIN_STEP (-3)
#define MMX_MIN_LS_MAX_STEP (-4)
#define MMX_MIN_LS_MAX_LINESEARCH (-5)
#define MMX_MIN_NAN (-6)
#define MMX_MIN_KERNEL (-7) /* a force kernel reported that it could not do its work (e.g. the half-shell pair kernel's
                               work-unit protocol timed out): the evaluation was void, nothing was decided on it;
                               mmx_minimize / mmx_compute / mmx_md_step return MMX_ERR_STATE */

typedef struct mmx_handle_s *mmx_handle;

typedef struct {
    int32_t iterations;  /* accepted L-BFGS iterations */
    int32_t evaluations; /* energy+force evaluations */
    int32_t status;      /* MMX_MIN_* */
    int32_t n_beads;
    double e_initial, e_final;
    double gnorm_final; /* ||grad||_2 over 3N */
    double xnorm_final; /* max(1, ||x||_2) */
    double rms_force;   /* gnorm / sqrt(N), kJ/mol/nm */
    double seconds;     /* wall time of the call */
    double energy_terms[MMX_N_TERMS]; /* at the final point */
    /* Live per-kernel HIP-event timing (only filled when option "profile" > 0). */
    double kernel_ns[MMX_N_KERNELS];     /* summed event-measured duration of sampled launches */
    int64_t kernel_samples[MMX_N_KERNELS]; /* number of sampled launches behind kernel_ns */
    int64_t kernel_launches[MMX_N_KERNELS];/* launches issued during the call */
} mmx_stats;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* Replaces: Platform.getPlatformByName + Simulation(...) context creation, model.py:863-876. */
int mmx_create(int32_t n_beads, int32_t device_id, mmx_handle *out);
int mmx_destroy(mmx_handle h);
/* Message of the last error on h (h == NULL: last error of a failed mmx_create on this thread). */
const char *mmx_last_error(mmx_handle h);
/* Library/ABI version, for loaders. */
int mmx_abi_version(void);

/* ---- multi-GPU (BASELINE config 5; no counterpart in the reference, which is single-device) ----
 * One process and one handle per GPU.  The beads are cut into SEGMENTS of 62 consecutive beads; a rank owns a set of
 * segments: forces, energies and the L-BFGS state of their beads live on its GPU, in ascending bead order (the "local
 * order": what mmx_compute's forces_out and mmx_dd_owned_beads use).  At creation rank r owns the contiguous bead range
 * [r*slice, min(N, (r+1)*slice)), slice = 62 * ceil(ceil(N/62) / world) -- the Hilbert start makes index ranges
 * compact bricks.  While a minimization deforms the structure, mmx_minimize RE-ASSIGNS the segments to the ranks by
 * recursive bisection of their centroids (every rank a compact region of space again; the L-BFGS vectors of a segment
 * migrate with it, so the optimisation itself is unaffected; option "dd_spatial", default 1).  Ownership only changes
 * inside mmx_minimize; ask mmx_dd_owned_beads after it.  Every setter still takes the arrays of the WHOLE system.
 * Per evaluation the
 * ranks exchange ghost beads (what lies within the cutoff of a peer's owned beads, by the peers' need-maps; backbone
 * neighbours and loop partners across segment ends) with ncclSend/ncclRecv and all-reduce once: 59 doubles in the
 * minimizer (energies, the three Gram rows an accepted step would change, g.d, x.x, two flags), 16 in a plain
 * evaluation -- RCCL, issued on the handle's stream, no host round trip (csrc/mmx_dd.hpp).  Without a communicator a
 * multi-rank handle still evaluates its owned beads against the positions last set by the host (unit tests). */
/* (MMX_ERR_BAD_ARG when some rank would own no bead at creation: (world - 1) * slice must be below N.) */
int mmx_create_dd(int32_t n_beads, int32_t rank, int32_t world, int32_t device_id, mmx_handle *out);
/* n_own: beads this handle owns now; own_lo: its first owned bead (the whole range while the ownership is the initial one). */
int mmx_dd_info(mmx_handle h, int32_t *own_lo, int32_t *n_own, int32_t *rank, int32_t *world);
/* bead_ids[n_own]: the owned beads in local order (ascending).  Single-domain handles: 0 .. N-1. */
int mmx_dd_owned_beads(mmx_handle h, int32_t *bead_ids);
/* rank 0: 128-byte ncclUniqueId to hand to every rank (e.g. by torch.distributed broadcast). */
int mmx_comm_unique_id(uint8_t *id128);
/* every rank: ncclCommInitRank on the handle's device with the handle's rank/world (collective). */
int mmx_comm_init(mmx_handle h, const uint8_t *id128);
/* Test/rehearsal communicator: the `world` handles are ranks 0..world-1 of one system created in ONE process on
 * ONE device; afterwards every call that issues collectives (compute, minimize, md_step) must be made for all
 * ranks concurrently, one host thread per handle.  Collectives are host barriers + HIP events + device copies,
 * reduced in rank order.  Exercises the multi-rank control flow on a single GPU; production runs use RCCL. */
int mmx_comm_init_local(mmx_handle *handles, int32_t world);

/* ---- system description (replaces the OpenMM Force-object construction in model.py) ----------- */
/* context.setPositions, model.py:877.  xyz_nm is [N,3] row-major (always the whole system). */
int mmx_set_positions(mmx_handle h, const float *xyz_nm);
/* context.getState(getPositions=True).getPositions(), model.py:889-892. */
int mmx_get_positions(mmx_handle h, float *xyz_nm);
/* Per-particle parameter "s" of the compartment/lamina forces, model.py:236-239, 485.  s in {-2..2}. */
int mmx_set_labels(mmx_handle h, const int8_t *s);
/* HarmonicBondForce + HarmonicAngleForce over the backbone with the reference's chr_ends index
 * quirks (bond (i,i+1) dropped when i in chr_ends; angle (i,i+1,i+2) dropped when i in chr_ends or
 * chr_ends-1): model.py:625-636, 708-720. */
int mmx_set_backbone(mmx_handle h, const int32_t *chr_ends, int32_t n_ends, float bond_r0, float bond_k,
                     float angle_theta0, float angle_k, int32_t use_bond, int32_t use_angle);
/* Same, with explicit per-bead masks: bit0 = bond (i,i+1) present, bit1 = angle (i,i+1,i+2) present. */
int mmx_set_backbone_masks(mmx_handle h, const uint8_t *flags, float bond_r0, float bond_k, float angle_theta0,
                           float angle_k, int32_t use_bond, int32_t use_angle);
/* Loop HarmonicBondForce: bonds (m[l], n[l]) with rest length r0[l], constant k.  model.py:651-659. */
int mmx_set_loops(mmx_handle h, const int32_t *m, const int32_t *n, const float *r0, int32_t n_loops, float k_loop);
/* CustomNonbondedForce "epsilon*(sigma/(r+r_small))^EV_POWER", model.py:181-201.
 * cutoff_nm <= 0: NoCutoff (what the reference does); > 0: plain truncation (CutoffNonPeriodic). */
int mmx_set_excluded_volume(mmx_handle h, float eps, float sigma, float r_small, float power, float cutoff_nm);
/* CustomNonbondedForce "-E*exp(-r^2/(2*rc^2))" with the label-pair amplitude table of the mode.
 * May be called once per mode; the amplitudes of COB and SCB add (both use rc = r_comp). */
int mmx_set_compartments(mmx_handle h, int32_t mode, const float *E, float rc, float cutoff_nm);
/* CustomExternalForce "C*(max(0,r-R2)^2+max(0,R1-r)^2)", model.py:453-466. */
int mmx_set_container(mmx_handle h, float C, float R1, float R2, const float centre[3]);
/* CustomExternalForce "B*(sin(pi*(r-R1)/(R2-R1))^8-1)*(delta(s+1)+delta(s+2))", model.py:499-507. */
int mmx_set_lamina(mmx_handle h, float B, float R1, float R2, const float centre[3]);
/* CustomExternalForce "G*chrom_s*(r-R1)^2", model.py:579-586; w is [N] chrom_strength. */
int mmx_set_central(mmx_handle h, float G, float R1, const float centre[3], const float *w);
/* CustomNonbondedForce "E*(k_C*r^4 - r^3 + r^2); E = dE*delta(chrom1-chrom2)", model.py:397-419
 * (polynomial form).  chrom is [N] chrom_spin: two beads interact iff their values are equal; beads of
 * one chromosome must be contiguous (they are: model.py:158-162 assigns by chr_ends ranges).  The
 * potential grows with r, so there is no cutoff: every pair inside a chromosome is evaluated. */
int mmx_set_chromosomal_blocks(mmx_handle h, float k_C, float dE, const int32_t *chrom);
/* ---- alternative functional forms (the *_FORCE_TYPE ini keys, config.py:269-312) -----
 * form 0 is the default the mmx_set_* call of the term documents; the others follow model.py:
 *   MMX_SEL_EV       1 gaussian_core   eps*exp(-r^2/(2 sigma^2))                                  :205-209
 *   MMX_SEL_COB/SCB  1 yukawa          -E*exp(-r/r_comp)/r          2 theta  -E*step(r_comp - r)    :262-288, 340-377
 *   MMX_SEL_CHB      1 gaussian        -dE*exp(-k_C r^2)            2 saturating  -dE/(1+k_C r^2)   :424-443
 *   MMX_SEL_LAMINA   1 gaussian_shell  2 harmonic_shell  3 logistic_shell                          :508-539
 *   MMX_SEL_CENTRAL  1 gaussian        2 logistic                                                  :588-612
 *   MMX_SEL_LOOPS    1 fene_soft       2 gaussian_tether                                           :662-701
 * The COB yukawa expression of the reference reads the label of particle 1 twice (model.py:266-267), i.e. it is
 * not symmetric in the pair; the lower bead index is taken as particle 1 (OpenMM Reference platform order). */
#define MMX_SEL_EV 0
#define MMX_SEL_COB 1
#define MMX_SEL_SCB 2
#define MMX_SEL_CHB 3
#define MMX_SEL_LAMINA 4
#define MMX_SEL_CENTRAL 5
#define MMX_SEL_LOOPS 6
#define MMX_N_SELECTORS 7
int mmx_set_functional_form(mmx_handle h, int32_t selector, int32_t form);

/* Removes a term again (term = MMX_T_*). */
int mmx_disable_term(mmx_handle h, int32_t term);

/* ---- tunables that are not part of the physics ------------------------------------------------
 * key                 meaning                                                      default
 * "deterministic"     0: systems of >= 100 000 beads use the half-shell pair kernel (every pair once, reaction
 *                     through LDS and float atomics: results reproducible to rounding, as OpenMM's GPU
 *                     platforms with DeterministicForces=false); 1: always the full-shell pair kernel with
 *                     a fixed summation order (bitwise reproducible runs, ~5 % slower at 200 000 beads)   0
 * "profile"           k>0: HIP-event time the kernel slots of every k-th evaluation of a
 *                     minimization (every k-th launch of a slot elsewhere)            0
 * "profile_nb"        k>0 (with "profile" > 0): also time the pair-kernel slot alone in every k-th
 *                     evaluation of a minimization (one event pair: more samples of the dominant kernel) 0
 * "use_graph"         1: replay the minimizer's trial evaluations from a hipGraph ("graph_evals" of them
 *                     per graph, even) instead of launching them one by one; same bits; slower on
 *                     ROCm 7.2 at every size measured (DESIGN_HISTORY.md 5b), kept for A/B     0
 * "dd_halo"           decomposed runs with a communicator: 1 = ghost-bead halo exchange (ghosts chosen by the
 *                     peers' need-maps -- coarse-cell occupancy grown by the cutoff --, ncclSend/ncclRecv of the
 *                     listed beads per evaluation); 0 = all-gather of every position per evaluation (round-1 path,
 *                     A/B; also what a run with chromosomal blocks uses: that term has no cutoff)  1
 * "dd_adaptive"       1: the polls of mmx_minimize choose how many evaluations K (1 .. "dd_rebuild_every", default 4) a set of
 *                     ghost lists serves, from the largest trial move any rank saw since the last poll (all-reduced);
 *                     K = 1 -- exact lists, no skin -- while the structure collapses; MD steps always get K = 1     1
 * "dd_rebuild_every"  K: the ghost lists are rebuilt, on the stream, before every K-th evaluation; setting it fixes the
 *                     lifetime (switches "dd_adaptive" off).  1: before each one -- exact lists, no skin, one 32 KB
 *                     all-gather more per evaluation; K > 1: the lists reach cutoff + dd_skin and hold while no bead has
 *                     moved more than dd_skin / 2 (checked by the pack; a violation voids the evaluation, which is
 *                     repeated with fresh lists)                                                             4
 * "dd_skin"           nm, K > 1 only; doubled (up to 0.8) for the rest of a call whenever a list went stale   0.15
 * "dd_lists_serve", "dd_move_seen"   (get only) K in force; largest trial move (nm) the last poll read back
 * "dd_overlap"        1: half-shell kernel on decomposed ranks -- list kernels, need-map all-gather, message pack, send / recv
 *                     and ghost count go to a second stream of the handle, beside the owned beads' share of the cell build;
 *                     the ghosts' clusters, the work items and the bonded pass follow in a second launch.  Same cluster
 *                     list.  Off by default: on one MI355X the fork / join of the two streams and the second launch cost
 *                     more than the 25 us they hide (profiles/r05/dd_overlap_ab.txt).  3: the halo's
 *                     launches are enqueued before the owned build (A/B); "dd_overlap_go": workgroups of the owned
 *                     build's launch (A/B); "dd_overlapped" (get only): evaluations that ran this way               0
 * "dd_spatial"        1: mmx_minimize re-assigns the 62-bead segments to the ranks by recursive bisection of their
 *                     centroids while the structure deforms (first attempt after "dd_reassign_first" evaluations, the
 *                     interval doubling up to "dd_reassign_max"; an attempt that would move < 2 % of the segments
 *                     changes nothing); 0: ownership stays the initial index ranges                         1
 * "dd_reassign_first", "dd_reassign_max"   see "dd_spatial"                                                48, 768
 * "dd_reassignments", "dd_reassign_attempts", "dd_segments_moved"   (get only) statistics of it
 * "cell_reuse"        1: single-domain minimizations keep the cell structure of a full build -- membership, cluster composition,
 *                     work items -- for up to 16 evaluations once the structure has thinned out (the grid is then 1.3 x
 *                     (half-shell kernel) / 1.45 x (full-shell) the cutoff wide: "cell_reuse_factor"); in between only
 *                     cluster positions and boxes are refreshed.  Exact while no bead has moved more than half the skin from
 *                     where it was binned: k_pack checks every evaluation, a violation voids the evaluation, which is
 *                     repeated after a full build; how long a structure serves follows the displacements read back at the
 *                     polls.  Iterations 1000-2000: chr1_50k +15 %, gw_200k +4.5 % iterations/s; the first few hundred
 *                     iterations from the lattice are unaffected (no skin there).  0: a full build per evaluation      1
 * "cell_builds", "cell_reuses", "cell_stale_halts", "cell_reuse_K"   (get only) statistics of it
 * "cell_slots"        1: the trial moves of a single-domain minimization write their 64-bit sort keys straight into per-cell
 *                     rows of a slot table (ONE allocation of 256 B per bead, cut at every poll into rows of twice the
 *                     fullest cell), so the counting sort's fill launch disappears; same keys, same clusters, bitwise the
 *                     same minimization (+1-2 % iterations/s).  A cell that outgrows its row voids the evaluation, which
 *                     is repeated with longer rows ("cell_slot_halts"; the repeat bins on another grid: from there on
 *                     the run differs from the fill path by rounding); 0: k_cell_fill after the scan               1
 * "fused_tail"        1: what follows the pair kernel of a minimizer's evaluation is ONE launch (k_tail): the half-shell
 *                     kernel's forces leave their cluster slots through a bead -> slot table, the history pass runs with the
 *                     four column groups of k_history sharing x / xp / gp / g through LDS (same summation order: bitwise the
 *                     same partials), and the workgroup with the highest index folds the others' TAGGED partials -- it polls
 *                     the values themselves -- and decides (line search, direction coefficients); decomposed ranks: it
 *                     forms the all-reduce's input.  0: k_nb_n3_unsort, k_history, k_decide as separate launches (A/B)   1
 * "fused_build"       1: the cell build of a trial move is ONE launch behind the pack (k_build_direct; decomposed ranks:
 *                     k_dd_unpack_count + k_build_direct_dd): the pack keeps per-row cluster totals, every workgroup finds its
 *                     offsets by itself (no scan stage), cells of <= 256 beads are sorted by one wave in registers, the bonded
 *                     pass and the half-shell kernel's work items ride along.  Same clusters in the same order as the
 *                     scan-based build (bitwise the same `deterministic` runs).  Needs "cell_slots"; grids beyond 64 cells per
 *                     row or 2048 (decomposed: 1024) rows void one evaluation and fall back.  0: the scan-based build (A/B)   1
 * "key32"             1: the direct build of systems of <= 2^20 beads sorts 32-bit keys (12-bit Hilbert index << 20 | bead: the same
 *                     order as the 64-bit ones, half the shuffles); 0: 64-bit keys (A/B)                                 1
 * "direct_builds"     (get only) full builds enqueued through it
 * "cell_edge_auto"    1: once a poll finds fewer than 32 beads per cutoff-sized grid cell (systems of >= 20 000 beads) the grid
 *                     switches to cells 1.12 x wider (same results: the box tests are exact; the in-cell ordering is a
 *                     latency chain per cell, fewer and fuller cells take 8-10 us off the cell build for +2 us of pair
 *                     kernel at 200 000 beads: profiles/r04_cell_edge_cost.txt; chr1_50k +7 %, gw_200k +1.5 % iterations/s
 *                     from 1 000 iterations on); 0: cells of edge cutoff throughout                              1
 * "cell_edge_scale"   measurement only: grid cells of edge scale x cutoff (>= 1).  Results are unchanged (the box tests
 *                     stay exact); the pair kernels see more candidates -- what cells of edge cutoff + skin, the price of
 *                     keeping the cell structure over several evaluations, would cost them (scripts/cell_edge_cost.py)   1
 * "md_step"           (get only) MD steps integrated so far; after an MMX_ERR_STATE of mmx_md_step on a decomposed run (a ghost
 *                     list went out of date: the steps since the last poll were taken back) it says where to go on from
 * "dd_us_needmap_allgather", "dd_us_halo_exchange", "dd_us_allreduce", "dd_collective_samples"
 *                     (get only; with "profile" > 0) mean HIP-event time in us of the three collectives of an evaluation
 *                     -- need-map all-gather, grouped send/recv of the halo, the 59-double all-reduce -- over the
 *                     sampled evaluations of mmx_minimize, each from the end of the work before it to its own end on
 *                     this rank's stream (transfer + waiting for the peers); they lie INSIDE the "cell_build" /
 *                     "reduce" brackets of mmx_stats.kernel_ns
 * "dd_ghosts", "dd_ghost_slots", "dd_exchanges", "dd_bytes_sent", "dd_redecompositions", "dd_sync_rebuilds",
 * "dd_halts", "dd_capacity_updates", "dd_skin_now"
 *                     (get only) statistics of the decomposed run: ghosts listed for this rank / slots of its
 *                     incoming messages (capacity) at the last poll; halo exchanges; bytes this rank put on the
 *                     wire in them; list rebuilds (all / host-synchronous); evaluations voided and repeated;
 *                     polls that resized a message
 * nb_variant bits:    4096 force the half-shell pair kernel, 8192 force the full-shell one (default: chosen by
 *                     system size and cell occupancy, DESIGN_HISTORY.md 5c); bits 24-30 configure the tail shares of the
 *                     half-shell kernel (A/B); the other bits select round-1 A/B and diagnosis instances
 * "poll_interval"     evaluations enqueued between host polls of the device state     32
 * "nb_variant"        non-bonded kernel variant (0 = default)                         0
 * "fused_bonded"      1: backbone + loops + confinement in one pass; 0: the three kernels
 *                     separately (per-kernel timing)                                  1
 * "overlap_bonded"    1: that one pass rides in the launch of the cell scan (cell-list mode; booked
 *                     in the cell-build slot); 0: its own launch ("confine" slot)     1
 * "order_fallbacks"   (get only) cells of the last call that were too large for the in-LDS sort and
 *                     kept arrival order: 0 means the summation order was bitwise reproducible
 * "dd_freeze"         measurement only (scripts/dd_projection.py): 1 = a rank of a decomposed run issues no collective any
 *                     more -- ghost lists and the ghost positions last received stay, sums are not all-reduced -- so
 *                     that mmx_time_kernel can time ONE rank's kernels on exactly the beads it holds in the run   0
 * "n3_long_items"     work items of the half-shell pair kernel: -1 = by size (24 clusters from 150 000 local beads,
 *                     16 below), 0 / 1 = short / long forced (tests, A/B), 2 = 32 clusters (measurement)          -1
 * "n3_pass_records"   1: a run of i-clusters whose candidates exceed the LDS window is emitted as one record of the item list
 *                     per window pass (equal slices; any workgroup takes them); 0: one record per run, the workgroup that
 *                     takes it walks the passes (A/B, tests)                                                      1
 * "n3_slice_cap"      measurement: longest slice of such a record in clusters (0 = the LDS window, 424)            0
 * "dd_split"          decomposed ranks on the half-shell kernel: 1 = the ghosts' clusters in a region of their own behind the
 *                     owned ones, never i-clusters (owned clusters sweep the ghost clusters of their full 3 x 3 x 3
 *                     neighbourhood); 0 = ghost clusters interleaved per cell and taking part as i-clusters (A/B)    1
 * "n_clusters", "n_cells", "n3_items"  (get only) 8-bead clusters, grid cells and half-shell work items of the last
 *                     cell build, as of the last poll
 * "inject_fault"      tests only: bit 0 makes every wait of the half-shell pair kernel's unit protocol time out at
 *                     once, bit 1 shrinks its work-item list to one entry -- both must surface as MMX_ERR_STATE;
 *                     bit 2 sizes the halo messages of a decomposed run without slack, so that any growth of a
 *                     ghost list exercises the halt-and-repeat protocol; bit 3 gives a kept cell structure ("cell_reuse")
 *                     a skin of nothing, so that every evaluation on one is voided and repeated after a full build; bit 4 cuts the
 *                     slot table ("cell_slots") in rows of 64 slots at every poll: crowded cells overflow, halt, repeat;
 *                     bit 5 makes k_tail's folding workgroup give up before its first poll (MMX_ERR_STATE); bit 6 makes the
 *                     direct build find its grid too large (one void evaluation, then the scan-based build)          0
 * "slot_cap", "slot_cells", "cell_edge"   (get only) rows of the slot table as cut at the last poll; cell edge of the last build
 */
int mmx_set_option(mmx_handle h, const char *key, double value);
int mmx_get_option(mmx_handle h, const char *key, double *value);

/* ---- compute (parity hook) --------------------------------------------------------------------
 * One evaluation of all enabled terms at the current positions: context.getState(getEnergy=True,
 * getForces=True).  forces_out is [N,3] (may be NULL; multi-GPU: [n_own,3], the owned beads),
 * energy_terms_out is [MMX_N_TERMS] (multi-GPU: all-reduced when a communicator exists, else this
 * rank's share). */
int mmx_compute(mmx_handle h, float *forces_out, double *energy_terms_out);

/* ---- minimize ---------------------------------------------------------------------------------
 * simulation.minimizeEnergy(), model.py:886 == OpenMM LocalEnergyMinimizer.minimize(context,
 * tolerance = 10 kJ/mol/nm, maxIterations = 0): liblbfgs L-BFGS (m = 6, backtracking strong-Wolfe
 * line search).  Runs entirely on the device; blocks until finished.  Positions are updated in
 * place (read them with mmx_get_positions).  max_iters == 0: until converged. */
int mmx_minimize(mmx_handle h, double tolerance, int32_t max_iters, mmx_stats *out);

/* ---- molecular dynamics -----------------------------------------------------------
 * Replaces the integrator objects of model.py:768-808 and `simulation.step(n)` of run_md(), model.py:907-995,
 * on the same force kernels.  Update rules are OpenMM's leap-frog integrators without constraints:
 *   MMX_INT_LANGEVIN  mm.LangevinIntegrator(T, friction, dt)   (the default, config.py:258-266)
 *   MMX_INT_VERLET    mm.VerletIntegrator(dt)
 *   MMX_INT_BROWNIAN  mm.BrownianIntegrator(T, friction, dt)
 *   MMX_INT_AMD       mm.amd.AMDIntegrator(dt, alpha, E), model.py:794-800: accelerated MD, a leap-frog step
 *                     on the boosted force f' = f (alpha / (alpha + E - U))^2 while the potential energy U of the
 *                     current positions is below E, f' = f otherwise; alpha and E from mmx_md_set_amd
 *                     (temperature and friction of mmx_md_configure are ignored)
 * (the variable-step integrators -- which the reference itself cannot construct, model.py:776,791 read a missing
 * attribute -- are not provided: MMX_ERR_BAD_ARG).
 * Units: ps, K, 1/ps, amu (ff.xml:5: 16427.889 for every bead); velocities nm/ps.  Noise comes from
 * Philox4x32-10 keyed by `seed` and indexed by (bead, step): independent of the launch geometry and of the
 * decomposition over GPUs (it is not OpenMM's generator: trajectories agree in distribution only). */
#define MMX_INT_LANGEVIN 0
#define MMX_INT_VERLET 1
#define MMX_INT_BROWNIAN 2
#define MMX_INT_AMD 3

typedef struct {
    int64_t step_count;  /* steps integrated since mmx_md_configure (State.getStepCount(), model.py:939) */
    int32_t n_steps;     /* steps of this call */
    int32_t integrator;
    double potential;    /* kJ/mol at the final positions (State.getPotentialEnergy(), model.py:943) */
    double kinetic;      /* kJ/mol; leap-frog velocities shifted by dt/2 as OpenMM reports them (model.py:944);
                            brownian and aMD (a CustomIntegrator: plain m v^2 / 2): unshifted */
    double temperature;  /* 2 K / (3 N kB) in kelvin (the fallback formula of model.py:966-970) */
    double seconds;      /* wall time of the call */
    double energy_terms[MMX_N_TERMS];
} mmx_md_stats;

int mmx_md_configure(mmx_handle h, int32_t integrator, double dt_ps, double temperature_K, double friction_per_ps,
                     double mass_amu, uint64_t seed);
/* Boost parameters of MMX_INT_AMD in kJ/mol: SIM_AMD_ALPHA, SIM_AMD_E (config.py:255-256: 100, 1000), the
 * alpha and E of mm.amd.AMDIntegrator(dt, alpha, E), model.py:796-800.  alpha > 0. */
int mmx_md_set_amd(mmx_handle h, double alpha_kj_per_mol, double e_kj_per_mol);
/* context.setVelocitiesToTemperature(T, seed), model.py:878: v = sqrt(kB T / m) N(0,1) per component. */
int mmx_md_set_velocities_to_temperature(mmx_handle h, double temperature_K, uint64_t seed);
/* [N,3] nm/ps, whole system (multi-GPU: the owned rows are used / filled, other rows are left untouched). */
int mmx_set_velocities(mmx_handle h, const float *v_nm_per_ps);
int mmx_get_velocities(mmx_handle h, float *v_nm_per_ps);
/* simulation.step(n_steps) followed by getState(getEnergy=True): n force evaluations (+1 when positions or
 * parameters changed since the last step), all enqueued on the device without host round trips. */
int mmx_md_step(mmx_handle h, int32_t n_steps, mmx_md_stats *out);

/* ---- measurement ------------------------------------------------------------------------------
 * Launches kernel slot `kernel` (MMX_K_*) `reps` times back to back at the current positions on the
 * handle's stream between two HIP events and returns the mean duration per launch and the
 * algorithmic bytes one launch moves (DESIGN.md "Kernels").  Results of the launches are discarded.
 * Slots 0..4 and MMX_K_FORCES; slots 2..4 as standalone launches (inside an evaluation they ride in the cell scan's launch). */
int mmx_time_kernel(mmx_handle h, int32_t kernel, int32_t reps, double *mean_us, double *algorithmic_bytes);
/* Diagnostics of the last cell build: n_cells, max beads per cell, cell edge (nm), pair-candidate
 * count (bead x stencil occupancy) and pairs inside the cutoff; any pointer may be NULL. */
int mmx_nb_census(mmx_handle h, int64_t *n_cells, int32_t *max_per_cell, double *cell_edge, double *pair_candidates,
                  double *pairs_within_cutoff);

/* Diagnostics of the cluster-pair kernel at the current positions: number of 8-bead clusters, number of
 * (i-cluster, j-cluster) tiles in the 27-cell stencils, number of tiles surviving the box-box cull and
 * number of j beads surviving the per-bead cull (each is swept against the 8 beads of its i-cluster). */
int mmx_cluster_census(mmx_handle h, int64_t *n_clusters, double *tiles_candidate, double *tiles_accepted,
                       double *beads_swept);

#ifdef __cplusplus
}
#endif
#endif /* MMX_H */
