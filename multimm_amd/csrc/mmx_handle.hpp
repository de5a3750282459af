// mmx_handle.hpp -- the engine handle behind `mmx_handle` (device buffers, options, profiling state), the RCCL entry
// points resolved at run time, and the in-process loopback communicator.  Host code of libmmx.so; included by
// mmx_api.hip only (one translation unit).
#pragma once
#include "../../include/mmx.h"
#include "mmx_bonded.hpp"
#include "mmx_build.hpp"
#include "mmx_cells.hpp"
#include "mmx_common.hpp"
#include "mmx_dd.hpp"
#include "mmx_lbfgs.hpp"
#include "mmx_md.hpp"
#include "mmx_nonbonded.hpp"
#include "mmx_nonbonded_n3.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h> // types only: the symbols are resolved with dlopen/dlsym when a communicator is requested

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>


using namespace mmx;

namespace {

constexpr int kChunk = 64;
constexpr int kMaxEvents = 8192;
thread_local std::string g_create_error;

struct EventPair {
    hipEvent_t a, b;
    int slot;
};

} // namespace

// RCCL entry points, loaded lazily (single-GPU users never need librccl.so)
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr; // optional
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};
static RcclApi g_rccl;
static bool load_rccl(std::string &err) {
    static std::mutex mu; // handles may be driven from several threads (one per rank in loopback runs)
    std::lock_guard<std::mutex> lock(mu);
    if (g_rccl.lib) return true;
    // the soname first: a process that already carries RCCL (PyTorch bundles its own copy) gets that very copy back,
    // not a second instance from /opt/rocm
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        err = std::string("cannot load librccl.so: ") + dlerror();
        return false;
    }
#define RSYM(field, name)                                                                                  \
    g_rccl.field = (decltype(g_rccl.field))dlsym(lib, name);                                               \
    if (!g_rccl.field) {                                                                                   \
        err = std::string("librccl.so lacks ") + name;                                                     \
        return false;                                                                                      \
    }
    RSYM(GetUniqueId, "ncclGetUniqueId");
    RSYM(CommInitRank, "ncclCommInitRank");
    RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(AllGather, "ncclAllGather");
    RSYM(AllReduce, "ncclAllReduce");
    RSYM(GetErrorString, "ncclGetErrorString");
    RSYM(Send, "ncclSend");
    RSYM(Recv, "ncclRecv");
    RSYM(GroupStart, "ncclGroupStart");
    RSYM(GroupEnd, "ncclGroupEnd");
#undef RSYM
    g_rccl.CommGetAsyncError = (decltype(g_rccl.CommGetAsyncError))dlsym(lib, "ncclCommGetAsyncError");
    g_rccl.lib = lib;
    return true;
}

// In-process loopback communicator (mmx_comm_init_local): the ranks of a decomposed system are handles of ONE
// process on ONE device, each driven by its own host thread.  Collectives = host barrier + HIP events +
// device-to-device copies, summed in rank order.  It exists so that the multi-rank control flow (slices, ghosts,
// identical decisions on every rank) can be executed and tested on a single GPU; production runs use RCCL.
struct mmx_handle_s;
struct LocalComm {
    int world = 0;
    std::vector<mmx_handle_s *> h;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    bool broken = false;
    std::vector<hipEvent_t> ready, done; // [rank*2 + parity]
    std::vector<double *> mailbox;       // [rank*2 + parity] -> 64 doubles on the device
    ~LocalComm() {
        for (auto e : ready) (void)hipEventDestroy(e);
        for (auto e : done) (void)hipEventDestroy(e);
        for (auto m : mailbox) (void)hipFree(m);
    }
    // Host barrier over the driving threads; false when a rank gave up (timeout / error) so that nobody hangs.
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const unsigned long long gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return generation != gen || broken; })) broken = true;
        if (broken) cv.notify_all();
        return !broken;
    }
};

struct mmx_handle_s {
    int n = 0, n4 = 0, device = 0;
    // domain decomposition (single GPU: rank 0 of 1, owns every bead)
    int rank = 0, world = 1, slice = 0; // slice = beads per rank (n padded to world * slice = n_all)
    int n_all = 0, own_lo = 0, n_own = 0;
    // Ownership in segments of kSeg beads (Own, mmx_common.hpp).  At creation rank r owns the contiguous segments
    // [r * seg_per, (r + 1) * seg_per) (slice = seg_per * kSeg beads; the tables are empty / null: "identity").  During a
    // minimization with a halo the segments are re-assigned by recursive bisection of their centroids (dd_reassign).
    int seg_per = 0;              // segments per rank of the initial assignment = capacity of a rank's vectors in segments
    int nseg = 0;                 // segments of the padded system: world * seg_per
    std::vector<int> seg_owner;   // [nseg] owner of every segment (identical on all ranks; empty: identity); segments past the last bead: -1
    std::vector<int> seg_lidx;    // [nseg] local segment index at its owner
    std::vector<int> my_segs;     // owned segments, ascending
    int *d_seg_own = nullptr, *d_seg_local = nullptr; // device tables of this rank (in use once seg_owner is set: own_of)
    int dd_spatial = 1;           // option: re-assign segments while minimizing (0: ownership stays the initial slices)
    int dd_reassign_first = 48;   // evaluations of a minimization before the first attempt; the interval doubles up to ...
    int dd_reassign_max = 768;
    long long dd_reassignments = 0, dd_reassign_attempts = 0, dd_segments_moved = 0; // statistics
    float *mig = nullptr;         // staging of a migration: [world][3 * slice] floats (allocated at the first one)
    float4 *seg_cent = nullptr;   // [world][seg_per] centroids {x, y, z, beads} of the owned segments, all-gathered
    float4 *seg_cent_host = nullptr; // pinned copy
    int *d_mig_src = nullptr;     // [seg_per] where each owned segment of the new assignment sits in `mig`
    std::vector<int> loop_m, loop_n; // the loops as given (mmx_set_loops): the per-rank CSR is rebuilt after a re-assignment
    std::vector<float> loop_r0v;
    ncclComm_t comm = nullptr;
    std::shared_ptr<LocalComm> lcomm; // in-process loopback communicator (tests on one GPU)
    unsigned long long coll_seq = 0;  // collectives issued so far (parity selects the event / mailbox set)
    bool coll_failed = false;
    std::string coll_error;           // what failed (RCCL error string); empty: loopback time-out
    double **lbox[2] = {nullptr, nullptr}; // device arrays [world] of the ranks' mailboxes, per parity
    // ghost-bead halo of a decomposed run with a communicator (mmx_dd.hpp); dd_halo = 0 keeps the all-gather of every position
    int dd_halo = 1;
    int dd_every = 4;                           // ghost lists are rebuilt (on the stream) before every dd_every-th evaluation at most
                                                // (dd_adaptive: the polls choose 1 .. dd_every; 1 = lists exact for the positions they serve, no skin)
    int dd_adaptive = 1;                        // option (default on; setting dd_rebuild_every fixes the lifetime and switches it off): the polls choose how many evaluations (1 .. dd_every) the ghost lists serve (dd_K)
    int dd_k_cur = 1;
    double dd_move_seen = 0.0;                  // largest trial move (nm) the last poll read back
    int dd_since = 0;                           // evaluations enqueued since the last rebuild
    float dd_skin = 0.15f;                      // nm (lists kept over more than one evaluation): the lists hold while no bead has moved more than half of it
    float dd_skin_cur = 0.15f;                  // ... as adapted by the minimizer: doubled (up to 0.8) when a list went stale
    float *dd_boxes = nullptr;                  // [world][6] owned bounding boxes (all-gathered at a synchronous rebuild)
    DDGrid *dd_grid = nullptr;                  // coarse grid of the need-maps (device)
    unsigned long long *dd_occ = nullptr;       // [kDDWords] coarse cells of the owned beads
    unsigned long long *dd_maps = nullptr;      // [world][kDDPayload] all-gathered need-maps + send-list lengths
    unsigned long long *dd_static = nullptr;    // [n_own] ranks that always need this bead (backbone / loop partners)
    std::vector<unsigned long long> dd_loop_mask; // host: the loop-partner part of it (mmx_set_loops)
    int *dd_send_ids = nullptr, *dd_send_cnt = nullptr, *dd_cntmat = nullptr, *dd_ghost_ids = nullptr;
    int *dd_cnt_host = nullptr;                 // pinned [world][world]: list lengths as of the last poll
    float4 *dd_sendbuf = nullptr, *dd_recvbuf = nullptr; // [world][slice]
    float *dd_xref = nullptr;                   // [3 n_own] owned positions when the lists were built (dd_every > 1)
    DDCaps dd_scap{}, dd_rcap{};                // entries per message to / from rank q (host-known: ncclSend/ncclRecv sizes)
    int dd_nghost = 0;                          // sum of dd_rcap: ghost slots binned per evaluation (padding included)
    bool dd_frozen = false;                     // measurement only (option dd_freeze): no collectives, ghosts as last received
    int dd_slack_div = 8;                       // a message has room for 1 / dd_slack_div more entries than its list had
    long long dd_halts = 0, dd_cap_updates = 0, dd_sync_rebuilds = 0; // statistics
    DDOffsets dd_off{};
    bool dd_lists_valid = false, dd_static_dirty = true;
    bool dd_occ_clean = false;                  // the occupancy words are zero (the last list rebuild left them so)
    bool dd_ref_in_pack = false;                // the pack of the evaluation being enqueued checks / records the lists' reference positions
    int dd_rc = 0;                              // first error of a re-decomposition inside a launch sequence
    long long dd_redecompositions = 0, dd_exchanges = 0, dd_bytes_sent = 0; // statistics (options dd_*)
    float *xg = nullptr;      // [3 * n_all] global positions as last set by the host (multi-GPU only)
    bool pos4_dirty = false;  // pos4 of non-owned beads must be refilled from xg before the next evaluation
    hipStream_t stream = nullptr;
    // decomposed ranks, halo beside the owned beads' share of the cell build (enqueue_build): the list kernels, the two collectives
    // of the halo and the ghost count run on stream2 between ev_pack (the pack is done) and ev_halo (the ghosts are in place)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    int dd_overlap = 0;                         // option (default OFF: measured slower; half-shell kernel's split cluster list only): 1 = the halo on stream2 beside the owned build
    int dd_overlap_go = 0;                      // option (A/B): workgroups of the owned build's launch while the halo runs beside it (0: what fills the device)
    long long dd_overlapped = 0;                // statistics: evaluations whose halo ran beside the owned build
    FFParams P{};
    bool have_pos = false;
    // vectors (float, padded to n4*4)
    float *x = nullptr, *xp = nullptr, *g = nullptr, *gp = nullptr, *d = nullptr, *S = nullptr, *Y = nullptr;
    float4 *pos4 = nullptr;
    int8_t *labels = nullptr;
    uint8_t *flags = nullptr;
    float *cf_w = nullptr;
    int *chrom_of = nullptr, *chrom_lo = nullptr, *chrom_hi = nullptr; // chromosomal blocks: id per bead, range per id
    // cells
    int *count_own = nullptr; // decomposed handles: owned beads per cell (a cell's ghosts form clusters of their own)
    int *cell_of = nullptr, *count = nullptr, *rank_in_cell = nullptr, *start = nullptr, *istart = nullptr,
        *perm = nullptr;
    int2 *items = nullptr;
    int *cstart = nullptr;                       // cluster offsets per cell
    unsigned long long *okeys = nullptr;         // sort keys in cell order (written by k_cell_fill)
    bool nb_force_plain = false;                 // scaled units unusable (extreme r_comp): unscaled, non-lean instance
    bool nb_skip_energy = false;                 // MD steps between reports: pair forces only (default instance)
    int *biglist = nullptr;                      // ids of the cells of > 64 beads (written by k_cell_scan)
    int last_max_per_cell = -1;                  // largest cell seen at the last poll (sizes the in-LDS sort)
    float4 *spos4 = nullptr, *cl_lo = nullptr, *cl_hi = nullptr; // padded cell-sorted positions, cluster boxes
    int last_clusters = -1;
    long long n3_launches = 0;                   // launches of the half-shell kernel since the handle was created (option n3_launches)
    long long gn3_launches = 0;                  // ... per replay of the captured graph
    bool n3_build = false;                       // the last enqueued cell build prepared the half-shell kernel's work items
    int last_ncells = -1;                        // cells of the grid at the last poll (picks the pair kernel, see use_n3)
    int *sbead = nullptr;                        // half-shell kernel: bead id per cluster slot (-1: padding), written with spos4
    float *fsort = nullptr;                      // half-shell kernel: force per cluster slot, SoA [3][fstride], zero between evaluations
    int fstride = 0;
    bool fsort_dirty = false;                    // a minimization ended abnormally: k_tail only clears the slots of beads the build
                                                 // binned, so fsort is cleared wholesale before the next call (prepare)
    int *slot_of = nullptr;                      // [owned beads, local order] cluster slot of the bead (emit_clusters): what k_tail gathers by
    unsigned tail_epoch = 0u;                    // tag of the last k_tail launch's partials (TailArgs::epoch)
    int fused_tail = 1;                          // option: unsort + history + decision in one launch (k_tail); 0: the separate kernels (A/B)
    int *dcount = nullptr;                       // direct build: two sets of cell populations [2][maxcells + 1] ...
    int *drows = nullptr;                        // ... and of row totals [2][4][kDirectMaxRows] (clusters, large cells, ghost clusters, cells of many ghosts), alternating with the build's parity
    int *dcount_g = nullptr;                     // decomposed ranks: the ghosts' populations [2][maxcells + 1]
    int direct_slots[4] = {0, 0, 0, 0};          // workgroups of k_build_direct<CAP, N3> resident at once (occupancy x CUs), per instance
    int direct_dd_slots[4] = {0, 0, 0, 0};       // ... of k_build_direct_dd<N3, PHASE>: [N3 ? 1 : 0] one launch, [2] / [3] the two launches of the overlapped build
    bool dset_dirty[2] = {false, false};         // the direct build's counter set of that parity holds the counts of an earlier build
                                                 // (a direct build leaves its own set behind and zeroes the other one; builds through
                                                 // the scan in between flip the parity without touching either)
    int key32 = 1;                               // option: 32-bit sort keys in the direct build (systems of <= 2^20 beads); 0: 64-bit (A/B)
    bool direct_key32 = false;                   // ... the build being enqueued uses them
    bool direct_ok = true;                       // this call's grids have fitted the direct build so far (nx <= 64, rows <= kDirectMaxRows)
    bool last_build_direct = false;              // the last enqueued full build was a direct one (the polls fetch its fullest cell: k_poll_stats)
    int last_direct_parity = 0;
    long long direct_builds = 0;                 // statistics
    int fused_build = 1;                         // option: scan + bonded pass + fill + order + work items in one launch (k_build); 0: separate (A/B)
    bool nb_lean = false;                        // the lean pair loop applies (default forms, one cutoff): refresh_params
    int n3_long_items = -1;                      // work items of k_nb_n3: -1 by size (kN3LongItemsFrom), 0 short (16 clusters), 1 long (24)
    int n3_cap = 0;                              // LDS force window of k_nb_n3 in clusters (0: not usable on this device)
    N3Item *n3_items = nullptr;                  // its work items (k_n3_items, after every cell scan)
    int n3_max_items = 0, n_cus = 0;
    GridParams *grid = nullptr;  // [2]: grid of this build / of the next one (ping-pong)
    GridParams *gcur = nullptr;  // grid the last enqueued build used (what the pair kernel reads)
    int build_idx = 0;
    bool grid_ready = false;     // grid[build_idx & 1] was computed by a build of this system as it is now (no setter since)
    float *bbox_part = nullptr;  // [6][ceil(n/256)] per-block bounding boxes of k_pack
    int maxcells = 262144;
    int max_items = 0;
    int last_items = -1;
    // reductions / state
    double *part = nullptr, *rows = nullptr;
    MinState *st = nullptr;      // device
    MinState *st_host = nullptr; // pinned
    // loops (CSR over beads carrying a loop end)
    int n_loops = 0, n_rows = 0;
    int *row_bead = nullptr, *row_start = nullptr, *partner = nullptr;
    int *lstart = nullptr; // [n_own + 1] loop entries per owned bead (same partner / r0 arrays; fused bonded kernel)
    float *loop_r0 = nullptr;
    // all-pairs scratch
    float4 *fpart = nullptr;
    float2 *epart = nullptr;
    int ap_slices = 0;
    // compartments
    float tab_cob[25]{}, tab_scb[25]{};
    bool has_cob = false, has_scb = false;
    float ev_cut = 0.f, g_cut = 0.f, g_rc = 0.15f;
    // molecular dynamics (mmx_md_*): velocities, low-order position bits, integrator constants
    float *v = nullptr, *xlo = nullptr;
    float *md_snap = nullptr;     // decomposed runs: x, v, xlo as of the last poll that found the ghost lists in order (3 vectors)
    uint64_t md_snap_step = 0;
    bool md_sync_next = false;    // the first step after a roll-back rebuilds its ghost lists synchronously (fresh capacities): progress
    double *ke_part = nullptr, *ke_out = nullptr;
    bool md_configured = false, md_forces_valid = false;
    int md_kind = 0;
    double md_dt = 0.0, md_temp = 0.0, md_friction = 0.0, md_mass = 1.0;
    double amd_alpha = 100.0, amd_e = 1000.0; // config.py:255-256
    uint64_t md_seed = 0, md_step = 0;
    MdParams md{};
    int forms[MMX_N_SELECTORS]{}; // functional form per term selector (0 = default)
    FormParams Q{};               // derived constants of the non-default forms (host copy)
    FormParams *formp = nullptr;  // device copy read by the FORMS instances of the pair kernels
    // options
    int deterministic = 0, profile = 0, poll_interval = 32, nb_variant = 0, fused_bonded = 1, overlap_bonded = 1;
    // hipGraph of consecutive minimizer evaluations (captured per mmx_minimize call, see graph_capture).  OFF by default:
    // on ROCm 7.2 / MI355X replaying loses to launch-by-launch submission at every size measured (DESIGN_HISTORY.md 5b)
    int use_graph = 0, graph_evals = 2; // evaluations per graph (even: the cell grid ping-pongs)
    // Grid cells wider than the cutoff once the structure has thinned out (set at the polls, see pull_state): below ~32 beads
    // per cutoff-sized cell the in-cell ordering is a latency chain per cell and 1.12 x wider cells (1.4 x the beads each)
    // take 8-10 us off the cell build for +2 us of pair kernel at 200 000 beads (profiles/r04_cell_edge_cost.txt); denser
    // states and systems below 20 000 beads lose or gain nothing.
    // Kept cell structure (option "cell_reuse", single-domain minimizations): membership, cluster composition, cluster order
    // and work items of a full build serve up to reuse_K evaluations; in between only the cluster positions and boxes are
    // refreshed (k_refresh_clusters) -- exact while no bead has moved more than half the skin (cell edge - cutoff) from
    // where it was binned, which k_pack checks; a violation voids the evaluation (PH_HALT) and it is repeated after a full
    // build.  reuse_K follows the displacements the polls read back (pull_state).
    // Slot table (SlotArgs, mmx_cells.hpp): trial moves write their sort keys straight into per-cell slots, no k_cell_fill
    int cell_slots = 1;           // option
    int n3_slice_cap = 0;         // option: longest window slice of a record, in clusters (0: the LDS window, kN3MaxCap)
    int n3_pass_records = 1;      // option: a run's window passes are records of the item list of their own (N3Item::w0); 0: for the A/B
    int dd_split = 1;             // option: decomposed ranks on the half-shell kernel keep the ghosts' clusters in a region of their own
                                  // (ghost clusters are never i-clusters: ScanArgs::split); 0 = interleaved per cell, for the A/B
    unsigned long long *slotkeys = nullptr;
    double wide_below = 0.0;      // option cell_wide_below (measurement): beads per cutoff-sized cell under which the wide grid is used (0: 32)
    bool md_active = false;       // the call in progress is mmx_md_step (the polls' cell-edge policy differs: pull_state)
    int slot_cap = 0, slot_cells = 0;
    size_t slot_total = 0;        // key slots of the one allocation (ensure_slots cuts it into slot_cells rows of slot_cap)
    bool slots_now = false;       // the build being enqueued uses the table
    long long slot_halts = 0;
    int cell_reuse = 1;
    float reuse_factor = 0.f;     // cell edge / cutoff once the structure has thinned out and structures are kept; 0: by pair kernel
                                  // (1.3 half shell, 1.45 full shell: scripts/cell_reuse_ab.py); option cell_reuse_factor
    int reuse_K = 1;              // evaluations a structure may serve (1: a full build per evaluation)
    int struct_evals = 0;         // evaluations the structure in use has served
    bool struct_valid = false;    // ... and cell_xref holds the positions it was built from
    float struct_factor = 1.f;    // cell edge / cutoff of the grid it was built on
    float grid_factor[2] = {1.f, 1.f}; // ... of the two ping-pong grids
    float *cell_xref = nullptr;   // [3 n_own]
    long long cell_builds = 0, cell_reuses = 0, cell_stale_halts = 0; // statistics
    int cell_edge_auto = 1;      // option: 0 = cells of edge cutoff throughout (A/B)
    float edge_auto = 1.f;       // the factor in force
    float cell_edge_scale = 1.f; // measurement only (option cell_edge_scale): grid cells of edge scale * cutoff -- what a Verlet skin would cost the pair kernels
    int inject_fault = 0; // tests only: bit 0 = every wait of k_nb_n3 times out at once, bit 1 = its item list holds one item,
                          // bit 2 = halo messages without slack, bit 3 = a kept cell structure is always stale, bit 4 = slot rows of 64,
                          // bit 5 = k_tail's fold gives up before its first poll, bit 6 = the direct build finds its grid too large
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    bool capturing = false;
    int gkey_parity = 0;
    int64_t glaunches[MMX_N_KERNELS]{};
    int prof_eval = -1; // minimizer: 1 / 0 = this evaluation is / is not a profiling sample, 2 = its pair kernel alone is;
                        // -1 = per-slot sampling
    int profile_nb = 0; // minimizer: every profile_nb-th evaluation samples the pair-kernel slot alone (one event pair)
    // profiling
    // decomposed runs: HIP-event time of the collectives of the sampled evaluations (option "profile"), slots kCollNeedmap..:
    // from the end of the work before it to its own end on this rank's stream, i.e. transfer + waiting for the peers
    double coll_ns[4]{};
    long long coll_samples[4]{};
    std::vector<EventPair> ev_pool, ev_used;
    int64_t launches[MMX_N_KERNELS]{};
    std::string err;
};
