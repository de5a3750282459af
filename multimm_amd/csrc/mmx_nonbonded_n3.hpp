// mmx_nonbonded_n3.hpp -- K2 (default from 100 000 beads): half-shell cluster-pair kernel, Newton's third law through LDS.
//
// Same pair physics and the same lean pair loop as k_nb_clusters_j (mmx_nonbonded.hpp; reference: model.py:199 EV
// power law, model.py:246-250 / 322-328 compartment Gaussians), but every unordered bead pair is evaluated ONCE: the
// pair force is accumulated on the i side (8 i beads in scalar registers, per-lane partial sums folded over the wave
// once per i-cluster) AND on the j side (the lane that holds the j bead keeps three more FMAs per pair).  The j-side
// sums of a 64-bead batch go to a per-workgroup force window in LDS, which is flushed to global memory once per work
// item: float atomics on global memory run at the memory side at ~1.3 TB/s on this part (MI355X_MICROARCH.md, "Global
// float atomics") -- one flush per 64-bead batch would need ~280 MB of them per evaluation at 200 000 beads, one flush
// per work item needs a tenth of that, in 256-byte contiguous wave instructions.
//
// Work item (n3_items_block builds the list after every cell scan, inside the launch of the in-cell ordering) = a run of
// up to 24 consecutive clusters (16 on systems below 150 000 beads; a dense cell by itself: equal runs of up to 30) of one
// ROW of the cell grid (x is the fastest cell index, so a row is one contiguous stretch of the cell-sorted cluster list);
// the run may span several cells when they are sparse.  Half shell by cells: the j candidates of the run are, in
// increasing cluster order, (0) the clusters of its own row from the run's first cluster to the end of the cell after
// its last one, (1) row y+1 and (2-4) the three rows of layer z+1, each from the cell before the run's first cell to
// the cell after its last one: five contiguous stretches of the cluster list.  Their concatenation is the index
// space of the LDS window.  An i-cluster takes every candidate with a cluster id >= its own (lower ids of its own row
// are i-clusters themselves and take the pair from their side; its own id = the self tile) that the box-box test
// accepts; candidates of cells that are not adjacent to the i-cluster's cell are at least one cell edge >= cutoff
// away, so geometry alone keeps the half shell exact.  A run whose candidates do not fit the window is processed in
// several passes over equal slices of the index space; every pass is a record of the item list of its own (N3Item::w0), so
// the passes of a run are taken by different workgroups at about the same time.
//
// Execution: ONE persistent workgroup of 16 waves per CU pulls items from a global queue (from its far end: descending
// rows).  A unit = (item, slice of its candidate index space).  Two units are in flight per workgroup, each with its own
// LDS force window and candidate buffers: waves grab the i-clusters of a unit one at a time from an LDS counter and move on
// to the next unit when none is left; the LAST wave to leave a unit opens the flush of its window and stages the unit
// after next (queue pop, item descriptor, candidate boxes and ids) while the others compute; the flush is done, a chunk
// at a time, by the waves that wait for that window.  No workgroup barrier in the steady state; every wait is a bounded
// spin that raises an error (failed evaluation) instead of hanging.
// Where a wave's time goes (gw_200k after 150 iterations, -DMMX_N3_TIMING build, scripts/dd_n3_tail.py and n3_trace.py):
// 74 % inside i-cluster visits (14.5 us each: set-up 0.5, culls 1.2, sweeps 11.6, fold + i-side atomics 1.3), 15 %
// waiting for its next unit -- a unit ends with its slowest visit (they range 5-48 us) and a wave may run only one unit
// ahead --, 11 % unit hand-over, staging, prologue and epilogue.  With 16-cluster items the waits were 24 %: the item
// length is what the measurement moved; staging ahead of the windows, splitting long visits between two waves and a
// queue in weight order were built and did not pay (profiles/r03_experiments/README.md).
//
// The tail.  A wave sweeps ~6 i-clusters per launch at 200 000 beads, ~35 us each, so without care the last sweep leaves
// most waves idle for a sixth of the kernel.  The items the queue hands out LAST are therefore handed out 2 or 4 times, as
// SHARES: share c of 2^s sweeps all i-clusters of the item against the candidates k = c (mod 2^s) of the window -- same
// staging, own LDS window, flushed like any other unit; the i-side sums are atomics anyway and the self tile belongs to
// exactly one share.  The ticket -> (item, share) map lives in stage_unit; the item list itself is not touched.
//
// LDS accumulation is int32 fixed point (2^-13 kJ/mol/nm): ds_add_f32 is serialised on gfx950 (measured,
// scripts/ubench/lds_atomic.hip: 193 cycles per wave instruction against 4.5-7 for ds_add_u32), and integer sums do
// not depend on the order in which the waves arrive.  The flush and the i side use float atomics on global memory,
// whose order is not fixed: results are reproducible to rounding, not bitwise.  `deterministic = 1` selects
// k_nb_clusters_j, which is.
//
// `diag` (nb_variant >> 16; timing diagnosis only, results are wrong with any bit but 64): 1 no flush atomics, 2 no LDS
// adds (and no j-side FMAs), 4 no pair arithmetic, 8 no i-side atomics, 16 cull only, 32 ghost i-clusters skipped, 64 items in ascending order.
//
// Self tile: the 8 beads of the i-cluster also enter the stream as j beads.  All 64 ordered pairs of that tile are
// evaluated, so the i side alone gets the complete intra-cluster force; the j-side sums of those lanes are dropped
// and their energies weighted 1/2.  The r = 0 self pair has zero force and a known energy that is taken out in the
// same batch it entered.
#pragma once
#include "mmx_nonbonded.hpp"

namespace mmx {

constexpr int kN3Waves = 16;        // waves per workgroup
constexpr int kN3Threads = kN3Waves * 64;
// i-clusters per work item (grabbed one at a time by the waves).  Large systems take LONG items (fewer units, fewer waits
// for a unit's slowest sweep: -4..-6 % of the kernel at 200 000 beads, -12 % at 1 M); below ~150 000 beads a workgroup
// would be left with 2-4 units per launch and the short ones win by 1-4 % (scripts/ab_choice.sh).  The host picks.
constexpr int kN3ItemClusters = 24, kN3ItemClustersSmall = 16, kN3LongItemsFrom = 150000;
struct N3ItemShape {
    int run;       // clusters per run of a sparse segment
    int dense;     // a cell of at least this many clusters is cut by itself (fewer still fit one window across a cell boundary)
    int dense_run; // ... into equal runs of at most this many clusters: 14 n candidates must fit the window
};
__host__ __device__ constexpr N3ItemShape n3_item_shape(int long_items) { // 0 short, 1 long, 2 (measurement) two visits per wave
    return long_items == 2 ? N3ItemShape{32, 22, 32} : long_items ? N3ItemShape{kN3ItemClusters, 17, 30} : N3ItemShape{kN3ItemClustersSmall, 22, 16};
}
constexpr int kN3List = 192;        // accepted j-clusters buffered per wave before a sweep (culled 128 candidates at a time)
constexpr int kN3MaxCap = 424;      // largest LDS window, in clusters (two windows in flight: 160 KB of LDS, all of it)
// A window slot receives at most one batch sum per i-cluster of the item, so sums below 2^31 / 16 units cannot
// overflow; a larger one (overlapping beads) bypasses LDS with a global float atomic.
constexpr float kN3Fix = 8192.f;
constexpr float kN3FixLim = (float)(2147483648.0 / 32) * 0.999f; // 2^26 units = 8192 kJ/mol/nm: items of up to 32 clusters
static_assert(n3_item_shape(1).dense_run <= 32 && n3_item_shape(2).dense_run <= 32 && kN3ItemClusters <= 32, "a window slot must not overflow");

struct N3Item { // 128 bytes; a single-domain launch reads the first 64 only
    int a, n;          // i-clusters [a, a + n)
    int T;             // END of this record's slice of the concatenated candidate runs (see w0)
    int w0;            // START of the slice.  A run whose candidates do not fit one LDS window is emitted as several records,
                       // one per window pass, with slices of equal length: the passes are queue entries of their own (any
                       // workgroup takes them, at the same time) instead of consecutive units of one workgroup
    int rlo[5], rn[5]; // the half-shell runs: clusters [rlo, rlo + rn)
    int pad1[2];
    // decomposed ranks (split cluster list: the ghosts' clusters in a region of their own behind the owned ones): the ghost
    // clusters of the 3 x 3 rows around the item, cells xa - 1 .. xb + 1 -- the FULL stencil, because ghost clusters are never
    // i-clusters: every owned-ghost pair is taken from the owned side
    int grlo[9];
    unsigned short grn[10];
    int pad2[2];
};
static_assert(sizeof(N3Item) == 128, "N3Item is two 64-byte halves");
constexpr int kN3Runs = 5, kN3GhostRuns = 9;

// dynamic LDS of k_nb_n3 for windows of `cap` clusters: two force windows, two box buffers, four id buffers
constexpr size_t n3_lds_bytes(int cap) {
    return sizeof(int) * 2 * 3 * ((size_t)cap * 8 + 8) + sizeof(float4) * 2 * 2 * ((size_t)cap + 1) + sizeof(int) * 4 * ((size_t)cap + 8);
}

// Control block of the unit pipeline (LDS).  Unit v uses force window v & 1, box buffer v & 1, id buffer v & 3 (while
// unit v + 2 is staged, the ids of v - 1 and v are still needed by their flushes and those of v + 1 by its compute).
struct N3Ctl {
    int grab[2];     // next i-cluster of the unit in this parity
    int done[2];     // waves that have run out of i-clusters of it
    int ready[2];    // highest unit number whose window, boxes and ids are in place
    int fl_epoch[2]; // unit whose window is being flushed (job open)
    int fl_next[2], fl_done[2], fl_total[2], fl_nwin[2], fl_ids[2];
    int error;
};
constexpr int kN3SpinLimit = 1 << 22; // s_sleep rounds before a waiting wave gives up (a bug, not a state of the data);
                                      // the kernel takes the limit as an argument so that a test can inject the failure

// ---- item builder ---------------------------------------------------------------------------------------------
// One wave per row of the cell grid, lanes = cells of the row.  A cell of >= S.dense clusters is cut into equal runs
// of at most S.dense_run clusters by itself (a run that left a dense cell would drag the candidates of five more
// cells into its window); the sparse cells between two dense ones (or row ends) form a segment that is cut into runs
// of S.run clusters from its start, wherever the cell boundaries fall.  Run starts follow from a segmented
// prefix sum over the lanes, so nothing walks the row sequentially.

struct N3Row {
    const int *cstart;
    int nx, base[5]; // cell index of x = 0 in the five candidate rows (-1: the row does not exist)
    const int *gstart; // split layout: ghost-cluster offsets per cell (nullptr: no ghost runs), gbase: id of the first ghost cluster
    int gbase, gb[9];  // cell index of x = 0 in the nine rows around the item (-1: outside the grid)
    int *err = nullptr; // MinState::kernel_error: a ghost run longer than its 16-bit length field voids the evaluation
    __device__ __forceinline__ int TG(int xa, int xb, int *grlo, unsigned short *grn) const {
        const int x0 = max(xa - 1, 0), x1 = min(xb + 1, nx - 1);
        int t = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            grlo[r] = 0;
            grn[r] = 0;
            if (gstart && gb[r] >= 0) {
                const int lo = gstart[gb[r] + x0], len = gstart[gb[r] + x1 + 1] - lo;
                grlo[r] = gbase + lo;
                // (cannot be reached today -- a cell of more than 4096 beads already raises KERR_ORDER_DD --; if that bound is ever
                //  relaxed, owned-ghost pairs must not be dropped in silence)
                if (len > 65535 && err) atomicOr(err, (int)KERR_N3_ITEMS);
                grn[r] = (unsigned short)min(len, 65535);
                t += (int)grn[r];
            }
        }
        return t;
    }
    // candidate runs of i-clusters [a, ...) that live in cells xa..xb of row base[0]; returns T
    __device__ __forceinline__ int T(int a, int xa, int xb, int *rlo, int *rn) const {
        const int x0 = max(xa - 1, 0), x1 = min(xb + 1, nx - 1);
        rlo[0] = a;
        rn[0] = cstart[base[0] + x1 + 1] - a;
        int t = rn[0];
#pragma unroll
        for (int r = 1; r < 5; ++r) {
            rlo[r] = 0;
            rn[r] = 0;
            if (base[r] >= 0) {
                rlo[r] = cstart[base[r] + x0];
                rn[r] = cstart[base[r] + x1 + 1] - rlo[r];
            }
            t += rn[r];
        }
        return t;
    }
};

// Per 64-cell chunk of a row: clusters of my cell, whether it is dense, the segment-relative cluster offset p of
// its first cluster (sparse cells), and the number of runs that START in it.  carry_p / carry_dense: state at the
// chunk's left edge (wave-uniform, updated for the next chunk).
struct N3Cell {
    int c0, n, p, runs;
    bool dense;
};
__device__ __forceinline__ N3Cell n3_cell(const N3ItemShape S, const int *__restrict__ cs /* cstart of the row */, int nx,
                                          int x, int lane, int &carry_p, bool &carry_dense) {
    N3Cell C;
    const bool valid = x < nx;
    C.c0 = valid ? cs[x] : 0;
    C.n = valid ? cs[x + 1] - C.c0 : 0;
    C.dense = C.n >= S.dense;
    bool prev_dense = __shfl_up((int)C.dense, 1, 64) != 0;
    if (lane == 0) prev_dense = carry_dense;
    // segmented inclusive sum of the sparse cells' cluster counts; a segment starts at a dense cell (which counts 0
    // itself) and at the cell after one
    int sum = C.dense ? 0 : C.n;
    bool head = C.dense || prev_dense;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int s2 = __shfl_up(sum, o, 64);
        const bool h2 = __shfl_up((int)head, o, 64) != 0;
        if (lane >= o) {
            if (!head) sum += s2;
            head = head || h2;
        }
    }
    if (!head) sum += carry_p; // no segment start at or before this lane inside the chunk
    C.p = sum - (C.dense ? 0 : C.n);
    C.runs = 0;
    if (valid && C.n > 0)
        C.runs = C.dense ? (C.n + S.dense_run - 1) / S.dense_run
                         : (C.p + C.n + S.run - 1) / S.run - (C.p + S.run - 1) / S.run;
    carry_p = __shfl(sum, 63, 64);
    carry_dense = __shfl((int)C.dense, 63, 64) != 0;
    return C;
}

// `bid` of `nblk` workgroups of 256 threads.  row_setup(row, y, z, R): fills the N3Row of grid row `row` -- where the cluster
// offsets of its five (+ nine) candidate rows are found: the scan's cstart array (n3_items_block), or offsets the wave itself
// derives from the populations (the direct build, mmx_build.hpp).
template <class RowSetup>
__device__ __forceinline__ void n3_items_rows(const int bid, const int nblk, const GridParams &G, N3Item *__restrict__ items,
                                              int max_items, MinState *__restrict__ st, const int long_items,
                                              const bool pass_records, const int slice_cap, RowSetup row_setup) {
    const N3ItemShape S = n3_item_shape(long_items);
    const int nrows = G.ny * G.nz, nx = G.nx;
    // (the wave index in a scalar register: everything that depends on the row only -- N3Row -- then lives in SGPRs; this
    //  function shares a kernel with the in-cell order, whose occupancy its registers set)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int row = bid * 4 + wave; row < nrows; row += nblk * 4) {
        N3Row R;
        const int y = row % G.ny, z = row / G.ny;
        row_setup(row, y, z, R);
        const int *const cstart = R.cstart;
        const int *cs = cstart + R.base[0];
        const int c_hi = cs[nx];
        if (c_hi == cs[0]) continue; // empty row
        // geometry of run j of a cell: i-clusters [a, a + n) in cells x..xb of the row
        auto run_of = [&](const N3Cell &C, int x, int j, int &a, int &n, int &xb) {
            xb = x;
            if (C.dense) { // runs of equal length (27 clusters: 14 + 13, not 16 + 11)
                a = C.c0 + (int)(((long long)C.n * j) / C.runs);
                n = C.c0 + (int)(((long long)C.n * (j + 1)) / C.runs) - a;
            } else {
                const int m0 = (C.p + S.run - 1) / S.run * S.run; // first run start at or after p
                a = C.c0 + (m0 - C.p) + j * S.run;
                n = min(S.run, c_hi - a);
                while (a + n > cs[xb + 1]) { // the run goes on into the next cell, unless that one is dense
                    if (cs[xb + 2] - cs[xb + 1] >= S.dense) {
                        n = cs[xb + 1] - a;
                        break;
                    }
                    ++xb;
                }
            }
        };
        // run j of a cell as a record (slice not yet set); returns the length of its candidate index space
        auto build = [&](const N3Cell &C, int x, int j, N3Item &it) {
            int xb;
            run_of(C, x, j, it.a, it.n, xb);
            it.pad1[0] = it.pad1[1] = 0;
            it.grn[9] = 0;
            it.pad2[0] = it.pad2[1] = 0;
            return R.T(it.a, x, xb, it.rlo, it.rn) + R.TG(x, xb, it.grlo, it.grn);
        };
        // one record per window pass (option n3_pass_records = 0, for the A/B: one record per run, the workgroup that takes it
        // walks the passes itself)
        auto passes = [&](int T) { return pass_records ? max(1, (T + slice_cap - 1) / slice_cap) : 1; };
        auto emit = [&](N3Item &it, int T, int np, int &at) {
            for (int q = 0; q < np; ++q) { // slices of equal length (<= the window)
                it.w0 = (int)(((long long)T * q) / np);
                it.T = (int)(((long long)T * (q + 1)) / np);
                items[at++] = it;
            }
        };
        // One pass per 64-cell chunk of the row: the records a lane will write are counted first (the length of its first run's
        // index space -- nearly always its only run -- is kept), the chunk's records are reserved with one atomic, then written.
        int carry_p = 0;
        bool carry_dense = true; // the row start opens a segment
        for (int xc = 0; xc < nx; xc += 64) {
            const int x = xc + lane;
            const N3Cell C = n3_cell(S, cs, nx, x, lane, carry_p, carry_dense);
            int mine = 0;
            for (int j = 0; j < C.runs; ++j) {
                N3Item t;
                mine += passes(build(C, x, j, t));
            }
            int inc = mine; // inclusive scan of the record counts: where my records go
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(inc, o, 64);
                if (lane >= o) inc += u;
            }
            const int chunk_total = __shfl(inc, 63, 64);
            if (chunk_total == 0) continue;
            int first = 0;
            if (lane == 0) first = atomicAdd(&st->n3_items, chunk_total);
            first = __builtin_amdgcn_readfirstlane(first);
            if (first + chunk_total > max_items) { // cannot happen (see the allocation: a run per cell plus one per 16 clusters,
                                                   // and room for their window passes)
                if (lane == 0) atomicOr(&st->kernel_error, (int)KERR_N3_ITEMS); // the controller voids the evaluation
                break;
            }
            int at = first + inc - mine;
            for (int j = 0; j < C.runs; ++j) {
                N3Item t;
                const int T = build(C, x, j, t);
                emit(t, T, passes(T), at);
            }
        }
    }
}

__device__ __forceinline__ void n3_items_block(const int bid, const int nblk, const GridParams *__restrict__ grid,
                                               const int *__restrict__ cstart, N3Item *__restrict__ items, int max_items,
                                               MinState *__restrict__ st, const int long_items,
                                               const int *__restrict__ gstart = nullptr, const bool pass_records = true,
                                               const int slice_cap = kN3MaxCap) {
    const GridParams G = *grid;
    const int gbase = gstart ? st->n_clusters_own : 0;
    n3_items_rows(bid, nblk, G, items, max_items, st, long_items, pass_records, slice_cap,
                  [&](int row, int y, int z, N3Row &R) {
                      const int nx = G.nx;
                      R.cstart = cstart;
                      R.nx = nx;
                      R.gstart = gstart;
                      R.gbase = gbase;
                      R.err = &st->kernel_error;
#pragma unroll
                      for (int r = 0; r < 9; ++r) {
                          const int yy = y + r % 3 - 1, zz = z + r / 3 - 1;
                          R.gb[r] = (gstart && yy >= 0 && yy < G.ny && zz >= 0 && zz < G.nz) ? (zz * G.ny + yy) * nx : -1;
                      }
                      R.base[0] = row * nx;
                      R.base[1] = y + 1 < G.ny ? (row + 1) * nx : -1;
#pragma unroll
                      for (int dy = -1; dy <= 1; ++dy)
                          R.base[3 + dy] = (z + 1 < G.nz && y + dy >= 0 && y + dy < G.ny) ? (row + G.ny + dy) * nx : -1;
                  });
}

// The item builder needs the scan's cluster offsets only, the in-cell ordering needs nothing of the items: both run in
// ONE launch (horizontal fusion, as the bonded pass rides in the launch of the cell scan): the first `n_order`
// workgroups (in block-index order: the LAST n_order) order the cells, the first few build the items -- 9 us of latency-bound row walking disappear under the 18 us
// of the cell order instead of standing in the stream.
constexpr int kN3ItemBlocks = 128;
template <int CHUNK, int CAP>
__global__ __launch_bounds__(256) void k_order_items(const int n_order, const GridParams *__restrict__ grid,
                                                     const int *__restrict__ start, const int *__restrict__ istart,
                                                     int *__restrict__ count, int *__restrict__ perm, int2 *__restrict__ items,
                                                     const int *__restrict__ cstart, const float4 *__restrict__ pos4,
                                                     float4 *__restrict__ spos4, float4 *__restrict__ cl_lo,
                                                     float4 *__restrict__ cl_hi, const Own own,
                                                     const unsigned long long *__restrict__ okeys,
                                                     const int *__restrict__ biglist,
                                                     N3Item *__restrict__ n3_items, int n3_max_items,
                                                     MinState *__restrict__ st, int *__restrict__ count_own = nullptr,
                                                     int *__restrict__ sbead = nullptr, const int n3_long_items = 0,
                                                     const int slot_cap = 0, const int slot_cells = 0, const int split = 0,
                                                     int *__restrict__ slot_of = nullptr) {
    if (st->phase >= PH_DONE) return;
    const int n_items_blocks = (int)gridDim.x - n_order; // they come FIRST: dispatched at once, their latency chains
    if ((int)blockIdx.x < n_items_blocks) {              // run beside the cell order instead of behind it
        n3_items_block((int)blockIdx.x, n_items_blocks, grid, cstart, n3_items, n3_max_items, st, (n3_long_items & 4) ? 2 : (n3_long_items & 1),
                       split ? istart : nullptr, !(n3_long_items & 2),
                       (n3_long_items >> 8) > 0 ? min(n3_long_items >> 8, kN3MaxCap) : kN3MaxCap);
        return;
    }
    cell_order_block<CHUNK, CAP>((int)blockIdx.x - n_items_blocks, n_order, grid, start, istart, count, perm, items, cstart, pos4,
                                 spos4, cl_lo, cl_hi, own, okeys, biglist, st, count_own, sbead, slot_cap, slot_cells, split, slot_of);
}

#ifdef MMX_N3_TIMING
__device__ unsigned long long g_n3_t[512 * 20];
// cumulative work counters: [ghost i-cluster ? 4 : 0] + {i-clusters, candidates past the cluster cull, streamed steps of 64 beads,
// batches of 64 beads x 8 i beads through the pair loop}; [8] launches
__device__ unsigned long long g_n3_c[16];
// per wave: ticks (10 ns) spent waiting for a unit to become ready (flush help included), ticks inside i-cluster visits
__device__ unsigned g_n3_w[512 * 16 * 8]; // + ticks of a visit's phases: set-up (loads, scalar i beads), sweeps, fold + i-side atomics
// event trace of the first kN3TraceBlocks workgroups: {kind | wave << 8, unit, i-cluster of the unit, t0, t1 (10 ns ticks, low 32 bits), batches, -, -};
// kind 0 visit, 1 wait at the entry of a unit, 2 staging (of unit), 3 last wave left the unit (flush job open), 4 window ready for unit
constexpr int kN3TraceBlocks = 32, kN3TraceEvents = 1024;
__device__ unsigned g_n3_v[kN3TraceBlocks * kN3TraceEvents * 8];
__device__ unsigned g_n3_vn[kN3TraceBlocks];
#define N3_TRACE(kind, unit, gi, t0, t1, nb)                                                          \
    do {                                                                                               \
        if (blockIdx.x < kN3TraceBlocks && lane == 0) {                                                \
            const unsigned ix_ = atomicAdd(&g_n3_vn[blockIdx.x], 1u);                                  \
            if (ix_ < (unsigned)kN3TraceEvents) {                                                      \
                unsigned *e_ = g_n3_v + ((size_t)blockIdx.x * kN3TraceEvents + ix_) * 8;               \
                e_[0] = (unsigned)(kind) | ((unsigned)wave << 8);                                      \
                e_[1] = (unsigned)(unit);                                                              \
                e_[2] = (unsigned)(gi);                                                                \
                e_[3] = (unsigned)(t0);                                                                \
                e_[4] = (unsigned)(t1);                                                                \
                e_[5] = (unsigned)(nb);                                                                \
            }                                                                                          \
        }                                                                                              \
    } while (0)
#else
#define N3_TRACE(kind, unit, gi, t0, t1, nb) do {} while (0)
#endif
// float -> int, rounded to nearest (ties up): ONE operation where __float2int_rn is v_rndne + v_cvt
__device__ __forceinline__ int cvt_nearest(float v) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// fsort: force (not gradient) per cluster slot, SoA [3][fstride]; zero on entry, k_nb_n3_unsort zeroes it again.
//
// DD (decomposed runs: the cell list holds this rank's owned beads AND its ghosts, in separate clusters -- emit_clusters).
// Split layout (default, ScanArgs::split): the owned clusters come first in the cluster list, the ghosts' behind them; work
// items are cut from the owned clusters only, and an item's window is its five half-shell runs of owned clusters followed by
// nine runs of ghost clusters -- the FULL 3 x 3 rows around it -- so that every owned-ghost pair is taken from the owned side
// and a ghost cluster is never an i-cluster (no visit, no cull, no flush for it: its window sums are dropped in LDS).
// Interleaved layout (option dd_split = 0, and what a single domain forced onto this instance sees): every pair with at
// least one owned bead is evaluated once, by the cluster with the lower id, whichever side owns what; an all-ghost
// i-cluster culls all-ghost candidates, and forces that land on ghost slots are dropped by k_nb_n3_unsort.
// Either way pairs of two ghosts are never evaluated and energies are weighted 1/2 (own_i + own_j): each rank books half of
// every cross-rank pair, all of every pair it owns both beads of.  (A/B of the two layouts: profiles/r04_dd_split_ab.txt.)
template <int PMODE, bool EV, bool GAUSS, bool NOENERGY, bool DD = false>
__global__ __launch_bounds__(kN3Threads, 4) void k_nb_n3(const FFParams P, const float4 *__restrict__ spos4,
                                                          const float4 *__restrict__ cl_box,
                                                          const N3Item *__restrict__ items, MinState *__restrict__ st,
                                                          float *__restrict__ fsort, const int fstride,
                                                          double *__restrict__ part, const int cap_arg,
                                                          const int diag = 0, const int tail_items = 0,
                                                          const int tail_sh = 0, const int tail2_items = 0,
                                                          const int tail2_sh = 0, const int spin_limit = kN3SpinLimit) {
    if (st->phase >= PH_DONE) return;
#ifdef MMX_N3_TIMING
    if (threadIdx.x == 0) {
        g_n3_t[blockIdx.x * 20] = wall_clock64();
        g_n3_t[blockIdx.x * 20 + 18] = g_n3_t[blockIdx.x * 20 + 19] = 0ull;
        if (blockIdx.x < kN3TraceBlocks) g_n3_vn[blockIdx.x] = 0u;
    }
    __syncthreads();
#endif
    // dynamic LDS: two force windows [3][cap*8 + 8] int (x | y | z per window slot, fixed point, see kN3Fix; the last
    // 8 slots are a dummy cluster), two box buffers [cap + 1][2] float4, four id buffers [cap + 8] int
    extern __shared__ __attribute__((aligned(16))) int s_f[];
    __shared__ unsigned short s_list[kN3Waves][kN3List + 72];
    __shared__ float4 s_ring[kN3Waves][128];
    __shared__ __attribute__((aligned(32))) float s_tab[5 * 8];
    __shared__ float s_arow[kN3Waves][kCl * 8];
    __shared__ double s_e[2][kN3Waves];
    __shared__ int s_desc[2][DD ? 40 : 16]; // unit descriptors: a, n, T, wlo, rlo[5], rn[5], id buffer, share | log2(shares) << 8
                                            // (decomposed ranks: + the nine ghost runs, grlo at 16, grn at 25)
    __shared__ N3Ctl ctl;
    // the wave index in a scalar register: hipcc cannot prove threadIdx.x >> 6 uniform and would otherwise keep the
    // scalar i beads in vector registers
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane >> 3, slot = lane & 7;
    // the window size is a compile-time constant (the host refuses the kernel when the device cannot give it that much
    // LDS): window strides become immediate offsets of the LDS instructions
    constexpr int cap = kN3MaxCap;
    (void)cap_arg;
    constexpr int fstr = cap * 8 + 8;
    int *const box0 = s_f + 2 * 3 * fstr;           // box buffers
    int *const ids0 = box0 + 2 * 8 * (cap + 1);     // id buffers
    const int far_cl = P.n_all; // a resident all-padding cluster (8 beads at -1e18)
    // decomposed ranks: clusters from here on are ghosts' (the split cluster list of cell_scan_block): what lands on their
    // slots is nobody's business on this rank -- the flush and the large-sum bypass skip them, the unsort never reads them
    const int n_own_cl = DD ? st->n_clusters_own : 0x7fffffff;
    // (an item list that overflowed -- the evaluation is void already -- is not walked: its count runs past the buffer)
    const int n_items = (st->kernel_error & (int)KERR_N3_ITEMS) ? 0 : st->n3_items;
    if (threadIdx.x < 40) {
        const bool in5 = (threadIdx.x & 7) < 5;
        s_tab[threadIdx.x] = in5 ? P.table[(threadIdx.x >> 3) * 5 + (threadIdx.x & 7)] : 0.f;
    }
    for (int q = threadIdx.x * 4; q < 2 * 3 * fstr; q += kN3Threads * 4) *reinterpret_cast<int4 *>(s_f + q) = make_int4(0, 0, 0, 0);
    if (threadIdx.x < 32) // list padding points at window entry `cap` of every id buffer: the dummy cluster
        ids0[(threadIdx.x >> 3) * (cap + 8) + cap + (threadIdx.x & 7)] = far_cl;
    if (threadIdx.x == 0) {
        ctl.grab[0] = ctl.grab[1] = ctl.done[0] = ctl.done[1] = 0;
        ctl.ready[0] = ctl.ready[1] = -1;
        ctl.fl_epoch[0] = ctl.fl_epoch[1] = -100;
        ctl.fl_next[0] = ctl.fl_next[1] = ctl.fl_done[0] = ctl.fl_done[1] = ctl.fl_total[0] = ctl.fl_total[1] = 0;
        ctl.error = 0;
    }
    __syncthreads();
    unsigned short *list = s_list[wave];
    float4 *ring = s_ring[wave];
    const float *arow = s_arow[wave];
    // factored constants: exactly the LEAN instance of k_nb_clusters_j
    const float rc2 = P.rc2max;
    const float s3 = P.ev_sigma * P.ev_sigma * P.ev_sigma;
    const float ev_c = P.ev_eps * s3 * s3;
    const float tiny = 1e-20f;
    const float escale = (EV && PMODE == 6) ? ev_c : 1.f;
    const float pscale = EV ? P.ev_power * escale : 1.f;
    const float g_k = P.g_inv_rc2 / pscale;
    const float fix_k = pscale * kN3Fix;     // pair-loop units -> fixed point
    const float fix_lim = kN3FixLim / fix_k; // largest |batch sum| the fixed-point path takes
    const float unfix = -1.f / kN3Fix;       // fixed point -> force on the j bead (reaction: minus)
    const float nbig = -1e30f;
    const float cut_all = 1e30f * fminf(rc2, 1e6f);
    // energy of the r = 0 self pair, by the very operations of the pair loop
    float eself = 0.f;
    if (EV && !NOENERGY) {
        const float us = __builtin_amdgcn_rcpf(fmaf(tiny, __builtin_amdgcn_rsqf(tiny), P.ev_rs));
        if (PMODE == 6) {
            const float u2 = us * us;
            eself = (u2 * u2) * u2;
        } else {
            eself = P.ev_eps * ev_pow<PMODE>(P.ev_sigma * us, P.ev_power);
        }
    }
    double acc_ev = 0.0, acc_g = 0.0;
#ifdef MMX_N3_TIMING
    unsigned cnt_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (blockIdx.x == 0 && threadIdx.x == 0 && (diag & 128)) atomicAdd(&g_n3_c[8], 1ull);
#endif

    volatile int *vready = ctl.ready, *vepoch = ctl.fl_epoch;
    // Flush job of parity p (window of unit `unit`): takes one chunk of 256 window slots if there is one left.
    // fsort[slot] -= sum / 2^13; the 8 slots of a cluster are contiguous, neighbouring candidates mostly too: 256-byte
    // atomic wave instructions.  Zeroes what it flushes; whoever hands in the last token declares unit + 2 ready.
    auto help_flush = [&](int p, int unit) {
        if (vepoch[p] != unit) return;
        wg_lds_acquire(); // the job description and the window sums of every wave that left the unit
        const int total = ctl.fl_total[p];
        int c = total;
        if (lane == 0 && *(volatile int *)&ctl.fl_next[p] < total) c = atomicAdd(&ctl.fl_next[p], 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= total) return;
        int *wx = s_f + p * 3 * fstr, *wy = wx + fstr, *wz = wy + fstr;
        const int *jcs = ids0 + ctl.fl_ids[p] * (cap + 8);
        const int nsl = ctl.fl_nwin[p] * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = c * 256 + r * 64 + lane;
            if (e < nsl) {
                const int vx = wx[e], vy = wy[e], vz = wz[e];
                if ((vx | vy | vz) != 0) {
                    wx[e] = 0;
                    wy[e] = 0;
                    wz[e] = 0;
                    if (!(diag & 1) && jcs[e >> 3] < n_own_cl) {
                        const int gs = jcs[e >> 3] * kCl + (e & 7);
                        atomicAdd(fsort + gs, unfix * (float)vx);
                        atomicAdd(fsort + fstride + gs, unfix * (float)vy);
                        atomicAdd(fsort + 2 * fstride + gs, unfix * (float)vz);
                    }
                }
            }
        }
        if (c == 0 && lane < 8) wx[cap * 8 + lane] = wy[cap * 8 + lane] = wz[cap * 8 + lane] = 0; // the dummy cluster
        wg_lds_release(); // the zeroed chunk, before the token that may declare the window ready
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && atomicAdd(&ctl.fl_done[p], 1) == total) {
            vready[p] = unit + 2;
#ifdef MMX_N3_TIMING
            N3_TRACE(4, unit + 2, 0, wall_clock64(), 0, 0);
#endif
        }
    };
    // Decides what unit `v` is (the next pass of the item of unit v - 1, or the next item of the queue), stages its
    // window -- candidate k of the concatenated runs -> cluster id, box -- and writes its descriptor.  One wave.
    auto stage_unit = [&](int v) {
        const int p = v & 1, ib = v & 3;
        constexpr int NR = DD ? kN3Runs + kN3GhostRuns : kN3Runs; // candidate runs of an item (decomposed ranks: + the ghosts')
        int a = 0, n = 0, T = 0, wlo = 0, rlo[NR], rn[NR], shr = 0;
        bool fetch = true;
        if (v > 0) { // unit v - 1 sits in the other parity
            const int *pd = s_desc[p ^ 1];
            n = __builtin_amdgcn_readfirstlane(pd[1]);
            T = __builtin_amdgcn_readfirstlane(pd[2]);
            wlo = __builtin_amdgcn_readfirstlane(pd[3]) + cap;
            if (n == 0) fetch = false; // the queue has run dry: stays dry
            else if (wlo < T) {        // next pass of the same item
                fetch = false;
                a = __builtin_amdgcn_readfirstlane(pd[0]);
                shr = __builtin_amdgcn_readfirstlane(pd[15]);
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    rlo[r] = __builtin_amdgcn_readfirstlane(pd[4 + r]);
                    rn[r] = __builtin_amdgcn_readfirstlane(pd[9 + r]);
                }
                if (DD) {
#pragma unroll
                    for (int r = 0; r < kN3GhostRuns; ++r) {
                        rlo[(DD ? 5 : 0) + r] = __builtin_amdgcn_readfirstlane(pd[(DD ? 16 : 0) + r]);
                        rn[(DD ? 5 : 0) + r] = __builtin_amdgcn_readfirstlane(pd[(DD ? 25 : 0) + r]);
                    }
                }
            }
        }
        if (fetch) {
            int q = 0;
            if (lane == 0) q = atomicAdd(&st->n3_queue, 1);
            q = __builtin_amdgcn_readfirstlane(q);
            // the list is in (roughly) ascending row order; taking it from the far end measured ~5 % faster at 200 000
            // beads (an item's flush targets its own row and the rows above it: descending order keeps the workgroups
            // in flight off each other's target rows); diag & 64: ascending, for the A/B
            // The tail: the `tail_items` items taken last are each taken 2^tail_sh times, as SHARES -- every share sweeps all the
            // i-clusters of the item against every 2^tail_sh-th candidate cluster of the window (its own LDS window, flushed
            // like any other) -- so that the work left when the queue runs dry comes in pieces of a quarter of an i-cluster
            // sweep and spreads over all workgroups: without it the last of ~6 sweeps per wave (~35 us each at 200 000
            // beads) leaves the waves idle for 17 % of the kernel on average (scripts/n3_tail.py).
            if (!(diag & 64)) {
                const int K2 = min(tail2_items, n_items), K1 = min(tail_items, n_items - K2);
                if (q < n_items - K1 - K2) q = n_items - 1 - q;
                else {
                    int r = q - (n_items - K1 - K2);
                    if (r < (K1 << tail_sh)) {
                        q = K2 + K1 - 1 - (r >> tail_sh);
                        shr = (r & ((1 << tail_sh) - 1)) | (tail_sh << 8);
                    } else {
                        r -= K1 << tail_sh;
                        q = r < (K2 << tail2_sh) ? K2 - 1 - (r >> tail2_sh) : n_items;
                        shr = (r & ((1 << tail2_sh) - 1)) | (tail2_sh << 8);
                    }
                }
            }
            n = 0;
            wlo = 0;
            if (q < n_items) {
                const int4 *pi = reinterpret_cast<const int4 *>(items + q);
                const int4 v0 = pi[0], v1 = pi[1], v2 = pi[2], v3 = pi[3];
                a = __builtin_amdgcn_readfirstlane(v0.x);
                n = __builtin_amdgcn_readfirstlane(v0.y);
                T = __builtin_amdgcn_readfirstlane(v0.z);
                wlo = __builtin_amdgcn_readfirstlane(v0.w); // the record's slice of the candidate index space: [w0, T)
                rlo[0] = __builtin_amdgcn_readfirstlane(v1.x);
                rlo[1] = __builtin_amdgcn_readfirstlane(v1.y);
                rlo[2] = __builtin_amdgcn_readfirstlane(v1.z);
                rlo[3] = __builtin_amdgcn_readfirstlane(v1.w);
                rlo[4] = __builtin_amdgcn_readfirstlane(v2.x);
                rn[0] = __builtin_amdgcn_readfirstlane(v2.y);
                rn[1] = __builtin_amdgcn_readfirstlane(v2.z);
                rn[2] = __builtin_amdgcn_readfirstlane(v2.w);
                rn[3] = __builtin_amdgcn_readfirstlane(v3.x);
                rn[4] = __builtin_amdgcn_readfirstlane(v3.y);
                if (DD) { // second half of the record: the ghost runs (ints 16..24: first clusters, 25..29: nine 16-bit lengths)
                    const int4 g0 = pi[4], g1 = pi[5], g2 = pi[6], g3 = pi[7];
                    const int gw[14] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y};
#pragma unroll
                    for (int r = 0; r < kN3GhostRuns; ++r) {
                        rlo[(DD ? 5 : 0) + r] = __builtin_amdgcn_readfirstlane(gw[r]);
                        rn[(DD ? 5 : 0) + r] = (__builtin_amdgcn_readfirstlane(gw[9 + (r >> 1)]) >> ((r & 1) * 16)) & 0xffff;
                    }
                }
            }
        }
        if (n > 0) {
            float4 *box = reinterpret_cast<float4 *>(box0 + p * 8 * (cap + 1));
            int *jcs = ids0 + ib * (cap + 8);
            const int nwin = min(T, wlo + cap) - wlo;
            int woff[NR];
            woff[0] = 0;
#pragma unroll
            for (int r = 1; r < NR; ++r) woff[r] = woff[r - 1] + rn[r - 1];
            for (int k = lane; k < nwin; k += 64) {
                const int kk = wlo + k;
                int jc = rlo[0] + kk;
#pragma unroll
                for (int r = 1; r < NR; ++r) jc = kk >= woff[r] ? rlo[r] + (kk - woff[r]) : jc;
                jcs[k] = jc;
                box[2 * k] = cl_box[2 * jc];
                box[2 * k + 1] = cl_box[2 * jc + 1];
            }
        }
#ifdef MMX_N3_TIMING
        if (lane == 0 && n > 0) {
            atomicAdd(&g_n3_t[blockIdx.x * 20 + 19], 1ull);
            if (wlo > 0) atomicAdd(&g_n3_t[blockIdx.x * 20 + 18], 1ull);
        }
#endif
        if (lane < (DD ? 40 : 16)) {
            int val = 0;
            if (n > 0) {
                val = lane == 0 ? a : lane == 1 ? n : lane == 2 ? T : lane == 3 ? wlo : lane == 14 ? ib : lane == 15 ? shr : 0;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    val = lane == 4 + r ? rlo[r] : val;
                    val = lane == 9 + r ? rn[r] : val;
                }
                if (DD) {
#pragma unroll
                    for (int r = 0; r < kN3GhostRuns; ++r) {
                        val = lane == 16 + r ? rlo[(DD ? 5 : 0) + r] : val;
                        val = lane == 25 + r ? rn[(DD ? 5 : 0) + r] : val;
                    }
                }
            }
            s_desc[p][lane] = val;
        }
        wg_lds_release(); // descriptor, ids and boxes, before the caller raises `ready` (or hands in its flush token)
        __builtin_amdgcn_wave_barrier();
    };

    // ---- unit pipeline.  No workgroup barrier: a wave that runs out of i-clusters of unit v moves on to unit v + 1
    // (other window, other buffers); the LAST wave to leave unit v opens the flush of its window and stages unit
    // v + 2 meanwhile; the flush itself is done, a chunk at a time, by the waves that want to start unit v + 2.
    if (wave == 0) {
        stage_unit(0);
        if (lane == 0) vready[0] = 0;
        stage_unit(1);
        if (lane == 0) vready[1] = 1;
    }
#ifdef MMX_N3_TIMING
    unsigned long long t_wait = 0ull, t_visit = 0ull, t_stage = 0ull, n_stage = 0ull, t_setup = 0ull, t_sweep = 0ull, t_fold = 0ull;
#endif
    for (int v = 0;; ++v) {
        const int p = v & 1;
        bool failed = false;
#ifdef MMX_N3_TIMING
        const unsigned long long tw0 = wall_clock64();
#endif
        for (int spins = 0; vready[p] < v; ++spins) {
            help_flush(p, v - 2);
            __builtin_amdgcn_s_sleep(2);
            if (spins >= spin_limit || *(volatile int *)&ctl.error) {
                failed = true;
                break;
            }
        }
        if (failed) {
            if (lane == 0) {
                ctl.error = 1;
                atomicOr(&st->kernel_error, (int)KERR_N3_SPIN); // the controller voids the evaluation (MMX_MIN_KERNEL)
            }
            break;
        }
#ifdef MMX_N3_TIMING
        {
            const unsigned long long tw1 = wall_clock64();
            t_wait += tw1 - tw0;
            N3_TRACE(1, v, 0, tw0, tw1, 0);
        }
#endif
        wg_lds_acquire(); // what the stager / the flushers wrote before they raised `ready`
        __builtin_amdgcn_wave_barrier();
        const int *dd = s_desc[p];
        const int D_n = __builtin_amdgcn_readfirstlane(dd[1]);
        if (D_n == 0) { // the queue is dry; unit v - 1 may still need its window flushed
            for (int spins = 0; v > 0 && vready[p ^ 1] < v + 1 && spins < spin_limit && !*(volatile int *)&ctl.error; ++spins) {
                help_flush(p ^ 1, v - 1);
                __builtin_amdgcn_s_sleep(2);
            }
            break;
        }
        const int D_a = __builtin_amdgcn_readfirstlane(dd[0]), D_T = __builtin_amdgcn_readfirstlane(dd[2]),
                  wlo = __builtin_amdgcn_readfirstlane(dd[3]), D_ib = __builtin_amdgcn_readfirstlane(dd[14]),
                  D_shr = __builtin_amdgcn_readfirstlane(dd[15]);
        const int sh = D_shr >> 8, share = D_shr & 255; // this unit sweeps candidates k = share (mod 2^sh) of the window
        const float4 *s_box = reinterpret_cast<const float4 *>(box0 + p * 8 * (cap + 1));
        const int *s_jc = ids0 + D_ib * (cap + 8);
        int *sfx = s_f + p * 3 * fstr;
        const int nwin = min(D_T, wlo + cap) - wlo;
        // ---- compute: grab i-clusters of the item one at a time
        for (;;) {
            int gi = 0;
            if (lane == 0) gi = atomicAdd(&ctl.grab[p], 1);
            gi = __builtin_amdgcn_readfirstlane(gi);
            if (gi >= D_n) break;
#ifdef MMX_N3_TIMING
            const unsigned long long tv0 = wall_clock64();
            const unsigned nb0 = cnt_w[3] + cnt_w[7];
#endif
            const int icl = D_a + gi;
            const float4 lo_i = cl_box[2 * icl], hi_i = cl_box[2 * icl + 1];
            // decomposed runs: is this a cluster of owned beads or of ghosts (never both)
            const bool i_own = !DD || __builtin_amdgcn_readfirstlane(__float_as_int(hi_i.w) >> 8) != 0;
            if (DD && !i_own && (diag & 32)) continue; // timing diagnosis only: what the kernel costs without ghost i-clusters
            float4 pv = spos4[(size_t)icl * kCl + slot];
            const int own_w = __float_as_int(pv.w);
            if (own_w < 0) { // padding slots: far away on the i side (they are j entries at +1e18 too)
                pv.x = pv.y = pv.z = 3e18f;
                pv.w = __int_as_float(-8 + 2);
            }
            float xi[kCl], yi[kCl], zi[kCl];
            float fx[kCl], fy[kCl], fz[kCl];
#pragma unroll
            for (int s = 0; s < kCl; ++s) {
                xi[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.x), s));
                yi[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.y), s));
                zi[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.z), s));
                fx[s] = fy[s] = fz[s] = 0.f;
            }
            if (GAUSS) { // lane l: row (l >> 3) = i bead, column (l & 7) = label of the j bead
                const int wrow = __shfl(__float_as_int(pv.w), lane >> 3, 64) & 7;
                wave_lds_sync(); // the previous i-cluster's sweeps have finished reading the rows
                s_arow[wave][lane] = s_tab[wrow * 8 + (lane & 7)];
                wave_lds_sync();
            }
            float ee = 0.f, eg = 0.f;
#ifdef MMX_N3_TIMING
            const int cg = i_own ? 0 : 4;
            cnt_w[cg]++;
            t_setup += wall_clock64() - tv0;
#endif
            // the i-cluster's own place in the window: candidates before it have lower cluster ids (i-clusters of this
            // item or of an earlier one: they take those pairs); it is itself a candidate (the self tile)
            const int own_k = (icl - D_a) - wlo;
            if (own_k < nwin) {
                const int own_lc = own_k >= 0 ? own_k : -1;
                const int k0 = max(own_k, 0);
                int nlist = 0, rcount = 0, rhead = 0;
                for (int g = (k0 >> sh) & ~63; (g << sh) < nwin; g += 128) {
                    // ---- cull: 128 candidate clusters of the window per step (two independent box reads in flight: the LDS
                    // round trip is what a step waits for), boxes from LDS
                    bool ok[2], jown[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = ((g + 64 * h + lane) << sh) + share;
                        ok[h] = false;
                        jown[h] = false;
                        if (k >= k0 && k < nwin) {
                            const float4 lo_j = s_box[2 * k], hi_j = s_box[2 * k + 1];
                            const float dx = fmaxf(fmaxf(lo_j.x - hi_i.x, lo_i.x - hi_j.x), 0.f);
                            const float dy = fmaxf(fmaxf(lo_j.y - hi_i.y, lo_i.y - hi_j.y), 0.f);
                            const float dz = fmaxf(fmaxf(lo_j.z - hi_i.z, lo_i.z - hi_j.z), 0.f);
                            ok[h] = fmaf(dx, dx, fmaf(dy, dy, dz * dz)) < rc2;
                            // decomposed runs: a cluster is all-owned or all-ghost; never ghosts against ghosts.  Which of the
                            // two the candidate is rides in bit 15 of its list entry (window indices stay below 2^9): the sweep
                            // needs it per j bead for the energy weight, and a look-up in the ownership tables there would be a
                            // dependent global load in every step of the stream
                            if (DD) {
                                jown[h] = (__float_as_int(hi_j.w) >> 8) != 0;
                                ok[h] = ok[h] && (i_own || jown[h]);
                            }
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned long long mask = __ballot(ok[h]);
                        if (ok[h])
                            list[nlist + prefix_count(mask)] =
                                (unsigned short)((((g + 64 * h + lane) << sh) + share) | ((DD && jown[h]) ? 0x8000 : 0));
                        nlist += __builtin_amdgcn_readfirstlane(__popcll(mask));
                    }
                    const bool last = ((g + 128) << sh) >= nwin;
                    if (nlist < kN3List - 128 && !last) continue;
                    if (diag & 16) { // timing diagnosis only: cull without sweep
                        fx[0] += (float)nlist;
                        nlist = 0;
                        continue;
                    }
                    if (lane < 8) list[nlist + lane] = (unsigned short)cap; // pad to a multiple of 8: the dummy cluster
                    wave_lds_sync();
                    const int nsteps = max((nlist + 7) >> 3, 1);
#ifdef MMX_N3_TIMING
                    cnt_w[cg + 1] += nlist;
                    cnt_w[cg + 2] += nsteps;
                    const unsigned long long tsw0 = wall_clock64();
#endif
                    // ---- sweep: 8 j-clusters (64 j beads) per step; ids two steps ahead, positions one
                    constexpr int kIdx = DD ? 0x7fff : 0xffff; // (list entry: window index | owned-cluster flag, see the cull)
                    int ln = list[sub];
                    float4 qn = spos4[(unsigned)s_jc[ln & kIdx] * kCl + slot];
                    int ln2 = nsteps > 1 ? list[8 + sub] : cap;
                    int jn = s_jc[ln2 & kIdx];
                    for (int t = 0; t < nsteps; ++t) {
                        float4 q = qn;
                        const int lq = ln;
                        if (t + 1 < nsteps) {
                            qn = spos4[(unsigned)jn * kCl + slot];
                            ln = ln2;
                            ln2 = t + 2 < nsteps ? list[(t + 2) * 8 + sub] : cap;
                            jn = s_jc[ln2 & kIdx];
                        }
                        // per-bead cull against the i box; survivors are compacted through the ring with their LDS slot
                        {
                            const float bx = fmaxf(fmaxf(lo_i.x - q.x, q.x - hi_i.x), 0.f);
                            const float by = fmaxf(fmaxf(lo_i.y - q.y, q.y - hi_i.y), 0.f);
                            const float bz = fmaxf(fmaxf(lo_i.z - q.z, q.z - hi_i.z), 0.f);
                            const bool okb = fmaf(bx, bx, fmaf(by, by, bz * bz)) < rc2;
                            const unsigned long long mb = __ballot(okb);
                            if (okb) {
                                // window slot (< 2^12) << 3 | label, bit 15: the j bead is owned
                                const int wj = ((((lq & kIdx) << 3) | slot) << 3) | (__float_as_int(q.w) & 7) | (DD ? lq & 0x8000 : 0);
                                q.w = __int_as_float(wj);
                                ring[(rhead + rcount + prefix_count(mb)) & 127] = q;
                            }
                            rcount += __builtin_amdgcn_readfirstlane(__popcll(mb));
                        }
                        const bool fin = last && (t + 1 == nsteps);
                        while (rcount >= 64 || (fin && rcount > 0)) {
                            wave_lds_sync();
                            q = ring[(rhead + lane) & 127];
                            if (rcount < 64) { // wave-uniform: only the very last, partial batch
                                if (lane >= rcount) q = make_float4(-1e18f, -1e18f, -1e18f, __int_as_float(((cap * 8) << 3) | 2));
                            }
                            const int took = min(rcount, 64);
#ifdef MMX_N3_TIMING
                            cnt_w[cg + 3]++;
#endif
                            rhead = (rhead + took) & 127;
                            rcount -= took;
                            const int wq = __float_as_int(q.w);
                            const int lj = wq & 7;
                            const int jslot = DD ? (wq >> 3) & 0xfff : wq >> 3; // (window slots: < 2^12)
                            // energy weight of this lane's pairs: half per owned side
                            const float wl = DD ? 0.5f * ((i_own ? 1.f : 0.f) + (float)((wq >> 15) & 1)) : 1.f;
                            float fjx = 0.f, fjy = 0.f, fjz = 0.f, eb = 0.f, gb = 0.f;
                            if (diag & 4) { // timing diagnosis only: no pair arithmetic
                                fx[0] += q.x;
                                continue;
                            }
#pragma unroll
                            for (int s = 0; s < kCl; ++s) {
                                const float dx = xi[s] - q.x, dy = yi[s] - q.y, dz = zi[s] - q.z;
                                const float r2t = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, tiny)));
                                const float in = fma_sat(r2t, nbig, cut_all);
                                const float rinv = __builtin_amdgcn_rsqf(r2t);
                                float fs = 0.f;
                                if (EV) {
                                    const float uu = __builtin_amdgcn_rcpf(fmaf(r2t, rinv, P.ev_rs));
                                    float E;
                                    if (PMODE == 6) {
                                        const float u2 = uu * uu;
                                        E = (u2 * u2) * u2;
                                    } else {
                                        E = P.ev_eps * ev_pow<PMODE>(P.ev_sigma * uu, P.ev_power);
                                    }
                                    if (!NOENERGY) eb = fmaf(E, in, eb);
                                    fs = E * (uu * rinv);
                                }
                                if (GAUSS) {
                                    const float gg = arow[s * 8 + lj] * __builtin_amdgcn_exp2f(r2t * P.g_c2);
                                    if (!NOENERGY) gb = fmaf(-gg, in, gb);
                                    fs = fmaf(-gg, g_k, fs);
                                }
                                fs *= in;
                                fx[s] = fmaf(fs, dx, fx[s]);
                                fy[s] = fmaf(fs, dy, fy[s]);
                                fz[s] = fmaf(fs, dz, fz[s]);
                                fjx = fmaf(fs, dx, fjx);
                                fjy = fmaf(fs, dy, fjy);
                                fjz = fmaf(fs, dz, fjz);
                            }
                            const bool self = (jslot >> 3) == own_lc;
                            const bool big = fmaxf(fmaxf(fabsf(fjx), fabsf(fjy)), fabsf(fjz)) >= fix_lim;
                            // Nearly every batch holds neither a bead of the i-cluster itself nor a sum too large for the
                            // fixed-point window: one wave-uniform test takes it past the ~12 operations of both.
                            if (__builtin_expect(__ballot(self || big) == 0ull, 1)) {
                                if (!NOENERGY) {
                                    if (EV) ee = DD ? fmaf(eb, wl, ee) : ee + eb;
                                    if (GAUSS) eg = DD ? fmaf(gb, wl, eg) : eg + gb;
                                }
                                if (!(diag & 2)) { // reaction on the j beads (sign and unit: at the flush)
                                    atomicAdd(sfx + jslot, cvt_nearest(fjx * fix_k));
                                    atomicAdd(sfx + fstr + jslot, cvt_nearest(fjy * fix_k));
                                    atomicAdd(sfx + 2 * fstr + jslot, cvt_nearest(fjz * fix_k));
                                }
                                continue;
                            }
                            if (!NOENERGY) {
                                // self tile: both orders of a pair were swept (weight 1/2), and the r = 0 pair -- swept with
                                // the rest -- goes out again in the batch it came in with (its value, eself, is formed by
                                // the very operations of the loop: what is left of it is the rounding of the up to 7 small
                                // terms that shared an accumulator with it, ~1e-4 kJ/mol per bead, random in sign)
                                if (EV) ee = fmaf(self ? eb - eself : eb, (self ? 0.5f : 1.f) * wl, ee);
                                if (GAUSS) eg = fmaf(self ? gb + arow[(jslot & 7) * 8 + lj] : gb, (self ? 0.5f : 1.f) * wl, eg);
                            }
                            if (!(diag & 2)) {
                                if (big && !self && s_jc[jslot >> 3] < n_own_cl) { // rare (overlapping beads): straight to global memory
                                    const int gs = s_jc[jslot >> 3] * kCl + (jslot & 7);
                                    atomicAdd(fsort + gs, -pscale * fjx);
                                    atomicAdd(fsort + fstride + gs, -pscale * fjy);
                                    atomicAdd(fsort + 2 * fstride + gs, -pscale * fjz);
                                }
                                // lanes of the self tile (and the large sums) add into the dummy cluster: no branch around the
                                // adds, so the j-side FMAs stay in the pair loop instead of keeping all eight (fs, d) sets
                                // alive behind it
                                const int tslot = (self || big) ? cap * 8 + (lane & 7) : jslot;
                                atomicAdd(sfx + tslot, cvt_nearest(fjx * fix_k));
                                atomicAdd(sfx + fstr + tslot, cvt_nearest(fjy * fix_k));
                                atomicAdd(sfx + 2 * fstr + tslot, cvt_nearest(fjz * fix_k));
                            }
                        }
                    }
                    nlist = 0;
                    wave_lds_sync();
#ifdef MMX_N3_TIMING
                    t_sweep += wall_clock64() - tsw0;
#endif
                }
            }
#ifdef MMX_N3_TIMING
            const unsigned long long tf0 = wall_clock64();
#endif
            // ---- i side: fold over the wave; lane s (< 8) ends up owning bead s of the i-cluster.  Transposed butterfly: every
            // step halves the values a lane carries (lane bit k picks which half it keeps summing), so the 8 sums of a
            // component cost ~27 operations instead of 8 full wave reductions (56)
            const float ofx = fold8(fx, lane) * pscale, ofy = fold8(fy, lane) * pscale, ofz = fold8(fz, lane) * pscale;
            if (lane < kCl && own_w >= 0 && i_own && !(diag & 8)) {
                atomicAdd(fsort + icl * kCl + lane, ofx);
                atomicAdd(fsort + fstride + icl * kCl + lane, ofy);
                atomicAdd(fsort + 2 * fstride + icl * kCl + lane, ofz);
            }
            if (!NOENERGY) { // per lane; summed over the wave once, at the end of the kernel
                acc_ev += (double)ee;
                acc_g += (double)eg;
            }
#ifdef MMX_N3_TIMING
            {
                const unsigned long long tv1 = wall_clock64();
                t_visit += tv1 - tv0;
                t_fold += tv1 - tf0;
                N3_TRACE(0, v, gi, tv0, tv1, cnt_w[3] + cnt_w[7] - nb0);
            }
#endif
        }
        // ---- out of i-clusters: leave the unit; the last wave to do so opens the flush and stages unit v + 2
        wg_lds_release(); // this wave's adds into the window, before it is counted out
        __builtin_amdgcn_wave_barrier();
        int d = 0;
        if (lane == 0) d = atomicAdd(&ctl.done[p], 1);
        d = __builtin_amdgcn_readfirstlane(d);
        if (d == kN3Waves - 1) {
            if (lane == 0) {
                ctl.grab[p] = 0;
                ctl.done[p] = 0;
                ctl.fl_nwin[p] = nwin;
                ctl.fl_ids[p] = D_ib;
                ctl.fl_total[p] = (nwin * 8 + 255) >> 8;
                ctl.fl_next[p] = 0;
                ctl.fl_done[p] = 0;
            }
            wg_lds_release(); // the job description, before the epoch that opens it
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) vepoch[p] = v; // the job is open
#ifdef MMX_N3_TIMING
            const unsigned long long ts0 = wall_clock64();
            N3_TRACE(3, v, 0, ts0, 0, 0);
#endif
            stage_unit(v + 2);
#ifdef MMX_N3_TIMING
            {
                const unsigned long long ts1 = wall_clock64();
                t_stage += ts1 - ts0;
                n_stage++;
                N3_TRACE(2, v + 2, 0, ts0, ts1, 0);
            }
#endif
            if (lane == 0 && atomicAdd(&ctl.fl_done[p], 1) == ctl.fl_total[p]) {
                vready[p] = v + 2;
#ifdef MMX_N3_TIMING
                N3_TRACE(4, v + 2, 0, wall_clock64(), 0, 0);
#endif
            }
        }
    }
#ifdef MMX_N3_TIMING
    if (lane == 0) {
        g_n3_t[blockIdx.x * 20 + 1 + wave] = wall_clock64();
        unsigned *gw_ = g_n3_w + (blockIdx.x * 16 + wave) * 8;
        gw_[0] = (unsigned)t_wait;
        gw_[1] = (unsigned)t_visit;
        gw_[2] = (unsigned)t_stage;
        gw_[3] = (unsigned)n_stage;
        gw_[4] = (unsigned)t_setup;
        gw_[5] = (unsigned)t_sweep;
        gw_[6] = (unsigned)t_fold;
        if (diag & 128) // (4096 waves x 8 contended atomics: a launch that counts is not one to time)
            for (int k = 0; k < 8; ++k) atomicAdd(&g_n3_c[k], (unsigned long long)cnt_w[k]);
    }
#endif
    acc_ev = (double)escale * wave_sum(acc_ev);
    acc_g = wave_sum(acc_g);
    if (lane == 0) {
        s_e[0][wave] = acc_ev;
        s_e[1][wave] = acc_g;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int w = 0; w < kN3Waves; ++w) {
            a += s_e[0][w];
            b += s_e[1][w];
        }
        part[P_EV * kPartStride + blockIdx.x] = a;
        part[P_GAUSS * kPartStride + blockIdx.x] = b;
#ifdef MMX_N3_TIMING
        g_n3_t[blockIdx.x * 20 + 17] = wall_clock64();
#endif
    }
}

// g[bead] -= fsort[slot] for every real bead of the cluster list, fsort back to zero for the next evaluation, and the
// item queue rewound (the pair kernel may be launched again on the same cell build: mmx_time_kernel).
// Decomposed runs: g holds the owned beads only; what landed on a ghost's slot is dropped (its owner computes it).
__global__ __launch_bounds__(256) void k_nb_n3_unsort(const int *__restrict__ sbead, float *__restrict__ fsort,
                                                      const int fstride, float *__restrict__ g,
                                                      MinState *__restrict__ st, const Own own) {
    if (st->phase >= PH_DONE) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->n3_queue = 0;
    const int nsl = st->n_clusters_own * kCl; // (decomposed ranks: the ghosts' clusters lie behind and are never written)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nsl; i += gridDim.x * 256) {
        const int bead = sbead[i];
        const float fx = fsort[i], fy = fsort[fstride + i], fz = fsort[2 * fstride + i];
        fsort[i] = 0.f;
        fsort[fstride + i] = 0.f;
        fsort[2 * fstride + i] = 0.f;
        const int li = own.local(bead);
        if (li >= 0) {
            float *gb = g + 3 * (size_t)li; // the bonded terms wrote the gradient first
            gb[0] -= fx;
            gb[1] -= fy;
            gb[2] -= fz;
        }
    }
}

} // namespace mmx
