// mmx_dd.hpp -- K7: ghost-bead halo of a decomposed run (BASELINE config 5).
//
// Rank r owns the bead slice [own_lo, own_lo + n_own) -- the Hilbert start makes index slices spatially compact -- and
// needs, per evaluation, the positions of (a) every foreign bead within the pair cutoff of its owned beads, (b) the two
// backbone neighbours beyond each slice end and (c) the loop partners of its owned beads.  Instead of gathering every
// position on every rank, each owner SENDS what the others need, and WHAT they need is decided by geometry:
//
//   need-maps.  Space is cut into a coarse grid of 64^3 cells of edge >= reach / 2 (reach = cutoff + skin; one 64-bit
//     word per (z, y) row, x = the bit: 32 KB per rank).  A rank marks the cells its owned beads sit in
//     (k_dd_occupancy), grows the set by ceil(reach / edge) <= 2 cells in every direction (k_dd_dilate: "I need every
//     bead of these cells") and the ranks all-gather their maps.  Owned bead i then goes to rank q iff the bit of
//     its cell is set in q's map, or it is a static partner of q ((b), (c): one bit per rank and bead).  A map follows
//     the SHAPE of a slice -- a concave blob of the collapsed globule as well as a brick of the lattice start; round 2
//     tested against the slice's bounding box and sent a third of everybody else.  Cell indices wrap (& 63): a bead
//     outside the grid of the last synchronous rebuild aliases into it, which can only add ghosts, never lose one.
//   lists on the stream.  k_dd_build_lists appends bead ids to one list per destination (wave-aggregated atomics).
//     The whole rebuild -- occupancy, dilation, all-gather of the maps, lists -- is enqueued on the handle's stream with
//     no host round trip; by default it runs before EVERY evaluation ("dd_rebuild_every" = 1), so the lists are exact
//     for the positions they are used with and there is no skin at all.  With a rebuild every K > 1 evaluations the
//     lists hold while no bead has moved more than skin / 2 (k_dd_displacement).
//   messages of host-known capacity.  ncclSend / ncclRecv need their sizes on the host, the lists are sized on the
//     device: a message carries `cap` entries (the length of the list when the capacity was last set + 1/8 ... 1/1 of it
//     + 512), entries beyond the list are padding (bead id -1).  Every all-gather of the maps also carries the senders' current
//     list lengths, so each poll of the host sees the world x world matrix of lengths and both ends of a message
//     re-derive its capacity from the same numbers before a list outgrows it.
//   when it still goes wrong -- a list longer than its message (dd_overflow), or a bead beyond skin / 2 (dd_stale, K > 1)
//     -- the flag rides in the evaluation's one all-reduce, k_decide_reduced decides nothing and halts (PH_HALT) on
//     every rank, the host rebuilds synchronously at the trial point (fresh capacities) and repeats the evaluation.
//   every evaluation: k_dd_pack gathers the listed beads' float4 {x, y, z, id|label}; grouped ncclSend/ncclRecv, one
//     message per pair of ranks that share any; k_dd_unpack scatters what arrived into pos4 by the id each entry
//     carries and writes the ghost id list that the cell build bins next to the owned beads.  No position of a bead
//     outside the halo is touched, sent or binned; nothing but 6 floats per rank is ever all-gathered from everybody.
#pragma once
#include "mmx_cells.hpp"
#include "mmx_common.hpp"

namespace mmx {

constexpr int kDDMaxWorld = 64;                          // static partner masks are one bit per rank
constexpr int kDDWords = kDDGridN * kDDGridN;            // words of a need-map: [z][y]
constexpr int kDDPayload = kDDWords + kDDMaxWorld / 2;   // ... + the rank's send-list lengths (kDDMaxWorld ints)

struct DDOffsets {
    int off[kDDMaxWorld + 1]; // prefix of the per-source message capacities
};
struct DDCaps {
    int cap[kDDMaxWorld]; // entries per message (to / from rank q)
};
// owned bounding box of this rank from k_pack's per-block boxes -> out6 = {lo x y z, hi x y z}; one block
__global__ __launch_bounds__(256) void k_dd_bbox(const float *__restrict__ bbox_part, int nblk, float *__restrict__ out6) {
    __shared__ float s_red[6][4];
    const float big = 3.0e38f;
    float v[6] = {big, big, big, -big, -big, -big};
    for (int b = threadIdx.x; b < nblk; b += 256) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[k] = fminf(v[k], bbox_part[k * nblk + b]);
            v[k + 3] = fmaxf(v[k + 3], bbox_part[(k + 3) * nblk + b]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = wave_min(v[k]);
        v[k + 3] = wave_max(v[k + 3]);
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 6; ++k) s_red[k][threadIdx.x >> 6] = v[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float r = s_red[k][0];
        for (int w = 1; w < 4; ++w) r = k < 3 ? fminf(r, s_red[k][w]) : fmaxf(r, s_red[k][w]);
        out6[k] = r;
    }
}

// Coarse grid from the all-gathered owned boxes: every rank computes the same numbers from the same input.  The grid
// spans the whole system with two cells of margin; its edge is reach / 2 unless the system is larger than 60 such cells.
__global__ void k_dd_grid(const float *__restrict__ boxes, int world, float reach, DDGrid *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int q = 0; q < world; ++q) {
        const float *b = boxes + 6 * q;
        if (!(b[0] <= b[3]) || !(b[1] <= b[4]) || !(b[2] <= b[5])) continue; // a rank without (finite) beads
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], b[k]);
            hi[k] = fmaxf(hi[k], b[k + 3]);
        }
    }
    float ext = 0.f;
    for (int k = 0; k < 3; ++k) {
        if (!(lo[k] <= hi[k])) lo[k] = hi[k] = 0.f;
        ext = fmaxf(ext, hi[k] - lo[k]);
    }
    DDGrid G;
    // edge a hair above reach / 2, radius = ceil(reach / edge) with NO tolerance taken off: a map must reach at least `reach`
    // beyond the cell of an owned bead (round 3 took 1e-4 off before rounding up: with reach / edge in (2, 2.0001] a pair
    // between edge * 2 and reach apart could lose its ghost)
    G.edge = fmaxf(0.50005f * reach, ext * (1.f / (float)(kDDGridN - 4)));
    if (!(G.edge > 1e-6f) || !(G.edge < 1e30f)) G.edge = 1.f;
    G.inv_edge = 1.f / G.edge;
    G.radius = max(1, (int)ceilf(reach * G.inv_edge));
    G.ox = lo[0] - 2.f * G.edge;
    G.oy = lo[1] - 2.f * G.edge;
    G.oz = lo[2] - 2.f * G.edge;
    *out = G;
}

// occ |= the coarse cells of the owned beads (dd_mark, mmx_common.hpp).  The rebuilds on the stream do this inside k_pack, which
// has the new positions in registers; this kernel serves the synchronous rebuild and MD.
__global__ __launch_bounds__(256) void k_dd_occupancy(int n_own, const float *__restrict__ x, const DDGrid *__restrict__ grid,
                                                      unsigned long long *__restrict__ occ, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const DDGrid G = *grid;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < n_own;
    dd_mark(G, occ, act, act ? x[3 * i] : 0.f, act ? x[3 * i + 1] : 0.f, act ? x[3 * i + 2] : 0.f);
}

// need = occ grown by `radius` cells in every direction (indices wrap) -> this rank's part of the payload; the last
// block appends the lengths of the send lists in use (built by the previous rebuild) for the capacity bookkeeping of
// the host.  17 blocks of 256 threads.
__global__ __launch_bounds__(256) void k_dd_dilate(const unsigned long long *__restrict__ occ, const DDGrid *__restrict__ grid,
                                                   unsigned long long *__restrict__ payload, int *__restrict__ send_cnt,
                                                   int world, const MinState *__restrict__ st, const int reset_cnt = 0) {
    if (st->phase >= PH_DONE) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= kDDWords) {
        const int q = t - kDDWords;
        if (q < kDDMaxWorld) {
            reinterpret_cast<int *>(payload + kDDWords)[q] = q < world ? send_cnt[q] : 0;
            if (reset_cnt && q < world) send_cnt[q] = 0; // (the lists that follow count from zero: no memset launch in between)
        }
        return;
    }
    const int R = grid->radius;
    const int z = t / kDDGridN, y = t % kDDGridN;
    unsigned long long acc = 0ull;
    for (int dz = -R; dz <= R; ++dz)
        for (int dy = -R; dy <= R; ++dy)
            acc |= occ[((z + dz) & (kDDGridN - 1)) * kDDGridN + ((y + dy) & (kDDGridN - 1))];
    unsigned long long out = acc;
    for (int r = 1; r <= R; ++r) out |= (acc << r) | (acc >> (64 - r)) | (acc >> r) | (acc << (64 - r));
    payload[t] = out;
}

// Send lists: owned bead i goes to rank q when it is a static partner of q (backbone neighbour beyond a slice end, loop
// partner) or its coarse cell is set in q's need-map.  Lanes of a wave that go to the same rank share one atomic; the
// order inside a list is arbitrary (the receiver's cell build sorts ghosts by position and id).  A list that would
// outgrow its message raises st->dd_overflow and stops growing (the evaluation will be repeated).
__global__ __launch_bounds__(256) void k_dd_build_lists(int n_own, const Own own, int rank, int world,
                                                        const float *__restrict__ x, const DDGrid *__restrict__ grid,
                                                        const unsigned long long *__restrict__ maps /* [world][kDDPayload] */,
                                                        const unsigned long long *__restrict__ static_mask,
                                                        int *__restrict__ send_ids, int slice, int *__restrict__ send_cnt,
                                                        const DDCaps caps, MinState *__restrict__ st,
                                                        int *__restrict__ cntmat = nullptr,
                                                        unsigned long long *__restrict__ occ_zero = nullptr) {
    if (st->phase >= PH_DONE) return;
    // (the occupancy words have been dilated and sent: zero again for the pack of the next evaluation that marks them)
    if (occ_zero)
        for (int t = blockIdx.x * 256 + threadIdx.x; t < kDDWords; t += gridDim.x * 256) occ_zero[t] = 0ull;
    // rebuilds on the stream: block 0 also copies the list lengths that came with the maps -- rank r's send list for q as it
    // was until now -- into the world x world matrix the host reads at its next poll
    if (cntmat && blockIdx.x == 0)
        for (int t = threadIdx.x; t < world * world; t += 256)
            cntmat[t] = reinterpret_cast<const int *>(maps + (size_t)(t / world) * kDDPayload + kDDWords)[t % world];
    const DDGrid G = *grid;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < n_own;
    int word = 0, bit = 0;
    unsigned long long sm = 0ull;
    if (act) {
        dd_cell(G, x[3 * i], x[3 * i + 1], x[3 * i + 2], word, bit);
        if (static_mask) sm = static_mask[i];
    }
    const int lane = threadIdx.x & 63;
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        const bool need = act && ((((sm >> q) | (maps[(size_t)q * kDDPayload + word] >> bit)) & 1ull) != 0ull);
        const unsigned long long m = __ballot(need);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&send_cnt[q], __popcll(m));
        base = __shfl(base, leader, 64);
        if (base + __popcll(m) > caps.cap[q]) { // wave-uniform
            if (lane == leader) st->dd_overflow = 1;
            continue;
        }
        if (need) send_ids[(size_t)q * slice + base + __popcll(m & ((1ull << lane) - 1ull))] = own.bead(i);
    }
}

// sendbuf[q][k] = pos4[send_ids[q][k]] for the entries of the list, padding (bead id -1) up to the message's capacity;
// grid (blocks, world)
__global__ __launch_bounds__(256) void k_dd_pack(const int *__restrict__ send_ids, const int *__restrict__ send_cnt, int slice,
                                                 const float4 *__restrict__ pos4, float4 *__restrict__ sendbuf,
                                                 const DDCaps caps, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int q = blockIdx.y, cap = caps.cap[q], n = min(send_cnt[q], cap);
    const float4 pad = make_float4(3e18f, 3e18f, 3e18f, __int_as_float(-8 + 2));
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cap; k += gridDim.x * 256)
        sendbuf[(size_t)q * slice + k] = k < n ? pos4[send_ids[(size_t)q * slice + k]] : pad;
}

// What arrived from rank q (recvbuf[q][0 .. cap_q)) goes to pos4 by the bead id it carries; ghost_ids lists the ids in
// arrival order for the cell build (-1: padding).  grid (blocks, world)
__global__ __launch_bounds__(256) void k_dd_unpack(const float4 *__restrict__ recvbuf, const DDOffsets O, int slice,
                                                   float4 *__restrict__ pos4, int *__restrict__ ghost_ids, int n_all,
                                                   const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int q = blockIdx.y, n = O.off[q + 1] - O.off[q];
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        const float4 p = recvbuf[(size_t)q * slice + k];
        int id = __float_as_int(p.w) >> 3;
        if ((unsigned)id >= (unsigned)n_all) id = -1;
        if (id >= 0) pos4[id] = p;
        ghost_ids[O.off[q] + k] = id;
    }
}

// k_dd_unpack + the ghosts' share of the direct build's counting (mmx_build.hpp; the pack has counted the owned beads): every
// arriving ghost is binned on the build's grid -- CLAMPED like an owned bead: two beads within the cutoff of each other are at
// most one cell apart per axis before the clamp and therefore after it, so a grid laid out from an earlier evaluation's box stays
// exact; a ghost beyond the box merely lands in a boundary cell --, takes a place behind the cell's owned beads in its row of
// the slot table, and moves the per-row totals of ghost clusters / large cells.  One workgroup = 256 consecutive entries of one
// peer's message (whole waves: cell_rank).  grid (blocks, world)
struct GhostCount {
    const GridParams *grid;
    int *cell_of, *rank;
    int *count_g;            // [cells] ghosts per cell (this build's set)
    const int *count_o;      // [cells] owned beads per cell: final, the pack ran before
    int *rowclg, *rowbig;    // [rows]
    unsigned long long *keys;
    int cap, cells;
    int split_big;           // two-launch build (k_build_direct_dd, PHASE 1 / 2): rowbig counts the cells of many GHOSTS, not of many beads altogether
};
__global__ __launch_bounds__(256) void k_dd_unpack_count(const float4 *__restrict__ recvbuf, const DDOffsets O, int slice,
                                                         float4 *__restrict__ pos4, int *__restrict__ ghost_ids, int n_all,
                                                         const GhostCount C, MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    __shared__ int s_rows[3 * kRowAgg];
    if (threadIdx.x < 3 * kRowAgg) s_rows[threadIdx.x] = threadIdx.x < kRowAgg ? -1 : 0;
    __syncthreads();
    const GridParams G = *C.grid;
    const float excess = __uint_as_float(st->dd_excess_bits) * 1.0001f + 1e-6f; // (final: the pack ran before)
    const int q = blockIdx.y, n = O.off[q + 1] - O.off[q];
    for (int k0 = blockIdx.x * 256; k0 < n; k0 += gridDim.x * 256) { // (block-uniform trip count: whole waves reach cell_rank)
        const int k = k0 + (int)threadIdx.x;
        int id = -1, c = 0, cx = 0, cy = 0, cz = 0;
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < n) {
            p = recvbuf[(size_t)q * slice + k];
            id = __float_as_int(p.w) >> 3;
            if ((unsigned)id >= (unsigned)n_all) id = -1;
            if (id >= 0) pos4[id] = p;
            ghost_ids[O.off[q] + k] = id;
        }
        bool todo = id >= 0;
        if (todo) { // farther outside the grid box than any owned bead reaches beyond the box shrunk by the cutoff: no partner here
            const float hx = G.ox + (float)G.nx * G.h, hy = G.oy + (float)G.ny * G.h, hz = G.oz + (float)G.nz * G.h;
            const float out = fmaxf(fmaxf(fmaxf(G.ox - p.x, p.x - hx), fmaxf(G.oy - p.y, p.y - hy)), fmaxf(G.oz - p.z, p.z - hz));
            todo = !(out > excess);
            if (!todo) C.cell_of[id] = -1;
        }
        if (todo) {
            cx = cell_coord(p.x, G.ox, G.inv_h, G.nx);
            cy = cell_coord(p.y, G.oy, G.inv_h, G.ny);
            cz = cell_coord(p.z, G.oz, G.inv_h, G.nz);
            c = (cz * G.ny + cy) * G.nx + cx;
            C.cell_of[id] = c;
        }
        const int r = cell_rank(todo, c, todo ? id : 0, C.rank, C.count_g, nullptr, true, C.rowclg, C.rowbig, G.nx, s_rows,
                                C.split_big ? nullptr : C.count_o);
        if (todo) {
            const int at = C.count_o[c] + r; // behind the cell's owned beads
            if (c < C.cells && at < C.cap) C.keys[(size_t)c * C.cap + at] = order_key(p, G, cx, cy, cz, id, true);
            else atomicOr(&st->cell_stale, 2); // the row is too short for this cell: the evaluation is void, longer rows for the repeat
        }
    }
    __syncthreads();
    if (threadIdx.x < kRowAgg) {
        const int row = s_rows[threadIdx.x];
        if (row >= 0) {
            if (s_rows[kRowAgg + threadIdx.x]) atomicAdd(&C.rowclg[row], s_rows[kRowAgg + threadIdx.x]);
            if (s_rows[2 * kRowAgg + threadIdx.x]) atomicAdd(&C.rowbig[row], s_rows[2 * kRowAgg + threadIdx.x]);
        }
    }
}

// Raises st->dd_stale when an owned bead is farther than sqrt(thr2) from where it was when the lists were built.
__global__ __launch_bounds__(256) void k_dd_displacement(int n_own, const float *__restrict__ x, const float *__restrict__ xref,
                                                         float thr2, MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool far = false;
    if (i < n_own) {
        const float dx = x[3 * i] - xref[3 * i], dy = x[3 * i + 1] - xref[3 * i + 1], dz = x[3 * i + 2] - xref[3 * i + 2];
        far = fmaf(dx, dx, fmaf(dy, dy, dz * dz)) > thr2; // (false for NaN: a non-finite state is the controller's to report)
    }
    if (__ballot(far) != 0ull && (threadIdx.x & 63) == 0) st->dd_stale = 1;
}

// Cell id + slot of the owned beads and of the listed ghosts (the decomposed twin of k_cell_count: nothing outside
// the halo is looked at).  Ghosts outside the grid of this build get cell -1.  Whole waves must call (cell_rank).
__global__ __launch_bounds__(256) void k_cell_count_dd(int n_own, const Own own, int n_ghost, const int *__restrict__ ghost_ids,
                                                       const float4 *__restrict__ pos4,
                                                       const GridParams *__restrict__ grid, int *__restrict__ cell_of,
                                                       int *__restrict__ rank, int *__restrict__ count,
                                                       int *__restrict__ count_own, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const GridParams G = *grid;
    const int t = blockIdx.x * 256 + threadIdx.x;
    bool todo = t < n_own + n_ghost;
    int bead = 0, c = 0;
    if (todo) {
        const bool owned = t < n_own;
        bead = owned ? own.bead(t) : ghost_ids[t - n_own];
        todo = bead >= 0; // padding of a message
    }
    if (todo) {
        const bool owned = t < n_own;
        const float4 p = pos4[bead];
        if (!owned) {
            const float fx = (p.x - G.ox) * G.inv_h, fy = (p.y - G.oy) * G.inv_h, fz = (p.z - G.oz) * G.inv_h;
            todo = fx >= 0.f && fx < (float)G.nx && fy >= 0.f && fy < (float)G.ny && fz >= 0.f && fz < (float)G.nz;
        }
        if (todo)
            c = (cell_coord(p.z, G.oz, G.inv_h, G.nz) * G.ny + cell_coord(p.y, G.oy, G.inv_h, G.ny)) * G.nx +
                cell_coord(p.x, G.ox, G.inv_h, G.nx);
        cell_of[bead] = todo ? c : -1;
    }
    cell_rank(todo, c, bead, rank, count, count_own, t < n_own);
}

// Bead ids and sort keys into the cells' slices, for the owned beads and the listed ghosts (twin of k_cell_fill).
__global__ __launch_bounds__(256) void k_cell_fill_dd(int n_own, const Own own, int n_ghost, const int *__restrict__ ghost_ids,
                                                      const int *__restrict__ cell_of, const int *__restrict__ rank,
                                                      const int *__restrict__ start, int *__restrict__ perm,
                                                      unsigned long long *__restrict__ okeys,
                                                      const float4 *__restrict__ pos4,
                                                      const GridParams *__restrict__ grid, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_own + n_ghost) return;
    const int i = t < n_own ? own.bead(t) : ghost_ids[t - n_own];
    if (i < 0) return;
    const int c = cell_of[i];
    if (c < 0) return;
    const GridParams G = *grid;
    const int slot = start[c] + rank[i];
    perm[slot] = i;
    okeys[slot] = order_key(pos4[i], G, c % G.nx, (c / G.nx) % G.ny, c / (G.nx * G.ny), i, t >= n_own);
}

} // namespace mmx
