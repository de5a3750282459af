// mmx_dd.hpp -- K7: ghost-bead halo of a decomposed run (BASELINE config 5).
//
// Rank r owns the bead slice [own_lo, own_lo + n_own) -- the Hilbert start makes index slices spatially compact -- and
// needs, per evaluation, the positions of (a) every foreign bead within the pair cutoff of its owned beads, (b) the two
// backbone neighbours beyond each slice end and (c) the loop partners of its owned beads.  Instead of gathering every
// position on every rank, each owner SENDS what the others need:
//
//   re-decomposition (host-synchronous; at the start of a call, and whenever an evaluation finds the lists stale):
//     all-gather of the positions and of the ranks' owned bounding boxes; every rank lists, per destination q, its
//     owned beads that lie inside q's box grown by cutoff + skin, plus the static partners of (b) and (c); the
//     world x world matrix of list lengths is all-gathered and read by the host: these are the message sizes.
//   every evaluation: k_dd_pack gathers the listed beads' float4 {x, y, z, id|label} into one send buffer per
//     destination; grouped ncclSend/ncclRecv of exactly the listed entries; k_dd_unpack scatters what arrived into
//     pos4 by the id each entry carries and writes the ghost id list that the cell build bins next to the owned beads.
//     No position of a bead outside the halo is touched, sent or binned.
//   validity: the lists hold for as long as no bead has moved more than skin / 2 since they were built
//     (k_dd_displacement, every evaluation, on the owned beads -- every bead is owned by someone).  The flag rides in
//     the evaluation's one all-reduce; when it is up, k_decide_reduced decides nothing and halts (PH_HALT), the host
//     re-decomposes at the trial point and repeats the evaluation.
#pragma once
#include "mmx_cells.hpp"
#include "mmx_common.hpp"

namespace mmx {

constexpr int kDDMaxWorld = 64; // static partner masks are one bit per rank

struct DDOffsets {
    int off[kDDMaxWorld + 1]; // prefix of the per-source ghost counts
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

// Send lists: owned bead i goes to rank q when it is a static partner of q (backbone neighbour beyond a slice end, loop
// partner) or lies inside q's owned box grown by `reach` = cutoff + skin.  Lanes of a wave that go to the same rank
// share one atomic; the order inside a list is arbitrary (the receiver's cell build sorts ghosts by position and id).
__global__ __launch_bounds__(256) void k_dd_build_lists(int n_own, int own_lo, int rank, int world,
                                                        const float *__restrict__ x, const float *__restrict__ boxes,
                                                        float reach, const unsigned long long *__restrict__ static_mask,
                                                        int *__restrict__ send_ids, int slice, int *__restrict__ send_cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < n_own;
    float px = 0.f, py = 0.f, pz = 0.f;
    unsigned long long sm = 0ull;
    if (act) {
        px = x[3 * i];
        py = x[3 * i + 1];
        pz = x[3 * i + 2];
        if (static_mask) sm = static_mask[i];
    }
    const int lane = threadIdx.x & 63;
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        const float *b = boxes + 6 * q;
        const bool need = act && (((sm >> q) & 1ull) || (px >= b[0] - reach && px <= b[3] + reach && py >= b[1] - reach &&
                                                            py <= b[4] + reach && pz >= b[2] - reach && pz <= b[5] + reach));
        const unsigned long long m = __ballot(need);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&send_cnt[q], __popcll(m));
        base = __shfl(base, leader, 64);
        if (need) send_ids[(size_t)q * slice + base + __popcll(m & ((1ull << lane) - 1ull))] = own_lo + i;
    }
}

// sendbuf[q][k] = pos4[send_ids[q][k]]; grid (blocks, world)
__global__ __launch_bounds__(256) void k_dd_pack(const int *__restrict__ send_ids, const int *__restrict__ send_cnt, int slice,
                                                 const float4 *__restrict__ pos4, float4 *__restrict__ sendbuf,
                                                 const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int q = blockIdx.y, n = send_cnt[q];
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256)
        sendbuf[(size_t)q * slice + k] = pos4[send_ids[(size_t)q * slice + k]];
}

// What arrived from rank q (recvbuf[q][0 .. cnt_q)) goes to pos4 by the bead id it carries; ghost_ids lists the ids in
// arrival order for the cell build.  grid (blocks, world)
__global__ __launch_bounds__(256) void k_dd_unpack(const float4 *__restrict__ recvbuf, const DDOffsets O, int slice,
                                                   float4 *__restrict__ pos4, int *__restrict__ ghost_ids,
                                                   const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int q = blockIdx.y, n = O.off[q + 1] - O.off[q];
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
        const float4 p = recvbuf[(size_t)q * slice + k];
        const int id = __float_as_int(p.w) >> 3;
        pos4[id] = p;
        ghost_ids[O.off[q] + k] = id;
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
__global__ __launch_bounds__(256) void k_cell_count_dd(int n_own, int own_lo, int n_ghost, const int *__restrict__ ghost_ids,
                                                       const float4 *__restrict__ pos4,
                                                       const GridParams *__restrict__ grid, int *__restrict__ cell_of,
                                                       int *__restrict__ rank, int *__restrict__ count,
                                                       const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const GridParams G = *grid;
    const int t = blockIdx.x * 256 + threadIdx.x;
    bool todo = t < n_own + n_ghost;
    int bead = 0, c = 0;
    if (todo) {
        const bool owned = t < n_own;
        bead = owned ? own_lo + t : ghost_ids[t - n_own];
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
    cell_rank(todo, c, bead, rank, count);
}

// Bead ids and sort keys into the cells' slices, for the owned beads and the listed ghosts (twin of k_cell_fill).
__global__ __launch_bounds__(256) void k_cell_fill_dd(int n_own, int own_lo, int n_ghost, const int *__restrict__ ghost_ids,
                                                      const int *__restrict__ cell_of, const int *__restrict__ rank,
                                                      const int *__restrict__ start, int *__restrict__ perm,
                                                      unsigned long long *__restrict__ okeys,
                                                      const float4 *__restrict__ pos4,
                                                      const GridParams *__restrict__ grid, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_own + n_ghost) return;
    const int i = t < n_own ? own_lo + t : ghost_ids[t - n_own];
    const int c = cell_of[i];
    if (c < 0) return;
    const GridParams G = *grid;
    const int slot = start[c] + rank[i];
    perm[slot] = i;
    okeys[slot] = order_key(pos4[i], G, c % G.nx, (c / G.nx) % G.ny, c / (G.nx * G.ny), i, own_lo, n_own);
}

} // namespace mmx
