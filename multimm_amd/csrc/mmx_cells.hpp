// mmx_cells.hpp -- K1: position pack + bounding box, cell hash, count, scan, fill, in-cell ordering.
// All device-driven: no host round trip between an L-BFGS trial step and the pair kernel.
#pragma once
#include "mmx_common.hpp"

namespace mmx {

constexpr int kWaveCellMax = 256; // direct build: cells up to this many beads are sorted by ONE wave in registers; rowbig counts the larger ones
constexpr int kRowAgg = 8; // rows a workgroup of the pack sums in LDS before it updates the global row totals (cell_rank)
// Slot of a bead inside its cell: group the lanes of the wave by cell with ballots only (no memory traffic inside
// the loop), then let the first lane of every group issue its atomicAdd in ONE instruction -- one round trip per
// wave instead of one per distinct cell (Hilbert-ordered beads: ~3 distinct cells per wave).  Whole wave must call.
// count_own (decomposed runs): the cell's OWNED beads are counted separately -- they and the ghosts of a cell form
// separate clusters (cell_scan_block, emit_clusters).
// rowcl / rowbig (the direct build, mmx_build.hpp): per ROW of the cell grid (row = c / nx), the clusters of 8 its cells need
// and its cells of more than kWaveCellMax beads, kept current with every increment -- the population goes from `base` to `base + m`, so
// the row's totals move by the difference of the derived quantities -- which is what lets every workgroup of the build find a
// cell's place in the cluster list by itself (a prefix over <= 2048 row totals + one row of populations), without a scan pass.
__device__ __forceinline__ int cell_rank(bool todo, int c, int i, int *__restrict__ rank, int *__restrict__ count,
                                         int *__restrict__ count_own = nullptr, bool owned = true,
                                         int *__restrict__ rowcl = nullptr, int *__restrict__ rowbig = nullptr, const int nx = 1,
                                         int *s_rows = nullptr /* LDS [3][kRowAgg]: row (-1: free), clusters, large cells */,
                                         const int *__restrict__ count_before = nullptr /* decomposed ranks, ghosts: the cell's owned
                                         beads, counted by the pack before -- a cell is large by what it holds altogether */) {
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long own_lanes = count_own ? __ballot(todo && owned) : 0ull;
    unsigned long long pending = __ballot(todo), mine = 0ull;
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int c0 = __shfl(c, leader, 64);
        const unsigned long long same = __ballot(todo && c == c0);
        if (todo && c == c0) mine = same;
        pending &= ~same;
    }
    const int first = todo ? __ffsll((long long)mine) - 1 : lane;
    int base = 0;
    if (todo && lane == first) {
        const int m = __popcll(mine);
        base = atomicAdd(&count[c], m);
        if (count_own && (mine & own_lanes)) atomicAdd(&count_own[c], __popcll(mine & own_lanes));
        if (rowcl) {
            const int nc = base + m, dcl = ((nc + 7) >> 3) - ((base + 7) >> 3), b0 = count_before ? count_before[c] : 0;
            const int dbig = (b0 + nc > kWaveCellMax ? 1 : 0) - (b0 + base > kWaveCellMax ? 1 : 0);
            // The rows are few (121 at the densest): straight global atomics put ~80 updates on every address, one after the
            // other at the memory side (measured: the pack took 27.5 us instead of 13).  A workgroup's 256 chain-consecutive beads
            // sit in a handful of rows: they are summed in LDS first, one global update per row and workgroup at the end.
            if (dcl | dbig) {
                const int row = c / nx;
                int slot = -1;
                if (s_rows) {
                    for (int q = 0; q < kRowAgg && slot < 0; ++q) {
                        int k = s_rows[q];
                        if (k == -1) k = atomicCAS(&s_rows[q], -1, row);
                        if (k == -1 || k == row) slot = q;
                    }
                }
                if (slot >= 0) {
                    if (dcl) atomicAdd(&s_rows[kRowAgg + slot], dcl);
                    if (dbig) atomicAdd(&s_rows[2 * kRowAgg + slot], dbig);
                } else {
                    if (dcl) atomicAdd(&rowcl[row], dcl);
                    if (dbig) atomicAdd(&rowbig[row], dbig);
                }
            }
        }
    }
    base = __shfl(base, first, 64);
    const int r = base + __popcll(mine & lt);
    if (todo) rank[i] = r;
    return r;
}

// 12-bit Hilbert index of a position inside its cell (16 sub-cells per axis; Skilling's axes -> transpose, 4 bits).  Eight
// consecutive beads of a Hilbert-ordered cell form a tighter cluster than eight of a Morton-ordered one (the Z curve
// jumps): measured on gw_200k states, 53 % of the swept lanes inside the cutoff instead of 48 % (DESIGN_HISTORY.md 5c).
__device__ __forceinline__ unsigned spread4(unsigned v) { // abcd -> 00a00b00c00d
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ unsigned hilbert12(unsigned x, unsigned y, unsigned z) {
#pragma unroll
    for (unsigned q = 8u; q > 1u; q >>= 1) {
        const unsigned m = q - 1u;
        if (x & q) x ^= m; // (the step for the first axis exchanges it with itself otherwise)
        if (y & q) x ^= m;
        else {
            const unsigned t = (x ^ y) & m;
            x ^= t;
            y ^= t;
        }
        if (z & q) x ^= m;
        else {
            const unsigned t = (x ^ z) & m;
            x ^= t;
            z ^= t;
        }
    }
    y ^= x; // Gray encode
    z ^= y;
    unsigned t = 0u;
#pragma unroll
    for (unsigned q = 8u; q > 1u; q >>= 1)
        if (z & q) t ^= q - 1u;
    x ^= t;
    y ^= t;
    z ^= t;
    return (spread4(x) << 2) | (spread4(y) << 1) | spread4(z);
}
__device__ __forceinline__ unsigned long long order_key(const float4 p, const GridParams &G, int cx, int cy, int cz,
                                                        int bead, const bool is_ghost) {
    const float fx = ((p.x - G.ox) * G.inv_h - (float)cx) * 16.f;
    const float fy = ((p.y - G.oy) * G.inv_h - (float)cy) * 16.f;
    const float fz = ((p.z - G.oz) * G.inv_h - (float)cz) * 16.f;
    const unsigned qx = (unsigned)min(max((int)fx, 0), 15), qy = (unsigned)min(max((int)fy, 0), 15),
                   qz = (unsigned)min(max((int)fz, 0), 15);
    const unsigned m = hilbert12(qx, qy, qz);
    // owned beads first (multi-GPU: clusters are then all-owned, one mixed, all-ghost), then along the curve, then id
    const unsigned long long ghost = is_ghost ? 1ull : 0ull;
    return (ghost << 63) | ((unsigned long long)m << 32) | (unsigned)bead;
}

// x = xp + step*d (MOVE) or x as given; builds pos4 = {x,y,z, bits((bead<<3)|(label+2))} and the bbox.
// One thread per bead.  Algorithmic traffic: read 12(+24 when MOVE) B, write 16(+12) B per bead.
// COUNT (single-GPU runs): the cell assignment of k_cell_count is done here as well -- the grid of this build
// was fixed by the previous build's scan, so nothing between the two kernels is needed and one launch goes away.
// DIR (minimizer, trial moves): when the previous evaluation ended an iteration (st->accepted) the new L-BFGS
// direction d = sum_a coef[a] B_a over the basis {S_0..5, Y_0..5, g} is formed here, per bead, right before it
// is used -- with xp <- x and gp <- g -- instead of in a separate elementwise kernel (same fp32 operations in
// the same order as the former stand-alone direction kernel, so the same bits; one launch and one pass over xp, d fewer per iteration).
struct DirArgs {
    const float *g;
    float *gp;
    const float *S, *Y; // [MMX_M] vectors of nv floats
    size_t nv;
};
// Kept cell structure ("cell_reuse"): xref = the positions the structure in use was built from.  mode 1: an evaluation that
// keeps it -- a bead farther than sqrt(thr2) from xref voids the evaluation (st->cell_stale); mode 2: an evaluation that
// builds anew -- xref <- x (mode 3: the same without a previous reference to measure from).  Modes 1 and 2 record the largest squared displacement (st->disp2_bits) for the host's choice of how
// many evaluations a structure may serve.
struct RefArgs {
    float *xref;
    int mode;
    float thr2;
    int move_track = 0; // decomposed ranks (option dd_adaptive): record the largest squared trial move (MinState::dd_move2_bits)
    int dd = 0; // decomposed ranks: xref = where the owned beads were when the ghost lists were built (dd_rebuild_every > 1); a bead
                // beyond half the skin raises MinState::dd_stale (k_dd_displacement's job, without its launch and its pass over x)
};
// Slot table (trial moves of a single-domain minimization): the pack itself writes the bead's 64-bit sort key into its cell's
// slot -- keys[cell * cap + rank in cell], cap from the largest cell the last poll saw -- which is what k_cell_fill would
// write into the cell's slice of the counting sort after the scan: that launch disappears (5 us + a dispatch gap per
// evaluation).  A cell beyond the table or fuller than cap voids the evaluation (st->cell_stale bit 1: PH_HALT, the host
// enlarges the table and the evaluation is repeated).
constexpr int kDirectMaxRows = 2048; // rows (ny * nz) of a grid the direct build handles
struct SlotArgs {
    unsigned long long *keys;
    int cap, cells;
    int *rowcl = nullptr, *rowbig = nullptr; // direct build (mmx_build.hpp): per-row totals kept by the pack (cell_rank)
    int force_void = 0;                      // tests (inject_fault bit 6): behave as if the grid were beyond the direct build
    int max_rows = kDirectMaxRows;           // rows (ny * nz) the build that follows can take
    float dd_margin = -1.f;                  // decomposed ranks: the cutoff the grid box was grown by (>= 0: record the owned beads' excess)
    int key32 = 0;                           // single domain, <= 2^20 beads: 32-bit keys (12-bit Hilbert index << 20 | bead): same order, half the sort
};

template <bool MOVE, bool COUNT = false, bool DIR = false>
__global__ __launch_bounds__(256) void k_pack(int n_own, const Own own, float *__restrict__ x, float *__restrict__ xp,
                                              float *__restrict__ d, const int8_t *__restrict__ labels,
                                              float4 *__restrict__ pos4, float *__restrict__ bbox_part,
                                              const MinState *__restrict__ st,
                                              const GridParams *__restrict__ grid = nullptr,
                                              int *__restrict__ cell_of = nullptr, int *__restrict__ rank = nullptr,
                                              int *__restrict__ count = nullptr, const DirArgs D = DirArgs{},
                                              const RefArgs R = RefArgs{nullptr, 0, 0.f}, MinState *__restrict__ stw = nullptr,
                                              const SlotArgs T = SlotArgs{nullptr, 0, 0},
                                              const DDGrid *__restrict__ ddgrid = nullptr,
                                              unsigned long long *__restrict__ ddocc = nullptr) {
    if (st->phase >= PH_DONE) return;
    __shared__ float s_bb[6][4];
    __shared__ float4 s_xp[MOVE ? 192 : 1], s_d[MOVE ? 192 : 1]; // the block's 768 floats of xp and d
    __shared__ int s_rows[3 * kRowAgg];
    if (COUNT && threadIdx.x < 3 * kRowAgg) s_rows[threadIdx.x] = threadIdx.x < kRowAgg ? -1 : 0;
    if (COUNT && !MOVE) __syncthreads(); // (MOVE: the barrier of the hand-over below)
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // local index of an owned bead
    float px = 0.f, py = 0.f, pz = 0.f;
    const bool act = i < n_own;
    if (MOVE) {
        // the 256 beads of a block own 768 consecutive floats = 192 float4 of the flat vectors: threads 0..191
        // fetch (or, after an accepted step, form) them with coalesced 16-byte accesses and hand them over in LDS
        const int n4 = (3 * n_own + 3) >> 2;
        const int e4 = blockIdx.x * 192 + threadIdx.x;
        if (threadIdx.x < 192 && e4 < n4) {
            float4 xp4, d4;
            if (DIR && st->accepted && st->phase != PH_IDLE) {
                float c[MMX_NBASIS];
#pragma unroll
                for (int b = 0; b < MMX_NBASIS; ++b) c[b] = (float)st->coef[b];
                // ALL 14 loads first, unconditionally (history slots not yet written hold zeros: mmx_minimize clears S and Y):
                // behind `if (c[a] != 0)` every load sat in a basic block of its own and the kernel was 13 dependent memory
                // round trips long (12.7 us at 200 000 beads, 10.5 at 50 000: latency, not bandwidth)
                float4 sv[MMX_M], yv[MMX_M];
#pragma unroll
                for (int a = 0; a < MMX_M; ++a) {
                    sv[a] = reinterpret_cast<const float4 *>(D.S + (size_t)a * D.nv)[e4];
                    yv[a] = reinterpret_cast<const float4 *>(D.Y + (size_t)a * D.nv)[e4];
                }
                const float4 G = reinterpret_cast<const float4 *>(D.g)[e4];
                xp4 = reinterpret_cast<const float4 *>(x)[e4];
                float4 o = make_float4(c[2 * MMX_M] * G.x, c[2 * MMX_M] * G.y, c[2 * MMX_M] * G.z, c[2 * MMX_M] * G.w);
#pragma unroll
                for (int a = 0; a < MMX_M; ++a) { // same operations in the same order (a zero coefficient adds c * v = 0)
                    o.x = fmaf(c[a], sv[a].x, o.x);
                    o.y = fmaf(c[a], sv[a].y, o.y);
                    o.z = fmaf(c[a], sv[a].z, o.z);
                    o.w = fmaf(c[a], sv[a].w, o.w);
                    o.x = fmaf(c[MMX_M + a], yv[a].x, o.x);
                    o.y = fmaf(c[MMX_M + a], yv[a].y, o.y);
                    o.z = fmaf(c[MMX_M + a], yv[a].z, o.z);
                    o.w = fmaf(c[MMX_M + a], yv[a].w, o.w);
                }
                d4 = o;
                reinterpret_cast<float4 *>(xp)[e4] = xp4;
                reinterpret_cast<float4 *>(D.gp)[e4] = G;
                reinterpret_cast<float4 *>(d)[e4] = o;
            } else {
                xp4 = reinterpret_cast<const float4 *>(xp)[e4];
                d4 = reinterpret_cast<const float4 *>(d)[e4];
            }
            s_xp[threadIdx.x] = xp4;
            s_d[threadIdx.x] = d4;
        }
        __syncthreads();
    }
    if (act) {
        if (MOVE) {
            const double step = st->step;
            const float *lx = reinterpret_cast<const float *>(s_xp) + 3 * threadIdx.x;
            const float *ld = reinterpret_cast<const float *>(s_d) + 3 * threadIdx.x;
            px = (float)((double)lx[0] + step * (double)ld[0]);
            py = (float)((double)lx[1] + step * (double)ld[1]);
            pz = (float)((double)lx[2] + step * (double)ld[2]);
            x[3 * i] = px;
            x[3 * i + 1] = py;
            x[3 * i + 2] = pz;
        } else {
            px = x[3 * i];
            py = x[3 * i + 1];
            pz = x[3 * i + 2];
        }
        const int bead = own.bead(i);
        const int w = (bead << 3) | ((int)labels[bead] + 2);
        pos4[bead] = make_float4(px, py, pz, __int_as_float(w));
    }
    if (MOVE && R.move_track) { // how far this trial move takes the owned beads from the last accepted point
        float m2 = 0.f;
        if (act) {
            const float *lx = reinterpret_cast<const float *>(s_xp) + 3 * threadIdx.x;
            const float dx = px - lx[0], dy = py - lx[1], dz = pz - lx[2];
            m2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
            if (!(m2 >= 0.f)) m2 = 3e38f;
        }
        m2 = wave_max(m2);
        if ((threadIdx.x & 63) == 0 && m2 > 0.f && __float_as_uint(m2) > stw->dd_move2_bits) atomicMax(&stw->dd_move2_bits, __float_as_uint(m2));
    }
    if (R.mode) { // (block-uniform) displacement from where the cell structure in use binned the beads
        float d2 = 0.f;
        if (act) {
            const float dx = px - R.xref[3 * i], dy = py - R.xref[3 * i + 1], dz = pz - R.xref[3 * i + 2];
            d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
            if (!(d2 >= 0.f)) d2 = 3e38f; // NaN: never "within the skin"
            if (R.mode == 3) d2 = 0.f; // no reference yet: store only
            if (R.mode >= 2) {
                R.xref[3 * i] = px;
                R.xref[3 * i + 1] = py;
                R.xref[3 * i + 2] = pz;
            }
        }
        const float m = wave_max(d2);
        if ((threadIdx.x & 63) == 0 && m > 0.f) {
            if (R.dd) {
                if (R.mode == 1 && m > R.thr2) stw->dd_stale = 1;
            } else {
                if (__float_as_uint(m) > stw->disp2_bits) atomicMax(&stw->disp2_bits, __float_as_uint(m));
                if (R.mode == 1 && m > R.thr2) atomicOr(&stw->cell_stale, 1);
            }
        }
    }
    if (COUNT) { // the owned beads (single domain: all of them, bead == i; decomposed ranks with the direct build: the ghosts are
                 // counted when they arrive, k_dd_unpack_count)
        const GridParams G = *grid;
        int c = 0, cx = 0, cy = 0, cz = 0;
        const int bead_c = act ? own.bead(i) : 0;
        if (act) {
            cx = cell_coord(px, G.ox, G.inv_h, G.nx);
            cy = cell_coord(py, G.oy, G.inv_h, G.ny);
            cz = cell_coord(pz, G.oz, G.inv_h, G.nz);
            c = (cz * G.ny + cy) * G.nx + cx;
            cell_of[bead_c] = c;
        }
        const int r = cell_rank(act, c, bead_c, rank, count, nullptr, true, T.rowcl, T.rowbig, G.nx, T.rowcl ? s_rows : nullptr);
        if (T.rowcl && i == 0) stw->n3_items = 0; // (direct build: no scan resets the work-item count the item builders append to)
        // the direct build reads a row of populations with one lane per cell and keeps the row prefixes in LDS: a grid beyond
        // that voids the evaluation (the host falls back to the scan-based build for the rest of the call)
        if (T.rowcl && i == 0 && (G.nx > 64 || G.ny * G.nz > T.max_rows || T.force_void)) atomicOr(&stw->cell_stale, 4);
        if (T.dd_margin >= 0.f) { // how far owned beads reach beyond the grid box shrunk by the cutoff (MinState::dd_excess_bits)
            float ex = 0.f;
            if (act) {
                const float hx = G.ox + (float)G.nx * G.h, hy = G.oy + (float)G.ny * G.h, hz = G.oz + (float)G.nz * G.h, m = T.dd_margin;
                ex = fmaxf(fmaxf(fmaxf(G.ox + m - px, px - (hx - m)), fmaxf(G.oy + m - py, py - (hy - m))),
                           fmaxf(fmaxf(G.oz + m - pz, pz - (hz - m)), 0.f));
                if (!(ex >= 0.f)) ex = 3e38f; // (NaN: keep every ghost)
            }
            ex = wave_max(ex);
            if ((threadIdx.x & 63) == 0 && ex > 0.f && __float_as_uint(ex) > stw->dd_excess_bits) atomicMax(&stw->dd_excess_bits, __float_as_uint(ex));
        }
        if (T.keys && act) {
            if (c < T.cells && r < T.cap) {
                const unsigned long long k64 = order_key(make_float4(px, py, pz, 0.f), G, cx, cy, cz, bead_c, false);
                if (T.key32) reinterpret_cast<unsigned *>(T.keys)[(size_t)c * T.cap + r] = ((unsigned)(k64 >> 32) << 20) | (unsigned)bead_c;
                else T.keys[(size_t)c * T.cap + r] = k64;
            }
            else
                atomicOr(&stw->cell_stale, 2); // the table is too small for this state: the evaluation is void (k_decide halts)
        }
    }
    // decomposed ranks, list rebuild on the stream: the occupancy of the need-map from the positions just formed (k_dd_occupancy's job)
    if (ddocc) dd_mark(*ddgrid, ddocc, act, px, py, pz);
    // Block bounding box -> bbox_part[k][block] (k = minx,miny,minz,maxx,maxy,maxz); no atomics.
    const float big = 3.0e38f;
    const float fin = (act && fabsf(px) < big && fabsf(py) < big && fabsf(pz) < big) ? 1.f : 0.f;
    float v[6] = {fin ? px : big, fin ? py : big, fin ? pz : big, fin ? px : -big, fin ? py : -big, fin ? pz : -big};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = wave_min(v[k]);
        v[k + 3] = wave_max(v[k + 3]);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_bb[k][wave] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float r = s_bb[k][0];
        for (int w = 1; w < 4; ++w) r = k < 3 ? fminf(r, s_bb[k][w]) : fmaxf(r, s_bb[k][w]);
        bbox_part[k * gridDim.x + blockIdx.x] = r;
    }
    if (COUNT && T.rowcl && threadIdx.x >= 64 && threadIdx.x < 64 + kRowAgg) { // (the barrier above: every wave's updates are in)
        const int q = threadIdx.x - 64, row = s_rows[q];
        if (row >= 0) {
            if (s_rows[kRowAgg + q]) atomicAdd(&T.rowcl[row], s_rows[kRowAgg + q]);
            if (s_rows[2 * kRowAgg + q]) atomicAdd(&T.rowbig[row], s_rows[2 * kRowAgg + q]);
        }
    }
}

// Reduces the per-block bounding boxes of k_pack (nblk blocks) into a grid; called by one block.
template <int BLOCK>
__device__ __forceinline__ GridParams grid_from_parts(const float *__restrict__ bbox_part, int nblk, float hmin,
                                                       int maxcells, float *s_red /* [6][BLOCK/64] */,
                                                       float expand = 0.f) {
    const float big = 3.0e38f;
    float v[6] = {big, big, big, -big, -big, -big};
    for (int b = threadIdx.x; b < nblk; b += BLOCK) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[k] = fminf(v[k], bbox_part[k * nblk + b]);
            v[k + 3] = fmaxf(v[k + 3], bbox_part[(k + 3) * nblk + b]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = wave_min(v[k]);
        v[k + 3] = wave_max(v[k + 3]);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) s_red[k * (BLOCK / 64) + wave] = v[k];
    }
    __syncthreads();
    float r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        r[k] = s_red[k * (BLOCK / 64)];
        for (int w = 1; w < BLOCK / 64; ++w)
            r[k] = k < 3 ? fminf(r[k], s_red[k * (BLOCK / 64) + w]) : fmaxf(r[k], s_red[k * (BLOCK / 64) + w]);
    }
    return grid_from_box(r[0] - expand, r[1] - expand, r[2] - expand, r[3] + expand, r[4] + expand,
                         r[5] + expand, hmin, maxcells);
}

// Exact grid for the current positions of the owned beads, grown by `expand` on every side (multi-GPU:
// expand = cutoff, so that every bead within the cutoff of an owned bead falls inside the grid).
__global__ __launch_bounds__(256) void k_grid_init(const float *__restrict__ bbox_part, int nblk, float hmin,
                                                   int maxcells, float expand, GridParams *__restrict__ grid,
                                                   const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    __shared__ float s_red[6 * 4];
    const GridParams G = grid_from_parts<256>(bbox_part, nblk, hmin, maxcells, s_red, expand);
    if (threadIdx.x == 0) *grid = G;
}

// Cell id per bead + per-cell population.  The grid may stem from the previous evaluation's bounding
// box: coordinates are clamped, which keeps every pair within the cutoff inside the 27-cell stencil
// (a bead outside the box by more than one cell edge cannot be within the cutoff of an interior cell
// two layers in).  Lanes of a wave that share a cell issue one atomicAdd (Hilbert-ordered beads: ~3
// distinct cells per wave) and derive their slot in the cell from the returned base.
__global__ __launch_bounds__(256) void k_cell_count(int n_all, const Own own,
                                                    const float4 *__restrict__ pos4,
                                                    const GridParams *__restrict__ grid, int *__restrict__ cell_of,
                                                    int *__restrict__ rank, int *__restrict__ count,
                                                    const MinState *__restrict__ st, int *__restrict__ count_own = nullptr) {
    if (st->phase >= PH_DONE) return;
    const GridParams G = *grid;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool todo = i < n_all;
    int c = 0;
    const bool owned = own.owns(i);
    if (todo) {
        const float4 p = pos4[i];
        if (!owned) {
            // ghost candidate (multi-GPU): kept only when strictly inside the (cutoff-expanded) grid
            const float fx = (p.x - G.ox) * G.inv_h, fy = (p.y - G.oy) * G.inv_h, fz = (p.z - G.oz) * G.inv_h;
            todo = fx >= 0.f && fx < (float)G.nx && fy >= 0.f && fy < (float)G.ny && fz >= 0.f && fz < (float)G.nz;
        }
        if (todo) {
            const int cx = cell_coord(p.x, G.ox, G.inv_h, G.nx);
            const int cy = cell_coord(p.y, G.oy, G.inv_h, G.ny);
            const int cz = cell_coord(p.z, G.oz, G.inv_h, G.nz);
            c = (cz * G.ny + cy) * G.nx + cx;
        }
        cell_of[i] = todo ? c : -1;
    }
    cell_rank(todo, c, i, rank, count, count_own, owned);
}

// Clusters of a cell of k beads, ko of them owned: owned beads and ghosts never share a cluster (single-domain runs: ko = k).
__device__ __forceinline__ int cell_clusters(int k, int ko) { return ((ko + 7) >> 3) + ((k - ko + 7) >> 3); }

// Single-block exclusive scan of the cell populations (bead offsets) and of the per-cell chunk
// counts (work-item offsets); publishes the item count and the grid of the next build.
struct ScanArgs {
    const float *bbox_part;
    int nblk;
    float hmin;
    int maxcells;
    const int *count;
    int *start, *istart, *cstart, *biglist;
    const GridParams *grid;
    GridParams *grid_next;
    const int *count_own; // decomposed runs: owned beads per cell (nullptr: every bead is owned)
    float expand_next = 0.f; // decomposed ranks: the NEXT build's grid is this evaluation's owned box grown by the cutoff (the direct
                             // build bins on it, ghosts included: mmx_build.hpp)
    int split;            // decomposed ranks running the half-shell kernel: the clusters of owned beads come first in the cluster
                          // list (cstart: their offsets per cell), the ghosts' clusters behind them (istart: THEIR offsets per
                          // cell, counted from the first ghost cluster) -- every owned-ghost pair is then taken from the owned
                          // side and ghost clusters are never i-clusters (mmx_nonbonded_n3.hpp)
};
template <int CHUNK>
__device__ __forceinline__ void cell_scan_block(const ScanArgs &a, MinState *__restrict__ st) {
    const float *__restrict__ bbox_part = a.bbox_part;
    const int nblk = a.nblk, maxcells = a.maxcells;
    const float hmin = a.hmin;
    const int *__restrict__ count = a.count, *__restrict__ count_own = a.count_own;
    int *__restrict__ start = a.start, *__restrict__ istart = a.istart, *__restrict__ cstart = a.cstart,
                      *__restrict__ biglist = a.biglist;
    const GridParams *__restrict__ grid = a.grid;
    GridParams *__restrict__ grid_next = a.grid_next;
    __shared__ int s_a[64], s_b[64], s_c[64], s_d[64], s_m[16];
    __shared__ float s_red[6 * 16];
    const GridParams G = *grid;
    // grid of the NEXT build from this evaluation's bounding box (see k_cell_count on staleness)
    const GridParams GN = grid_from_parts<1024>(bbox_part, nblk, hmin, maxcells, s_red, a.expand_next);
    const int t = threadIdx.x;
    const int per = (G.ncells + 1023) / 1024;
    const int c0 = t * per, c1 = min(c0 + per, G.ncells);
    int sa = 0, sb = 0, sc = 0, sd = 0, mx = 0; // beads, 64-bead chunks, clusters, cells of > 64 beads
    for (int c = c0; c < c1; ++c) {
        const int k = count[c];
        const int ko = count_own ? count_own[c] : k;
        sa += k;
        sb += a.split ? ((k - ko + 7) >> 3) : (k + CHUNK - 1) / CHUNK;
        sc += a.split ? ((ko + 7) >> 3) : cell_clusters(k, ko);
        sd += k > 64 ? 1 : 0;
        mx = max(mx, k);
    }
    mx = wave_max_i(mx);
    if ((t & 63) == 0) s_m[t >> 6] = mx;
    // inclusive scan over the 1024 partials: shuffles inside each wave, then the 16 wave totals by wave 0
    const int lane = t & 63, wave = t >> 6;
    int ia = sa, ib = sb, ic = sc, id = sd;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64), uc = __shfl_up(ic, o, 64),
                  ud = __shfl_up(id, o, 64);
        if (lane >= o) {
            ia += ua;
            ib += ub;
            ic += uc;
            id += ud;
        }
    }
    if (lane == 63) {
        s_a[wave] = ia;
        s_b[wave] = ib;
        s_c[wave] = ic;
        s_d[wave] = id;
    }
    __syncthreads();
    if (wave == 0) {
        int wa = lane < 16 ? s_a[lane] : 0, wb = lane < 16 ? s_b[lane] : 0, wc = lane < 16 ? s_c[lane] : 0,
            wd = lane < 16 ? s_d[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int ua = __shfl_up(wa, o, 64), ub = __shfl_up(wb, o, 64), uc = __shfl_up(wc, o, 64),
                      ud = __shfl_up(wd, o, 64);
            if (lane >= o) {
                wa += ua;
                wb += ub;
                wc += uc;
                wd += ud;
            }
        }
        if (lane < 16) { // inclusive totals up to and including wave `lane`
            s_a[32 + lane] = wa;
            s_b[32 + lane] = wb;
            s_c[32 + lane] = wc;
            s_d[32 + lane] = wd;
        }
    }
    __syncthreads();
    const int oa = wave ? s_a[32 + wave - 1] : 0, ob = wave ? s_b[32 + wave - 1] : 0, oc = wave ? s_c[32 + wave - 1] : 0,
              od = wave ? s_d[32 + wave - 1] : 0;
    const int tot_a = s_a[32 + 15], tot_b = s_b[32 + 15], tot_c = s_c[32 + 15], tot_d = s_d[32 + 15];
    ia += oa;
    ib += ob;
    ic += oc;
    id += od;
    int ra = ia - sa, rb = ib - sb, rcl = ic - sc, rd = id - sd; // exclusive prefixes
    for (int c = c0; c < c1; ++c) {
        const int k = count[c];
        start[c] = ra;
        istart[c] = rb;
        cstart[c] = rcl;
        if (k > 64) biglist[rd++] = c; // cells the order kernel sorts with a whole block, one per block
        const int ko = count_own ? count_own[c] : k;
        ra += k;
        rb += a.split ? ((k - ko + 7) >> 3) : (k + CHUNK - 1) / CHUNK;
        rcl += a.split ? ((ko + 7) >> 3) : cell_clusters(k, ko);
    }
    if (t == 1023) {
        start[G.ncells] = tot_a;
        istart[G.ncells] = tot_b;
        cstart[G.ncells] = tot_c;
        st->n_clusters = a.split ? tot_c + tot_b : tot_c;
        st->n_clusters_own = tot_c;
        st->n_big = tot_d;
        st->n3_items = 0; // k_n3_items (next in the stream) counts them
        st->n3_queue = 0;
        *grid_next = GN;
        int m = 0;
        for (int w = 0; w < 16; ++w) m = max(m, s_m[w]);
        st->n_items = tot_b;
        st->ncells = G.ncells;
        st->max_per_cell = m;
        st->cell_edge = (double)G.h;
    }
}
template <int CHUNK>
__global__ __launch_bounds__(1024) void k_cell_scan(const ScanArgs a, MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    cell_scan_block<CHUNK>(a, st);
}

// Scatter bead ids AND their 64-bit sort keys into their cell's slice (arrival order; k_cell_order makes it
// canonical).  Writing the key here -- where the position is read coalesced by bead -- takes the perm -> pos4 gather
// out of the latency chain of the per-cell sort.
__global__ __launch_bounds__(256) void k_cell_fill(int n_all, const int *__restrict__ cell_of,
                                                   const int *__restrict__ rank, const int *__restrict__ start,
                                                   int *__restrict__ perm, unsigned long long *__restrict__ okeys,
                                                   const float4 *__restrict__ pos4, const GridParams *__restrict__ grid,
                                                   const Own own, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_all) return;
    const int c = cell_of[i];
    if (c < 0) return;
    const GridParams G = *grid;
    const int slot = start[c] + rank[i];
    perm[slot] = i;
    okeys[slot] = order_key(pos4[i], G, c % G.nx, (c / G.nx) % G.ny, c / (G.nx * G.ny), i, !own.owns(i));
}

// Orders every cell's beads along a Hilbert curve of 16^3 sub-cells (ties by bead id: bitwise
// reproducible summation order, and 8 consecutive entries form a spatially compact cluster), emits the
// cell's work items {cell, chunk}, the padded cluster positions and cluster boxes and clears count for
// the next build.  Two grid-stride passes in one launch: cells of <= 64 beads are sorted by ONE WAVE in
// registers (bitonic over __shfl_xor); larger cells by the WHOLE BLOCK in LDS (bitonic, <= CAP beads).  CAP is
// 1024 (8 KB of LDS: 8 blocks per CU, one cell per block in flight for ~2000 cells) when the largest cell of the
// previous poll leaves >= 60 % headroom, else 4096; a cell above CAP keeps arrival order (still correct, not
// bitwise reproducible) and is counted in st->order_fallbacks.

// padded cluster positions + boxes of one sorted cell; `nthr` cooperating threads, thread index `tid`
// `no`: owned beads of the cell (they come first in the sorted order: order_key); the ghosts start a new cluster, so that
// a cluster is all-owned or all-ghost (the half-shell kernel weights energies and drops reactions per cluster).
// cbg: first cluster of the cell's ghosts (split layout), or -1: right behind the owned ones.
__device__ __forceinline__ void emit_clusters(int c, int s, int cnt, int no, int cb, const int cbg, const int *__restrict__ perm,
                                              const float4 *__restrict__ pos4, float4 *__restrict__ spos4,
                                              float4 *__restrict__ cl_lo, float4 *__restrict__ cl_hi, int tid,
                                              int nthr, const Own &own,
                                              const unsigned long long *keys = nullptr, int *__restrict__ sbead = nullptr,
                                              int *__restrict__ slot_of = nullptr, const int cap_slots = 0x7fffffff,
                                              MinState *__restrict__ st_err = nullptr, const int n_beads = 0x7fffffff) {
    // cap_slots (a multiple of 8): slots of the cluster list.  Offsets derive from the cell counters; counters that a void
    // evaluation left behind, or any corruption of them, must end in an error code, never in a store past the arrays
    const int o8 = ((no + 7) >> 3) << 3; // slots of the owned clusters
    const int ncl = cell_clusters(cnt, no);
    for (int e = tid; e < ncl * 8; e += nthr) {
        float4 p = make_float4(1e18f, 1e18f, 1e18f, __int_as_float(-8)); // padding: far away, bead id -1
        int nown = 0, bead = -1;
        const int src = e < o8 ? (e < no ? e : -1) : (e - o8 < cnt - no ? no + (e - o8) : -1); // place in the sorted cell
        const bool real = src >= 0;
        if (real) {
            // sorted bead id: from the sorted keys still in LDS when the caller has them (no global round trip)
            const int b = keys ? (int)(unsigned)(keys[src] & 0xffffffffull) : perm[s + src];
            if ((unsigned)b < (unsigned)n_beads) {
                p = pos4[b]; // as it is: the pair kernels see the state bit for bit (k_nb_clusters_j)
                bead = b;
            } else if (st_err) { // (a key that is no bead: counters and keys of different builds)
                atomicOr(&st_err->kernel_error, (int)KERR_BOUNDS);
            }
        }
        const int lb = bead >= 0 ? own.local(bead) : -1;
        nown = lb >= 0 ? 1 : 0;
        // slot of entry e: the owned clusters at cb, the ghosts' either right behind them or in their own region
        const size_t sl = (cbg < 0 || e < o8) ? (size_t)cb * 8 + e : (size_t)cbg * 8 + (e - o8);
        if (sl >= (size_t)cap_slots) { // (whole clusters: the eight lanes of a cluster agree)
            if (st_err && (e & 7) == 0) atomicOr(&st_err->kernel_error, (int)KERR_BOUNDS);
            continue;
        }
        spos4[sl] = p;
        if (sbead) sbead[sl] = bead; // slot -> bead (k_nb_n3_unsort reads 4 bytes per slot, not a float4)
        if (slot_of && lb >= 0) slot_of[lb] = (int)sl; // owned bead (local order) -> slot: what k_tail gathers the pair forces by
        const float big = 3.0e38f;
        float lx = real ? p.x : big, ly = real ? p.y : big, lz = real ? p.z : big;
        float hx = real ? p.x : -big, hy = real ? p.y : -big, hz = real ? p.z : -big;
        int nreal = real ? 1 : 0;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            lx = fminf(lx, __shfl_xor(lx, o, 64));
            ly = fminf(ly, __shfl_xor(ly, o, 64));
            lz = fminf(lz, __shfl_xor(lz, o, 64));
            hx = fmaxf(hx, __shfl_xor(hx, o, 64));
            hy = fmaxf(hy, __shfl_xor(hy, o, 64));
            hz = fmaxf(hz, __shfl_xor(hz, o, 64));
            nown += __shfl_xor(nown, o, 64);
            nreal += __shfl_xor(nreal, o, 64);
        }
        if ((e & 7) == 0) { // lo.w = cell id, hi.w = (owned beads << 8) | beads of the cluster
            // one 32-byte record per cluster: {lo, hi} interleaved in cl_lo (cl_hi is unused: both halves of a box are
            // then fetched by two back-to-back 16-byte loads from one address)
            const size_t cl = sl >> 3;
            cl_lo[2 * cl] = make_float4(lx, ly, lz, __int_as_float(c));
            cl_lo[2 * cl + 1] = make_float4(hx, hy, hz, __int_as_float((nown << 8) | nreal));
        }
    }
}

// Kept cell structure: cluster positions and boxes from the current pos4 -- membership, cluster composition, cluster
// order and work items stay as the last full build left them (cells of edge cutoff + skin make that exact while no bead has
// moved more than skin / 2 from where it was binned: k_pack checks).  One thread per cluster slot.
__global__ __launch_bounds__(256) void k_refresh_clusters(const int *__restrict__ sbead, const float4 *__restrict__ pos4,
                                                          float4 *__restrict__ spos4, float4 *__restrict__ cl_lo,
                                                          const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    const int nsl = st->n_clusters * 8;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nsl; e += gridDim.x * 256) { // (nsl is a multiple of 8: whole clusters per octet)
        const int b = sbead[e];
        const bool real = b >= 0;
        float4 p = make_float4(1e18f, 1e18f, 1e18f, __int_as_float(-8));
        if (real) p = pos4[b];
        spos4[e] = p;
        const float big = 3.0e38f;
        float lx = real ? p.x : big, ly = real ? p.y : big, lz = real ? p.z : big;
        float hx = real ? p.x : -big, hy = real ? p.y : -big, hz = real ? p.z : -big;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            lx = fminf(lx, __shfl_xor(lx, o, 64));
            ly = fminf(ly, __shfl_xor(ly, o, 64));
            lz = fminf(lz, __shfl_xor(lz, o, 64));
            hx = fmaxf(hx, __shfl_xor(hx, o, 64));
            hy = fmaxf(hy, __shfl_xor(hy, o, 64));
            hz = fmaxf(hz, __shfl_xor(hz, o, 64));
        }
        if ((e & 7) == 0) { // lo.w (cell id) and hi.w (bead counts) are what the build wrote
            const int c = e >> 3;
            const float cw = cl_lo[2 * c].w, nw = cl_lo[2 * c + 1].w;
            cl_lo[2 * c] = make_float4(lx, ly, lz, cw);
            cl_lo[2 * c + 1] = make_float4(hx, hy, hz, nw);
        }
    }
}

// Block-wide bitonic sort of n2 = 256*H/..  keys held in REGISTERS (H keys per lane; wave w owns the contiguous chunk
// [w*C, (w+1)*C), C = 64*H; element index i = w*C + h*64 + lane).  Stages with j < 64 are wave shuffles, stages
// with 64 <= j < C are compares between a lane's own registers, and only the <= 3 stages with j >= C (partner in
// another wave) go through LDS -- against 45 LDS round trips of the all-LDS network this replaces (measured on
// gw_200k: 15 us of the 23 us of k_cell_order were the sort).  On return s_buf[0..n2) holds the sorted keys.
// Whole block must call; waves with w >= n2 / C hold padding keys and only take part in the barriers.
// 32-bit keys of the direct build (12-bit Hilbert index << 20 | bead) widened to the 64-bit layout (index << 32 | bead)
__device__ __forceinline__ unsigned long long widen_key(unsigned long long k) { return k; }
__device__ __forceinline__ unsigned long long widen_key(unsigned k) { return ((unsigned long long)(k >> 20) << 32) | (k & 0xfffffu); }
template <int H, class KeyT = unsigned long long>
__device__ __forceinline__ void block_sort_regs(unsigned long long *s_buf, const KeyT *__restrict__ src,
                                                int cnt, int n2) {
    constexpr int C = 64 * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = wave * C;
    const bool live = base < n2; // wave-uniform
    unsigned long long v[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int i = base + h * 64 + lane;
        v[h] = ~0ull;
        if (live && i < cnt) v[h] = widen_key(src[i]);
    }
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= C) { // partner lives in another wave: one exchange through LDS
                __syncthreads();
                if (live) {
#pragma unroll
                    for (int h = 0; h < H; ++h) s_buf[base + h * 64 + lane] = v[h];
                }
                __syncthreads();
                if (live) {
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const int i = base + h * 64 + lane;
                        const unsigned long long o = s_buf[i ^ j];
                        const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
                        v[h] = keep_min ? (v[h] < o ? v[h] : o) : (v[h] < o ? o : v[h]);
                    }
                }
            } else if (j >= 64) { // partner is another register of the same lane: h ^ (j / 64), static indices
                const int dh = j >> 6; // 1 or 2 (H <= 4)
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const bool up = ((base + h * 64 + lane) & k) == 0; // h is the lower index of its pair
                    if (h + 1 < H && (h & 1) == 0 && dh == 1) {
                        const unsigned long long a = v[h], b = v[(h + 1) % H];
                        if ((a > b) == up) {
                            v[h] = b;
                            v[(h + 1) % H] = a;
                        }
                    }
                    if (h + 2 < H && (h & 2) == 0 && dh == 2) {
                        const unsigned long long a = v[h], b = v[(h + 2) % H];
                        if ((a > b) == up) {
                            v[h] = b;
                            v[(h + 2) % H] = a;
                        }
                    }
                }
            } else { // partner is lane ^ j of the same wave, same register
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const unsigned long long o = __shfl_xor(v[h], j, 64);
                    const int i = base + h * 64 + lane;
                    const bool keep_min = ((lane & j) == 0) == ((i & k) == 0);
                    v[h] = keep_min ? (v[h] < o ? v[h] : o) : (v[h] < o ? o : v[h]);
                }
            }
        }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int h = 0; h < H; ++h) s_buf[base + h * 64 + lane] = v[h];
    }
    __syncthreads();
}

// Sort of <= 512 keys by COUNTING (whole workgroup of 256 threads): a key's place in the sorted cell is the number of keys below it,
// counted against broadcast reads of the keys in LDS; no exchange network.  The bitonic network above takes 45 dependent stages
// for 257..512 keys (9.7 us measured, profiles/r05/build_stages.txt: the ~300 large cells of a collapsing gw_200k ended the
// build's launch 5 us after everything else); this takes ~2.
// Keys of a valid evaluation are unique (the bead id is part of them).  A VOID one may hold the same key twice -- a send list that
// outgrew its message keeps stale ids behind its last entry, and the ghost arrives twice: equal keys get the same place and leave
// the next one unwritten, i.e. an arbitrary bead id in the cluster list of an evaluation that is merely to be repeated.  The places
// are therefore pre-filled with a value no key has and checked afterwards: false (block-uniform) = some place stayed empty, the
// caller sorts the cell with the network instead (tests/test_gpu_faults.py: ..._outgrows_its_message_under_the_direct_build).
// s_in: LDS [512] KeyT (input copy); s_out: LDS [512] sorted keys, widened to 64 bits (what emit_clusters reads).  Same order as
// the network: ascending keys.
template <class KeyT>
__device__ __forceinline__ bool block_rank_sort(KeyT *s_in, unsigned long long *s_out, const KeyT *__restrict__ src, const int cnt) {
    const int t = threadIdx.x;
    const bool one = t < cnt, two = t + 256 < cnt;
    KeyT a0 = (KeyT)~0ull, a1 = (KeyT)~0ull; // (padding: never below a real key)
    if (one) a0 = src[t];
    if (two) a1 = src[t + 256];
    s_in[t] = a0;
    s_in[t + 256] = a1;
    s_out[t] = ~0ull;
    s_out[t + 256] = ~0ull;
    __syncthreads();
    const bool wave_two = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u)) + 256 < cnt; // any lane of this wave holds a second key
    const int cnt4 = (cnt + 3) & ~3;
    int r0 = 0, r1 = 0;
    if (wave_two) {
        for (int q = 0; q < cnt4; q += 4) {
            KeyT k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = s_in[q + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                r0 += k[u] < a0 ? 1 : 0;
                r1 += k[u] < a1 ? 1 : 0;
            }
        }
    } else {
        for (int q = 0; q < cnt4; q += 4) {
            KeyT k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = s_in[q + u];
#pragma unroll
            for (int u = 0; u < 4; ++u) r0 += k[u] < a0 ? 1 : 0;
        }
    }
    if (one) s_out[r0] = widen_key(a0);
    if (two) s_out[r1] = widen_key(a1);
    __syncthreads();
    const bool hole = (one && s_out[t] == ~0ull) || (two && s_out[t + 256] == ~0ull);
    return __syncthreads_or(hole ? 1 : 0) == 0;
}

// `bid` of `nblk` workgroups (k_cell_order: blockIdx / gridDim; k_order_items: the launch's first nblk workgroups)
template <int CHUNK, int CAP>
__device__ __forceinline__ void cell_order_block(const int bid, const int nblk, const GridParams *__restrict__ grid,
                                                 const int *__restrict__ start, const int *__restrict__ istart,
                                                 int *__restrict__ count, int *__restrict__ perm, int2 *__restrict__ items,
                                                 const int *__restrict__ cstart, const float4 *__restrict__ pos4,
                                                 float4 *__restrict__ spos4, float4 *__restrict__ cl_lo,
                                                 float4 *__restrict__ cl_hi, const Own own,
                                                 const unsigned long long *__restrict__ okeys,
                                                 const int *__restrict__ biglist,
                                                 MinState *__restrict__ st, int *__restrict__ count_own = nullptr,
                                                 int *__restrict__ sbead = nullptr, const int slot_cap = 0,
                                                 const int slot_cells = 0, const int split = 0,
                                                 int *__restrict__ slot_of = nullptr) {
    // split: the ghosts' clusters of cell c start at n_clusters_own + istart[c] (ScanArgs::split)
    const int gbase = split ? st->n_clusters_own : 0;
    // slot_cap > 0: `okeys` is the slot table the pack wrote (keys of cell c at c * slot_cap), not the counting sort's slices
    __shared__ unsigned long long s_buf[CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const GridParams G = *grid;
    const int ncells = G.ncells;
    const unsigned long long kmax = ~0ull;

    // Large cells (block per cell, pass B) and small cells (wave per cell, pass A) go to DIFFERENT blocks, all of them
    // resident at once: the launch lasts as long as the longer of the two dependent-load chains, not their sum (a block
    // that did both first walked its small cells, then sorted its large one: 18.8 us for the launch against ~11 now).
    // The first nB blocks take the large cells; at least a quarter of the blocks stays with the small ones.
    const int nbig = st->n_big;
    const int nB = min(nbig, nblk - (nblk >> 2));
    // ---- pass A: one wave per small cell
    for (int c = (bid - nB) * 4 + wave; bid >= nB && c < ncells; c += (nblk - nB) * 4) {
        const int s = start[c], cnt = start[c + 1] - s;
        if (cnt > 64) continue;
        const size_t kb = slot_cap ? (size_t)c * slot_cap : (size_t)s;
        const int no = count_own ? count_own[c] : cnt;
        if (lane == 0) { // (also for a cell that is skipped below: the next build counts from zero)
            count[c] = 0;
            if (count_own) count_own[c] = 0;
        }
        if (slot_cap && (c >= slot_cells || cnt > slot_cap)) continue; // (a void evaluation: k_pack flagged it)
        if (cnt == 0) continue;
        if (lane == 0 && !split) items[istart[c]] = make_int2(c, 0);
        if (cnt > 1 || slot_cap) { // (slot table: perm has no fill behind it, a single bead is written here too)
            unsigned long long v = kmax;
            if (lane < cnt) v = okeys[kb + lane];
            if (cnt > 1) {
#pragma unroll
                for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        const unsigned long long o = __shfl_xor(v, j, 64);
                        const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                        v = keep_min ? (v < o ? v : o) : (v < o ? o : v);
                    }
                }
            }
            if (lane < cnt) perm[s + lane] = (int)(unsigned)(v & 0xffffffffull);
            __threadfence_block(); // the sorted perm[] is re-read below by other lanes
        }
        emit_clusters(c, s, cnt, no, cstart[c], split ? gbase + istart[c] : -1, perm, pos4, spos4, cl_lo, cl_hi, lane, 64, own, nullptr, sbead, slot_of);
    }

    // ---- pass B: the whole block per large cell, taken from the list the scan compacted (one cell per block in
    // flight as long as there are fewer large cells than resident blocks, wherever they sit in the grid)
    for (int bi = bid; bid < nB && bi < nbig; bi += nB) {
        const int c = biglist[bi];
        const int s = start[c], cnt = start[c + 1] - s;
        if (cnt <= 64) continue; // (cannot happen; block-uniform)
        const size_t kb = slot_cap ? (size_t)c * slot_cap : (size_t)s;
        const int no = count_own ? count_own[c] : cnt;
        __syncthreads(); // every thread has read count_own[c]
        if (threadIdx.x == 0) { // (also for a cell that is skipped below: the next build counts from zero)
            count[c] = 0;
            if (count_own) count_own[c] = 0;
        }
        if (slot_cap && (c >= slot_cells || cnt > slot_cap)) continue; // (a void evaluation: k_pack flagged it)
        const int nchunk = (cnt + CHUNK - 1) / CHUNK, ib = istart[c];
        for (int k = threadIdx.x; k < nchunk && !split; k += 256) items[ib + k] = make_int2(c, k);
        if (cnt <= 1024) { // keys in registers, <= 3 exchanges through LDS
            int n2 = 128;
            while (n2 < cnt) n2 <<= 1;
            __syncthreads(); // s_buf free (emit of the previous cell has read it)
            if (n2 <= 256) block_sort_regs<1>(s_buf, okeys + kb, cnt, n2);
            else if (n2 == 512) block_sort_regs<2>(s_buf, okeys + kb, cnt, n2);
            else block_sort_regs<4>(s_buf, okeys + kb, cnt, n2);
            for (int q = threadIdx.x; q < cnt; q += 256) perm[s + q] = (int)(unsigned)(s_buf[q] & 0xffffffffull);
            emit_clusters(c, s, cnt, no, cstart[c], split ? gbase + istart[c] : -1, perm, pos4, spos4, cl_lo, cl_hi, threadIdx.x, 256, own, s_buf, sbead, slot_of);
            continue;
        }
        if (cnt <= CAP) { // 1025..4096 beads (CAP = 4096 instances only): the all-LDS network
            int n2 = 128;
            while (n2 < cnt) n2 <<= 1;
            __syncthreads(); // s_buf free
            for (int q = threadIdx.x; q < n2; q += 256) {
                s_buf[q] = q < cnt ? okeys[kb + q] : kmax;
            }
            __syncthreads();
            // bitonic network; stages with j < 128 stay inside 128-element segments, each owned by one wave
            // (wave-level LDS ordering only), so only the few stages with j >= 128 need a block barrier
            for (int k = 2; k <= n2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    if (j >= 128) {
                        __syncthreads();
                        for (int q = threadIdx.x; q < (n2 >> 1); q += 256) {
                            const int i0 = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                            const int i1 = i0 | j;
                            const unsigned long long a = s_buf[i0], b = s_buf[i1];
                            const bool up = (i0 & k) == 0;
                            if ((a > b) == up) {
                                s_buf[i0] = b;
                                s_buf[i1] = a;
                            }
                        }
                        __syncthreads();
                    } else {
                        for (int seg = wave; seg < (n2 >> 7); seg += 4) {
                            const int i0 = (seg << 7) + (((lane & ~(j - 1)) << 1) | (lane & (j - 1)));
                            const int i1 = i0 | j;
                            const unsigned long long a = s_buf[i0], b = s_buf[i1];
                            const bool up = (i0 & k) == 0;
                            if ((a > b) == up) {
                                s_buf[i0] = b;
                                s_buf[i1] = a;
                            }
                        }
                        wave_lds_sync();
                    }
                }
            }
            __syncthreads();
            for (int q = threadIdx.x; q < cnt; q += 256) perm[s + q] = (int)(unsigned)(s_buf[q] & 0xffffffffull);
            __threadfence_block();
            __syncthreads();
        }
        else if (slot_cap) { // above CAP: arrival order -- which, without a fill, has to be written out first
            for (int q = threadIdx.x; q < cnt; q += 256) perm[s + q] = (int)(unsigned)(okeys[kb + q] & 0xffffffffull);
            __threadfence_block();
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(&st->order_fallbacks, 1);
        }
        else if (threadIdx.x == 0) {
            atomicAdd(&st->order_fallbacks, 1);
            // cells above CAP beads keep arrival order: still correct on a single domain (not bitwise reproducible); on a
            // decomposed rank the clusters below would mix owned beads and ghosts -- emit_clusters takes the first `no`
            // entries of the sorted cell for the owned ones -- and the half-shell kernel's per-cluster ownership with them:
            // the evaluation is void (decomposed handles always run the CAP = 4096 instance: > 4096 beads in one cell)
            if (count_own) atomicOr(&st->kernel_error, (int)KERR_ORDER_DD);
        }
        emit_clusters(c, s, cnt, no, cstart[c], split ? gbase + istart[c] : -1, perm, pos4, spos4, cl_lo, cl_hi, threadIdx.x, 256, own, nullptr, sbead, slot_of);
    }
}

template <int CHUNK, int CAP>
__global__ __launch_bounds__(256) void k_cell_order(const GridParams *__restrict__ grid,
                                                    const int *__restrict__ start, const int *__restrict__ istart,
                                                    int *__restrict__ count,
                                                    int *__restrict__ perm, int2 *__restrict__ items,
                                                    const int *__restrict__ cstart,
                                                    const float4 *__restrict__ pos4, float4 *__restrict__ spos4,
                                                    float4 *__restrict__ cl_lo, float4 *__restrict__ cl_hi,
                                                    const Own own, const unsigned long long *__restrict__ okeys,
                                                    const int *__restrict__ biglist,
                                                    MinState *__restrict__ st, int *__restrict__ count_own = nullptr,
                                                    int *__restrict__ sbead = nullptr, const int slot_cap = 0,
                                                    const int slot_cells = 0, int *__restrict__ slot_of = nullptr) {
    if (st->phase >= PH_DONE) return;
    cell_order_block<CHUNK, CAP>((int)blockIdx.x, (int)gridDim.x, grid, start, istart, count, perm, items, cstart, pos4, spos4,
                                 cl_lo, cl_hi, own, okeys, biglist, st, count_own, sbead, slot_cap, slot_cells, 0, slot_of);
}

} // namespace mmx
