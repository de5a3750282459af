// mmx_cells.hpp -- K1: position pack + bounding box, cell hash, count, scan, fill, in-cell ordering.
// All device-driven: no host round trip between an L-BFGS trial step and the pair kernel.
#pragma once
#include "mmx_common.hpp"

namespace mmx {

// x = xp + step*d (MOVE) or x as given; builds pos4 = {x,y,z, bits((bead<<3)|(label+2))} and the bbox.
// One thread per bead.  Algorithmic traffic: read 12(+24 when MOVE) B, write 16(+12) B per bead.
template <bool MOVE>
__global__ __launch_bounds__(256) void k_pack(int n, float *__restrict__ x, const float *__restrict__ xp,
                                              const float *__restrict__ d, const int8_t *__restrict__ labels,
                                              float4 *__restrict__ pos4, unsigned *__restrict__ bbox,
                                              const MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float px = 0.f, py = 0.f, pz = 0.f;
    const bool act = i < n;
    if (act) {
        if (MOVE) {
            const double step = st->step;
            px = (float)((double)xp[3 * i] + step * (double)d[3 * i]);
            py = (float)((double)xp[3 * i + 1] + step * (double)d[3 * i + 1]);
            pz = (float)((double)xp[3 * i + 2] + step * (double)d[3 * i + 2]);
            x[3 * i] = px;
            x[3 * i + 1] = py;
            x[3 * i + 2] = pz;
        } else {
            px = x[3 * i];
            py = x[3 * i + 1];
            pz = x[3 * i + 2];
        }
        const int w = (i << 3) | ((int)labels[i] + 2);
        pos4[i] = make_float4(px, py, pz, __int_as_float(w));
    }
    const float big = 3.0e38f;
    float mnx = wave_min(act ? px : big), mny = wave_min(act ? py : big), mnz = wave_min(act ? pz : big);
    float mxx = wave_max(act ? px : -big), mxy = wave_max(act ? py : -big), mxz = wave_max(act ? pz : -big);
    if ((threadIdx.x & 63) == 0 && mnx <= mxx) {
        atomicMin(&bbox[0], enc_ordered(mnx));
        atomicMin(&bbox[1], enc_ordered(mny));
        atomicMin(&bbox[2], enc_ordered(mnz));
        atomicMax(&bbox[3], enc_ordered(mxx));
        atomicMax(&bbox[4], enc_ordered(mxy));
        atomicMax(&bbox[5], enc_ordered(mxz));
    }
}

// Cell id per bead + per-cell population.
__global__ __launch_bounds__(256) void k_cell_count(int n, const float4 *__restrict__ pos4,
                                                    const unsigned *__restrict__ bbox, float hmin, int maxcells,
                                                    int *__restrict__ cell_of, int *__restrict__ count,
                                                    const MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    const GridParams G = grid_from_bbox(bbox, hmin, maxcells);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos4[i];
    const int cx = cell_coord(p.x, G.ox, G.inv_h, G.nx);
    const int cy = cell_coord(p.y, G.oy, G.inv_h, G.ny);
    const int cz = cell_coord(p.z, G.oz, G.inv_h, G.nz);
    const int c = (cz * G.ny + cy) * G.nx + cx;
    cell_of[i] = c;
    atomicAdd(&count[c], 1);
}

// Single-block exclusive scan of the cell populations (bead offsets) and of the per-cell chunk
// counts (work-item offsets); publishes the grid, the item count and resets the bbox.
template <int CHUNK>
__global__ __launch_bounds__(1024) void k_cell_scan(unsigned *__restrict__ bbox, float hmin, int maxcells,
                                                    const int *__restrict__ count, int *__restrict__ start,
                                                    int *__restrict__ istart, GridParams *__restrict__ grid,
                                                    MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    __shared__ int s_a[1024], s_b[1024], s_m[16];
    const GridParams G = grid_from_bbox(bbox, hmin, maxcells);
    const int t = threadIdx.x;
    const int per = (G.ncells + 1023) / 1024;
    const int c0 = t * per, c1 = min(c0 + per, G.ncells);
    int sa = 0, sb = 0, mx = 0;
    for (int c = c0; c < c1; ++c) {
        const int k = count[c];
        sa += k;
        sb += (k + CHUNK - 1) / CHUNK;
        mx = max(mx, k);
    }
    s_a[t] = sa;
    s_b[t] = sb;
    mx = wave_max_i(mx);
    if ((t & 63) == 0) s_m[t >> 6] = mx;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials.
    for (int o = 1; o < 1024; o <<= 1) {
        int va = 0, vb = 0;
        if (t >= o) {
            va = s_a[t - o];
            vb = s_b[t - o];
        }
        __syncthreads();
        s_a[t] += va;
        s_b[t] += vb;
        __syncthreads();
    }
    int ra = s_a[t] - sa, rb = s_b[t] - sb; // exclusive prefixes
    for (int c = c0; c < c1; ++c) {
        const int k = count[c];
        start[c] = ra;
        istart[c] = rb;
        ra += k;
        rb += (k + CHUNK - 1) / CHUNK;
    }
    if (t == 1023) {
        start[G.ncells] = s_a[1023];
        istart[G.ncells] = s_b[1023];
        *grid = G;
        int m = 0;
        for (int w = 0; w < 16; ++w) m = max(m, s_m[w]);
        st->n_items = s_b[1023];
        st->ncells = G.ncells;
        st->max_per_cell = m;
        st->cell_edge = (double)G.h;
        bbox[0] = bbox[1] = bbox[2] = kEncPosInf;
        bbox[3] = bbox[4] = bbox[5] = kEncNegInf;
    }
}

// Scatter bead ids into their cell's slice (arrival order; k_cell_order makes it canonical).
__global__ __launch_bounds__(256) void k_cell_fill(int n, const int *__restrict__ cell_of,
                                                   const int *__restrict__ start, int *__restrict__ cursor,
                                                   int *__restrict__ perm, const MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of[i];
    perm[start[c] + atomicAdd(&cursor[c], 1)] = i;
}

// One wave per cell (grid-stride): sorts the cell's bead ids ascending (bitwise reproducible pair
// summation order), emits the cell's work items {cell, chunk} and clears count/cursor for the next build.
constexpr int kOrderLds = 2048;
template <int CHUNK>
__global__ __launch_bounds__(256) void k_cell_order(const GridParams *__restrict__ grid,
                                                    const int *__restrict__ start, const int *__restrict__ istart,
                                                    int *__restrict__ count, int *__restrict__ cursor,
                                                    int *__restrict__ perm, int2 *__restrict__ items,
                                                    int deterministic, const MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    __shared__ int s_buf[4][kOrderLds];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ncells = grid->ncells;
    const int nwaves = gridDim.x * 4;
    int *buf = s_buf[wave];
    for (int c = blockIdx.x * 4 + wave; c < ncells; c += nwaves) {
        const int s = start[c], cnt = start[c + 1] - s;
        if (lane == 0) {
            count[c] = 0;
            cursor[c] = 0;
        }
        if (cnt == 0) continue;
        const int nchunk = (cnt + CHUNK - 1) / CHUNK, ib = istart[c];
        for (int k = lane; k < nchunk; k += 64) items[ib + k] = make_int2(c, k);
        if (!deterministic || cnt == 1) continue;
        if (cnt <= 64) {
            int v = lane < cnt ? perm[s + lane] : 0x7fffffff;
#pragma unroll
            for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const int o = __shfl_xor(v, j, 64);
                    const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
                    v = keep_min ? min(v, o) : max(v, o);
                }
            }
            if (lane < cnt) perm[s + lane] = v;
        } else if (cnt <= kOrderLds) {
            int n2 = 128;
            while (n2 < cnt) n2 <<= 1;
            for (int q = lane; q < n2; q += 64) buf[q] = q < cnt ? perm[s + q] : 0x7fffffff;
            wave_lds_sync();
            for (int k = 2; k <= n2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int q = lane; q < (n2 >> 1); q += 64) {
                        const int i0 = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                        const int i1 = i0 | j;
                        const int a = buf[i0], b = buf[i1];
                        const bool up = (i0 & k) == 0;
                        if ((a > b) == up) {
                            buf[i0] = b;
                            buf[i1] = a;
                        }
                    }
                    wave_lds_sync();
                }
            }
            for (int q = lane; q < cnt; q += 64) perm[s + q] = buf[q];
            wave_lds_sync();
        }
        // cells above kOrderLds beads keep arrival order (still correct, not bitwise reproducible)
    }
}

} // namespace mmx
