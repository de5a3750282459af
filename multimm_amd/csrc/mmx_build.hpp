// mmx_build.hpp -- K1, the DIRECT cell build of a single-domain trial move: ONE launch behind the pack.
//
// The scan-based build (mmx_cells.hpp) is pack -> [single-workgroup scan of the cell populations || bonded pass] -> in-cell
// order + work items: the scan is a serial stage of ~10 us with a grid drain and a dispatch on either side of it, and every
// consumer then chases offsets through memory (start -> keys -> perm -> positions: four dependent round trips per cell).  Here
// nothing is scanned by anybody for everybody:
//   * the pack keeps, next to the cell populations, two totals per ROW of the grid (cell_rank: clusters of 8 the row's cells
//     need, cells of more than 64 beads), in a set of counters that alternates with the build's parity -- the set a build reads
//     is never written during its launch, the other one is zeroed by it for the next pack;
//   * every workgroup of this launch turns the <= 2048 row totals into prefixes in LDS (one coalesced load round + a
//     workgroup scan), and whoever needs the place of cell (x, row) in the cluster list adds to the row's prefix the clusters
//     of the row's cells before x -- one lane per cell of the row, one masked wave reduction;
//   * sort keys come from the slot table at cell * cap (an address that needs no offset), in the same load round as the row's
//     populations; a cell of <= 64 beads is sorted by one wave in registers and every lane then holds the bead of its own cluster
//     slot (no permutation array); larger cells are found through the prefix of the rows' large-cell counts;
//   * the half-shell kernel's work items are cut per row from cluster offsets the wave derives for the five rows it looks at
//     (LDS scratch instead of the scan's cstart array);
//   * the bonded pass rides in the same launch, as it did in the scan's.
// A cell's dependent chain is grid -> {row totals, populations, keys} -> positions: three round trips instead of five, and the
// evaluation has one launch less.  What the scan also produced: cstart (written here by whoever owns the cell: the full-shell
// kernel walks it), cluster / large-cell totals and the next build's grid (workgroup 0), the fullest cell (k_poll_stats at the
// host's polls).  Not produced: start, perm, the 64-bead chunk items (nb_variant 1, census: they build through the scan).
// Limits: nx <= 64 (one lane per cell of a row), ny * nz <= kDirectMaxRows; a grid beyond them voids the evaluation (k_pack)
// and the host falls back to the scan-based build.
#pragma once
#include <type_traits>
#include "mmx_bonded.hpp"
#include "mmx_nonbonded_n3.hpp"

namespace mmx {

struct DirectArgs {
    const GridParams *grid;    // grid of this build (fixed by the previous build)
    GridParams *grid_next;     // ... of the next one: from this evaluation's bounding box
    int parity;                // counter set of this build (the other one is zeroed here for the next pack)
    const float *bbox_part;
    int nblk_bbox;
    float hmin;
    int maxcells;
    const int *count;          // [cells] populations of this build (read-only here)
    const int *rowcl, *rowbig; // [rows] totals of this build
    int *count_zero, *rowcl_zero, *rowbig_zero; // the other set: zeroed for the next pack
    const unsigned long long *keys; // slot table: keys of cell c at c * slot_cap
    int slot_cap, slot_cells;
    const float4 *pos4;
    float4 *spos4, *cl_lo;
    int *cstart, *sbead, *slot_of;
    int cap_clusters;          // clusters the cluster list holds
    int n_beads;               // beads of the system (a sort key that names no bead is an error, never an address)
    N3Item *n3_items;
    int n3_max_items, n3_flags; // (flags as k_order_items: bit 0 long items, bit 1 no pass records, bit 2 two-visit items, >> 8 slice cap)
    int n_items_blocks, n_bonded_blocks, n_order;
};
struct BondedArgs {
    const uint8_t *flags;
    const int *lstart, *partner;
    const float *r0, *cf_w;
    float *g;
    double *part;
    int loop_form, lam_form, cf_form, nvb;
};

// Exclusive prefixes of the row totals in LDS: s_cl[r], s_big[r] for r < nrows, the totals at [nrows].  Whole workgroup (256).
__device__ __forceinline__ void direct_row_prefix(const int *__restrict__ rowcl, const int *__restrict__ rowbig, const int nrows,
                                                  int *s_cl, int *s_big) {
    constexpr int PER = kDirectMaxRows / 256;
    __shared__ int s_wcl[4], s_wbig[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int vcl[PER], vbig[PER], sa = 0, sb = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { // rows t * PER + j: consecutive per thread (two 16-byte loads per array)
        const int r = t * PER + j;
        vcl[j] = r < nrows ? rowcl[r] : 0;
        vbig[j] = r < nrows ? rowbig[r] : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        sa += vcl[j];
        sb += vbig[j];
    }
    int ia = sa, ib = sb;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64);
        if (lane >= o) {
            ia += ua;
            ib += ub;
        }
    }
    if (lane == 63) {
        s_wcl[wave] = ia;
        s_wbig[wave] = ib;
    }
    __syncthreads();
    int oa = 0, ob = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        oa += w < wave ? s_wcl[w] : 0;
        ob += w < wave ? s_wbig[w] : 0;
    }
    int ra = oa + ia - sa, rb = ob + ib - sb;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int r = t * PER + j;
        if (r < nrows) {
            s_cl[r] = ra;
            s_big[r] = rb;
        }
        ra += vcl[j];
        rb += vbig[j];
    }
    if (t == 255) {
        s_cl[nrows] = ra;
        s_big[nrows] = rb;
    }
    __syncthreads();
}

// populations of the cells of row `row` (lane = x; 0 beyond nx), this lane's cell's clusters, and the clusters of the row's
// cells before x0: whole wave
__device__ __forceinline__ int direct_row_before(const int cl_lane, const int lane, const int x0) {
    return wave_sum_i(lane < x0 ? cl_lane : 0);
}

// Bitonic sort of 64 * H keys held by ONE wave, H per lane (element h * 64 + lane in v[h]; the first H registers of the array):
// partners at distance < 64 by wave shuffles, at 64 / 128 in another register of the same lane.  Ascending.
template <int H, int HMAX, class KeyT>
__device__ __forceinline__ void wave_sort_keys(KeyT (&v)[HMAX], const int lane) {
    constexpr int N = 64 * H;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                constexpr int dummy = 0;
                (void)dummy;
                const int dh = j >> 6;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    if ((h & dh) == 0 && h + dh < H) {
                        const bool up = (((h * 64 + lane) & k) == 0);
                        const KeyT a = v[h], b = v[h + dh];
                        if ((a > b) == up) {
                            v[h] = b;
                            v[h + dh] = a;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const KeyT o = __shfl_xor(v[h], j, 64);
                    const bool keep_min = ((lane & j) == 0) == (((h * 64 + lane) & k) == 0);
                    v[h] = keep_min ? (v[h] < o ? v[h] : o) : (v[h] < o ? o : v[h]);
                }
            }
        }
    }
}

// Cluster slots of one sorted cell from the keys in registers: entry e = h * 64 + lane of the cell's clusters is lane's v[h].
template <int H, int HMAX, class DA, class KeyT>
__device__ __forceinline__ void direct_emit_wave(const DA &D, MinState *__restrict__ st, const KeyT (&v)[HMAX],
                                                 const int c, const int cnt, const int cb, const int cap_slots, const int lane) {
    constexpr unsigned kBeadMask = sizeof(KeyT) == 4 ? 0xfffffu : 0xffffffffu;
    const int ncl = (cnt + 7) >> 3;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int e = h * 64 + lane;
        if (e < ncl * 8) {
            bool real = e < cnt;
            int bead = real ? (int)((unsigned)v[h] & kBeadMask) : -1;
            if (real && (unsigned)bead >= (unsigned)D.n_beads) { // (a key that is no bead: counters and keys of different builds)
                atomicOr(&st->kernel_error, (int)KERR_BOUNDS);
                real = false;
                bead = -1;
            }
            float4 p = make_float4(1e18f, 1e18f, 1e18f, __int_as_float(-8)); // padding: far away, bead id -1
            if (real) p = D.pos4[bead];
            const int sl = cb * 8 + e;
            const bool fits = sl < cap_slots; // (whole clusters: the eight lanes of a cluster agree)
            if (fits) {
                D.spos4[sl] = p;
                D.sbead[sl] = bead;
                if (real) D.slot_of[bead] = sl;
            }
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
                nreal += __shfl_xor(nreal, o, 64);
            }
            if ((lane & 7) == 0) {
                if (fits) {
                    const int cl = sl >> 3;
                    D.cl_lo[2 * cl] = make_float4(lx, ly, lz, __int_as_float(c));
                    D.cl_lo[2 * cl + 1] = make_float4(hx, hy, hz, __int_as_float((nreal << 8) | nreal));
                } else {
                    atomicOr(&st->kernel_error, (int)KERR_BOUNDS);
                }
            }
        }
    }
}

template <int CHUNK, int CAP, bool N3, bool KEY32 = false>
__global__ __launch_bounds__(256, 6) void k_build_direct(const DirectArgs D, MinState *__restrict__ st, const FFParams P,
                                                       const BondedArgs B) {
    if (st->phase >= PH_DONE) return;
    // KEY32 (systems of <= 2^20 beads): the pack wrote 32-bit keys (12-bit Hilbert index << 20 | bead) -- the same order as the
    // 64-bit ones, half the shuffles and simpler compares in every stage of the sorts
    using KeyT = typename std::conditional<KEY32, unsigned, unsigned long long>::type;
    const KeyT *const keys = reinterpret_cast<const KeyT *>(D.keys);
    __shared__ unsigned long long s_buf[CAP]; // order: the block sort; item builders: the waves' cluster-offset scratch
    __shared__ int s_cl[kDirectMaxRows + 1], s_big[kDirectMaxRows + 1];
    __shared__ double s_w[4];
    const int nbb = D.n_bonded_blocks;
    BUILD_STAMP(0); // (timing build: scripts/stage_build.py)
    // ---- bonded pass: workgroups [n_items_blocks, n_items_blocks + nbb)
    if ((int)blockIdx.x >= D.n_items_blocks && (int)blockIdx.x < D.n_items_blocks + nbb) {
        // (a virtual block of the bonded pass per round: the partials stay those of the stand-alone pass, workgroup by workgroup)
        for (int vb = (int)blockIdx.x - D.n_items_blocks; vb < B.nvb; vb += nbb)
            bonded_fused_block<256>(P, D.pos4, B.flags, B.lstart, B.partner, B.r0, B.cf_w, B.g, B.part, B.loop_form, B.lam_form,
                                    B.cf_form, vb, B.nvb, s_w);
        BUILD_STAMP(2);
        return;
    }
#ifdef MMX_STAGE_TIMING
    const int bid_o = (int)blockIdx.x - D.n_items_blocks - nbb;
    int tbase = -1; // which of the traced workgroups this is (decided once the large-cell count is known, below)
    const unsigned long long t_start = wall_clock64();
#endif
    const GridParams G = *D.grid;
    const int nx = G.nx, nrows = G.ny * G.nz, ncells = G.ncells;
    if (nx > 64 || nrows > kDirectMaxRows) return; // (k_pack has voided the evaluation)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cap_slots = D.cap_clusters * 8;
#ifdef MMX_STAGE_TIMING
    const unsigned long long t_grid = wall_clock64();
#endif
    direct_row_prefix(D.rowcl, D.rowbig, nrows, s_cl, s_big);
    const int tot_cl = s_cl[nrows], nbig = s_big[nrows];
    BUILD_STAMP(1);
#ifdef MMX_STAGE_TIMING
    {
        const int nB_ = min(nbig, D.n_order - (D.n_order >> 2));
        if (blockIdx.x == 0) tbase = 4200;
        else if (blockIdx.x == 1) tbase = 4212;
        else if (bid_o == nB_) tbase = 4204;
        else if (bid_o == 0 && nB_ > 0) tbase = 4208;
        if (tbase >= 0 && threadIdx.x == 0) {
            g_stage_t[tbase] = t_start;
            g_stage_t[tbase + 1] = t_grid;
            g_stage_t[tbase + 2] = wall_clock64();
        }
    }
#define BUILD_DONE_STAMP()                                                                            \
    do {                                                                                               \
        if (tbase >= 0 && threadIdx.x == 0) g_stage_t[tbase + 3] = wall_clock64();                     \
        BUILD_STAMP(2);                                                                                \
    } while (0)
#else
#define BUILD_DONE_STAMP() do {} while (0)
#endif

    if ((int)blockIdx.x < D.n_items_blocks) {
        // ---- workgroup 0: what the scan published for everybody
        if (blockIdx.x == 0) {
            __shared__ float s_red[6 * 4];
            const GridParams GN = grid_from_parts<256>(D.bbox_part, D.nblk_bbox, D.hmin, D.maxcells, s_red);
            if (threadIdx.x == 0) {
                *D.grid_next = GN;
                st->n_clusters = tot_cl;
                st->n_clusters_own = tot_cl;
                st->n_big = nbig;
                st->ncells = ncells;
                st->ncells_set[D.parity] = ncells;
                st->cell_edge = (double)G.h;
                D.cstart[ncells] = min(tot_cl, D.cap_clusters);
                if (tot_cl > D.cap_clusters) atomicOr(&st->kernel_error, (int)KERR_BOUNDS); // (counters beyond the beads there are)
            }
        }
        if (!N3) {
            BUILD_DONE_STAMP();
            return;
        }
        // ---- work items of the half-shell kernel: a wave per row; cluster offsets of its five rows from the populations
        int *const cs_w = reinterpret_cast<int *>(s_buf) + wave * (kN3Runs * 66);
        n3_items_rows((int)blockIdx.x, D.n_items_blocks, G, D.n3_items, D.n3_max_items, st, (D.n3_flags & 4) ? 2 : (D.n3_flags & 1),
                      !(D.n3_flags & 2), (D.n3_flags >> 8) > 0 ? min(D.n3_flags >> 8, kN3MaxCap) : kN3MaxCap,
                      [&](int row, int y, int z, N3Row &R) {
                          int rb[kN3Runs];
                          rb[0] = row;
                          rb[1] = y + 1 < G.ny ? row + 1 : -1;
#pragma unroll
                          for (int dy = -1; dy <= 1; ++dy)
                              rb[3 + dy] = (z + 1 < G.nz && y + dy >= 0 && y + dy < G.ny) ? row + G.ny + dy : -1;
                          int cnt[kN3Runs];
#pragma unroll
                          for (int r = 0; r < kN3Runs; ++r) cnt[r] = (rb[r] >= 0 && lane < nx) ? D.count[rb[r] * nx + lane] : 0;
                          wave_lds_sync(); // the previous row's offsets have been read
#pragma unroll
                          for (int r = 0; r < kN3Runs; ++r) {
                              const int cl = (cnt[r] + 7) >> 3;
                              int inc = cl;
#pragma unroll
                              for (int o = 1; o < 64; o <<= 1) {
                                  const int u = __shfl_up(inc, o, 64);
                                  if (lane >= o) inc += u;
                              }
                              if (rb[r] >= 0) {
                                  const int base = s_cl[rb[r]];
                                  if (lane < nx) cs_w[r * 66 + lane] = base + inc - cl;
                                  if (lane == 63) cs_w[r * 66 + nx] = base + inc; // (lanes beyond nx add nothing)
                              }
                              R.base[r] = rb[r] >= 0 ? r * 66 : -1;
                          }
                          wave_lds_sync();
                          R.cstart = cs_w;
                          R.nx = nx;
                          R.gstart = nullptr;
                          R.gbase = 0;
#pragma unroll
                          for (int r = 0; r < 9; ++r) R.gb[r] = -1;
                      });
        BUILD_DONE_STAMP();
        return;
    }

    // ---- in-cell order: workgroups behind the bonded ones
    const int bid = (int)blockIdx.x - D.n_items_blocks - nbb, nblk = D.n_order;
    // the other counter set, for the next pack (the grid it was used with may have had more cells than this one)
    {
        const int nz_cells = min(st->ncells_set[D.parity ^ 1], D.maxcells); // (written by the build before this one)
        for (int q = bid * 256 + (int)threadIdx.x; q < nz_cells; q += nblk * 256) D.count_zero[q] = 0;
        for (int q = bid * 256 + (int)threadIdx.x; q < kDirectMaxRows; q += nblk * 256) {
            D.rowcl_zero[q] = 0;
            D.rowbig_zero[q] = 0;
        }
    }
    const int nB = min(nbig, nblk - (nblk >> 2)); // the first nB workgroups take the cells of more than kWaveCellMax beads (rare)
    // ---- pass A: one wave per cell of <= kWaveCellMax beads, keys in registers (1, 2 or 4 per lane: element h * 64 + lane of the
    // sorted cell ends up in register h of lane `lane`): no LDS, no barrier, four cells in flight per workgroup
    const int kvec = min(D.slot_cap >> 6, kWaveCellMax / 64); // 64-key vectors a row of the slot table holds (of those a wave can take)
    for (int c = (bid - nB) * 4 + wave; bid >= nB && c < ncells; c += (nblk - nB) * 4) {
        const int row = c / nx, x = c - row * nx;
        // one load round: the row's populations (lane = cell) and -- at an address that needs no offset -- the cell's keys
        const int cnt_l = lane < nx ? D.count[row * nx + lane] : 0;
        const bool in_table = c < D.slot_cells;
        KeyT v[kWaveCellMax / 64];
#pragma unroll
        for (int h = 0; h < kWaveCellMax / 64; ++h) v[h] = (KeyT)~0ull;
        if (in_table) v[0] = keys[(size_t)c * D.slot_cap + lane]; // (slot_cap >= 64: the first 64 keys of the row are there)
        const int cnt = __shfl(cnt_l, x, 64);
#ifdef MMX_STAGE_TIMING
        if (threadIdx.x == 0 && blockIdx.x < 4096 && c == (bid - nB) * 4) {
            g_stage_c[blockIdx.x * 2] = c;
            g_stage_c[blockIdx.x * 2 + 1] = cnt;
        }
#endif
        const int cb = s_cl[row] + direct_row_before((cnt_l + 7) >> 3, lane, x);
        if (lane == 0) D.cstart[c] = min(cb, D.cap_clusters);
        if (cnt > kWaveCellMax || cnt == 0) continue; // (larger: pass B)
        if (!in_table || cnt > D.slot_cap) continue; // (a void evaluation: k_pack flagged it)
        if (cnt > 64) { // the rest of a larger cell's keys: a second load round, for those cells only
#pragma unroll
            for (int h = 1; h < kWaveCellMax / 64; ++h)
                if (h * 64 < cnt && h < kvec) v[h] = keys[(size_t)c * D.slot_cap + h * 64 + lane];
        }
#pragma unroll
        for (int h = 0; h < kWaveCellMax / 64; ++h)
            if (h * 64 + lane >= cnt) v[h] = (KeyT)~0ull;
        if (cnt <= 64) {
            if (cnt > 1) wave_sort_keys<1>(v, lane);
            direct_emit_wave<1>(D, st, v, c, cnt, cb, cap_slots, lane);
        } else if (cnt <= 128) {
            wave_sort_keys<2>(v, lane);
            direct_emit_wave<2>(D, st, v, c, cnt, cb, cap_slots, lane);
        } else { // (a wave sorting 512 keys in 8 registers was measured too: the launch got 6 us longer than with those cells in pass B)
            wave_sort_keys<4>(v, lane);
            direct_emit_wave<4>(D, st, v, c, cnt, cb, cap_slots, lane);
        }
    }

    // ---- pass B: the whole workgroup per cell of > kWaveCellMax beads; the bi-th such cell is found through the prefix of the rows'
    // large-cell counts (every wave does the search and the row's bookkeeping for itself: no barrier for it)
    const Own own{0, 0x7fffffff, 0, nullptr, nullptr}; // single domain: every bead is owned, local index = bead
    for (int bi = bid; bid < nB && bi < nbig; bi += nB) {
        int lo = 0, hi = nrows; // last row with s_big[row] <= bi
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_big[mid] <= bi) lo = mid;
            else hi = mid;
        }
        const int row = lo;
        const int cnt_l = lane < nx ? D.count[row * nx + lane] : 0;
        unsigned long long m = __ballot(cnt_l > kWaveCellMax);
        for (int k = bi - s_big[row]; k > 0 && m; --k) m &= m - 1; // its (bi - prefix)-th large cell
        if (!m) continue; // (cannot happen: the row totals count exactly these cells; block-uniform)
        const int x = __ffsll((long long)m) - 1;
        const int c = row * nx + x;
        const int cnt = __shfl(cnt_l, x, 64);
        const int cb = s_cl[row] + direct_row_before((cnt_l + 7) >> 3, lane, x);
        if (threadIdx.x == 0) D.cstart[c] = min(cb, D.cap_clusters);
        if (c >= D.slot_cells || cnt > D.slot_cap) continue; // (a void evaluation: k_pack flagged it)
        const KeyT *kp = keys + (size_t)c * D.slot_cap;
        __syncthreads(); // s_buf free (emit of the previous cell has read it)
#ifdef MMX_STAGE_TIMING
        if (bid == 0 && bi == bid && threadIdx.x == 0) {
            g_stage_t[4216] = wall_clock64(); // the cell is known (row search, population row)
            g_stage_t[4219] = (unsigned long long)cnt;
        }
#endif
        if (cnt <= 512) { // (nearly all of them) by counting: block_rank_sort
            static_assert(CAP >= 1024, "input copy in the first 512 entries of s_buf, sorted keys in the next 512");
            const unsigned long long *sorted = s_buf + 512;
            if (!block_rank_sort<KeyT>(reinterpret_cast<KeyT *>(s_buf), s_buf + 512, kp, cnt)) { // equal keys (a void evaluation's)
                block_sort_regs<2>(s_buf, kp, cnt, 512);
                sorted = s_buf;
            }
#ifdef MMX_STAGE_TIMING
            if (bid == 0 && bi == bid && threadIdx.x == 0) g_stage_t[4217] = wall_clock64(); // sorted
#endif
            emit_clusters(c, 0, cnt, cnt, cb, -1, nullptr, D.pos4, D.spos4, D.cl_lo, nullptr, threadIdx.x, 256, own, sorted, D.sbead,
                          D.slot_of, cap_slots, st, D.n_beads);
#ifdef MMX_STAGE_TIMING
            if (bid == 0 && bi == bid && threadIdx.x == 0) g_stage_t[4218] = wall_clock64(); // emitted
#endif
            continue;
        }
        if (cnt <= 1024) { // 513..1024: the bitonic network, keys in registers, three exchanges through LDS
            block_sort_regs<4>(s_buf, kp, cnt, 1024);
            emit_clusters(c, 0, cnt, cnt, cb, -1, nullptr, D.pos4, D.spos4, D.cl_lo, nullptr, threadIdx.x, 256, own, s_buf, D.sbead,
                          D.slot_of, cap_slots, st, D.n_beads);
            continue;
        }
        if (cnt <= CAP) { // 1025..4096 beads (CAP = 4096 instances only): the all-LDS network of cell_order_block
            int n2 = 128;
            while (n2 < cnt) n2 <<= 1;
            for (int q = threadIdx.x; q < n2; q += 256) {
                unsigned long long kq = ~0ull;
                if (q < cnt) kq = widen_key(kp[q]);
                s_buf[q] = kq;
            }
            __syncthreads();
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
            emit_clusters(c, 0, cnt, cnt, cb, -1, nullptr, D.pos4, D.spos4, D.cl_lo, nullptr, threadIdx.x, 256, own, s_buf, D.sbead,
                          D.slot_of, cap_slots, st, D.n_beads);
            continue;
        }
        // above CAP: arrival order (still correct, not bitwise reproducible), straight from the slot table
        if (KEY32) { // (its rows hold 32-bit keys: no place to widen a cell that large -- one void evaluation, then the scan-based build)
            if (threadIdx.x == 0) atomicOr(&st->cell_stale, 4);
            continue;
        }
        if (threadIdx.x == 0) atomicAdd(&st->order_fallbacks, 1);
        emit_clusters(c, 0, cnt, cnt, cb, -1, nullptr, D.pos4, D.spos4, D.cl_lo, nullptr, threadIdx.x, 256, own,
                      reinterpret_cast<const unsigned long long *>(kp), D.sbead, D.slot_of, cap_slots, st, D.n_beads);
    }
    BUILD_DONE_STAMP();
}

// ---- decomposed ranks: the same build over owned beads AND ghosts -------------------------------------------------------
// The pack counts the owned beads (counter set `count`), k_dd_unpack_count the ghosts as they arrive (set `count_g`, keys
// behind the owned ones in the cell's row of the slot table); both keep per-row totals of the clusters they need -- owned beads
// and ghosts never share a cluster, and the ghosts' clusters lie in a region of their own behind all the owned ones (the split
// layout of the half-shell kernel's DD instance) -- and of the cells that hold more than kWaveCellMax beads ALTOGETHER.  The grid
// is the one the previous build laid out from its owned box grown by the cutoff; beads and ghosts beyond it are clamped into
// its boundary cells, which keeps every pair within the cutoff inside the 27-cell stencil (k_dd_unpack_count).
constexpr int kDirectRowSet = 4 * kDirectMaxRows; // ints of one set of row totals: clusters, large cells, ghost clusters, cells of many ghosts
constexpr int kDirectDDCap = 2048;  // beads of the largest cell its block sort takes (LDS); beyond: KERR_ORDER_DD
constexpr int kDirectDDRows = 1024; // rows of a rank's grid the build handles (LDS: three prefix arrays)
struct DirectDD {
    const int *count_g, *rowclg; // ghosts per cell, ghost clusters per row (this build's set)
    int *count_g_zero, *rowclg_zero;
    int *rowbigg_zero;           // the other set's count of cells of many ghosts (what a two-launch build's PHASE 2 takes as D.rowbig)
    int *istart;                 // [cells + 1] out: ghost-cluster offsets per cell (counted from the first ghost cluster)
    float expand;                // the next grid: owned box grown by this much (the cutoff)
    Own own;
};

// Exclusive prefixes of one array of row totals (<= kDirectDDRows) in LDS, total at [nrows].  Whole workgroup (256).
__device__ __forceinline__ void direct_row_prefix1(const int *__restrict__ rows, const int nrows, int *s_out) {
    constexpr int PER = kDirectDDRows / 256;
    __shared__ int s_wt[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int v[PER], sa = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int r = t * PER + j;
        v[j] = r < nrows ? rows[r] : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) sa += v[j];
    int ia = sa;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ua = __shfl_up(ia, o, 64);
        if (lane >= o) ia += ua;
    }
    __syncthreads(); // (s_wt of the previous call has been read)
    if (lane == 63) s_wt[wave] = ia;
    __syncthreads();
    int oa = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) oa += w < wave ? s_wt[w] : 0;
    int ra = oa + ia - sa;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int r = t * PER + j;
        if (r < nrows) s_out[r] = ra;
        ra += v[j];
    }
    if (t == 255) s_out[nrows] = ra;
    __syncthreads();
}

// Cluster slots of one sorted cell of a decomposed rank from the keys in registers (sorted: the ko owned beads first, then the
// kg ghosts): the owned clusters at cb, the ghosts' at cbg.
template <int H, int HMAX>
__device__ __forceinline__ void direct_emit_wave_dd(const DirectArgs &D, const DirectDD &X, MinState *__restrict__ st,
                                                    const unsigned long long (&v)[HMAX], const int c, const int ko, const int kg,
                                                    const int cb, const int cbg, const int cap_slots, const int lane) {
    const int o8 = ((ko + 7) >> 3) << 3, g8 = ((kg + 7) >> 3) << 3;
    for (int e0 = 0; e0 < o8 + g8; e0 += 64) { // (wave-uniform trip count)
        const int e = e0 + lane;
        const bool in = e < o8 + g8;
        const int src = !in ? -1 : e < o8 ? (e < ko ? e : -1) : (e - o8 < kg ? ko + (e - o8) : -1); // place in the sorted cell
        unsigned long long key = 0ull;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const unsigned long long t = __shfl(v[h], max(src, 0) & 63, 64);
            if ((max(src, 0) >> 6) == h) key = t;
        }
        if (in) {
            bool real = src >= 0;
            int bead = real ? (int)(unsigned)(key & 0xffffffffull) : -1;
            if (real && (unsigned)bead >= (unsigned)D.n_beads) {
                atomicOr(&st->kernel_error, (int)KERR_BOUNDS);
                real = false;
                bead = -1;
            }
            float4 p = make_float4(1e18f, 1e18f, 1e18f, __int_as_float(-8));
            if (real) p = D.pos4[bead];
            const int lb = real ? X.own.local(bead) : -1;
            const int sl = e < o8 ? cb * 8 + e : cbg * 8 + (e - o8);
            const bool fits = sl < cap_slots;
            if (fits) {
                D.spos4[sl] = p;
                D.sbead[sl] = bead;
                if (lb >= 0) D.slot_of[lb] = sl;
            }
            const float big = 3.0e38f;
            float lx = real ? p.x : big, ly = real ? p.y : big, lz = real ? p.z : big;
            float hx = real ? p.x : -big, hy = real ? p.y : -big, hz = real ? p.z : -big;
            int nreal = real ? 1 : 0, nown = lb >= 0 ? 1 : 0;
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                lx = fminf(lx, __shfl_xor(lx, o, 64));
                ly = fminf(ly, __shfl_xor(ly, o, 64));
                lz = fminf(lz, __shfl_xor(lz, o, 64));
                hx = fmaxf(hx, __shfl_xor(hx, o, 64));
                hy = fmaxf(hy, __shfl_xor(hy, o, 64));
                hz = fmaxf(hz, __shfl_xor(hz, o, 64));
                nreal += __shfl_xor(nreal, o, 64);
                nown += __shfl_xor(nown, o, 64);
            }
            if ((lane & 7) == 0) {
                if (fits) {
                    const int cl = sl >> 3;
                    D.cl_lo[2 * cl] = make_float4(lx, ly, lz, __int_as_float(c));
                    D.cl_lo[2 * cl + 1] = make_float4(hx, hy, hz, __int_as_float((nown << 8) | nreal));
                } else {
                    atomicOr(&st->kernel_error, (int)KERR_BOUNDS);
                }
            }
        }
    }
}

// PHASE (the build in two launches, so that the halo exchange runs beside the first: enqueue_build, dd_overlap; split layout
// only): 0 = owned beads and ghosts in ONE launch; 1 = the owned beads' share -- next grid, owned clusters, cstart: nothing in it
// waits for a peer --; 2 = the ghosts' share once they have arrived -- ghost clusters, istart, work items, bonded pass (its loop
// partners and chain neighbours may be ghosts).  A cell is "large" (block sort) by what the launch sorts: D.rowbig counts the cells
// of > kWaveCellMax beads altogether (0), owned beads (1), ghosts (2).  The result is the same list bit for bit: owned beads
// sort before ghosts in a cell's keys, and the two kinds never share a cluster.
template <int CAP, bool N3, int PHASE = 0>
__global__ __launch_bounds__(256, 5) void k_build_direct_dd(const DirectArgs D, const DirectDD X, MinState *__restrict__ st,
                                                          const FFParams P, const BondedArgs B) {
    if (st->phase >= PH_DONE) return;
    static_assert(PHASE == 0 || N3, "the two-launch build needs the split layout of the half-shell kernel");
    constexpr bool OWN = PHASE != 2, GHO = PHASE != 1;
    __shared__ unsigned long long s_buf[CAP];
    __shared__ int s_cl[kDirectDDRows + 1], s_clg[kDirectDDRows + 1], s_big[kDirectDDRows + 1];
    __shared__ double s_w[4];
    BUILD_STAMP(0); // (timing build: scripts/stage_build_dd.py)
    static_assert(sizeof(unsigned long long) * CAP >= sizeof(int) * 4 * (kN3Runs + kN3GhostRuns) * 66, "item builders' scratch");
    const int nbb = D.n_bonded_blocks;
    if ((int)blockIdx.x >= D.n_items_blocks && (int)blockIdx.x < D.n_items_blocks + nbb) {
        for (int vb = (int)blockIdx.x - D.n_items_blocks; vb < B.nvb; vb += nbb)
            bonded_fused_block<256>(P, D.pos4, B.flags, B.lstart, B.partner, B.r0, B.cf_w, B.g, B.part, B.loop_form, B.lam_form,
                                    B.cf_form, vb, B.nvb, s_w);
        BUILD_STAMP(2);
        return;
    }
    const GridParams G = *D.grid;
    const int nx = G.nx, nrows = G.ny * G.nz, ncells = G.ncells;
    if (nx > 64 || nrows > kDirectDDRows) return; // (k_pack has voided the evaluation)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cap_slots = D.cap_clusters * 8;
    // Layout of the cluster list.  Half-shell kernel (N3): SPLIT -- all owned clusters first (offsets from s_cl), the ghosts' in a
    // region of their own behind them (offsets from s_clg, counted from the first ghost cluster).  Full-shell kernel: a cell's
    // ghost clusters right behind its owned ones (k_nb_clusters_j walks cstart[c] .. cstart[c + 1]): ONE offset space, s_cl holds
    // the prefix of both kinds together and s_clg stays zero.
    constexpr bool SPLIT = N3;
    direct_row_prefix1(D.rowcl, nrows, s_cl);
    if (GHO) direct_row_prefix1(X.rowclg, nrows, s_clg);
    direct_row_prefix1(D.rowbig, nrows, s_big);
    if (!SPLIT) {
        for (int r = threadIdx.x; r <= nrows; r += 256) {
            s_cl[r] += s_clg[r];
            s_clg[r] = 0;
        }
        __syncthreads();
    }
    const int tot_cl = s_cl[nrows], tot_clg = GHO ? s_clg[nrows] : 0, nbig = s_big[nrows];
    BUILD_STAMP(1);

    if ((int)blockIdx.x < D.n_items_blocks) {
        if (blockIdx.x == 0) {
            if (OWN) {
                __shared__ float s_red[6 * 4];
                const GridParams GN = grid_from_parts<256>(D.bbox_part, D.nblk_bbox, D.hmin, D.maxcells, s_red, X.expand);
                if (threadIdx.x == 0) {
                    *D.grid_next = GN;
                    st->n_clusters_own = tot_cl; // (interleaved layout: all of them, as the scan-based build reports it)
                    st->n_big = nbig;
                    st->ncells = ncells;
                    st->ncells_set[D.parity] = ncells;
                    st->cell_edge = (double)G.h;
                    D.cstart[ncells] = min(tot_cl, D.cap_clusters);
                }
            }
            if (GHO && threadIdx.x == 0) {
                st->n_clusters = tot_cl + tot_clg;
                if (PHASE == 2) atomicAdd(&st->n_big, nbig);
                X.istart[ncells] = min(tot_clg, D.cap_clusters);
                st->dd_excess_bits = 0u; // (read by this build's ghost count, which ran before; the next pack starts from zero)
                if (tot_cl + tot_clg > D.cap_clusters) atomicOr(&st->kernel_error, (int)KERR_BOUNDS);
            }
        }
        if (!N3 || PHASE == 1) return;
        // work items: offsets of the five owned rows and the nine ghost rows around the wave's row, from the populations
        int *const cs_w = reinterpret_cast<int *>(s_buf) + wave * ((kN3Runs + kN3GhostRuns) * 66);
        n3_items_rows((int)blockIdx.x, D.n_items_blocks, G, D.n3_items, D.n3_max_items, st, (D.n3_flags & 4) ? 2 : (D.n3_flags & 1),
                      !(D.n3_flags & 2), (D.n3_flags >> 8) > 0 ? min(D.n3_flags >> 8, kN3MaxCap) : kN3MaxCap,
                      [&](int row, int y, int z, N3Row &R) {
                          int rb[kN3Runs], rg[kN3GhostRuns];
                          rb[0] = row;
                          rb[1] = y + 1 < G.ny ? row + 1 : -1;
#pragma unroll
                          for (int dy = -1; dy <= 1; ++dy)
                              rb[3 + dy] = (z + 1 < G.nz && y + dy >= 0 && y + dy < G.ny) ? row + G.ny + dy : -1;
#pragma unroll
                          for (int r = 0; r < kN3GhostRuns; ++r) {
                              const int yy = y + r % 3 - 1, zz = z + r / 3 - 1;
                              rg[r] = (yy >= 0 && yy < G.ny && zz >= 0 && zz < G.nz) ? zz * G.ny + yy : -1;
                          }
                          int cnt[kN3Runs], cng[kN3GhostRuns];
#pragma unroll
                          for (int r = 0; r < kN3Runs; ++r) cnt[r] = (rb[r] >= 0 && lane < nx) ? D.count[rb[r] * nx + lane] : 0;
#pragma unroll
                          for (int r = 0; r < kN3GhostRuns; ++r) cng[r] = (rg[r] >= 0 && lane < nx) ? X.count_g[rg[r] * nx + lane] : 0;
                          wave_lds_sync(); // the previous row's offsets have been read
                          auto put = [&](int slot, int rowid, int cnt_l, const int *s_pref) {
                              const int cl = (cnt_l + 7) >> 3;
                              int inc = cl;
#pragma unroll
                              for (int o = 1; o < 64; o <<= 1) {
                                  const int u = __shfl_up(inc, o, 64);
                                  if (lane >= o) inc += u;
                              }
                              if (rowid >= 0) {
                                  const int base = s_pref[rowid];
                                  if (lane < nx) cs_w[slot * 66 + lane] = base + inc - cl;
                                  if (lane == 63) cs_w[slot * 66 + nx] = base + inc;
                              }
                          };
#pragma unroll
                          for (int r = 0; r < kN3Runs; ++r) {
                              put(r, rb[r], cnt[r], s_cl);
                              R.base[r] = rb[r] >= 0 ? r * 66 : -1;
                          }
#pragma unroll
                          for (int r = 0; r < kN3GhostRuns; ++r) {
                              put(kN3Runs + r, rg[r], cng[r], s_clg);
                              R.gb[r] = rg[r] >= 0 ? (kN3Runs + r) * 66 : -1;
                          }
                          wave_lds_sync();
                          R.cstart = cs_w;
                          R.nx = nx;
                          R.gstart = cs_w;
                          R.gbase = tot_cl;
                          R.err = &st->kernel_error;
                      });
        BUILD_STAMP(2);
        return;
    }

    const int bid = (int)blockIdx.x - D.n_items_blocks - nbb, nblk = D.n_order;
    {
        const int nz_cells = min(st->ncells_set[D.parity ^ 1], D.maxcells);
        for (int q = bid * 256 + (int)threadIdx.x; q < nz_cells; q += nblk * 256) {
            if (OWN) D.count_zero[q] = 0;
            if (GHO) X.count_g_zero[q] = 0;
        }
        for (int q = bid * 256 + (int)threadIdx.x; q < kDirectDDRows; q += nblk * 256) {
            if (OWN) {
                D.rowcl_zero[q] = 0;
                D.rowbig_zero[q] = 0;
            }
            if (GHO) {
                X.rowclg_zero[q] = 0;
                X.rowbigg_zero[q] = 0; // (whichever kind of build used that set: a one-launch build never counted into it)
            }
        }
    }
    const int nB = min(nbig, nblk - (nblk >> 2));
    const int kvec = min(D.slot_cap >> 6, kWaveCellMax / 64);
    for (int c = (bid - nB) * 4 + wave; bid >= nB && c < ncells; c += (nblk - nB) * 4) {
        const int row = c / nx, x = c - row * nx;
        const int ko_l = lane < nx ? D.count[row * nx + lane] : 0, kg_l = (GHO && lane < nx) ? X.count_g[row * nx + lane] : 0;
        const bool in_table = c < D.slot_cells;
        unsigned long long v[kWaveCellMax / 64];
#pragma unroll
        for (int h = 0; h < kWaveCellMax / 64; ++h) v[h] = ~0ull;
        const int ko = __shfl(ko_l, x, 64), kg = __shfl(kg_l, x, 64);
        const int k = PHASE == 0 ? ko + kg : PHASE == 1 ? ko : kg; // keys this launch sorts ...
        const int koff = PHASE == 2 ? ko : 0;                       // ... from this place of the cell's row of the slot table
        const int cb = s_cl[row] + direct_row_before(SPLIT ? (ko_l + 7) >> 3 : ((ko_l + 7) >> 3) + ((kg_l + 7) >> 3), lane, x);
        const int cg = SPLIT ? tot_cl + s_clg[row] + direct_row_before((kg_l + 7) >> 3, lane, x) : cb + ((ko + 7) >> 3);
        if (lane == 0) {
            if (OWN) D.cstart[c] = min(cb, D.cap_clusters);
            if (GHO) X.istart[c] = SPLIT ? min(cg - tot_cl, D.cap_clusters) : 0;
        }
        if (k > kWaveCellMax || k == 0) continue;
        if (!in_table || koff + k > D.slot_cap) continue; // (a void evaluation: the counting kernels flagged it)
#pragma unroll
        for (int h = 0; h < kWaveCellMax / 64; ++h)
            if (h * 64 < k && (h == 0 || h < kvec) && koff + h * 64 + lane < D.slot_cap)
                v[h] = D.keys[(size_t)c * D.slot_cap + koff + h * 64 + lane];
#pragma unroll
        for (int h = 0; h < kWaveCellMax / 64; ++h)
            if (h * 64 + lane >= k) v[h] = ~0ull;
        const int eo = PHASE == 2 ? 0 : ko, eg = PHASE == 1 ? 0 : kg;
        if (k <= 64) {
            if (k > 1) wave_sort_keys<1>(v, lane);
            direct_emit_wave_dd<1>(D, X, st, v, c, eo, eg, cb, cg, cap_slots, lane);
        } else if (k <= 128) {
            wave_sort_keys<2>(v, lane);
            direct_emit_wave_dd<2>(D, X, st, v, c, eo, eg, cb, cg, cap_slots, lane);
        } else {
            wave_sort_keys<4>(v, lane);
            direct_emit_wave_dd<4>(D, X, st, v, c, eo, eg, cb, cg, cap_slots, lane);
        }
    }
    for (int bi = bid; bid < nB && bi < nbig; bi += nB) {
        int lo = 0, hi = nrows;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_big[mid] <= bi) lo = mid;
            else hi = mid;
        }
        const int row = lo;
        const int ko_l = lane < nx ? D.count[row * nx + lane] : 0, kg_l = (GHO && lane < nx) ? X.count_g[row * nx + lane] : 0;
        unsigned long long m = __ballot((PHASE == 0 ? ko_l + kg_l : PHASE == 1 ? ko_l : kg_l) > kWaveCellMax);
        for (int k = bi - s_big[row]; k > 0 && m; --k) m &= m - 1;
        if (!m) continue;
        const int x = __ffsll((long long)m) - 1;
        const int c = row * nx + x;
        const int ko = __shfl(ko_l, x, 64), kg = __shfl(kg_l, x, 64);
        const int cnt = PHASE == 0 ? ko + kg : PHASE == 1 ? ko : kg, koff = PHASE == 2 ? ko : 0;
        const int cb = s_cl[row] + direct_row_before(SPLIT ? (ko_l + 7) >> 3 : ((ko_l + 7) >> 3) + ((kg_l + 7) >> 3), lane, x);
        const int cg = SPLIT ? tot_cl + s_clg[row] + direct_row_before((kg_l + 7) >> 3, lane, x) : cb + ((ko + 7) >> 3);
        if (threadIdx.x == 0) {
            if (OWN) D.cstart[c] = min(cb, D.cap_clusters);
            if (GHO) X.istart[c] = SPLIT ? min(cg - tot_cl, D.cap_clusters) : 0;
        }
        if (c >= D.slot_cells || koff + cnt > D.slot_cap) continue;
        const unsigned long long *kp = D.keys + (size_t)c * D.slot_cap + koff;
        __syncthreads();
        if (cnt > CAP) { // owned beads and ghosts cannot be kept in separate clusters without the sort: the evaluation is void
            if (threadIdx.x == 0) atomicOr(&st->kernel_error, (int)KERR_ORDER_DD);
            continue;
        }
        const unsigned long long *sorted = s_buf;
        if (cnt <= 512) { // by counting (block_rank_sort): input copy in s_buf[0 .. 512), sorted keys behind it
            static_assert(CAP >= 1024, "two regions of 512 keys");
            sorted = s_buf + 512;
            if (!block_rank_sort<unsigned long long>(s_buf, s_buf + 512, kp, cnt)) { // equal keys (a void evaluation's: a ghost that arrived twice)
                block_sort_regs<2>(s_buf, kp, cnt, 512);
                sorted = s_buf;
            }
        } else if (cnt <= 1024) {
            block_sort_regs<4>(s_buf, kp, cnt, 1024);
        } else {
            int n2 = 128;
            while (n2 < cnt) n2 <<= 1;
            for (int q = threadIdx.x; q < n2; q += 256) {
                unsigned long long kq = ~0ull;
                if (q < cnt) kq = kp[q];
                s_buf[q] = kq;
            }
            __syncthreads();
            for (int k = 2; k <= n2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
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
                }
            }
            __syncthreads();
        }
        emit_clusters(c, 0, cnt, PHASE == 2 ? 0 : ko, cb, PHASE == 1 ? -1 : cg, nullptr, D.pos4, D.spos4, D.cl_lo, nullptr, threadIdx.x,
                      256, X.own, sorted, D.sbead, D.slot_of, cap_slots, st, D.n_beads);
    }
    BUILD_STAMP(2);
}

// The fullest cell of the last direct build, for the host's polls (the scan used to publish it per build): the counter set is
// intact until the next build zeroes it.
__global__ __launch_bounds__(256) void k_poll_stats(const int *__restrict__ count, const GridParams *__restrict__ grid,
                                                    MinState *__restrict__ st, const int *__restrict__ count_g = nullptr) {
    __shared__ int s_m[4];
    const int n = grid->ncells;
    int m = 0;
    for (int q = threadIdx.x; q < n; q += 256) m = max(m, count[q] + (count_g ? count_g[q] : 0));
    m = wave_max_i(m);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) st->max_per_cell = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
}

} // namespace mmx
