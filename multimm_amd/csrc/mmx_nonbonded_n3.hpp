// mmx_nonbonded_n3.hpp -- K2 (default): half-shell cluster-pair kernel, Newton's third law through LDS.
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
// Work item (k_n3_items builds the list after every cell scan) = a run of up to 32 consecutive clusters of one ROW of
// the cell grid (x is the fastest cell index, so a row is one contiguous stretch of the cell-sorted cluster list); the
// run may span several cells when they are sparse.  Half shell by cells: the j candidates of the run are, in
// increasing cluster order, (0) the clusters of its own row from the run's first cluster to the end of the cell after
// its last one, (1) row y+1 and (2-4) the three rows of layer z+1, each from the cell before the run's first cell to
// the cell after its last one: five contiguous stretches of the cluster list.  Their concatenation is the index
// space of the LDS window.  An i-cluster takes every candidate with a cluster id >= its own (lower ids of its own row
// are i-clusters themselves and take the pair from their side; its own id = the self tile) that the box-box test
// accepts; candidates of cells that are not adjacent to the i-cluster's cell are at least one cell edge >= cutoff
// away, so geometry alone keeps the half shell exact.  A run whose candidates do not fit the window is processed in
// several passes over slices of the index space (only the densest cells, e.g. the lattice start).
//
// Execution: ONE persistent workgroup of 16 waves per CU pulls items from a global queue.  The candidates' boxes and
// ids of the NEXT unit (item, pass) are staged into a second LDS buffer by each wave as soon as it runs out of
// i-clusters of the current one (which it grabs one at a time from an LDS counter), so the dependent global loads of
// a unit's set-up hide under the pair arithmetic of the other waves; a unit costs two workgroup barriers.
//
// LDS accumulation is int32 fixed point (2^-13 kJ/mol/nm): ds_add_f32 is serialised on gfx950 (measured,
// scripts/ubench/lds_atomic.hip: 193 cycles per wave instruction against 4.5-7 for ds_add_u32), and integer sums do
// not depend on the order in which the waves arrive.  The flush and the i side use float atomics on global memory,
// whose order is not fixed: results are reproducible to rounding, not bitwise.  `deterministic = 1` selects
// k_nb_clusters_j, which is.
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
constexpr int kN3ItemClusters = 32; // i-clusters per work item (grabbed one at a time by the waves)
constexpr int kN3List = 128;        // accepted j-clusters buffered per wave before a sweep
constexpr int kN3MaxCap = 608;      // largest LDS window, in clusters (14 cells of 43 clusters: the lattice start)
// A window slot receives at most one batch sum per i-cluster of the item, so sums below 2^31 / 32 units cannot
// overflow; a larger one (overlapping beads) bypasses LDS with a global float atomic.
constexpr float kN3Fix = 8192.f;
constexpr float kN3FixLim = 67108864.f * 0.999f; // 2^26 units = 8192 kJ/mol/nm

struct N3Item { // 64 bytes
    int a, n;          // i-clusters [a, a + n)
    int T;             // length of the concatenated candidate runs
    int pad0;
    int rlo[5], rn[5]; // the runs: clusters [rlo, rlo + rn)
    int pad1[2];
};

// dynamic LDS of k_nb_n3 for a window of `cap` clusters: force sums, two candidate buffers (boxes + ids)
constexpr size_t n3_lds_bytes(int cap) {
    return sizeof(int) * 3 * ((size_t)cap * 8 + 8) + 2 * (sizeof(float4) * 2 * ((size_t)cap + 1) + sizeof(int) * ((size_t)cap + 8));
}

// ---- item builder ---------------------------------------------------------------------------------------------
// One thread per row of the cell grid walks the row's clusters and cuts them into runs: at most kN3ItemClusters
// clusters, and spanning more than one cell only while the candidate set of the run still fits the LDS window.
struct N3Row {
    const int *cstart;
    int nx, base[5]; // cell index of x = 0 in the five candidate rows (-1: the row does not exist)
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

template <bool EMIT>
__device__ __forceinline__ int n3_walk_row(const N3Row &R, int cap, N3Item *__restrict__ out) {
    const int *cs = R.cstart + R.base[0];
    const int c_hi = cs[R.nx];
    int a = cs[0], xa = 0, count = 0;
    while (a < c_hi) {
        while (cs[xa + 1] <= a) ++xa; // cell of cluster a
        int n = min(kN3ItemClusters, c_hi - a);
        int xb = xa;
        while (cs[xb + 1] < a + n) ++xb; // cell of the run's last cluster
        int rlo[5], rn[5];
        int t = R.T(a, xa, xb, rlo, rn);
        while (xb > xa && t > cap) { // too many candidates: end the run with the cell before xb
            n = cs[xb] - a;
            xb = xa;
            while (cs[xb + 1] < a + n) ++xb;
            t = R.T(a, xa, xb, rlo, rn);
        }
        if (EMIT) {
            N3Item it;
            it.a = a;
            it.n = n;
            it.T = t;
            it.pad0 = 0;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                it.rlo[r] = rlo[r];
                it.rn[r] = rn[r];
            }
            it.pad1[0] = it.pad1[1] = 0;
            out[count] = it;
        }
        ++count;
        a += n;
    }
    return count;
}

__global__ __launch_bounds__(256) void k_n3_items(const GridParams *__restrict__ grid, const int *__restrict__ cstart,
                                                  N3Item *__restrict__ items, int cap, int max_items,
                                                  MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    __shared__ int s_w[4], s_base;
    const GridParams G = *grid;
    const int nrows = G.ny * G.nz;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // rows are dealt out in blocks of 256 to the workgroups (usually one round)
    for (int r0 = blockIdx.x * 256; r0 < nrows; r0 += gridDim.x * 256) {
        const int row = r0 + threadIdx.x;
        N3Row R;
        R.cstart = cstart;
        R.nx = G.nx;
        int cnt = 0;
        if (row < nrows) {
            const int y = row % G.ny, z = row / G.ny;
            R.base[0] = row * G.nx;
            R.base[1] = y + 1 < G.ny ? (row + 1) * G.nx : -1;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
                R.base[3 + dy] = (z + 1 < G.nz && y + dy >= 0 && y + dy < G.ny) ? (row + G.ny + dy) * G.nx : -1;
            cnt = n3_walk_row<false>(R, cap, nullptr);
        }
        // exclusive scan of the counts over the workgroup, then one atomic for the workgroup's slice of the item list
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        __syncthreads();
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_w[w];
        if (threadIdx.x == 0) s_base = atomicAdd(&st->n3_items, s_w[0] + s_w[1] + s_w[2] + s_w[3]);
        __syncthreads();
        const int first = s_base + woff + inc - cnt;
        if (row < nrows && cnt > 0) {
            if (first + cnt <= max_items) n3_walk_row<true>(R, cap, items + first);
            else st->nan_seen = 1; // cannot happen (the list holds one item per cluster): surface it as a failed evaluation
        }
    }
}

// fsort: force (not gradient) per cluster slot, SoA [3][fstride]; zero on entry, k_nb_n3_unsort zeroes it again.
template <int PMODE, bool EV, bool GAUSS, bool NOENERGY>
__global__ __launch_bounds__(kN3Threads, 4) void k_nb_n3(const FFParams P, const float4 *__restrict__ spos4,
                                                          const float4 *__restrict__ cl_box,
                                                          const N3Item *__restrict__ items, MinState *__restrict__ st,
                                                          float *__restrict__ fsort, const int fstride,
                                                          double *__restrict__ part, const float sc, const int cap,
                                                          const int diag = 0, unsigned long long *__restrict__ dbg = nullptr) {
    if (st->phase == PH_DONE) return;
    unsigned long long t_cmp = 0, t_stage = 0, t_bar = 0, t_flush = 0, t_mark = 0; // diag & 128: where the time goes
    // dynamic LDS: [3][cap*8 + 8] int force sums per window slot, x | y | z (fixed point, see kN3Fix; the last 8 slots
    // are a dummy cluster), then two candidate buffers {[cap + 1][2] float4 boxes, [cap + 8] cluster ids}
    extern __shared__ __attribute__((aligned(16))) int s_f[];
    __shared__ unsigned short s_list[kN3Waves][kN3List + 72];
    __shared__ float4 s_ring[kN3Waves][128];
    __shared__ __attribute__((aligned(32))) float s_tab[5 * 8];
    __shared__ float s_arow[kN3Waves][kCl * 8];
    __shared__ double s_e[2][kN3Waves];
    __shared__ int s_item[4]; // queue positions of the units to come (ring of 3) ...
    __shared__ int s_grab[2]; // ... and the next i-cluster of the current unit (one counter per parity)
    // the wave index in a scalar register: hipcc cannot prove threadIdx.x >> 6 uniform and would otherwise keep the
    // scalar i beads in vector registers
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane >> 3, slot = lane & 7;
    const int fstr = cap * 8 + 8;
    int *sfx = s_f, *sfy = s_f + fstr, *sfz = s_f + 2 * fstr;
    const int cand_ints = 8 * (cap + 1) + (cap + 8); // one candidate buffer, in 4-byte words
    int *cand0 = s_f + 3 * fstr;
    const int far_cl = P.n_all; // a resident all-padding cluster (8 beads at -1e18)
    const int n_items = st->n3_items;
    if (threadIdx.x < 40) {
        const bool in5 = (threadIdx.x & 7) < 5;
        s_tab[threadIdx.x] = in5 ? P.table[(threadIdx.x >> 3) * 5 + (threadIdx.x & 7)] : 0.f;
    }
    for (int q = threadIdx.x * 4; q < 3 * fstr; q += kN3Threads * 4) *reinterpret_cast<int4 *>(s_f + q) = make_int4(0, 0, 0, 0);
    if (threadIdx.x < 16) { // list padding points at window entry `cap` of either buffer: the dummy cluster
        int *jc = cand0 + (threadIdx.x >> 3) * cand_ints + 8 * (cap + 1);
        jc[cap + (threadIdx.x & 7)] = far_cl;
    }
    if (threadIdx.x == 0) {
        // the first two units of this workgroup; everything after them is fetched two units ahead
        s_item[0] = atomicAdd(&st->n3_queue, 2);
        s_item[1] = s_item[0] + 1;
        s_grab[0] = s_grab[1] = 0;
    }
    __syncthreads();
    unsigned short *list = s_list[wave];
    float4 *ring = s_ring[wave];
    const float *arow = s_arow[wave];
    // scaled length units, factored constants: exactly the LEAN instance of k_nb_clusters_j
    const float sc2 = sc * sc;
    const float rc2 = P.rc2max * sc2;
    const float s3 = P.ev_sigma * P.ev_sigma * P.ev_sigma;
    const float ev_c = P.ev_eps * s3 * s3;
    const float tiny = 1e-20f;
    const float sc6 = sc2 * sc2 * sc2;
    const float escale = (EV && PMODE == 6) ? ev_c * sc6 : 1.f;
    const float pscale = EV ? P.ev_power * escale * sc : 1.f;
    const float g_k = P.g_inv_rc2 / (sc * pscale);
    const float fix_k = pscale * kN3Fix;     // scaled pair-loop units -> fixed point
    const float fix_lim = kN3FixLim / fix_k; // largest |batch sum| the fixed-point path takes
    const float unfix = -1.f / kN3Fix;       // fixed point -> force on the j bead (reaction: minus)
    const float rs_s = P.ev_rs * sc, sigma_s = P.ev_sigma * sc;
    const float nbig = -1e30f;
    const float cut_all = 1e30f * fminf(rc2, 1e6f);
    // energy of the r = 0 self pair, by the very operations of the pair loop
    float eself = 0.f;
    if (EV && !NOENERGY) {
        const float us = __builtin_amdgcn_rcpf(fmaf(tiny, __builtin_amdgcn_rsqf(tiny), rs_s));
        if (PMODE == 6) {
            const float u2 = us * us;
            eself = (u2 * u2) * u2;
        } else {
            eself = P.ev_eps * ev_pow<PMODE>(sigma_s * us, P.ev_power);
        }
    }
    double acc_ev = 0.0, acc_g = 0.0;

    // Stages window [wlo, wlo + cap) of item `it` into candidate buffer `buf`: this wave's share (every 16th group of
    // 64 candidates).  Candidate k of the concatenated runs -> cluster id, box.
    auto stage = [&](const N3Item &it, int wlo, int buf) {
        float4 *box = reinterpret_cast<float4 *>(cand0 + buf * cand_ints);
        int *jcs = cand0 + buf * cand_ints + 8 * (cap + 1);
        const int nwin = min(it.T, wlo + cap) - wlo;
        int woff[5];
        woff[0] = 0;
#pragma unroll
        for (int r = 1; r < 5; ++r) woff[r] = woff[r - 1] + it.rn[r - 1];
        for (int k = wave * 64 + lane; k < nwin; k += kN3Threads) {
            const int kk = wlo + k;
            int jc = it.rlo[0] + kk;
#pragma unroll
            for (int r = 1; r < 5; ++r) jc = kk >= woff[r] ? it.rlo[r] + (kk - woff[r]) : jc;
            jcs[k] = jc;
            box[2 * k] = cl_box[2 * jc];
            box[2 * k + 1] = cl_box[2 * jc + 1];
        }
    };
    auto load_item = [&](int q) {
        N3Item it;
        const int4 *p = reinterpret_cast<const int4 *>(items + (q < n_items ? q : 0));
        const int4 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
        it.a = __builtin_amdgcn_readfirstlane(v0.x);
        it.n = __builtin_amdgcn_readfirstlane(v0.y);
        it.T = __builtin_amdgcn_readfirstlane(v0.z);
        it.pad0 = 0;
        it.rlo[0] = __builtin_amdgcn_readfirstlane(v1.x);
        it.rlo[1] = __builtin_amdgcn_readfirstlane(v1.y);
        it.rlo[2] = __builtin_amdgcn_readfirstlane(v1.z);
        it.rlo[3] = __builtin_amdgcn_readfirstlane(v1.w);
        it.rlo[4] = __builtin_amdgcn_readfirstlane(v2.x);
        it.rn[0] = __builtin_amdgcn_readfirstlane(v2.y);
        it.rn[1] = __builtin_amdgcn_readfirstlane(v2.z);
        it.rn[2] = __builtin_amdgcn_readfirstlane(v2.w);
        it.rn[3] = __builtin_amdgcn_readfirstlane(v3.x);
        it.rn[4] = __builtin_amdgcn_readfirstlane(v3.y);
        it.pad1[0] = it.pad1[1] = 0;
        if (q >= n_items) it.n = 0; // past the end of the queue: an empty unit (every wave sees the same)
        return it;
    };

    // ---- unit pipeline: u = units done so far; unit u lives in candidate buffer u & 1
    int qpos = 0; // ring position of the current item's queue index in s_item
    N3Item cur = load_item(s_item[0]);
    int wlo = 0;
    if (cur.n > 0) stage(cur, 0, 0);
    __syncthreads();
    for (int u = 0; cur.n > 0; ++u) {
        const int buf = u & 1;
        const float4 *s_box = reinterpret_cast<const float4 *>(cand0 + buf * cand_ints);
        const int *s_jc = cand0 + buf * cand_ints + 8 * (cap + 1);
        const int nwin = min(cur.T, wlo + cap) - wlo;
        const bool last_pass = wlo + cap >= cur.T;
        // what comes after this unit: the next pass of the same item, or the next item of the queue
        if (threadIdx.x == 0 && last_pass) s_item[(qpos + 2) % 3] = atomicAdd(&st->n3_queue, 1); // two items ahead
        // ---- compute: grab i-clusters of the item one at a time
        if (diag & 128) t_mark = __builtin_amdgcn_s_memtime();
        for (;;) {
            int gi = 0;
            if (lane == 0) gi = atomicAdd(&s_grab[buf], 1);
            gi = __builtin_amdgcn_readfirstlane(gi);
            if (gi >= cur.n) break;
            const int icl = cur.a + gi;
            const float4 lo_i = cl_box[2 * icl], hi_i = cl_box[2 * icl + 1];
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
            // the i-cluster's own place in the window: candidates before it have lower cluster ids (i-clusters of this
            // item or of an earlier one: they take those pairs); it is itself a candidate (the self tile)
            const int own_k = (icl - cur.rlo[0]) - wlo;
            if (own_k < nwin) {
                const int own_lc = own_k >= 0 ? own_k : -1;
                const int k0 = max(own_k, 0);
                int nlist = 0, rcount = 0, rhead = 0;
                for (int g = k0 & ~63; g < nwin; g += 64) {
                    // ---- cull: 64 candidate clusters of the window per step, boxes from LDS
                    const int k = g + lane;
                    bool ok = false;
                    if (k >= k0 && k < nwin) {
                        const float4 lo_j = s_box[2 * k], hi_j = s_box[2 * k + 1];
                        const float dx = fmaxf(fmaxf(lo_j.x - hi_i.x, lo_i.x - hi_j.x), 0.f);
                        const float dy = fmaxf(fmaxf(lo_j.y - hi_i.y, lo_i.y - hi_j.y), 0.f);
                        const float dz = fmaxf(fmaxf(lo_j.z - hi_i.z, lo_i.z - hi_j.z), 0.f);
                        ok = fmaf(dx, dx, fmaf(dy, dy, dz * dz)) < rc2;
                    }
                    const unsigned long long mask = __ballot(ok);
                    if (ok) list[nlist + prefix_count(mask)] = (unsigned short)k;
                    nlist += __builtin_amdgcn_readfirstlane(__popcll(mask));
                    const bool last = g + 64 >= nwin;
                    if (nlist < kN3List - 64 && !last) continue;
                    if (diag & 16) { // timing diagnosis only: cull without sweep
                        fx[0] += (float)nlist;
                        nlist = 0;
                        continue;
                    }
                    if (lane < 8) list[nlist + lane] = (unsigned short)cap; // pad to a multiple of 8: the dummy cluster
                    wave_lds_sync();
                    const int nsteps = max((nlist + 7) >> 3, 1);
                    // ---- sweep: 8 j-clusters (64 j beads) per step; ids two steps ahead, positions one
                    int ln = list[sub];
                    float4 qn = spos4[(unsigned)s_jc[ln] * kCl + slot];
                    int ln2 = nsteps > 1 ? list[8 + sub] : cap;
                    int jn = s_jc[ln2];
                    for (int t = 0; t < nsteps; ++t) {
                        float4 q = qn;
                        const int lq = ln;
                        if (t + 1 < nsteps) {
                            qn = spos4[(unsigned)jn * kCl + slot];
                            ln = ln2;
                            ln2 = t + 2 < nsteps ? list[(t + 2) * 8 + sub] : cap;
                            jn = s_jc[ln2];
                        }
                        // per-bead cull against the i box; survivors are compacted through the ring with their LDS slot
                        {
                            const float bx = fmaxf(fmaxf(lo_i.x - q.x, q.x - hi_i.x), 0.f);
                            const float by = fmaxf(fmaxf(lo_i.y - q.y, q.y - hi_i.y), 0.f);
                            const float bz = fmaxf(fmaxf(lo_i.z - q.z, q.z - hi_i.z), 0.f);
                            const bool okb = fmaf(bx, bx, fmaf(by, by, bz * bz)) < rc2;
                            const unsigned long long mb = __ballot(okb);
                            if (okb) {
                                q.w = __int_as_float((((lq << 3) | slot) << 3) | (__float_as_int(q.w) & 7));
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
                            rhead = (rhead + took) & 127;
                            rcount -= took;
                            const int wq = __float_as_int(q.w);
                            const int lj = wq & 7;
                            const int jslot = wq >> 3;
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
                                    const float uu = __builtin_amdgcn_rcpf(fmaf(r2t, rinv, rs_s));
                                    float E;
                                    if (PMODE == 6) {
                                        const float u2 = uu * uu;
                                        E = (u2 * u2) * u2;
                                    } else {
                                        E = P.ev_eps * ev_pow<PMODE>(sigma_s * uu, P.ev_power);
                                    }
                                    if (!NOENERGY) eb = fmaf(E, in, eb);
                                    fs = E * (uu * rinv);
                                }
                                if (GAUSS) {
                                    const float gg = arow[s * 8 + lj] * __builtin_amdgcn_exp2f(-r2t);
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
                            if (!NOENERGY) {
                                // self tile: both orders of a pair were swept (weight 1/2), and the r = 0 pair goes out again
                                if (EV) ee = fmaf(self ? eb - eself : eb, self ? 0.5f : 1.f, ee);
                                if (GAUSS) eg = fmaf(self ? gb + arow[(jslot & 7) * 8 + lj] : gb, self ? 0.5f : 1.f, eg);
                            }
                            if (!(diag & 2)) { // reaction on the j beads (sign and unit: at the flush)
                                const bool big = fmaxf(fmaxf(fabsf(fjx), fabsf(fjy)), fabsf(fjz)) >= fix_lim;
                                if (__builtin_expect(__ballot(big && !self) != 0ull, 0)) {
                                    if (big && !self) { // rare (overlapping beads): straight to global memory
                                        const int gs = s_jc[jslot >> 3] * kCl + (jslot & 7);
                                        atomicAdd(fsort + gs, -pscale * fjx);
                                        atomicAdd(fsort + fstride + gs, -pscale * fjy);
                                        atomicAdd(fsort + 2 * fstride + gs, -pscale * fjz);
                                    }
                                }
                                // lanes of the self tile (and the rare large sums) add into the dummy cluster: no branch
                                // around the adds, so the j-side FMAs stay in the pair loop instead of keeping all
                                // eight (fs, d) sets alive behind it
                                const int tslot = (self || big) ? cap * 8 + (lane & 7) : jslot;
                                atomicAdd(sfx + tslot, __float2int_rn(fjx * fix_k));
                                atomicAdd(sfy + tslot, __float2int_rn(fjy * fix_k));
                                atomicAdd(sfz + tslot, __float2int_rn(fjz * fix_k));
                            }
                        }
                    }
                    nlist = 0;
                    wave_lds_sync();
                }
            }
            // ---- i side: fold over the wave; lane s (< 8) ends up owning bead s of the i-cluster
            float ofx = 0.f, ofy = 0.f, ofz = 0.f;
#pragma unroll
            for (int s = 0; s < kCl; ++s) {
                const float a0 = wave_sum_dpp(fx[s]), a1 = wave_sum_dpp(fy[s]), a2 = wave_sum_dpp(fz[s]);
                if (lane == s) {
                    ofx = a0 * pscale;
                    ofy = a1 * pscale;
                    ofz = a2 * pscale;
                }
            }
            if (lane < kCl && own_w >= 0 && !(diag & 8)) {
                atomicAdd(fsort + icl * kCl + lane, ofx);
                atomicAdd(fsort + fstride + icl * kCl + lane, ofy);
                atomicAdd(fsort + 2 * fstride + icl * kCl + lane, ofz);
            }
            if (!NOENERGY) {
                acc_ev += (double)escale * wave_sum((double)ee);
                acc_g += (double)wave_sum(eg);
            }
        }
        // ---- out of i-clusters: stage the next unit into the other candidate buffer
        if (diag & 128) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_cmp += t - t_mark;
            t_mark = t;
        }
        N3Item nxt = cur;
        int nwlo = wlo + cap, nqpos = qpos;
        if (last_pass) {
            nqpos = (qpos + 1) % 3;
            nxt = load_item(s_item[nqpos]); // written two units ago, behind a barrier
            nwlo = 0;
        }
        if (nxt.n > 0) stage(nxt, nwlo, buf ^ 1);
        if (diag & 128) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_stage += t - t_mark;
            t_mark = t;
        }
        __syncthreads(); // every wave is done with the window and with the staging of the next unit
        if (diag & 128) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_bar += t - t_mark;
            t_mark = t;
        }
        // ---- flush the window: fsort[slot] -= sum / 2^13 (the 8 slots of a cluster are contiguous, neighbouring
        // candidates mostly too: 256-byte atomic wave instructions), and zero it for the next unit
        for (int e = threadIdx.x; e < ((diag & 32) ? 0 : nwin * 8); e += kN3Threads) {
            const int vx = sfx[e], vy = sfy[e], vz = sfz[e];
            if ((vx | vy | vz) != 0) {
                sfx[e] = 0;
                sfy[e] = 0;
                sfz[e] = 0;
                if (!(diag & 1)) {
                    const int gs = s_jc[e >> 3] * kCl + (e & 7);
                    atomicAdd(fsort + gs, unfix * (float)vx);
                    atomicAdd(fsort + fstride + gs, unfix * (float)vy);
                    atomicAdd(fsort + 2 * fstride + gs, unfix * (float)vz);
                }
            }
        }
        if (threadIdx.x < 8) sfx[cap * 8 + threadIdx.x] = sfy[cap * 8 + threadIdx.x] = sfz[cap * 8 + threadIdx.x] = 0;
        if (threadIdx.x == 0) s_grab[buf] = 0; // this parity is used again two units from now
        cur = nxt;
        wlo = nwlo;
        qpos = nqpos;
        if (diag & 128) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_flush += t - t_mark;
            t_mark = t;
        }
        __syncthreads();
        if (diag & 128) t_bar += __builtin_amdgcn_s_memtime() - t_mark;
    }
    if ((diag & 128) && dbg && lane == 0) {
        atomicAdd(dbg + 0, t_cmp);
        atomicAdd(dbg + 1, t_stage);
        atomicAdd(dbg + 2, t_bar);
        atomicAdd(dbg + 3, t_flush);
        atomicAdd(dbg + 4, 1ull);
    }
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
    }
}

// g[bead] -= fsort[slot] for every real bead of the cluster list, fsort back to zero for the next evaluation, and the
// item queue rewound (the pair kernel may be launched again on the same cell build: mmx_time_kernel).
__global__ __launch_bounds__(256) void k_nb_n3_unsort(const float4 *__restrict__ spos4, float *__restrict__ fsort,
                                                      const int fstride, float *__restrict__ g,
                                                      MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) st->n3_queue = 0;
    const int nsl = st->n_clusters * kCl;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nsl; i += gridDim.x * 256) {
        const int bead = __float_as_int(spos4[i].w) >> 3;
        const float fx = fsort[i], fy = fsort[fstride + i], fz = fsort[2 * fstride + i];
        fsort[i] = 0.f;
        fsort[fstride + i] = 0.f;
        fsort[2 * fstride + i] = 0.f;
        if (bead >= 0) {
            float *gb = g + 3 * (size_t)bead; // the bonded terms wrote the gradient first
            gb[0] -= fx;
            gb[1] -= fy;
            gb[2] -= fz;
        }
    }
}

} // namespace mmx
