// mmx_nonbonded.hpp -- K2 (cell-list pair kernel) and K2x (exact all-pairs kernel).
//
// Pair physics (reference: model.py:199 EV power law; model.py:246-250 / 322-328 compartment Gaussians):
//   E_ev  = eps*(sigma/(r+r_small))^p            F_i += p*E_ev/((r+r_small) r) * d,   d = x_i - x_j
//   E_g   = -A(s_i,s_j)*exp(-r^2/(2 rc^2))       F_i += -(A/rc^2)*exp(.) * d
// Every bead sums over its full neighbour shell (each pair is visited from both ends, energies are
// halved): no atomics, no write conflicts, summation order fixed by the in-cell bead order.
#pragma once
#include "mmx_common.hpp"
#include <type_traits>

namespace mmx {

template <int PMODE>
__device__ __forceinline__ float ev_pow(float t, float p) {
    if (PMODE == 6) {
        const float t2 = t * t;
        return t2 * t2 * t2;
    } else if (PMODE == 3) {
        return t * t * t;
    } else {
        return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(t)); // v_exp_f32(p * v_log_f32(t))
    }
}

struct PairAcc {
    float fx, fy, fz, eev, eg;
};

// ---- alternative functional forms of the pair terms ---------------------------------------
//   EV  gaussian_core  E = eps*exp(-r^2/(2 sigma^2))                       model.py:205-209
//   COB/SCB yukawa     E = -A*exp(-r/lambda)/r, lambda = r_comp            model.py:262-274, 340-356
//   COB/SCB theta      E = -A*step(r_comp - r)  (no force)                 model.py:277-288, 359-377
// The COB yukawa expression reads s1 twice (model.py:266-267): its amplitude depends on ONE bead of the pair,
// which OpenMM leaves undefined for a CustomNonbondedForce; we follow the Reference platform's evaluation order
// (particle 1 = the lower bead index).  Coincident beads (r = 0) get no yukawa term (the reference's is -inf).
// One attraction term of form `form` with amplitude A; returns the energy, adds -dE/dr / r to fs.
__device__ __forceinline__ float comp_form_term(const FFParams &P, const FormParams &Q, int form, float A, float r2, float r, float rinv,
                                                float &fs) {
    if (form == 0) {
        const float gg = A * __builtin_amdgcn_exp2f(r2 * P.g_c2);
        fs = fmaf(-gg, P.g_inv_rc2, fs);
        return -gg;
    } else if (form == 1) {
        const float y = r2 > 0.f ? A * __builtin_amdgcn_exp2f(r * Q.g_yuk) * rinv : 0.f;
        fs = fmaf(-y * rinv, __builtin_amdgcn_rcpf(Q.g_rcomp) + rinv, fs);
        return -y;
    }
    return r <= Q.g_rcomp ? -A : 0.f;
}

// Generic pair: every form by run-time (wave-uniform) switches.  wi/wj = packed (bead<<3 | label+2) words,
// tabc/tabs = COB/SCB amplitude tables with row stride `ts`.  in_ev/in_g = cutoff masks (0/1).
template <bool EV, bool GAUSS>
__device__ __forceinline__ void pair_generic(const FFParams &P, const FormParams &Q, float r2, int wi, int wj, const float *tabc,
                                             const float *tabs, int ts, float in_ev, float in_g, float &e_ev,
                                             float &e_g, float &fs) {
    const float r2s = r2 + 1e-20f;
    const float rinv = __builtin_amdgcn_rsqf(r2s);
    const float r = r2s * rinv;
    const float notself = wi != wj ? 1.f : 0.f;
    e_ev = 0.f;
    e_g = 0.f;
    fs = 0.f;
    if (EV) {
        float E, f1;
        if (Q.ev_form == 0) {
            const float u = __builtin_amdgcn_rcpf(r + P.ev_rs);
            E = P.ev_eps * __builtin_amdgcn_exp2f(P.ev_power * __builtin_amdgcn_logf(P.ev_sigma * u));
            f1 = P.ev_power * E * u * rinv;
        } else {
            E = P.ev_eps * __builtin_amdgcn_exp2f(r2 * Q.ev_gc2);
            f1 = E * Q.ev_inv_s2;
        }
        const float m = in_ev * notself;
        e_ev = E * m;
        fs = f1 * m;
    }
    if (GAUSS) {
        const int li = wi & 7, lj = wj & 7;
        float f2 = 0.f, e2 = 0.f;
        if (Q.has_cob) {
            const float A = Q.cob_form == 1 ? Q.cob_a[wi < wj ? li : lj] : tabc[li * ts + lj];
            e2 += comp_form_term(P, Q, Q.cob_form, A, r2, r, rinv, f2);
        }
        if (Q.has_scb) e2 += comp_form_term(P, Q, Q.scb_form, tabs[li * ts + lj], r2, r, rinv, f2);
        const float m = in_g * notself;
        e_g = e2 * m;
        fs = fmaf(f2, m, fs);
    }
}

// One (i,j) interaction accumulated on the i side.  `wi`/`wj` are the packed (bead<<3 | label+2) words;
// equal words mean the same bead (self term skipped).  tab5 points at row s_i of the amplitude table.
template <int PMODE, bool EV, bool GAUSS, bool FORMS = false>
__device__ __forceinline__ void pair_accum(const FFParams &P, const float4 pi, const int wi, const float4 q,
                                           const float *tab5, PairAcc &a, const FormParams *Q = nullptr,
                                           const float *tabc = nullptr, const float *tabs = nullptr) {
    const float dx = pi.x - q.x, dy = pi.y - q.y, dz = pi.z - q.z;
    const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const int wj = __float_as_int(q.w);
    if (FORMS) {
        if (r2 < P.rc2max && wj != wi) {
            float e1, e2, fs;
            pair_generic<EV, GAUSS>(P, *Q, r2, wi, wj, tabc, tabs, 5, r2 < P.ev_rc2 ? 1.f : 0.f,
                                    r2 < P.g_rc2 ? 1.f : 0.f, e1, e2, fs);
            a.eev += e1;
            a.eg += e2;
            a.fx = fmaf(fs, dx, a.fx);
            a.fy = fmaf(fs, dy, a.fy);
            a.fz = fmaf(fs, dz, a.fz);
        }
        return;
    }
    if (r2 < P.rc2max && wj != wi) {
        const float r2s = fmaxf(r2, 1e-20f);
        const float rinv = __builtin_amdgcn_rsqf(r2s);
        const float r = r2s * rinv;
        float fs = 0.f;
        if (EV) {
            const float u = __builtin_amdgcn_rcpf(r + P.ev_rs);
            float E = P.ev_eps * ev_pow<PMODE>(P.ev_sigma * u, P.ev_power);
            E = r2 < P.ev_rc2 ? E : 0.f;
            a.eev += E;
            fs = P.ev_power * E * u * rinv;
        }
        if (GAUSS) {
            const float A = tab5[wj & 7];
            float g = A * __builtin_amdgcn_exp2f(r2 * P.g_c2);
            g = r2 < P.g_rc2 ? g : 0.f;
            a.eg -= g;
            fs = fmaf(-g, P.g_inv_rc2, fs);
        }
        a.fx = fmaf(fs, dx, a.fx);
        a.fy = fmaf(fs, dy, a.fy);
        a.fz = fmaf(fs, dz, a.fz);
    }
}

// ------------------------------------------------------------------------------------------------
// K2 v1: one 192-thread block per work item {cell, chunk of <=64 home beads}; wave w sweeps the
// z-layer cz+w-1 of the 27-cell stencil (3 x-contiguous runs of sorted beads each).  Neighbour
// beads are gathered 64 at a time with one coalesced float4 load per lane into a per-wave LDS tile
// and consumed as broadcast ds_read_b128.  The three partial forces meet in LDS and wave 0 writes
// g = -F; energies leave through wave-shuffle reductions as one double per block.
// ------------------------------------------------------------------------------------------------
template <int PMODE, bool EV, bool GAUSS>
__global__ __launch_bounds__(192) void k_nb_cells(const FFParams P, const float4 *__restrict__ pos4,
                                                  const int *__restrict__ perm, const int *__restrict__ start,
                                                  const int2 *__restrict__ items,
                                                  const GridParams *__restrict__ grid,
                                                  const MinState *__restrict__ st, float *__restrict__ g,
                                                  double *__restrict__ part) {
    if (st->phase >= PH_DONE) return;
    __shared__ float4 s_tile[3][64];
    __shared__ float s_red[2][5][64];
    __shared__ float s_tab[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 25) s_tab[threadIdx.x] = P.table[threadIdx.x];
    const GridParams G = *grid;
    const int n_items = st->n_items;
    __syncthreads();
    double acc_ev = 0.0, acc_g = 0.0;

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int2 it = items[item];
        const int c = it.x;
        const int cs = start[c], ce = start[c + 1];
        const int hidx = cs + it.y * 64 + lane;
        const bool act = hidx < ce;
        const int bead = act ? perm[hidx] : 0;
        float4 pi = make_float4(1e18f, 1e18f, 1e18f, __int_as_float(-1));
        if (act) pi = pos4[bead];
        const int wi = __float_as_int(pi.w);
        const float *tab5 = s_tab + 5 * (act ? (wi & 7) : 0);
        const int cx = c % G.nx, cy = (c / G.nx) % G.ny, cz = c / (G.nx * G.ny);
        const int zz = cz + wave - 1;
        PairAcc a = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (zz >= 0 && zz < G.nz) {
            const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.nx - 1);
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = cy + dy;
                if (yy < 0 || yy >= G.ny) continue;
                const int row = (zz * G.ny + yy) * G.nx;
                const int rs = start[row + x0], re = start[row + x1 + 1];
                for (int base = rs; base < re; base += 64) {
                    const int j = base + lane;
                    float4 pj = make_float4(-1e18f, -1e18f, -1e18f, __int_as_float(-2));
                    if (j < re) pj = pos4[perm[j]];
                    wave_lds_sync(); // previous tile fully consumed
                    s_tile[wave][lane] = pj;
                    wave_lds_sync();
                    const int cnt = min(64, re - base);
#pragma unroll 4
                    for (int t = 0; t < cnt; ++t) {
                        const float4 q = s_tile[wave][t];
                        pair_accum<PMODE, EV, GAUSS>(P, pi, wi, q, tab5, a);
                    }
                }
            }
        }
        __syncthreads(); // s_red free (previous item consumed)
        if (wave > 0) {
            s_red[wave - 1][0][lane] = a.fx;
            s_red[wave - 1][1][lane] = a.fy;
            s_red[wave - 1][2][lane] = a.fz;
            s_red[wave - 1][3][lane] = a.eev;
            s_red[wave - 1][4][lane] = a.eg;
        }
        __syncthreads();
        if (wave == 0) {
            const float fx = a.fx + s_red[0][0][lane] + s_red[1][0][lane];
            const float fy = a.fy + s_red[0][1][lane] + s_red[1][1][lane];
            const float fz = a.fz + s_red[0][2][lane] + s_red[1][2][lane];
            const float ev = a.eev + s_red[0][3][lane] + s_red[1][3][lane];
            const float eg = a.eg + s_red[0][4][lane] + s_red[1][4][lane];
            if (act) {
                g[3 * bead] -= fx; // the bonded terms wrote the gradient first
                g[3 * bead + 1] -= fy;
                g[3 * bead + 2] -= fz;
            }
            const float sev = wave_sum(act ? ev : 0.f), seg = wave_sum(act ? eg : 0.f);
            acc_ev += 0.5 * (double)sev;
            acc_g += 0.5 * (double)seg;
        }
    }
    if (threadIdx.x == 0) {
        part[P_EV * kPartStride + blockIdx.x] = acc_ev;
        part[P_GAUSS * kPartStride + blockIdx.x] = acc_g;
    }
}

// ------------------------------------------------------------------------------------------------
// K2 (default): cluster-pair kernel.
// Measured on MI355X the pair arithmetic is VALU-issue bound (v1 below: ~100 % issue; packed
// v_pk_*_f32 issues in 4 cycles, i.e. no gain over two plain ops; v_cmp/v_cndmask/VOP3 integer ops cost
// ~4.3 cycles, transcendentals ~9.3) and only ~18 % of the 27-cell stencil candidates of a 64-bead home
// chunk lie inside the cutoff while 83 % of them are within the cutoff of SOME home bead, so neither
// wave-uniform skipping nor per-lane bit-mask compaction pays (both were built and measured, see
// DESIGN.md).  What pays is making both sides of a tile spatially small: beads are grouped in clusters
// of 8 consecutive entries of the cell-sorted, Hilbert-ordered list (k_cell_order writes their padded
// positions `spos4` and bounding boxes).  One wave owns one i-cluster: it tests the candidate
// j-clusters of the 27-cell stencil box-against-box (lanes = candidate clusters), compacts the
// survivors with a ballot into an LDS list, keeps its 8 i beads in scalar registers (v_readlane once)
// and lets its 64 lanes hold the beads of EIGHT accepted j-clusters at a time (lane -> cluster
// t*8 + lane/8, slot lane%8: one coalesced 128-B piece per cluster, loaded straight into registers,
// prefetched one step ahead and reused for all 8 i beads).  The inner loop over the 8 i beads is pure
// VALU with scalar i operands: no LDS tile, no staging pass, no barrier.  The 8x5 per-i accumulators
// are folded over the wave by shuffles once per i-cluster; energies leave as one double per block.
// ------------------------------------------------------------------------------------------------
// sat(a*b + c) in ONE VALU issue (clamp output modifier).  hipcc lowers __saturatef(fmaf()) to two
// v_cmp/v_cndmask pairs (~17 cycles on gfx950, measured); v_fma_f32 ... clamp is a plain 2-cycle op.
__device__ __forceinline__ float fma_sat(float a, float b_sgpr, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "s"(b_sgpr), "v"(c));
    return r;
}

// Number of set bits of a wave mask below this lane: v_mbcnt_lo + v_mbcnt_hi (2 ops instead of 2 ands + 2 bcnts).
__device__ __forceinline__ int prefix_count(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Sums v[0..7] over the 64 lanes; every lane l returns the total of v[l & 7].  Butterfly with a halving payload: after the
// step over lane bit k a lane only carries the values whose index has its own bit k.  Fixed order => deterministic.
__device__ __forceinline__ float fold8(const float (&v)[8], const int lane) {
    float w[4], u[2];
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) { // lane ^ 1: even lanes keep v[2k], odd lanes v[2k+1]
        const float keep = b0 ? v[2 * k + 1] : v[2 * k], give = b0 ? v[2 * k] : v[2 * k + 1];
        w[k] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0xb1, 0xf, 0xf, false)); // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) { // lane ^ 2: w[k] holds index 2k + b0; keep the one whose bit 1 is b1
        const float keep = b1 ? w[2 * k + 1] : w[2 * k], give = b1 ? w[2 * k] : w[2 * k + 1];
        u[k] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(give), 0x4e, 0xf, 0xf, false)); // quad_perm [2,3,0,1]
    }
    const float keep = b2 ? u[1] : u[0], give = b2 ? u[0] : u[1]; // lane ^ 4
    float t = keep + __shfl_xor(give, 4, 64);
    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xf, 0xf, false)); // row_ror:8 = lane ^ 8
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    return t;
}

constexpr int kCl = 8;        // beads per cluster
constexpr int kListCap = 448; // accepted j-clusters buffered per wave before a sweep

// Occupancy bound: 6 waves/SIMD = 80 VGPRs for the energy instances (7 would spill 32 B per lane, measured -3 %); the
// forces-only instances of MD (OPT & 512) fit 72 VGPRs without scratch and take 7 (+0.5 %; p = 3 would spill 12 B).
template <int PMODE, bool EV, bool GAUSS, bool SAMECUT, int OPT, bool FORMS = false>
__global__ __launch_bounds__(256, FORMS ? 2 : (((OPT & 512) && PMODE != 3) ? 7 : 6)) void k_nb_clusters_j(const FFParams P, const float4 *__restrict__ spos4,
                                                       const float4 *__restrict__ cl_lo,
                                                       const float4 *__restrict__ cl_hi,
                                                       const int *__restrict__ cstart,
                                                       const GridParams *__restrict__ grid,
                                                       const MinState *__restrict__ st, float *__restrict__ g,
                                                       double *__restrict__ part,
                                                       const FormParams *__restrict__ Qd = nullptr) {
    if (st->phase >= PH_DONE) return;
    FormParams Q;
    if (FORMS) Q = *Qd; // uniform loads into scalar registers; the default instances never touch it
    constexpr bool RANK2 = (OPT & 1) != 0;  // amplitude = aA_i*alpha_j + aB_i*beta_j instead of an LDS lookup
    constexpr bool SATMASK = (OPT & 2) != 0; // cutoff by v_fma clamp instead of v_cmp + v_cndmask
    constexpr bool ESPLIT = (OPT & 4) == 0;  // per-i energy accumulators (else one pair per lane)
    constexpr bool BEADCULL = (OPT & 8) != 0; // per-bead second-level cull + LDS ring compaction
    constexpr bool NOSWEEP = (OPT & 16) != 0; // diagnosis only: skip the pair arithmetic (times culls + fold)
    // LEAN: the default configuration gets a pair loop with the mask and the power factored out (see below)
    constexpr bool LEAN = SATMASK && SAMECUT && !ESPLIT && !RANK2 && !FORMS && !NOSWEEP;
    constexpr bool NOENERGY = (OPT & 512) != 0; // MD steps between reports: forces only (LEAN instances)
    constexpr bool XCDMAP = (OPT & 128) != 0; // blocks of one XCD (blockIdx % 8) take contiguous cluster ranges
    constexpr bool P1ONLY = (OPT & 64) != 0;  // diagnosis only: cluster cull + fold, no j stream at all
    constexpr bool SAMEJ = (OPT & 32) != 0;   // diagnosis only: every j load hits the same 64 clusters (L1-resident)
    __shared__ int s_list[4][kListCap + 72];
    __shared__ float4 s_ring[4][BEADCULL ? 128 : 1];
    __shared__ __attribute__((aligned(32))) float s_tab[5 * 8];
    __shared__ float s_tabc[FORMS ? 40 : 1], s_tabs[FORMS ? 40 : 1]; // COB / SCB tables of the generic forms
    __shared__ float s_arow[4][kCl * 8]; // per wave: amplitude row of each of its 8 i beads (address = s*8 + label_j)
    __shared__ double s_e[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 3, slot = lane & 7;
    if (threadIdx.x < 40) {
        const bool in5 = (threadIdx.x & 7) < 5;
        const int t5 = (threadIdx.x >> 3) * 5 + (threadIdx.x & 7);
        s_tab[threadIdx.x] = in5 ? P.table[t5] : 0.f;
        if (FORMS) {
            s_tabc[threadIdx.x] = in5 ? Qd->tab_cob[t5] : 0.f;
            s_tabs[threadIdx.x] = in5 ? Qd->tab_scb[t5] : 0.f;
        }
    }
    const GridParams G = *grid;
    const int ncl = st->n_clusters;
    __syncthreads();
    int *list = s_list[wave];
    float4 *ring = s_ring[wave];
    const float *arow = s_arow[wave];
    // spos4 holds the beads' positions as they are (nm, bit for bit): the pair terms see the state without any rounding of
    // their own.  (Round 1 worked in lengths scaled by sqrt(log2(e) / (2 r_comp^2)), which saves the multiply in front
    // of v_exp -- 1 of 27 operations per pair, ~2 % of the kernel -- and costs one rounding of every coordinate at its
    // full magnitude: 3x the force error in relative L2, 25x in the maximum)
    const float rc2 = P.rc2max;
    const float4 far4 = make_float4(-1e18f, -1e18f, -1e18f, __int_as_float(-8 + 2)); // label 0: no amplitude
    const int far_cl = P.n_all; // cluster slot n_all of spos4 holds 8 copies of far4 (written once by mmx_create)
    const float s3 = P.ev_sigma * P.ev_sigma * P.ev_sigma;
    const float ev_c = P.ev_eps * s3 * s3; // eps*sigma^6 (PMODE 6)
    const float tiny = 1e-20f;              // keeps r = 0 finite (self pair, coincident beads)
    // LEAN: forces are accumulated divided by the power p (EV on) and multiplied back at the fold
    const float escale = (LEAN && EV && PMODE == 6) ? ev_c : 1.f; // unit of the EV energies summed in the loop
    const float pscale = (LEAN && EV) ? P.ev_power * escale : 1.f; // unit of the accumulated forces
    const float g_k = P.g_inv_rc2 / pscale;
    // step(rc^2 - r^2) = sat(1e30*(rc^2 - r^2)): exact for every representable r^2 (1 ulp of 0.36 * 1e30 >> 1)
    const float nbig = -1e30f;
    const float cut_all = 1e30f * fminf(rc2, 1e6f), cut_ev = 1e30f * fminf(P.ev_rc2, 1e6f),
                cut_g = 1e30f * fminf(P.g_rc2, 1e6f);
    double acc_ev = 0.0, acc_g = 0.0;

    // Optional XCD-aware work mapping (nb_variant bit 2048; OFF by default): workgroups go round-robin to the 8
    // XCDs (blockIdx % 8), each with its own 4 MB L2.  Clusters are cell-sorted, so a contiguous range of clusters
    // is a spatial slab: XCD x takes slabs x and x+8 of 16 and its L2 then holds two slabs plus halo instead of
    // every position.  Measured at 200k beads: L2 fetches from the fabric / 2.5, kernel time + 1 % (lattice) to
    // + 5 % (relaxed sphere): slabs near the z ends carry less work, so some XCDs drain early, and the kernel is
    // not bandwidth-bound in the first place.
    const int nvb = (ncl + 3) >> 2, per16 = (nvb + 15) >> 4;
    const int jend = XCDMAP ? 2 * per16 : nvb, jstep = XCDMAP ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    for (int j = XCDMAP ? (int)(blockIdx.x >> 3) : (int)blockIdx.x; j < jend; j += jstep) {
        const int xcd = blockIdx.x & 7;
        const int vb = !XCDMAP ? j : (j < per16 ? xcd * per16 + j : (8 + xcd) * per16 + (j - per16));
        const int icl = vb * 4 + wave;
        if (icl >= ncl) continue;
        const float4 lo_i = cl_lo[2 * icl], hi_i = cl_lo[2 * icl + 1];
        // wave-uniform values are moved to scalar registers explicitly (v_readfirstlane): hipcc cannot prove
        // uniformity of loaded / ballot-derived values and would otherwise run the loop control on the VALU
        const int c = __builtin_amdgcn_readfirstlane(__float_as_int(lo_i.w));
        if (__builtin_amdgcn_readfirstlane(__float_as_int(hi_i.w) >> 8) == 0) continue; // no owned bead in this cluster (multi-GPU ghosts)
        // i-cluster -> scalar registers
        float4 pv = spos4[(size_t)icl * kCl + slot];
        // padding slots sit at +1e18 in spos4 (they are also j entries of this very cluster) and ghost beads
        // of a mixed cluster get no force here: as i beads move both far away so that they meet no pair
        if (!P.own().owns(__float_as_int(pv.w) >> 3)) {
            pv.x = pv.y = pv.z = 3e18f;
            pv.w = __int_as_float(-8 + (__float_as_int(pv.w) & 7));
        }
        float xi[kCl], yi[kCl], zi[kCl];
        int wi[kCl];
#pragma unroll
        for (int s = 0; s < kCl; ++s) {
            xi[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.x), s));
            yi[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.y), s));
            zi[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv.z), s));
            wi[s] = __builtin_amdgcn_readlane(__float_as_int(pv.w), s);
        }
        if (GAUSS && !FORMS) { // lane l: row (l >> 3) = i bead, column (l & 7) = label of the j bead
            const int wrow = __shfl(__float_as_int(pv.w), lane >> 3, 64) & 7;
            wave_lds_sync(); // the previous i-cluster's sweeps have finished reading the rows
            s_arow[wave][lane] = s_tab[wrow * 8 + (lane & 7)];
            wave_lds_sync();
        }
        float fx[kCl], fy[kCl], fz[kCl], ee[kCl], eg[kCl], aA[kCl], aB[kCl];
#pragma unroll
        for (int s = 0; s < kCl; ++s) {
            fx[s] = fy[s] = fz[s] = ee[s] = eg[s] = 0.f;
            const int li = wi[s] & 7; // label + 2: A compartments are 3,4; B compartments are 0,1
            aA[s] = (li == 3 || li == 4) ? P.table[3 * 5 + 3] : 0.f;
            aB[s] = (li == 0 || li == 1) ? P.table[0 * 5 + 0] : 0.f;
        }

        const int cx = c % G.nx, cy = (c / G.nx) % G.ny, cz = c / (G.nx * G.ny);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.nx - 1);
        const int z0 = max(cz - 1, 0), z1 = min(cz + 1, G.nz - 1), y0 = max(cy - 1, 0), y1 = min(cy + 1, G.ny - 1);
        const int nrows = (z1 - z0 + 1) * (y1 - y0 + 1);
        int nlist = 0;
        int row_i = 0, base = 0, c1 = 0;
        int rcount = 0, rhead = 0; // LDS ring of compacted j beads (BEADCULL)
        bool more = true;
        while (more) {
            // ---- cull: fill the list with accepted j-clusters (lanes = candidate clusters)
            while (nlist <= kListCap - 64) {
                if (base >= c1) {
                    if (row_i >= nrows) {
                        more = false;
                        break;
                    }
                    const int zz = z0 + row_i / (y1 - y0 + 1), yy = y0 + row_i % (y1 - y0 + 1);
                    const int row = (zz * G.ny + yy) * G.nx;
                    base = __builtin_amdgcn_readfirstlane(cstart[row + x0]);
                    c1 = __builtin_amdgcn_readfirstlane(cstart[row + x1 + 1]);
                    ++row_i;
                    continue;
                }
                const int jc = base + lane;
                bool ok = false;
                if (jc < c1) {
                    float4 lo_j = cl_lo[2 * jc], hi_j = cl_lo[2 * jc + 1]; // one 32-byte box record
                    const float dx = fmaxf(fmaxf(lo_j.x - hi_i.x, lo_i.x - hi_j.x), 0.f);
                    const float dy = fmaxf(fmaxf(lo_j.y - hi_i.y, lo_i.y - hi_j.y), 0.f);
                    const float dz = fmaxf(fmaxf(lo_j.z - hi_i.z, lo_i.z - hi_j.z), 0.f);
                    ok = fmaf(dx, dx, fmaf(dy, dy, dz * dz)) < rc2;
                }
                const unsigned long long mask = __ballot(ok);
                if (ok) list[nlist + prefix_count(mask)] = jc;
                nlist += __builtin_amdgcn_readfirstlane(__popcll(mask));
                base += 64;
            }
            if (P1ONLY) {
                fx[0] += (float)nlist;
                nlist = 0;
                wave_lds_sync();
                continue;
            }
            if (nlist == 0 && (!BEADCULL || rcount == 0)) break;
            // pad to a multiple of 8 clusters with "no cluster" (also gives the ring its flush step)
            if (lane < 8) list[nlist + lane] = far_cl; // a resident all-padding cluster: loads need no predicate
            wave_lds_sync();
            const int nsteps = max((nlist + 7) >> 3, 1);
            // ---- sweep: 8 j-clusters (64 j beads) per step against the 8 scalar i beads
            int jn = list[sub];
            if (SAMEJ) jn &= 63;
            float4 qn = spos4[(unsigned)jn * kCl + slot]; // 32-bit offsets: n_all <= 2^24 beads (checked on the host)
            jn = nsteps > 1 ? list[8 + sub] : far_cl;     // cluster ids are read TWO steps ahead, positions one
            for (int t = 0; t < nsteps; ++t) {
                float4 q = qn;
                if (t + 1 < nsteps) { // prefetch the next 8 clusters (their ids are already in a register)
                    if (SAMEJ) jn &= 63;
                    qn = spos4[(unsigned)jn * kCl + slot];
                    jn = t + 2 < nsteps ? list[(t + 2) * 8 + sub] : far_cl;
                }
                if (BEADCULL) {
                    // second-level cull per j BEAD against the i-cluster box; survivors are compacted
                    // through a 128-entry LDS ring so that every swept lane holds a useful neighbour
                    const float bx = fmaxf(fmaxf(lo_i.x - q.x, q.x - hi_i.x), 0.f);
                    const float by = fmaxf(fmaxf(lo_i.y - q.y, q.y - hi_i.y), 0.f);
                    const float bz = fmaxf(fmaxf(lo_i.z - q.z, q.z - hi_i.z), 0.f);
                    const bool okb = fmaf(bx, bx, fmaf(by, by, bz * bz)) < rc2;
                    const unsigned long long mb = __ballot(okb);
                    if (okb) ring[(rhead + rcount + prefix_count(mb)) & 127] = q;
                    rcount += __builtin_amdgcn_readfirstlane(__popcll(mb));
                }
                const bool last = !more && (t + 1 == nsteps);
                // BEADCULL: drain the ring 64 beads at a time (everything at the very last step)
                for (int pass = 0; BEADCULL ? (rcount >= 64 || (last && rcount > 0)) : (pass == 0); ++pass) {
                if (BEADCULL) {
                    wave_lds_sync();
                    q = ring[(rhead + lane) & 127]; // one 16-byte LDS read; stale slots are replaced below
                    if (rcount < 64) {              // wave-uniform: only the very last, partial batch
                        if (lane >= rcount) q = far4;
                    }
                    const int took = min(rcount, 64);
                    rhead = (rhead + took) & 127;
                    rcount -= took;
                }
                if (NOSWEEP) {
                    fx[0] += q.x;
                    continue;
                }
                const int lj = __float_as_int(q.w) & 7;
                // rank-2 amplitude (compartment blocks only): A(s_i,s_j) = aA_i*alpha_j + aB_i*beta_j
                const float alpha_j = (lj == 3 || lj == 4) ? 1.f : 0.f, beta_j = (lj == 0 || lj == 1) ? 1.f : 0.f;
                if (FORMS) { // non-default functional forms: run-time switches, self pair masked (not subtracted)
                    const int wj = __float_as_int(q.w);
#pragma unroll
                    for (int s = 0; s < kCl; ++s) {
                        const float dx = xi[s] - q.x, dy = yi[s] - q.y, dz = zi[s] - q.z;
                        const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                        float e1, e2, fs;
                        pair_generic<EV, GAUSS>(P, Q, r2, wi[s], wj, s_tabc, s_tabs, 8, fma_sat(r2, nbig, cut_ev),
                                                fma_sat(r2, nbig, cut_g), e1, e2, fs);
                        ee[0] += e1;
                        eg[0] += e2;
                        fx[s] = fmaf(fs, dx, fx[s]);
                        fy[s] = fmaf(fs, dy, fy[s]);
                        fz[s] = fmaf(fs, dz, fz[s]);
                    }
                    continue;
                }
#pragma unroll
                for (int s = 0; s < kCl; ++s) {
                    const float dx = xi[s] - q.x, dy = yi[s] - q.y, dz = zi[s] - q.z;
                    const float r2 = LEAN ? 0.f : fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                    if (LEAN) {
                        // lean form of the default path (clamp mask, one cutoff, merged energies): the mask is applied
                        // ONCE to the force scale and rides on the energy FMAs, the power p is factored out of the
                        // pair loop (forces are accumulated as F/p and scaled at the fold): 3 VALU ops fewer per pair
                        const float r2t = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, tiny))); // r^2 + 1e-20 in the FMA chain
                        const float in = fma_sat(r2t, nbig, cut_all);
                        const float rinv = __builtin_amdgcn_rsqf(r2t);
                        float fs = 0.f;
                        if (EV) {
                            const float u = __builtin_amdgcn_rcpf(fmaf(r2t, rinv, P.ev_rs));
                            float E;
                            if (PMODE == 6) { // in units of eps*sigma^6: the constant joins the scale applied at the fold
                                const float u2 = u * u;
                                E = (u2 * u2) * u2;
                            } else {
                                E = P.ev_eps * ev_pow<PMODE>(P.ev_sigma * u, P.ev_power);
                            }
                            if (!NOENERGY) ee[0] = fmaf(E, in, ee[0]);
                            fs = E * (u * rinv);
                        }
                        if (GAUSS) {
                                                        const float gg = arow[s * 8 + lj] * __builtin_amdgcn_exp2f(r2t * P.g_c2);
                            if (!NOENERGY) eg[0] = fmaf(-gg, in, eg[0]);
                            fs = fmaf(-gg, g_k, fs);
                        }
                        fs *= in;
                        fx[s] = fmaf(fs, dx, fx[s]);
                        fy[s] = fmaf(fs, dy, fy[s]);
                        fz[s] = fmaf(fs, dz, fz[s]);
                        continue;
                    }
                    // exact step function of the cutoff without v_cmp/v_cndmask: sat(BIG*(rc^2 - r^2))
                    const float in = SATMASK ? fma_sat(r2, nbig, cut_all) : (r2 < rc2 ? 1.f : 0.f);
                    const float r2s = r2 + 1e-20f;
                    const float rinv = __builtin_amdgcn_rsqf(r2s);
                    float fs = 0.f;
                    if (EV) {
                        const float u = __builtin_amdgcn_rcpf(fmaf(r2s, rinv, P.ev_rs));
                        float E;
                        if (PMODE == 6) {
                            const float u2 = u * u;
                            E = (u2 * u2) * (u2 * ev_c);
                        } else {
                            E = P.ev_eps * ev_pow<PMODE>(P.ev_sigma * u, P.ev_power);
                        }
                        if (SATMASK) E *= SAMECUT ? in : fma_sat(r2, nbig, cut_ev);
                        else E = (r2 < (SAMECUT ? rc2 : P.ev_rc2)) ? E : 0.f;
                        ee[ESPLIT ? s : 0] += E;
                        fs = (P.ev_power * E) * (u * rinv);
                    }
                    if (GAUSS) {
                        float A;
                        if (RANK2) A = fmaf(aA[s], alpha_j, aB[s] * beta_j);
                        else A = arow[s * 8 + lj]; // ds_read_b32 with an immediate offset: no address arithmetic per pair
                        float gg = A * __builtin_amdgcn_exp2f(r2 * P.g_c2);
                        if (SATMASK) gg *= SAMECUT ? in : fma_sat(r2, nbig, cut_g);
                        else gg = (r2 < (SAMECUT ? rc2 : P.g_rc2)) ? gg : 0.f;
                        eg[ESPLIT ? s : 0] -= gg;
                        fs = fmaf(-gg, P.g_inv_rc2, fs);
                    }
                    fx[s] = fmaf(fs, dx, fx[s]);
                    fy[s] = fmaf(fs, dy, fy[s]);
                    fz[s] = fmaf(fs, dz, fz[s]);
                }
                } // drain passes
            }
            nlist = 0;
            wave_lds_sync();
        }
        // ---- fold over the wave; lane s (< 8) ends up owning bead s of the i-cluster (transposed butterfly, fold8:
        // ~27 operations per component instead of 8 full wave reductions)
        const float ofx = fold8(fx, lane) * pscale, ofy = fold8(fy, lane) * pscale, ofz = fold8(fz, lane) * pscale;
        int ow = -8;
#pragma unroll
        for (int s = 0; s < kCl; ++s) ow = lane == s ? wi[s] : ow;
        float tev = 0.f, teg = 0.f;
#pragma unroll
        for (int s = 0; s < (ESPLIT ? kCl : 1); ++s) {
            tev += ee[s];
            teg += eg[s];
        }
        const int bead = ow >> 3; // -1 for padding slots and for lanes >= 8
        const bool own = bead >= 0;
        // The self pair (r = 0, zero force) was swept with everything else: remove its energy (6400 kJ/mol per bead
        // with the default parameters, against a few kJ/mol of genuine pair energy).  The value subtracted is formed
        // by the very operations the loop used -- same constants, same order, before the common factor is
        // applied -- so that nothing systematic is left of it (a value that is merely equal in exact arithmetic
        // leaves ~1e-7 * 6400 kJ/mol per bead behind, all of one sign).
        if (own && EV && !FORMS && !NOENERGY) {
            const float us = __builtin_amdgcn_rcpf(fmaf(tiny, __builtin_amdgcn_rsqf(tiny), P.ev_rs));
            if (PMODE == 6) {
                const float u2 = us * us;
                tev -= LEAN ? (u2 * u2) * u2 : (u2 * u2) * (u2 * ev_c);
            } else {
                tev -= P.ev_eps * ev_pow<PMODE>(P.ev_sigma * us, P.ev_power);
            }
        }
        tev *= escale; // LEAN, p = 6: the loop summed (sigma-free) u^6
        if (own) {
            if (GAUSS && !FORMS && !NOENERGY) teg += s_tab[(ow & 7) * 8 + (ow & 7)];
            float *gb = g + 3 * P.own().local(bead); // the bonded terms wrote the gradient first (looked up here, not kept
                                                     // live across the pair loop: the kernel sits at exactly 80 VGPRs)
            const float g0 = gb[0], g1 = gb[1], g2 = gb[2];
            gb[0] = g0 - ofx;
            gb[1] = g1 - ofy;
            gb[2] = g2 - ofz;
        }
        // fp64 per lane, summed over the wave once at the end of the kernel: up to eight lanes still carry a self-pair
        // energy that the owners' lanes cancel
        if (!NOENERGY) acc_ev += 0.5 * (double)tev;
        acc_g += 0.5 * (double)teg;
    }
    acc_ev = wave_sum(acc_ev);
    acc_g = wave_sum(acc_g);
    if (lane == 0) {
        s_e[0][wave] = acc_ev;
        s_e[1][wave] = acc_g;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[P_EV * kPartStride + blockIdx.x] = (s_e[0][0] + s_e[0][1]) + (s_e[0][2] + s_e[0][3]);
        part[P_GAUSS * kPartStride + blockIdx.x] = (s_e[1][0] + s_e[1][1]) + (s_e[1][2] + s_e[1][3]);
    }
}

// ------------------------------------------------------------------------------------------------
// K2x: exact all-pairs (NoCutoff, what the reference does).  Block = 256 home beads, blockIdx.y = j
// slice; j tiles of 256 beads are staged in LDS and broadcast-read.  Partial forces per slice go to
// fpart[slice][bead] and are folded in a fixed order by k_nb_allpairs_fold.
// ------------------------------------------------------------------------------------------------
template <int PMODE, bool EV, bool GAUSS, bool FORMS = false>
__global__ __launch_bounds__(256) void k_nb_allpairs(const FFParams P, const float4 *__restrict__ pos4, int tiles_per_slice,
                                                     float4 *__restrict__ fpart, float2 *__restrict__ epart,
                                                     const MinState *__restrict__ st,
                                                     const FormParams *__restrict__ Qd = nullptr) {
    if (st->phase >= PH_DONE) return;
    __shared__ float4 s_tile[256];
    __shared__ float s_tab[32], s_tabc[FORMS ? 32 : 1], s_tabs[FORMS ? 32 : 1];
    FormParams Q;
    if (FORMS) Q = *Qd;
    if (threadIdx.x < 25) {
        s_tab[threadIdx.x] = P.table[threadIdx.x];
        if (FORMS) {
            s_tabc[threadIdx.x] = Qd->tab_cob[threadIdx.x];
            s_tabs[threadIdx.x] = Qd->tab_scb[threadIdx.x];
        }
    }
    const int n = P.n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < n;
    float4 pi = make_float4(1e18f, 1e18f, 1e18f, __int_as_float(-1));
    if (act) pi = pos4[i];
    const int wi = __float_as_int(pi.w);
    __syncthreads();
    const float *tab5 = s_tab + 5 * (act ? (wi & 7) : 0);
    PairAcc a = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int t0 = blockIdx.y * tiles_per_slice;
    for (int tt = 0; tt < tiles_per_slice; ++tt) {
        const int jb = (t0 + tt) * 256;
        if (jb >= n) break;
        const int j = jb + threadIdx.x;
        __syncthreads();
        // padded entries sit at -1e18: r2 ~ 1e36 stays finite; the inactive-i mask below removes them
        s_tile[threadIdx.x] = j < n ? pos4[j] : make_float4(-1e18f, -1e18f, -1e18f, __int_as_float(-2));
        __syncthreads();
        const int cnt = min(256, n - jb);
#pragma unroll 4
        for (int t = 0; t < cnt; ++t) {
            const float4 q = s_tile[t];
            pair_accum<PMODE, EV, GAUSS, FORMS>(P, pi, wi, q, tab5, a, &Q, s_tabc, s_tabs);
        }
    }
    if (act) {
        const size_t o = (size_t)blockIdx.y * (size_t)n + (size_t)i;
        fpart[o] = make_float4(a.fx, a.fy, a.fz, 0.f);
        epart[o] = make_float2(a.eev, a.eg);
    }
}

// Lean form of K2x for the pure NoCutoff case (no cutoff mask at all: every pair counts, which is exactly what the
// reference evaluates).  Same factoring as the LEAN cluster loop: eps*sigma^6 and the power p leave the pair loop, the
// r = 0 guard sits in the r^2 FMA chain, the self pair is masked in the one diagonal tile only:
// 17 VALU operations per pair with EV only, 21 with the Gaussians, against ~30 of the masked general form.
template <int PMODE, bool EV, bool GAUSS>
__global__ __launch_bounds__(256) void k_nb_allpairs_lean(const FFParams P, const float4 *__restrict__ pos4,
                                                          int tiles_per_slice, float4 *__restrict__ fpart,
                                                          float2 *__restrict__ epart, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    __shared__ float4 s_tile[256];
    __shared__ float s_tab[32];
    if (threadIdx.x < 25) s_tab[threadIdx.x] = P.table[threadIdx.x];
    const int n = P.n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool act = i < n;
    float4 pi = make_float4(1e18f, 1e18f, 1e18f, __int_as_float(2)); // inactive lanes: far away, label 0
    if (act) pi = pos4[i];
    const int li = __float_as_int(pi.w) & 7;
    __syncthreads();
    const float *tab5 = s_tab + 5 * li;
    const float s3 = P.ev_sigma * P.ev_sigma * P.ev_sigma;
    const float ev_c = P.ev_eps * s3 * s3, tiny = 1e-20f;
    const float escale = (EV && PMODE == 6) ? ev_c : 1.f;
    const float pscale = EV ? P.ev_power * escale : 1.f;
    const float g_k = P.g_inv_rc2 / pscale;
    float fx = 0.f, fy = 0.f, fz = 0.f, eev = 0.f, eg = 0.f;
    const int t0 = blockIdx.y * tiles_per_slice;
    for (int tt = 0; tt < tiles_per_slice; ++tt) {
        const int jb = (t0 + tt) * 256;
        if (jb >= n) break;
        const int j = jb + threadIdx.x;
        __syncthreads();
        s_tile[threadIdx.x] = pos4[min(j, n - 1)]; // unpredicated load; the loop below stops at the last real bead
        __syncthreads();
        const int cnt = min(256, n - jb);
        // the one tile that holds this block's own beads masks the self pair (adding and removing eps*(sigma/r_s)^p
        // = 6400 kJ/mol per bead would cost the small pair energies their last digits); all other tiles run unmasked
        const bool diag = (t0 + tt) == (int)blockIdx.x; // block-uniform
        auto sweep = [&](auto DIAG) {
#pragma unroll 4
            for (int t = 0; t < cnt; ++t) {
                const float4 q = s_tile[t];
                const float dx = pi.x - q.x, dy = pi.y - q.y, dz = pi.z - q.z;
                const float r2t = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, tiny)));
                const float rinv = __builtin_amdgcn_rsqf(r2t);
                const float m = (decltype(DIAG)::value && t == (int)threadIdx.x) ? 0.f : 1.f;
                float fs = 0.f;
                if (EV) {
                    const float u = __builtin_amdgcn_rcpf(fmaf(r2t, rinv, P.ev_rs));
                    float E;
                    if (PMODE == 6) {
                        const float u2 = u * u;
                        E = (u2 * u2) * u2;
                    } else {
                        E = P.ev_eps * ev_pow<PMODE>(P.ev_sigma * u, P.ev_power);
                    }
                    if (decltype(DIAG)::value) E *= m;
                    eev += E;
                    fs = E * (u * rinv);
                }
                if (GAUSS) {
                    float gg = tab5[__float_as_int(q.w) & 7] * __builtin_amdgcn_exp2f(r2t * P.g_c2);
                    if (decltype(DIAG)::value) gg *= m;
                    eg -= gg;
                    fs = fmaf(-gg, g_k, fs);
                }
                fx = fmaf(fs, dx, fx);
                fy = fmaf(fs, dy, fy);
                fz = fmaf(fs, dz, fz);
            }
        };
        if (diag) sweep(std::true_type{});
        else sweep(std::false_type{});
    }
    if (act) {
        eev *= escale;
        const size_t o = (size_t)blockIdx.y * (size_t)n + (size_t)i;
        fpart[o] = make_float4(fx * pscale, fy * pscale, fz * pscale, 0.f);
        epart[o] = make_float2(eev, eg);
    }
}

__global__ __launch_bounds__(256) void k_nb_allpairs_fold(int n, int nslices, const float4 *__restrict__ fpart,
                                                          const float2 *__restrict__ epart, float *__restrict__ g,
                                                          double *__restrict__ part,
                                                          const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    __shared__ double s_w[4];
    double ev = 0.0, eg = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float fx = 0.f, fy = 0.f, fz = 0.f, e1 = 0.f, e2 = 0.f;
        for (int s = 0; s < nslices; ++s) {
            const float4 f = fpart[(size_t)s * n + i];
            const float2 e = epart[(size_t)s * n + i];
            fx += f.x;
            fy += f.y;
            fz += f.z;
            e1 += e.x;
            e2 += e.y;
        }
        g[3 * i] -= fx; // the bonded terms wrote the gradient first
        g[3 * i + 1] -= fy;
        g[3 * i + 2] -= fz;
        ev += 0.5 * (double)e1;
        eg += 0.5 * (double)e2;
    }
    const double sev = block_sum<256>(ev, s_w);
    const double seg = block_sum<256>(eg, s_w);
    if (threadIdx.x == 0) {
        part[P_EV * kPartStride + blockIdx.x] = sev;
        part[P_GAUSS * kPartStride + blockIdx.x] = seg;
    }
}

} // namespace mmx
