// mmx_bonded.hpp -- K3 backbone bonds+angles, K4 loop restraints, K5 confinement/lamina/central.
// All three are per-bead gather kernels: the thread of bead i computes every contribution to
// the gradient of bead i itself (neighbours come from L1/L2), so there are no atomics and the result
// is bitwise reproducible.  The separate kernels each do g[i] += dE/dx_i; the fused pass (the default) WRITES g[i]:
// it is the first writer of the gradient in an evaluation and the pair kernels add to it (see k_scan_bonded).
#pragma once
#include "mmx_cells.hpp"
#include "mmx_common.hpp"

namespace mmx {

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(const float4 p) { return {p.x, p.y, p.z}; }
__device__ __forceinline__ F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(F3 a, F3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }

// HarmonicBondForce E = 1/2 k (r-r0)^2; returns E, adds dE/dx_i to (gx,gy,gz).  model.py:625-636, 651-659.
__device__ __forceinline__ float bond_grad(F3 pi, F3 pj, float r0, float k, float &gx, float &gy, float &gz) {
    const F3 d = sub(pi, pj);
    const float r2 = dot(d, d);
    const float rinv = __builtin_amdgcn_rsqf(fmaxf(r2, 1e-30f));
    const float r = r2 * rinv;
    const float dr = r - r0;
    const float s = k * dr * rinv; // dE/dr / r ; d == 0 => zero gradient
    gx = fmaf(s, d.x, gx);
    gy = fmaf(s, d.y, gy);
    gz = fmaf(s, d.z, gz);
    return 0.5f * k * dr * dr;
}

// Loop restraint with the alternative forms of add_loops (model.py:662-701); form 0 = the harmonic default.
//   1 fene_soft        E = k u^2/(1 + u^2/r0^2)          dE/du = 2 k u/(1 + u^2/r0^2)^2      (u = r - r0)
//   2 gaussian_tether  E = k (1 - exp(-u^2/sigma^2))     dE/du = 2 k u/sigma^2 exp(-u^2/sigma^2), sigma = r0/2
__device__ __forceinline__ float loop_grad(int form, F3 pi, F3 pj, float r0, float k, float &gx, float &gy, float &gz) {
    if (form == 0) return bond_grad(pi, pj, r0, k, gx, gy, gz);
    const F3 d = sub(pi, pj);
    const float r2 = dot(d, d);
    const float rinv = __builtin_amdgcn_rsqf(fmaxf(r2, 1e-30f));
    const float u = r2 * rinv - r0, u2 = u * u;
    float E, dEdu;
    if (form == 1) {
        const float den = __builtin_amdgcn_rcpf(fmaf(u2, __builtin_amdgcn_rcpf(r0 * r0), 1.f));
        E = k * u2 * den;
        dEdu = 2.f * k * u * den * den;
    } else {
        const float is2 = 4.f * __builtin_amdgcn_rcpf(r0 * r0);
        const float ex = __builtin_amdgcn_exp2f(-1.44269504f * u2 * is2);
        E = k * (1.f - ex);
        dEdu = 2.f * k * u * is2 * ex;
    }
    const float s = dEdu * rinv;
    gx = fmaf(s, d.x, gx);
    gy = fmaf(s, d.y, gy);
    gz = fmaf(s, d.z, gz);
    return E;
}

// HarmonicAngleForce E = 1/2 k (theta-theta0)^2, theta at the middle bead.  OpenMM force form:
// a = x_i-x_j, b = x_k-x_j, c = a x b (|c| clamped >= 1e-6),
//   F_i = -dEdth*(a x c)/(|a|^2|c|), F_k = -dEdth*(c x b)/(|b|^2|c|), F_j = -(F_i+F_k).   model.py:708-720.
// Returns E and the FORCES on the two end beads.
__device__ __forceinline__ float angle_forces(F3 pi, F3 pj, F3 pk, float th0, float kk, F3 &fi, F3 &fk) {
    const F3 a = sub(pi, pj), b = sub(pk, pj);
    const F3 c = cross(a, b);
    const float cn = sqrtf(dot(c, c));
    const float dt = dot(a, b);
    const float theta = atan2f(cn, dt);
    const float rp = fmaxf(cn, 1e-6f);
    const float aa = fmaxf(dot(a, a), 1e-30f), bb = fmaxf(dot(b, b), 1e-30f);
    const float dth = theta - th0;
    const float dE = kk * dth;
    const float ta = -dE / (aa * rp), tc = -dE / (bb * rp);
    const F3 ac = cross(a, c), cb = cross(c, b);
    fi = {ta * ac.x, ta * ac.y, ta * ac.z};
    fk = {tc * cb.x, tc * cb.y, tc * cb.z};
    return 0.5f * kk * dth * dth;
}

// K3.  flags[i] bit0: bond (i,i+1) present, bit1: angle (i,i+1,i+2) present.
// Every bond and every angle is evaluated ONCE, by the thread of the bead that starts it (round 2: the thread of bead i
// re-evaluated all it takes part in -- 2 bonds, 3 angles with their atan2 and cross products -- and the kernel ran at a
// quarter of the bandwidth the other per-bead kernels reach).  The other beads of a term get their share from that
// thread through wave shuffles: a wave is a TILE of 62 consecutive beads preceded by the two beads before them (lanes
// 0 and 1: they evaluate the terms they start, for the hand-over only, and write nothing).  No LDS, no barrier, no
// atomics; fixed order of the five fp32 additions per bead, so results are bitwise reproducible.
constexpr int kBBTile = 62; // beads written per wave

// For the bead of this lane (global index i; `valid`: it exists): dE/dx_i of the backbone terms into (gx,gy,gz);
// energies of the bond / angle the bead STARTS into eb / ea when `book` (output lanes).  Whole waves must call.
__device__ __forceinline__ void backbone_lane(const FFParams &P, const float4 *__restrict__ pos4,
                                              const uint8_t *__restrict__ flags, const int i, const bool valid,
                                              const bool book, double &eb, double &ea, float &gx, float &gy, float &gz) {
    const int n = P.n;
    float bx = 0.f, by = 0.f, bz = 0.f;       // gradient on i of bond (i, i+1); bead i+1 gets the opposite
    F3 fi = {0.f, 0.f, 0.f}, fk = {0.f, 0.f, 0.f}; // forces of angle (i, i+1, i+2) on its two end beads
    if (valid) {
        const int f0 = flags[i];
        const F3 p0 = f3(pos4[i]);
        F3 p1 = p0, p2 = p0;
        if (i + 1 < n) p1 = f3(pos4[i + 1]);
        if (i + 2 < n) p2 = f3(pos4[i + 2]);
        if (P.use_bond && (f0 & 1)) {
            const float e = bond_grad(p0, p1, P.bond_r0, P.bond_k, bx, by, bz);
            if (book) eb += (double)e;
        }
        if (P.use_angle && (f0 & 2)) {
            const float e = angle_forces(p0, p1, p2, P.ang_th0, P.ang_k, fi, fk);
            if (book) ea += (double)e;
        }
    }
    // what the two beads behind a term's first bead receive: from lane - 1 the bond's reaction and the angle's middle
    // share, from lane - 2 the angle's far-end force
    const float b1x = __shfl_up(bx, 1, 64), b1y = __shfl_up(by, 1, 64), b1z = __shfl_up(bz, 1, 64);
    const float m1x = __shfl_up(fi.x + fk.x, 1, 64), m1y = __shfl_up(fi.y + fk.y, 1, 64), m1z = __shfl_up(fi.z + fk.z, 1, 64);
    const float k2x = __shfl_up(fk.x, 2, 64), k2y = __shfl_up(fk.y, 2, 64), k2z = __shfl_up(fk.z, 2, 64);
    // same order as ever: bond (i-1,i), bond (i,i+1), angle ending here, angle centred here, angle starting here
    gx = (((gx - b1x) + bx) - k2x + m1x) - fi.x;
    gy = (((gy - b1y) + by) - k2y + m1y) - fi.y;
    gz = (((gz - b1z) + bz) - k2z + m1z) - fi.z;
}

// Tiles of a launch of `nwaves` waves over n_own owned beads: wave `gw` takes tiles gw, gw + nwaves, ...; lane l of tile t
// stands for owned bead t * kBBTile + l - 2 (lanes 0, 1: the hand-over beads, possibly of another rank or non-existent).
// A tile is one ownership SEGMENT (kSeg == kBBTile): its beads are consecutive global ids wherever the rank's other
// segments lie, and the two beads before it are the global predecessors of its first bead (ghosts when not owned).
static_assert(kBBTile == kSeg, "one backbone tile per ownership segment");
__device__ __forceinline__ int bb_tiles(int n_own) { return (n_own + kBBTile - 1) / kBBTile; }
__device__ __forceinline__ int bb_tile_bead(const FFParams &P, int t, int lane) { // global bead of lane `lane` of tile t
    return (P.seg_own ? P.seg_own[t] * kBBTile : P.own_lo + t * kBBTile) + lane - 2;
}

// Algorithmic traffic: read 12 B position + 1 B flag, write 12 B gradient = 25 B/bead (moved: 16 B pos4 + 1 + 12).
// first: this launch is the first writer of the gradient (stores; no memset before it), else it adds to it.
__global__ __launch_bounds__(256) void k_backbone(const FFParams P, const float4 *__restrict__ pos4,
                                                  const uint8_t *__restrict__ flags, float *__restrict__ g,
                                                  double *__restrict__ part, const MinState *__restrict__ st,
                                                  const int first) {
    if (st->phase >= PH_DONE) return;
    __shared__ double s_w[4];
    double eb = 0.0, ea = 0.0;
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int t = gw; t < bb_tiles(P.n_own); t += nw) {
        const int li = t * kBBTile + lane - 2, i = bb_tile_bead(P, t, lane);
        const bool out = lane >= 2 && li < P.n_own;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        backbone_lane(P, pos4, flags, i, i >= 0 && i < P.n && li < P.n_own, out, eb, ea, gx, gy, gz);
        if (out) {
            if (!first) {
                gx += g[3 * li];
                gy += g[3 * li + 1];
                gz += g[3 * li + 2];
            }
            g[3 * li] = gx;
            g[3 * li + 1] = gy;
            g[3 * li + 2] = gz;
        }
    }
    const double sb = block_sum<256>(eb, s_w);
    const double sa = block_sum<256>(ea, s_w);
    if (threadIdx.x == 0) {
        part[P_BOND * kPartStride + blockIdx.x] = sb;
        part[P_ANGLE * kPartStride + blockIdx.x] = sa;
    }
}

// K4.  Loops as a CSR over the beads that carry at least one loop end: row r = bead row_bead[r],
// entries [row_start[r], row_start[r+1]) = (partner, r0).  Each loop appears in two rows; each row
// books half of the loop's energy.  Algorithmic traffic 64 B/loop.
__global__ __launch_bounds__(256) void k_loops(const FFParams P, int n_rows, const float4 *__restrict__ pos4,
                                               const int *__restrict__ row_bead, const int *__restrict__ row_start,
                                               const int *__restrict__ partner, const float *__restrict__ r0,
                                               float *__restrict__ g, double *__restrict__ part,
                                               const MinState *__restrict__ st, const int loop_form) {
    if (st->phase >= PH_DONE) return;
    __shared__ double s_w[4];
    double e = 0.0;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n_rows; r += gridDim.x * 256) {
        const int b = row_bead[r];
        const F3 pb = f3(pos4[b]);
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int q = row_start[r]; q < row_start[r + 1]; ++q)
            e += 0.5 * (double)loop_grad(loop_form, pb, f3(pos4[partner[q]]), r0[q], P.loop_k, gx, gy, gz);
        const int lb = P.own().local(b); // (rows are built for owned beads only)
        g[3 * lb] += gx;
        g[3 * lb + 1] += gy;
        g[3 * lb + 2] += gz;
    }
    const double s = block_sum<256>(e, s_w);
    if (threadIdx.x == 0) part[P_LOOP * kPartStride + blockIdx.x] = s;
}

// K5.  Container (model.py:454-456), B-lamina "sin" shell (model.py:503-505), central force
// (model.py:584-586) on r = |x - centre|.
// Algorithmic traffic: read 16 B pos4, read-modify-write 12 B gradient.  (The reductions of the line search --
// g.d, g.g, x.x -- are taken by k_history once every term has been added.)
// External terms of one bead (container, lamina in form lam_form, central force in form cf_form, weight w_i) on
// r = |x - centre|: energies into (ec, el, ef), dE/dx added to (gx,gy,gz).
__device__ __forceinline__ void confine_bead(const FFParams &P, const float4 p, const float w_i, const int lam_form,
                                             const int cf_form, double &ec, double &el, double &ef, float &gx,
                                             float &gy, float &gz) {
    const float dx = p.x - P.cx, dy = p.y - P.cy, dz = p.z - P.cz;
    const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const float rinv = __builtin_amdgcn_rsqf(fmaxf(r2, 1e-30f));
    const float r = r2 * rinv;
    float dEdr = 0.f;
    if (P.use_container) {
        const float o = fmaxf(r - P.sc_R2, 0.f), in = fmaxf(P.sc_R1 - r, 0.f);
        ec += (double)(P.sc_C * (o * o + in * in));
        dEdr += 2.f * P.sc_C * (o - in);
    }
    if (P.use_lamina) {
        const int s = (__float_as_int(p.w) & 7) - 2;
        if (s < 0) {
            const float span = P.ibl_R2 - P.ibl_R1;
            if (lam_form == 0) { // sin^8 shell, model.py:503-505
                const float w = 3.14159265358979f / span;
                float sn, cs;
                sincosf(w * (r - P.ibl_R1), &sn, &cs);
                const float s2 = sn * sn, s4 = s2 * s2;
                el += (double)(P.ibl_B * (s4 * s4 - 1.f));
                dEdr += P.ibl_B * 8.f * s4 * s2 * sn * cs * w;
            } else if (lam_form == 1) { // gaussian_shell, sigma = 0.1 (R2-R1), model.py:511-518
                const float is2 = 100.f / (span * span);
                const float a = r - P.ibl_R1, b = r - P.ibl_R2;
                const float e1 = __expf(-0.5f * a * a * is2), e2 = __expf(-0.5f * b * b * is2);
                el += (double)(-P.ibl_B * (e1 + e2));
                dEdr += P.ibl_B * is2 * (a * e1 + b * e2);
            } else if (lam_form == 2) { // harmonic_shell, r0 = (R1+R2)/2, model.py:521-528
                const float u = r - 0.5f * (P.ibl_R1 + P.ibl_R2);
                el += (double)(P.ibl_B * u * u);
                dEdr += 2.f * P.ibl_B * u;
            } else { // logistic_shell, lambda = 0.05 (R2-R1), model.py:531-539
                const float il = 20.f / span;
                const float a = 1.f / (1.f + __expf((r - P.ibl_R2) * il));
                const float b = 1.f / (1.f + __expf(-(r - P.ibl_R1) * il));
                el += (double)(-P.ibl_B * (a + b));
                dEdr += -P.ibl_B * il * (b * (1.f - b) - a * (1.f - a));
            }
        }
    }
    if (P.use_central) {
        const float gw = P.cf_G * w_i;
        if (cf_form == 0) { // harmonic, model.py:579-586
            const float q = r - P.cf_R1;
            ef += (double)(gw * q * q);
            dEdr += 2.f * gw * q;
        } else if (cf_form == 1) { // gaussian, sigma = R1/2, model.py:591-599
            const float is2 = 4.f / (P.cf_R1 * P.cf_R1);
            const float e1 = __expf(-0.5f * r2 * is2);
            ef += (double)(-gw * e1);
            dEdr += gw * r * is2 * e1;
        } else { // logistic, lambda = 0.2 R1, model.py:604-612
            const float il = 5.f / P.cf_R1;
            const float a = 1.f / (1.f + __expf((r - P.cf_R1) * il));
            ef += (double)(-gw * a);
            dEdr += gw * il * a * (1.f - a);
        }
    }
    const float s = dEdr * rinv; // r == 0 => (dx,dy,dz) == 0 => zero gradient
    gx = fmaf(s, dx, gx);
    gy = fmaf(s, dy, gy);
    gz = fmaf(s, dz, gz);
}

__global__ __launch_bounds__(256) void k_confine(const FFParams P, const float4 *__restrict__ pos4,
                                                 const float *__restrict__ cf_w, float *__restrict__ g,
                                                 double *__restrict__ part, const MinState *__restrict__ st,
                                                 const int lam_form, const int cf_form) {
    if (st->phase >= PH_DONE) return;
    __shared__ double s_w[4];
    double ec = 0.0, el = 0.0, ef = 0.0;
    const bool any = P.use_container | P.use_lamina | P.use_central;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P.n_own; i += gridDim.x * 256) { // i: local index
        if (!any) break;
        const int b = P.own().bead(i);
        const float4 p = pos4[b];
        float gx = g[3 * i], gy = g[3 * i + 1], gz = g[3 * i + 2];
        confine_bead(P, p, P.use_central ? cf_w[b] : 0.f, lam_form, cf_form, ec, el, ef, gx, gy, gz);
        g[3 * i] = gx;
        g[3 * i + 1] = gy;
        g[3 * i + 2] = gz;
    }
    const double sc = block_sum<256>(ec, s_w);
    const double sl = block_sum<256>(el, s_w);
    const double sf = block_sum<256>(ef, s_w);
    if (threadIdx.x == 0) {
        part[P_CONT * kPartStride + blockIdx.x] = sc;
        part[P_LAM * kPartStride + blockIdx.x] = sl;
        part[P_CENT * kPartStride + blockIdx.x] = sf;
    }
}

// K3 + K4 + K5 in ONE pass over the owned beads (default; the separate kernels above stay selectable with the
// option "fused_bonded" = 0 and are what mmx_time_kernel measures): one write of the gradient instead of three
// read-modify-writes, two launches fewer per evaluation.  The bonded terms are the FIRST writers of the gradient in an
// evaluation (they only need pos4, so they run beside the cell build on a second stream); the pair kernels add to
// it afterwards.  Same per-bead arithmetic and the same order of the fp32 additions (0 + backbone + loops, then the
// confinement terms), so the result is bitwise the one of the separate kernels after a memset.
// lstart[li] .. lstart[li+1]: loop entries (partner, r0) of owned bead li in the CSR of mmx_set_loops.
// The work of one VIRTUAL block of 256 threads (vb of nvb): either a real 256-thread block (k_bonded_fused) or one
// quarter of a 1024-thread block of k_scan_bonded.  Partials are per virtual block, so both launches give bitwise
// the same sums.  Must be called by every thread of the real block (barriers); s_w holds one double per wave.
template <int NT>
__device__ __forceinline__ void bonded_fused_block(const FFParams &P, const float4 *__restrict__ pos4,
                                                   const uint8_t *__restrict__ flags, const int *__restrict__ lstart,
                                                   const int *__restrict__ partner, const float *__restrict__ r0,
                                                   const float *__restrict__ cf_w, float *__restrict__ g,
                                                   double *__restrict__ part, const int loop_form, const int lam_form,
                                                   const int cf_form, const int vb, const int nvb, double *s_w) {
    double eb = 0.0, ea = 0.0, elp = 0.0, ec = 0.0, el = 0.0, ef = 0.0;
    const bool bb = flags != nullptr && (P.use_bond | P.use_angle);
    const bool any = P.use_container | P.use_lamina | P.use_central;
    const int tv = threadIdx.x & 255;
    // the waves of the virtual blocks walk tiles of kBBTile beads (backbone_lane); lanes 0 and 1 of a tile only hand over
    const int lane = threadIdx.x & 63, gw = vb * 4 + (tv >> 6), nw = nvb * 4;
    for (int t = vb < nvb ? gw : bb_tiles(P.n_own); t < bb_tiles(P.n_own); t += nw) {
        const int li = t * kBBTile + lane - 2;
        const int i = bb_tile_bead(P, t, lane);
        const bool out = lane >= 2 && li < P.n_own;
        float gx = 0.f, gy = 0.f, gz = 0.f; // first writer of the gradient: the pair kernels add to it afterwards
        if (bb) {
            float tx = 0.f, ty = 0.f, tz = 0.f;
            backbone_lane(P, pos4, flags, i, i >= 0 && i < P.n && li < P.n_own, out, eb, ea, tx, ty, tz);
            gx += tx;
            gy += ty;
            gz += tz;
        }
        if (!out) continue;
        const float4 p = pos4[i];
        if (lstart) {
            const int q0 = lstart[li], q1 = lstart[li + 1];
            if (q1 > q0) {
                const F3 pb = f3(p);
                float tx = 0.f, ty = 0.f, tz = 0.f;
                for (int q = q0; q < q1; ++q)
                    elp += 0.5 * (double)loop_grad(loop_form, pb, f3(pos4[partner[q]]), r0[q], P.loop_k, tx, ty, tz);
                gx += tx;
                gy += ty;
                gz += tz;
            }
        }
        if (any) confine_bead(P, p, P.use_central ? cf_w[i] : 0.f, lam_form, cf_form, ec, el, ef, gx, gy, gz);
        g[3 * li] = gx;
        g[3 * li + 1] = gy;
        g[3 * li + 2] = gz;
    }
    const double v[6] = {eb, ea, elp, ec, el, ef};
    const int slot[6] = {P_BOND, P_ANGLE, P_LOOP, P_CONT, P_LAM, P_CENT};
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double w = wave_sum(v[k]);
        __syncthreads();
        if (lane == 0) s_w[wave] = w;
        __syncthreads();
        if (tv == 0 && vb < nvb) { // the four waves of this virtual block, in order (== block_sum<256>)
            const double *q = s_w + (wave & ~3);
            double r = 0.0;
#pragma unroll
            for (int j = 0; j < 4; ++j) r += q[j];
            part[slot[k] * kPartStride + vb] = r;
        }
    }
}

__global__ __launch_bounds__(256) void k_bonded_fused(const FFParams P, const float4 *__restrict__ pos4,
                                                      const uint8_t *__restrict__ flags, const int *__restrict__ lstart,
                                                      const int *__restrict__ partner, const float *__restrict__ r0,
                                                      const float *__restrict__ cf_w, float *__restrict__ g,
                                                      double *__restrict__ part, const MinState *__restrict__ st,
                                                      const int loop_form, const int lam_form, const int cf_form) {
    if (st->phase >= PH_DONE) return;
    __shared__ double s_w[4];
    bonded_fused_block<256>(P, pos4, flags, lstart, partner, r0, cf_w, g, part, loop_form, lam_form, cf_form,
                            (int)blockIdx.x, (int)gridDim.x, s_w);
}

// Horizontal fusion of the cell build's single-block scan with the bonded pass: block 0 scans the cell populations
// (k_cell_scan's work, 1024 threads), every other block is four virtual bonded blocks.  The scan would otherwise
// leave 255 CUs idle for ~10 us per evaluation and the bonded pass only needs pos4, so the two share one launch
// (option "overlap_bonded"; a second stream was measured instead and lost more in cross-stream events than it hid).
template <int CHUNK>
__global__ __launch_bounds__(1024) void k_scan_bonded(const ScanArgs a, MinState *__restrict__ st, const FFParams P,
                                                      const float4 *__restrict__ pos4,
                                                      const uint8_t *__restrict__ flags, const int *__restrict__ lstart,
                                                      const int *__restrict__ partner, const float *__restrict__ r0,
                                                      const float *__restrict__ cf_w, float *__restrict__ g,
                                                      double *__restrict__ part, const int loop_form,
                                                      const int lam_form, const int cf_form, const int nvb) {
    if (st->phase >= PH_DONE) return;
    if (blockIdx.x == 0) {
        cell_scan_block<CHUNK>(a, st);
        return;
    }
    __shared__ double s_w[16];
    bonded_fused_block<1024>(P, pos4, flags, lstart, partner, r0, cf_w, g, part, loop_form, lam_form, cf_form,
                             ((int)blockIdx.x - 1) * 4 + (int)(threadIdx.x >> 8), nvb, s_w);
}

// K7.  Chromosomal blocks (model.py:416-419): E = dE*(k_C r^4 - r^3 + r^2) for every pair of beads of the
// same chromosome, no cutoff (the potential grows with r).  Beads of a chromosome are contiguous, so this is
// a block-diagonal all-pairs sweep: a block owns 256 consecutive beads and walks the bead range spanned by the
// chromosomes of its first and last bead in LDS tiles; pairs of different chromosomes inside that range are
// masked.  F_i = -dE*(4 k_C r^2 - 3 r + 2)*d, one v_sqrt per pair.  Each thread adds to its own bead: no atomics.
template <int FORM> // 0 polynomial, 1 gaussian, 2 saturating (compile-time: the pair loop carries no branch)
__global__ __launch_bounds__(256) void k_chb(const FFParams P, const float4 *__restrict__ pos4,
                                             const int *__restrict__ chrom_of, const int *__restrict__ chrom_lo,
                                             const int *__restrict__ chrom_hi, float *__restrict__ g,
                                             double *__restrict__ part, const MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    __shared__ float4 s_tile[256];
    __shared__ int s_chr[256];
    __shared__ double s_w[4];
    const int li = blockIdx.x * 256 + threadIdx.x; // local index of an owned bead
    const bool act = li < P.n_own;
    const int i = P.own_lo + min(li, P.n_own - 1);
    const float4 pi = pos4[i];
    const int ci = chrom_of[i];
    // bead range covered by the chromosomes of this block's beads (block-uniform)
    const int ifirst = P.own_lo + blockIdx.x * 256, ilast = P.own_lo + min(blockIdx.x * 256 + 255, P.n_own - 1);
    const int jlo = chrom_lo[chrom_of[ifirst]], jhi = chrom_hi[chrom_of[ilast]];
    float fx = 0.f, fy = 0.f, fz = 0.f, e = 0.f;
    const float k4 = 4.f * P.chb_kc;
    for (int jb = jlo; jb < jhi; jb += 256) {
        const int j = jb + threadIdx.x;
        __syncthreads();
        s_tile[threadIdx.x] = j < jhi ? pos4[j] : pi;
        s_chr[threadIdx.x] = j < jhi ? chrom_of[j] : -1;
        __syncthreads();
        const int cnt = min(256, jhi - jb);
#pragma unroll 4
        for (int t = 0; t < cnt; ++t) {
            const float4 q = s_tile[t];
            const float dx = pi.x - q.x, dy = pi.y - q.y, dz = pi.z - q.z;
            const float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
            const float r = __builtin_amdgcn_sqrtf(r2);
            float m = (s_chr[t] == ci) ? P.chb_de : 0.f; // the self pair has r = 0: contributes nothing
            float fs;
            if (FORM == 0) { // polynomial, model.py:416-419
                e = fmaf(m * r2, fmaf(P.chb_kc, r2, 1.f - r), e);
                fs = -m * (fmaf(k4, r2, 2.f) - 3.f * r);
            } else {
                m = (jb + t == i) ? 0.f : m; // these forms are non-zero at r = 0: drop the self pair explicitly
                if (FORM == 1) { // gaussian: -dE exp(-k_C r^2), model.py:428-431
                    const float ex = m * __expf(-P.chb_kc * r2);
                    e -= ex;
                    fs = -2.f * P.chb_kc * ex;
                } else { // saturating: -dE/(1 + k_C r^2), model.py:440-443
                    const float den = 1.f / fmaf(P.chb_kc, r2, 1.f);
                    e -= m * den;
                    fs = -2.f * P.chb_kc * m * den * den;
                }
            }
            fx = fmaf(fs, dx, fx);
            fy = fmaf(fs, dy, fy);
            fz = fmaf(fs, dz, fz);
        }
    }
    if (act) {
        g[3 * li] -= fx;
        g[3 * li + 1] -= fy;
        g[3 * li + 2] -= fz;
    }
    const double se = block_sum<256>(act ? 0.5 * (double)e : 0.0, s_w);
    if (threadIdx.x == 0) part[P_CHB * kPartStride + blockIdx.x] = se;
}

} // namespace mmx
