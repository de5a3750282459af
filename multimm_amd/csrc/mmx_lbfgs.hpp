// mmx_lbfgs.hpp -- K6: device-resident L-BFGS (OpenMM LocalEnergyMinimizer == liblbfgs, m = 6,
// backtracking strong-Wolfe line search) as a state machine advanced once per evaluation.
//
// The two-loop recursion is done in coefficient space over the basis B = {S_0..S_5, Y_0..Y_5, g}:
// one streaming pass (k_history) stores the new (s,y) pair and produces the three Gram rows that
// changed, a single thread runs the recursion on the 13x13 Gram matrix (k_direction_coef), and one
// more streaming pass (k_direction) forms d = sum_a c_a B_a.  No host round trip, two reductions per
// iteration instead of 2m sequential ones; the same Gram rows are what a multi-GPU run all-reduces.
#pragma once
#include "mmx_common.hpp"

namespace mmx {

struct CtlArgs {
    int nblk[P_NSLOTS]; // block partials per slot written by the force kernels of this evaluation
};

constexpr int kCtlThreads = 1024;

// Deterministic sums of NS partial-sum slots by one 1024-thread block: every thread strides over
// every slot (independent loads, latency overlapped), wave-shuffle reduce, then a fixed-order fold of
// the 16 wave partials.  out[s] is valid on all threads after the call.
template <int NS>
__device__ __forceinline__ void multi_slot_sum(const double *__restrict__ part, int stride, const int *nblk,
                                               double *s_wave /* [NS][16] */, double *s_out /* [NS] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
    for (int s0 = 0; s0 < NS; s0 += 4) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u;
            if (s < NS) {
                const int n = nblk[s];
                const double *p = part + (size_t)s * stride;
                for (int i = threadIdx.x; i < n; i += kCtlThreads) v[u] += p[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u;
            if (s < NS) {
                const double w = wave_sum(v[u]);
                if (lane == 0) s_wave[s * 16 + wave] = w;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < NS) {
        double r = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) r += s_wave[threadIdx.x * 16 + w];
        s_out[threadIdx.x] = r;
    }
    __syncthreads();
}

// After every evaluation: fold the block partials, then (thread 0) advance the line search exactly as
// liblbfgs' line_search_backtracking with LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE does.
__global__ __launch_bounds__(1024) void k_controller(const CtlArgs A, const double *__restrict__ part,
                                                     MinState *__restrict__ st) {
    if (st->phase == PH_DONE) return;
    __shared__ double s_wave[P_NSLOTS * 16];
    __shared__ double s_out[P_NSLOTS];
    __shared__ int s_n[P_NSLOTS];
    if (threadIdx.x < P_NSLOTS) s_n[threadIdx.x] = A.nblk[threadIdx.x];
    __syncthreads();
    multi_slot_sum<P_NSLOTS>(part, kPartStride, s_n, s_wave, s_out);
    if (threadIdx.x != 0) return;
    double sums[P_NSLOTS];
#pragma unroll
    for (int s = 0; s < P_NSLOTS; ++s) sums[s] = s_out[s];

    double f = 0.0;
    for (int t = 0; t < 8; ++t) {
        st->eterms[t] = sums[t];
        f += sums[t];
    }
    st->ftrial = f;
    const double dg = sums[P_GD], gg = sums[P_GG], xx = sums[P_XX];
    const int phase = st->phase;
    if (phase == PH_IDLE) return; // plain mmx_compute()

    const double ftol = 1e-4, wolfe = 0.9, min_step = 1e-20, max_step = 1e20;
    const int max_linesearch = 40;
    const bool finite = (f - f == 0.0) && (gg - gg == 0.0);
    st->evals += 1;
    st->accepted = 0;
    st->store_hist = 0;

    if (phase == PH_INIT) {
        if (!finite) {
            st->status = -6; // MMX_MIN_NAN
            st->phase = PH_DONE;
            return;
        }
        st->fx = f;
        for (int t = 0; t < 8; ++t) st->eterms_acc[t] = sums[t];
        double xn = sqrt(xx);
        if (xn < 1.0) xn = 1.0;
        st->xnorm = xn;
        st->gnorm = sqrt(gg);
        if (st->gnorm / xn <= st->epsilon) {
            st->status = 0;
            st->phase = PH_DONE;
            return;
        }
        st->accepted = 1; // history kernel snapshots (xp,gp); direction becomes -g, step 1/|g|
        return;
    }

    // PH_LINESEARCH
    st->ls_count += 1;
    if (!finite) st->nan_seen += 1;
    double width;
    bool accept = false;
    if (!finite || f > st->finit + st->step * ftol * st->dginit) {
        width = 0.5;
    } else if (dg < wolfe * st->dginit) {
        width = 2.1;
    } else if (dg > -wolfe * st->dginit) {
        width = 0.5;
    } else {
        accept = true;
        width = 1.0;
    }
    if (!accept) {
        int err = 0;
        if (st->step < min_step) err = -3;
        else if (st->step > max_step) err = -4;
        else if (max_linesearch <= st->ls_count) err = -5;
        if (err) {
            st->status = (st->nan_seen > 0 && !finite) ? -6 : err;
            st->phase = PH_DONE; // host restores x = xp (liblbfgs reverts to the previous point)
            return;
        }
        st->step *= width;
        return;
    }
    // accepted: one L-BFGS iteration finished
    st->iters += 1;
    st->fx = f;
    for (int t = 0; t < 8; ++t) st->eterms_acc[t] = sums[t];
    double xn = sqrt(xx);
    if (xn < 1.0) xn = 1.0;
    st->xnorm = xn;
    st->gnorm = sqrt(gg);
    if (st->gnorm / xn <= st->epsilon) {
        st->status = 0;
        st->phase = PH_DONE;
        return;
    }
    if (st->max_iters != 0 && st->max_iters < st->k + 1) {
        st->status = 1;
        st->phase = PH_DONE;
        return;
    }
    st->accepted = 1;
    st->store_hist = 1;
}

// Accepted step: s = x - xp, y = g - gp into slot `end`; xp <- x, gp <- g; and the Gram rows of
// {s_new, y_new, g} against the whole basis as block partials rows[(r*13 + b)*stride + block].
// Streams 4+2m vectors once: (4 + 12) * 4 B reads + 4*4 B writes per float.
__global__ __launch_bounds__(256) void k_history(int n4, const float4 *__restrict__ x, float4 *__restrict__ xp,
                                                 const float4 *__restrict__ g, float4 *__restrict__ gp,
                                                 float4 *__restrict__ S, float4 *__restrict__ Y,
                                                 double *__restrict__ rows, const MinState *__restrict__ st) {
    if (st->phase == PH_DONE || !st->accepted) return;
    __shared__ double s_w[MMX_NROWS * MMX_NBASIS * 4];
    const int slot = st->end;
    const bool store = st->store_hist != 0;
    double acc[MMX_NROWS][MMX_NBASIS];
#pragma unroll
    for (int r = 0; r < MMX_NROWS; ++r)
#pragma unroll
        for (int b = 0; b < MMX_NBASIS; ++b) acc[r][b] = 0.0;

    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 X = x[i], XP = xp[i], G = g[i], GP = gp[i];
        float4 sn = make_float4(X.x - XP.x, X.y - XP.y, X.z - XP.z, X.w - XP.w);
        float4 yn = make_float4(G.x - GP.x, G.y - GP.y, G.z - GP.z, G.w - GP.w);
        if (!store) {
            sn = make_float4(0.f, 0.f, 0.f, 0.f);
            yn = sn;
        }
        xp[i] = X;
        gp[i] = G;
        float4 B[MMX_NBASIS];
#pragma unroll
        for (int a = 0; a < MMX_M; ++a) {
            B[a] = S[(size_t)a * n4 + i];
            B[MMX_M + a] = Y[(size_t)a * n4 + i];
        }
        B[2 * MMX_M] = G;
        if (store) {
#pragma unroll
            for (int a = 0; a < MMX_M; ++a)
                if (a == slot) {
                    B[a] = sn;
                    B[MMX_M + a] = yn;
                }
            S[(size_t)slot * n4 + i] = sn;
            Y[(size_t)slot * n4 + i] = yn;
        }
#pragma unroll
        for (int b = 0; b < MMX_NBASIS; ++b) {
            const float4 v = B[b];
            acc[0][b] += (double)(sn.x * v.x + sn.y * v.y) + (double)(sn.z * v.z + sn.w * v.w);
            acc[1][b] += (double)(yn.x * v.x + yn.y * v.y) + (double)(yn.z * v.z + yn.w * v.w);
            acc[2][b] += (double)(G.x * v.x + G.y * v.y) + (double)(G.z * v.z + G.w * v.w);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < MMX_NROWS; ++r)
#pragma unroll
        for (int b = 0; b < MMX_NBASIS; ++b) {
            const double s = wave_sum(acc[r][b]);
            if (lane == 0) s_w[(r * MMX_NBASIS + b) * 4 + wave] = s;
        }
    __syncthreads();
    if (threadIdx.x < MMX_NROWS * MMX_NBASIS) {
        const double *q = s_w + threadIdx.x * 4;
        rows[(size_t)threadIdx.x * kPartStride + blockIdx.x] = (q[0] + q[1]) + (q[2] + q[3]);
    }
}

// Gram update + two-loop recursion in coefficient space (liblbfgs lbfgs() main-loop tail).
__global__ __launch_bounds__(1024) void k_direction_coef(int nblk, const double *__restrict__ rows,
                                                         MinState *__restrict__ st) {
    if (st->phase == PH_DONE || !st->accepted) return;
    constexpr int NQ = MMX_NROWS * MMX_NBASIS;
    __shared__ double s_wave[NQ * 16];
    __shared__ double s_rows[NQ];
    __shared__ int s_n[NQ];
    if (threadIdx.x < NQ) s_n[threadIdx.x] = nblk;
    __syncthreads();
    multi_slot_sum<NQ>(rows, kPartStride, s_n, s_wave, s_rows);
    if (threadIdx.x != 0) return;
    constexpr int NB = MMX_NBASIS, M = MMX_M, IG = 2 * MMX_M;
    double *G = st->gram;
    const bool store = st->store_hist != 0;
    const int slot = st->end;
    if (store) {
        for (int b = 0; b < NB; ++b) {
            G[slot * NB + b] = G[b * NB + slot] = s_rows[0 * NB + b];
            G[(M + slot) * NB + b] = G[b * NB + (M + slot)] = s_rows[1 * NB + b];
        }
    }
    for (int b = 0; b < NB; ++b) G[IG * NB + b] = G[b * NB + IG] = s_rows[2 * NB + b];

    double c[NB], alpha[M];
    for (int b = 0; b < NB; ++b) c[b] = 0.0;
    c[IG] = -1.0;
    double step = 1.0;
    if (store) {
        const double ys = G[slot * NB + (M + slot)], yy = G[(M + slot) * NB + (M + slot)];
        st->ys[slot] = ys;
        const int bound = M <= st->k ? M : st->k;
        st->bound = bound;
        st->k += 1;
        const int end = (slot + 1) % M;
        st->end = end;
        int j = end;
        for (int i = 0; i < bound; ++i) {
            j = (j + M - 1) % M;
            double sd = 0.0;
            for (int b = 0; b < NB; ++b) sd += c[b] * G[j * NB + b];
            alpha[j] = sd / st->ys[j];
            c[M + j] -= alpha[j];
        }
        const double sc = ys / yy;
        for (int b = 0; b < NB; ++b) c[b] *= sc;
        for (int i = 0; i < bound; ++i) {
            double yd = 0.0;
            for (int b = 0; b < NB; ++b) yd += c[b] * G[(M + j) * NB + b];
            const double beta = yd / st->ys[j];
            c[j] += alpha[j] - beta;
            j = (j + 1) % M;
        }
    } else {
        step = 1.0 / sqrt(G[IG * NB + IG]); // first step of lbfgs(): 1/|d|, d = -g
    }
    double dginit = 0.0;
    for (int b = 0; b < NB; ++b) {
        st->coef[b] = c[b];
        dginit += c[b] * G[IG * NB + b];
    }
    st->dginit = dginit;
    st->finit = st->fx;
    st->step = step;
    st->ls_count = 0;
    st->phase = PH_LINESEARCH;
    if (!(dginit < 0.0) || !(step > 0.0)) { // LBFGSERR_INCREASEGRADIENT (or non-finite)
        st->status = -2;
        st->phase = PH_DONE;
    }
}

// d = sum_a coef[a] * B_a.  Streams 13 vectors in, one out.
__global__ __launch_bounds__(256) void k_direction(int n4, const float4 *__restrict__ g,
                                                   const float4 *__restrict__ S, const float4 *__restrict__ Y,
                                                   float4 *__restrict__ d, const MinState *__restrict__ st) {
    if (st->phase == PH_DONE || !st->accepted) return;
    float c[MMX_NBASIS];
#pragma unroll
    for (int b = 0; b < MMX_NBASIS; ++b) c[b] = (float)st->coef[b];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 G = g[i];
        float4 o = make_float4(c[2 * MMX_M] * G.x, c[2 * MMX_M] * G.y, c[2 * MMX_M] * G.z, c[2 * MMX_M] * G.w);
#pragma unroll
        for (int a = 0; a < MMX_M; ++a) {
            if (c[a] != 0.f) {
                const float4 s = S[(size_t)a * n4 + i];
                o.x = fmaf(c[a], s.x, o.x);
                o.y = fmaf(c[a], s.y, o.y);
                o.z = fmaf(c[a], s.z, o.z);
                o.w = fmaf(c[a], s.w, o.w);
            }
            if (c[MMX_M + a] != 0.f) {
                const float4 y = Y[(size_t)a * n4 + i];
                o.x = fmaf(c[MMX_M + a], y.x, o.x);
                o.y = fmaf(c[MMX_M + a], y.y, o.y);
                o.z = fmaf(c[MMX_M + a], y.z, o.z);
                o.w = fmaf(c[MMX_M + a], y.w, o.w);
            }
        }
        d[i] = o;
    }
}

__global__ __launch_bounds__(256) void k_copy4(int n4, const float4 *__restrict__ src, float4 *__restrict__ dst) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) dst[i] = src[i];
}

} // namespace mmx
