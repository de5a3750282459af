// mmx_lbfgs.hpp -- K6: device-resident L-BFGS (OpenMM LocalEnergyMinimizer == liblbfgs, m = 6,
// backtracking strong-Wolfe line search) as a state machine advanced once per evaluation.
//
// The two-loop recursion is done in coefficient space over the basis B = {S_0..S_5, Y_0..Y_5, g}:
// one streaming pass (k_history) right after the evaluation stores the (s,y) pair the step WOULD add and produces
// the three Gram rows that would change plus the line search's g.d and x.x; one block (k_decide) folds the energies
// and those rows, runs the line-search controller and -- when the step is accepted -- the recursion on the 13x13 Gram
// matrix; the next trial move (k_pack<.., DIR>) forms d = sum_a c_a B_a while it packs the positions.  No host
// round trip, ONE reduction point per evaluation (a multi-GPU run: one all-reduce of 57 doubles).
#pragma once
#include "mmx_common.hpp"

namespace mmx {

struct CtlArgs {
    int nblk[P_NSLOTS]; // block partials per slot written by the force kernels of this evaluation
};

constexpr int kCtlThreads = 1024;

// Deterministic sums of NS partial-sum slots by one 1024-thread block.  Work is cut in tasks of
// (slot, 1024-entry segment); each of the 16 waves takes tasks round-robin, issues its 16 loads per lane
// back to back and shuffle-reduces; thread s then folds the task results of slot s in task order.
// out[s] is valid on all threads after the call.
constexpr int kMaxTasks = 16 * 4 + 48; // segments of the few large slots + one per small slot
template <int NS>
__device__ __forceinline__ void multi_slot_sum(const double *__restrict__ part, int stride, const int *nblk,
                                               double *s_task /* [kMaxTasks] */, int *s_first /* [NS+1] */,
                                               double *s_out /* [NS] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        int t = 0;
        for (int s = 0; s < NS; ++s) {
            s_first[s] = t;
            t += (nblk[s] + 1023) >> 10;
        }
        s_first[NS] = min(t, kMaxTasks);
    }
    __syncthreads();
    const int ntask = s_first[NS];
    for (int task = wave; task < ntask; task += 16) {
        int s = 0;
        while (s + 1 < NS && s_first[s + 1] <= task) ++s;
        const int seg = task - s_first[s];
        const int n = nblk[s];
        const double *p = part + (size_t)s * stride + (size_t)seg * 1024;
        const int m = min(1024, n - seg * 1024);
        // (all 16 loads issued back to back, whatever m: the slot's row of kPartStride entries is allocated in full, what lies
        //  beyond its partials is discarded by value -- a load behind `i < m` is a branch and a wait of its own)
        double pv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) pv[k] = p[lane + 64 * k];
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += lane + 64 * k < m ? pv[k] : 0.0;
        v = wave_sum(v);
        if (lane == 0) s_task[task] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < NS) {
        double r = 0.0;
        for (int t = s_first[threadIdx.x]; t < min(s_first[threadIdx.x + 1], kMaxTasks); ++t) r += s_task[t];
        s_out[threadIdx.x] = r;
    }
    __syncthreads();
}

// Fold of k_history's partials: NQ rows of nblk (<= 256) block partials each, rows kPartStride apart.  Wave w
// takes rows w, w+16, w+32: every lane issues all its loads (<= 4 per row) back to back (rows_load; the caller puts
// the slot fold between load and finish, so both folds share one latency round), then three shuffle trees
// (rows_finish).  Fixed order => deterministic.  s_out valid after the barrier.
template <int NQ>
__device__ __forceinline__ void rows_load(const double *__restrict__ rows, int nblk, double (&acc)[3]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    acc[0] = acc[1] = acc[2] = 0.0;
    // (12 loads per lane issued back to back: rows beyond NQ and partials beyond nblk are read -- the buffer is allocated in
    //  full -- and discarded by value)
    double pv[3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double *p = rows + (size_t)min(wave + 16 * j, NQ - 1) * kPartStride;
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[j][k] = p[lane + 64 * k];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = wave + 16 * j;
        if (r < NQ) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[j] += lane + 64 * k < nblk ? pv[j][k] : 0.0;
        }
    }
}
template <int NQ>
__device__ __forceinline__ void rows_finish(const double (&acc)[3], double *s_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = wave + 16 * j;
        const double v = wave_sum(acc[j]);
        if (r < NQ && lane == 0) s_out[r] = v;
    }
    __syncthreads();
}

// k_tail's partials are TAGGED: a partial travels as two 64-bit words {low half | tag << 32, high half | tag << 32}, tag = the
// launch's epoch, stored with agent-scope atomic stores; the workgroup that folds them polls the words themselves until every
// tag is this launch's -- no ticket, no fence, no wait for store acknowledgements on the writers' side, and the fold's loads
// ARE the poll.  Layout: row r at rows + r * kPartStride (doubles), 16 bytes per workgroup.
__device__ __forceinline__ void rows_publish(double *__restrict__ rows, int r, int blk, double v, unsigned epoch) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)epoch << 32;
    unsigned long long *p = reinterpret_cast<unsigned long long *>(rows + (size_t)r * kPartStride) + 2 * blk;
    __hip_atomic_store(p, (b & 0xffffffffull) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, (b >> 32) | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Same assignment of rows to waves and of partials to lanes, same order of additions as rows_load: the same bits.
// false: a partial never arrived (bounded spin).
template <int NQ>
__device__ __forceinline__ bool rows_poll(const double *__restrict__ rows, int nblk, unsigned epoch, double (&acc)[3],
                                          const int spin_limit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long w0[3][4], w1[3][4];
    bool have[3][4]; // this lane's partial (j, k) has arrived: later rounds fetch only what is still missing -- the last rounds,
                     // the ones the decision waits for, are a load or two per lane instead of 24
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            have[j][k] = false;
            w0[j][k] = w1[j][k] = 0ull;
        }
    bool arrived = false;
    for (int spins = 0; spins <= spin_limit; ++spins) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const unsigned long long *p =
                reinterpret_cast<const unsigned long long *>(rows + (size_t)min(wave + 16 * j, NQ - 1) * kPartStride);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!have[j][k]) {
                    const int b = min(lane + 64 * k, nblk - 1);
                    w0[j][k] = __hip_atomic_load(p + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w1[j][k] = __hip_atomic_load(p + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                have[j][k] = (unsigned)(w0[j][k] >> 32) == epoch && (unsigned)(w1[j][k] >> 32) == epoch;
                ok = ok && have[j][k];
            }
        if (__all(ok)) {
            arrived = true;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    acc[0] = acc[1] = acc[2] = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = wave + 16 * j;
        if (r < NQ) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = __longlong_as_double((long long)((w0[j][k] & 0xffffffffull) | (w1[j][k] << 32)));
                acc[j] += lane + 64 * k < nblk ? v : 0.0;
            }
        }
    }
    return arrived;
}

// Line-search controller, run by ONE thread after every evaluation on the folded (and, in a multi-GPU
// run, all-reduced) slot sums: exactly liblbfgs' line_search_backtracking with
// LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE.  Every rank executes it on identical inputs.
// kerr: a force kernel of this evaluation reported that it could not do its work (MinState::kernel_error; multi-GPU: of
// any rank, through the all-reduce): the sums are partial, nothing may be decided on them.
// Everything the decision reads from *st.  The branches of controller_decide depend on loaded values, and a load behind each branch
// is a dependent L2 round trip of its own (four of them: 1.8 us of single-thread time); loaded up front they are ONE -- and
// k_tail's folding workgroup takes that one BEFORE it starts to poll for the other workgroups' partials (decide_block), so that
// nothing but arithmetic stands between the last partial and the decision.  Nobody else writes these fields in between: the
// kernels of the evaluation are done when the tail starts, the next evaluation's wait for it.
struct DecideIn {
    int phase, evals, ls_count, nan_seen, iters, k, max_iters, kernel_error, cell_stale;
    double finit, step, dginit, epsilon, tolerance, n_total;
};
__device__ __forceinline__ DecideIn decide_in(const MinState *__restrict__ st) {
    DecideIn in;
    in.phase = st->phase;
    in.evals = st->evals;
    in.ls_count = st->ls_count;
    in.nan_seen = st->nan_seen;
    in.iters = st->iters;
    in.k = st->k;
    in.max_iters = st->max_iters;
    in.kernel_error = st->kernel_error;
    in.cell_stale = st->cell_stale;
    in.finit = st->finit;
    in.step = st->step;
    in.dginit = st->dginit;
    in.epsilon = st->epsilon;
    in.tolerance = st->tolerance;
    in.n_total = st->n_total;
    return in;
}
__device__ __forceinline__ void controller_decide(MinState *__restrict__ st, const double *sums, const bool kerr, const DecideIn &in) {
    const int phase = in.phase;
    const int evals0 = in.evals, ls_count0 = in.ls_count, nan_seen0 = in.nan_seen, iters0 = in.iters, k0 = in.k, max_iters = in.max_iters;
    const double finit = in.finit, step0 = in.step, dginit0 = in.dginit, epsilon0 = in.epsilon, tolerance = in.tolerance,
                 n_total = in.n_total;
    double f = 0.0;
    for (int t = 0; t < 9; ++t) {
        st->eterms[t] = sums[t];
        f += sums[t];
    }
    st->ftrial = f;
    const double dg = sums[P_GD], gg = sums[P_GG], xx = sums[P_XX];
    if (kerr) {
        st->kernel_error |= 0x100; // (a rank without a failure of its own learns of it here)
        st->accepted = 0;
        st->store_hist = 0;
        if (phase != PH_IDLE) {
            st->status = -7; // MMX_MIN_KERNEL
            st->phase = PH_DONE;
        }
        return;
    }
    if (phase == PH_IDLE) return; // plain mmx_compute()

    const double ftol = 1e-4, wolfe = 0.9, min_step = 1e-20, max_step = 1e20;
    const int max_linesearch = 40;
    const bool finite = (f - f == 0.0) && (gg - gg == 0.0);
    st->evals = evals0 + 1;
    st->accepted = 0;
    st->store_hist = 0;

    if (phase == PH_INIT) {
        if (!finite) {
            st->status = -6; // MMX_MIN_NAN
            st->phase = PH_DONE;
            return;
        }
        st->fx = f;
        st->f0 = f;
        for (int t = 0; t < 9; ++t) st->eterms_acc[t] = sums[t];
        double xn = sqrt(xx);
        // OpenMM LocalEnergyMinimizer: epsilon = tolerance / max(1, sqrt(mean_i |x_i|^2)) of the starting positions
        // (x.x of this very evaluation: no copy of the positions to the host, no all-gather on a decomposed run)
        double rms = sqrt(xx / n_total);
        if (rms < 1.0) rms = 1.0;
        const double epsilon = tolerance / rms;
        st->epsilon = epsilon;
        if (xn < 1.0) xn = 1.0;
        st->xnorm = xn;
        const double gnorm = sqrt(gg);
        st->gnorm = gnorm;
        if (gnorm / xn <= epsilon) {
            st->status = 0;
            st->phase = PH_DONE;
            return;
        }
        st->accepted = 1; // history kernel snapshots (xp,gp); direction becomes -g, step 1/|g|
        return;
    }

    // PH_LINESEARCH
    const int ls_count = ls_count0 + 1;
    st->ls_count = ls_count;
    const int nan_seen = nan_seen0 + (finite ? 0 : 1);
    if (!finite) st->nan_seen = nan_seen;
    double width;
    bool accept = false;
    if (!finite || f > finit + step0 * ftol * dginit0) {
        width = 0.5;
    } else if (dg < wolfe * dginit0) {
        width = 2.1;
    } else if (dg > -wolfe * dginit0) {
        width = 0.5;
    } else {
        accept = true;
        width = 1.0;
    }
    if (!accept) {
        int err = 0;
        if (step0 < min_step) err = -3;
        else if (step0 > max_step) err = -4;
        else if (max_linesearch <= ls_count) err = -5;
        if (err) {
            st->status = (nan_seen > 0 && !finite) ? -6 : err;
            st->phase = PH_DONE; // host restores x = xp (liblbfgs reverts to the previous point)
            return;
        }
        st->step = step0 * width;
        return;
    }
    // accepted: one L-BFGS iteration finished
    st->iters = iters0 + 1;
    st->fx = f;
    for (int t = 0; t < 9; ++t) st->eterms_acc[t] = sums[t];
    double xn = sqrt(xx);
    if (xn < 1.0) xn = 1.0;
    st->xnorm = xn;
    const double gnorm = sqrt(gg);
    st->gnorm = gnorm;
    if (gnorm / xn <= epsilon0) {
        st->status = 0;
        st->phase = PH_DONE;
        return;
    }
    if (max_iters != 0 && max_iters < k0 + 1) {
        st->status = 1;
        st->phase = PH_DONE;
        return;
    }
    st->accepted = 1;
    st->store_hist = 1;
}


__device__ __forceinline__ void controller_decide(MinState *__restrict__ st, const double *sums, const bool kerr) {
    const DecideIn in = decide_in(st);
    controller_decide(st, sums, kerr, in);
}

// Runs right after every evaluation of the minimizer, BEFORE the line-search decision (so that decision and
// direction coefficients are one launch, k_decide): s = x - xp, y = g - gp into slot `end` -- the slot the step
// would occupy if accepted; a rejected trial just leaves values there that the next trial overwrites, nothing reads
// that slot in between (d is already formed, the Gram matrix only changes on acceptance) -- and the Gram rows of
// {s_new, y_new, g} against the whole basis as block partials rows[(r*13 + b)*stride + blockIdx.x].  The last column
// group also takes the two line-search reductions that are not Gram entries: g.d (row MMX_ROW_GD) and x.x
// (MMX_ROW_XX).  In PH_INIT (first evaluation) nothing is stored and s = y = 0.
// 2-D grid: blockIdx.y picks a group of <= 4 basis columns (12 fp64 accumulators per thread instead of
// 39: full occupancy); every group re-reads x, xp, g, gp (L2/MALL resident), only the last group
// stores the new pair.  xp <- x and gp <- g are done afterwards by the next trial move (no intra-kernel race).
constexpr int kHistGroups = 4;
__global__ __launch_bounds__(256) void k_history(int n4, const float4 *__restrict__ x, const float4 *__restrict__ xp,
                                                 const float4 *__restrict__ g, const float4 *__restrict__ gp,
                                                 const float4 *__restrict__ d, float4 *__restrict__ S,
                                                 float4 *__restrict__ Y, double *__restrict__ rows,
                                                 const MinState *__restrict__ st) {
    const int phase = st->phase;
    if (phase != PH_INIT && phase != PH_LINESEARCH) return;
    __shared__ double s_w[MMX_NROWS * 4 * 4];
    const int slot = st->end;
    const bool store = phase == PH_LINESEARCH;
    const bool last = blockIdx.y == kHistGroups - 1;
    double e_gd = 0.0, e_xx = 0.0;
    const int cg = blockIdx.y, col0 = cg * 4, ncol = min(4, MMX_NBASIS - col0);
    const float4 *colp[4];
    int colkind[4]; // 0: stored vector, 1: s_new, 2: y_new, 3: g
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = col0 + k;
        colp[k] = x;
        colkind[k] = 0;
        if (b < MMX_M) {
            colp[k] = S + (size_t)b * n4;
            if (store && b == slot) colkind[k] = 1;
        } else if (b < 2 * MMX_M) {
            colp[k] = Y + (size_t)(b - MMX_M) * n4;
            if (store && b - MMX_M == slot) colkind[k] = 2;
        } else {
            colkind[k] = 3;
        }
    }
    double acc[MMX_NROWS][4];
#pragma unroll
    for (int r = 0; r < MMX_NROWS; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[r][k] = 0.0;

    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const float4 X = x[i], XP = xp[i], G = g[i], GP = gp[i];
        float4 sn = make_float4(X.x - XP.x, X.y - XP.y, X.z - XP.z, X.w - XP.w);
        float4 yn = make_float4(G.x - GP.x, G.y - GP.y, G.z - GP.z, G.w - GP.w);
        if (!store) {
            sn = make_float4(0.f, 0.f, 0.f, 0.f);
            yn = sn;
        }
        if (last) {
            const float4 D = d[i];
            e_gd += ((double)G.x * D.x + (double)G.y * D.y) + ((double)G.z * D.z + (double)G.w * D.w);
            e_xx += ((double)X.x * X.x + (double)X.y * X.y) + ((double)X.z * X.z + (double)X.w * X.w);
            if (store) {
                S[(size_t)slot * n4 + i] = sn;
                Y[(size_t)slot * n4 + i] = yn;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < ncol) {
                float4 v;
                if (colkind[k] == 0) v = colp[k][i];
                else if (colkind[k] == 1) v = sn;
                else if (colkind[k] == 2) v = yn;
                else v = G;
                acc[0][k] += (double)(sn.x * v.x + sn.y * v.y) + (double)(sn.z * v.z + sn.w * v.w);
                acc[1][k] += (double)(yn.x * v.x + yn.y * v.y) + (double)(yn.z * v.z + yn.w * v.w);
                acc[2][k] += (double)(G.x * v.x + G.y * v.y) + (double)(G.z * v.z + G.w * v.w);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < MMX_NROWS; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double sm = wave_sum(acc[r][k]);
            if (lane == 0) s_w[(r * 4 + k) * 4 + wave] = sm;
        }
    __syncthreads();
    if (threadIdx.x < MMX_NROWS * 4) {
        const int r = threadIdx.x >> 2, k = threadIdx.x & 3;
        if (k < ncol) {
            const double *q = s_w + threadIdx.x * 4;
            rows[(size_t)(r * MMX_NBASIS + col0 + k) * kPartStride + blockIdx.x] = (q[0] + q[1]) + (q[2] + q[3]);
        }
    }
    if (last) { // block-uniform
        __syncthreads();
        const double s1 = wave_sum(e_gd), s2 = wave_sum(e_xx);
        if (lane == 0) {
            s_w[wave] = s1;
            s_w[4 + wave] = s2;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            const double *q = s_w + 4 * threadIdx.x;
            rows[(size_t)(MMX_ROW_GD + threadIdx.x) * kPartStride + blockIdx.x] = (q[0] + q[1]) + (q[2] + q[3]);
        }
    }
}

// Gram update + two-loop recursion in coefficient space (liblbfgs lbfgs() main-loop tail), run by ONE
// thread on the folded (multi-GPU: all-reduced) Gram rows.
// G: the 13x13 Gram matrix -- st->gram itself, or a copy of it in LDS that the caller loaded with all threads and
// stores back afterwards (the recursion reads 12 rows one after the other: 12 dependent L2 round trips on global
// memory, 5 us of single-thread time; ~0.5 us out of LDS).
// ysl: the curvatures <y_k, s_k> of the stored pairs, loaded by the caller BEFORE any store to *st (a load of
// st->ys[j] inside the recursion cannot be hoisted over those stores: 12 more dependent round trips).
__device__ __forceinline__ void coef_decide(MinState *__restrict__ st, const double *s_rows, double *G, double *ysl) {
    constexpr int NB = MMX_NBASIS, M = MMX_M, IG = 2 * MMX_M;
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
        ysl[slot] = ys;
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
            alpha[j] = sd / ysl[j];
            c[M + j] -= alpha[j];
        }
        const double sc = ys / yy;
        for (int b = 0; b < NB; ++b) c[b] *= sc;
        for (int i = 0; i < bound; ++i) {
            double yd = 0.0;
            for (int b = 0; b < NB; ++b) yd += c[b] * G[(M + j) * NB + b];
            const double beta = yd / ysl[j];
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


// Dots of the line search as k_history produced them (g.g is the Gram entry <g,g>).
__device__ __forceinline__ void dots_from_rows(double *sums, const double *rowsum) {
    sums[P_GD] = rowsum[MMX_ROW_GD];
    sums[P_GG] = rowsum[2 * MMX_NBASIS + 2 * MMX_M];
    sums[P_XX] = rowsum[MMX_ROW_XX];
}

__device__ __forceinline__ void slot_counts(const CtlArgs &A, int *s_n) {
    if (threadIdx.x < P_NSLOTS) s_n[threadIdx.x] = A.nblk[threadIdx.x];
    __syncthreads();
}

// Plain evaluation (mmx_compute, MD reports), single GPU: fold the energy partials.
__global__ __launch_bounds__(1024) void k_controller(const CtlArgs A, const double *__restrict__ part,
                                                     MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    __shared__ double s_task[kMaxTasks];
    __shared__ double s_out[P_NSLOTS];
    __shared__ int s_n[P_NSLOTS], s_first[P_NSLOTS + 1];
    slot_counts(A, s_n);
    multi_slot_sum<P_NSLOTS>(part, kPartStride, s_n, s_task, s_first, s_out);
    if (threadIdx.x != 0) return;
    double sums[P_NSLOTS];
#pragma unroll
    for (int s = 0; s < P_NSLOTS; ++s) sums[s] = s_out[s];
    controller_decide(st, sums, st->kernel_error != 0);
}

// Minimizer, single GPU: energies + k_history rows folded together, line-search decision, and on acceptance the
// direction coefficients.  The work of ONE 1024-thread workgroup: k_decide (a launch of its own), or the workgroup of k_tail
// that finishes last.
constexpr int kTailSpinLimit = 1 << 20; // polls of k_tail's folding workgroup before it gives up (a bug, not a state of the data)
template <bool POLL = false>
__device__ __forceinline__ void decide_block(const CtlArgs &A, const double *__restrict__ part, int nblk_rows,
                                             const double *__restrict__ rows, MinState *__restrict__ st,
                                             const unsigned epoch = 0u, const int spin_limit = kTailSpinLimit) {
    static_assert(MMX_NROWSUM <= 48, "rows_load covers 3 rows per wave");
    __shared__ double s_task[kMaxTasks];
    __shared__ double s_out[P_NSLOTS];
    __shared__ double s_rows[MMX_NROWSUM];
    __shared__ double s_G[MMX_NBASIS * MMX_NBASIS];
    __shared__ int s_n[P_NSLOTS], s_first[P_NSLOTS + 1];
    __shared__ int s_accepted;
    __shared__ int s_wait_failed;
    __shared__ DecideIn s_din;
    __shared__ double s_ys[MMX_M];
    double acc[3];
    if (threadIdx.x == 0) {
        s_din = decide_in(st); // (before the poll: see DecideIn)
        s_wait_failed = 0;
    }
    if (!POLL) rows_load<MMX_NROWSUM>(rows, nblk_rows, acc); // nblk_rows <= 256 (enqueue_history)
    // the Gram matrix rides in the same latency round as the partials
    const double gval = threadIdx.x < MMX_NBASIS * MMX_NBASIS ? st->gram[threadIdx.x] : 0.0;
    const double yval = threadIdx.x < MMX_M ? st->ys[threadIdx.x] : 0.0;
    slot_counts(A, s_n);
    multi_slot_sum<P_NSLOTS>(part, kPartStride, s_n, s_task, s_first, s_out);
    if (threadIdx.x < MMX_NBASIS * MMX_NBASIS) s_G[threadIdx.x] = gval;
    if (threadIdx.x < MMX_M) s_ys[threadIdx.x] = yval;
    // (k_tail: what the other workgroups of this very launch publish is waited for last -- everything above came from earlier launches)
    // (s_din / s_wait_failed were written before the barriers of multi_slot_sum)
    if (POLL && !rows_poll<MMX_NROWSUM>(rows, nblk_rows, epoch, acc, spin_limit) && (threadIdx.x & 63) == 0) {
        atomicOr(&st->kernel_error, (int)KERR_TAIL_WAIT);
        s_wait_failed = 1;
    }
    STAGE_STAMP(4098);
    rows_finish<MMX_NROWSUM>(acc, s_rows); // ends with a barrier: s_G is complete too
    STAGE_STAMP(4099);
    if (threadIdx.x == 0) {
        const DecideIn in = s_din;
        const bool kerr = in.kernel_error != 0 || s_wait_failed != 0;
        if (in.cell_stale && !kerr) {
            // the kept cell structure was out of date for this evaluation (k_pack): it never happened.  Nothing is decided;
            // every kernel of the evaluations already in the stream returns at once, the host builds anew and repeats it.
            st->halt_phase = in.phase;
            st->halt_reason = ((in.cell_stale & 1) ? 4 : 0) | ((in.cell_stale & 2) ? 8 : 0) | ((in.cell_stale & 4) ? 16 : 0); // stale structure / slot table too small / grid beyond the direct build
            st->phase = PH_HALT;
            st->accepted = 0; // the direction of this trial is already formed (k_pack): the repeat must not form it again
            s_accepted = 0;
        } else {
            double sums[P_NSLOTS];
#pragma unroll
            for (int s = 0; s < P_NSLOTS; ++s) sums[s] = s_out[s];
            dots_from_rows(sums, s_rows);
            controller_decide(st, sums, kerr, in);
            STAGE_STAMP(4100);
            s_accepted = st->accepted;
            if (s_accepted) coef_decide(st, s_rows, s_G, s_ys);
            STAGE_STAMP(4101);
        }
    }
    __syncthreads();
    if (s_accepted && threadIdx.x < MMX_NBASIS * MMX_NBASIS) st->gram[threadIdx.x] = s_G[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_decide(const CtlArgs A, const double *__restrict__ part, int nblk_rows,
                                                 const double *__restrict__ rows, MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) return;
    decide_block<false>(A, part, nblk_rows, rows, st);
}

// Entries 0..15 of the all-reduced array of a decomposed run: the P_NSLOTS slot sums, then the flags every rank must see
// alike (the host's policies read them from the all-reduced array too, never from a rank's own flag) -- 12: ranks on
// which a bead has moved beyond half the skin since the ghost lists were built (k_dd_displacement), 13: ranks whose force
// kernel failed (MinState::kernel_error), 14: ranks with a ghost list longer than its message (k_dd_build_lists).
constexpr int kSumStale = 12, kSumKernelError = 13, kSumOverflow = 14;
// 15: ranks whose direct cell build was void (MinState::cell_stale): + 1 per rank with a slot row too short, + 1024 per rank with a
// grid beyond the build (64 ranks at most: the two counts do not mix)
constexpr int kSumCellVoid = 15;
__device__ __forceinline__ double flag_or_sum(const MinState *__restrict__ st, const double *s_out, int t) {
    if (t < P_NSLOTS) return s_out[t];
    if (t == kSumStale) return st->dd_stale ? 1.0 : 0.0;
    if (t == kSumKernelError) return st->kernel_error ? 1.0 : 0.0;
    if (t == kSumOverflow) return st->dd_overflow ? 1.0 : 0.0;
    if (t == kSumCellVoid) return ((st->cell_stale & 2) ? 1.0 : 0.0) + ((st->cell_stale & 4) ? 1024.0 : 0.0);
    return 0.0;
}

// Multi-GPU: fold -> st->sums (+ st->rowsum) -> ncclAllReduce (fp64 sum, in place) -> decide on every rank.
__global__ __launch_bounds__(1024) void k_reduce_slots(const CtlArgs A, const double *__restrict__ part,
                                                       MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) {
        if (threadIdx.x < 16) st->sums[threadIdx.x] = 0.0; // keep the collective's input finite
        return;
    }
    __shared__ double s_task[kMaxTasks];
    __shared__ double s_out[P_NSLOTS];
    __shared__ int s_n[P_NSLOTS], s_first[P_NSLOTS + 1];
    slot_counts(A, s_n);
    multi_slot_sum<P_NSLOTS>(part, kPartStride, s_n, s_task, s_first, s_out);
    if (threadIdx.x < 16) st->sums[threadIdx.x] = flag_or_sum(st, s_out, threadIdx.x);
}
__global__ void k_controller_decide(MinState *__restrict__ st) {
    if (st->phase >= PH_DONE || threadIdx.x != 0) return;
    double sums[P_NSLOTS];
    for (int s = 0; s < P_NSLOTS; ++s) sums[s] = st->sums[s];
    controller_decide(st, sums, st->sums[kSumKernelError] > 0.5);
}
// (the work of one 1024-thread workgroup: k_reduce_all, or the workgroup of k_tail that finishes last)
template <bool POLL = false>
__device__ __forceinline__ void reduce_all_block(const CtlArgs &A, const double *__restrict__ part, int nblk_rows,
                                                 const double *__restrict__ rows, MinState *__restrict__ st,
                                                 const unsigned epoch = 0u, const int spin_limit = kTailSpinLimit) {
    __shared__ double s_task[kMaxTasks];
    __shared__ double s_out[P_NSLOTS];
    __shared__ double s_rows[MMX_NROWSUM];
    __shared__ int s_n[P_NSLOTS], s_first[P_NSLOTS + 1];
    double acc[3];
    if (!POLL) rows_load<MMX_NROWSUM>(rows, nblk_rows, acc);
    slot_counts(A, s_n);
    multi_slot_sum<P_NSLOTS>(part, kPartStride, s_n, s_task, s_first, s_out);
    if (POLL && !rows_poll<MMX_NROWSUM>(rows, nblk_rows, epoch, acc, spin_limit) && (threadIdx.x & 63) == 0)
        atomicOr(&st->kernel_error, (int)KERR_TAIL_WAIT);
    __syncthreads(); // (a time-out just raised is in the flags that go out below)
    rows_finish<MMX_NROWSUM>(acc, s_rows);
    // slot 12 of the all-reduced array: "some rank's ghost lists are out of date" (see k_dd_displacement)
    if (threadIdx.x < 16) st->sums[threadIdx.x] = flag_or_sum(st, s_out, threadIdx.x);
    if (threadIdx.x < MMX_NROWSUM) st->rowsum[threadIdx.x] = s_rows[threadIdx.x];
    if (threadIdx.x == 64) { // this rank's largest trial move of the evaluation rides in the same all-reduce
        // (the all-reduce SUMS: the 8th power of the squared move, so that the sum over the ranks is within 8^(1/8) = 1.3 of the
        //  largest term -- the host takes the 8th root)
        const float m2 = __uint_as_float(st->dd_move2_bits);
        const double q = (m2 >= 0.f && m2 < 1e4f) ? (double)m2 : 1e4, q2 = q * q, q4 = q2 * q2;
        st->dd_move = q4 * q4;
        st->dd_move2_bits = 0u;
    }
}
__global__ __launch_bounds__(1024) void k_reduce_all(const CtlArgs A, const double *__restrict__ part, int nblk_rows,
                                                     const double *__restrict__ rows, MinState *__restrict__ st) {
    if (st->phase >= PH_DONE) {
        if (threadIdx.x < 16) st->sums[threadIdx.x] = 0.0;
        if (threadIdx.x < MMX_NROWSUM) st->rowsum[threadIdx.x] = 0.0;
        if (threadIdx.x == 0) st->dd_move = 0.0;
        return;
    }
    reduce_all_block<false>(A, part, nblk_rows, rows, st);
}
__global__ void k_decide_reduced(MinState *__restrict__ st) {
    if (st->phase >= PH_DONE || threadIdx.x != 0) return;
    if (st->dd_move > st->dd_move2_max) st->dd_move2_max = st->dd_move;
    if (st->sums[kSumKernelError] <= 0.5 && (st->sums[kSumStale] > 0.5 || st->sums[kSumOverflow] > 0.5 || st->sums[kSumCellVoid] > 0.5)) { // a ghost is missing somewhere, or a rank's cell build was void: this evaluation never happened (every rank sees the same sums)
        const double cv = st->sums[kSumCellVoid];
        st->halt_phase = st->phase;
        st->halt_reason = (st->sums[kSumStale] > 0.5 ? 1 : 0) | (st->sums[kSumOverflow] > 0.5 ? 2 : 0) |
                          (cv - 1024.0 * floor(cv / 1024.0) > 0.5 ? 8 : 0) | (cv >= 1023.5 ? 16 : 0);
        st->phase = PH_HALT;
        st->accepted = 0; // the direction of this trial is already formed (k_pack): the repeat must not form it again
        return;
    }
    double ysl[MMX_M];
    for (int k = 0; k < MMX_M; ++k) ysl[k] = st->ys[k];
    double sums[P_NSLOTS];
    for (int s = 0; s < P_NSLOTS; ++s) sums[s] = st->sums[s];
    dots_from_rows(sums, st->rowsum);
    controller_decide(st, sums, st->sums[kSumKernelError] > 0.5);
    if (st->accepted) coef_decide(st, st->rowsum, st->gram, ysl);
}

// ---- k_tail: what follows the pair kernel of a minimizer's evaluation, in ONE launch --------------------------------------
// (1) the half-shell kernel's forces leave their cluster slots: g[li] -= fsort[slot_of[li]] (k_nb_n3_unsort's job -- here only
//     the slots of real owned beads are visited, through the bead -> slot table the cell build wrote; the padded sweep, its
//     launch and its pass over g are gone);
// (2) k_history's pass, with the same summation order (bitwise the same partials): a workgroup of 1024 threads is the four
//     column groups of k_history on the same 256 float4 -- x, xp, gp and the (merged) gradient are fetched ONCE per workgroup,
//     one by each group, and handed over in LDS instead of being read by four workgroups each;
// (3) the workgroup with the highest index folds the partials of all of them -- tagged, so that it can poll the values themselves
//     (rows_publish / rows_poll) -- and decides: k_decide's work (decomposed ranks: k_reduce_all's, the all-reduce and
//     k_decide_reduced follow), instead of a launch of one workgroup behind a grid drain.
struct TailArgs {
    const int *slot_of; // [n_own] cluster slot of every owned bead, local order (written by emit_clusters)
    float *fsort;       // [3][fstride] force per cluster slot; zero again afterwards
    int fstride, n_own;
    int *n3_queue;      // head of the half-shell kernel's work queue: rewound for the next launch on this cell build
    unsigned epoch;     // tag of this launch's partials (rows_publish): never 0, different from the previous launch's
    int spin_limit;     // polls before the folding workgroup gives up (tests inject 0)
};
constexpr int kTailKeep = 12; // tiles of a workgroup whose merged gradient stays in LDS between the merge and the history pass
                            // (48 KB: every tile of systems up to a million beads -- beyond, the rest is read back from g)
template <bool UNSORT, bool SOLO>
__global__ __launch_bounds__(1024) void k_tail(int n4, const float4 *__restrict__ x, const float4 *__restrict__ xp,
                                               float4 *g, const float4 *__restrict__ gp,
                                               const float4 *__restrict__ d, float4 *__restrict__ S, float4 *__restrict__ Y,
                                               double *__restrict__ rows, MinState *__restrict__ st, const TailArgs T,
                                               const CtlArgs A, const double *__restrict__ part) {
    const int phase = st->phase;
    if (phase != PH_INIT && phase != PH_LINESEARCH) {
        if (!SOLO && blockIdx.x == 0) { // keep the collective's input finite (k_reduce_all)
            if (threadIdx.x < 16) st->sums[threadIdx.x] = 0.0;
            if (threadIdx.x < MMX_NROWSUM) st->rowsum[threadIdx.x] = 0.0;
            if (threadIdx.x == 0) st->dd_move = 0.0;
        }
        return;
    }
    STAGE_STAMP(blockIdx.x * 4);
    __shared__ float4 s_v[4][256];
    __shared__ float4 s_g[UNSORT ? kTailKeep : 1][256];
    __shared__ double s_w[4][MMX_NROWS * 4 * 4];
    const int slot = st->end;
    const bool store = phase == PH_LINESEARCH;
    const int cg = threadIdx.x >> 8, tv = threadIdx.x & 255; // column group (k_history's blockIdx.y), thread of the group
    const bool last = cg == kHistGroups - 1;
    double e_gd = 0.0, e_xx = 0.0;
    const int col0 = cg * 4, ncol = min(4, MMX_NBASIS - col0);
    // columns of this group; a column that is not a stored vector (s_new, y_new, g, or none) points at x: its load is issued
    // like the others -- every thread has the same loads in flight, no branch around any of them -- and ignored
    const float4 *colp[4];
    int colkind[4]; // 0: stored vector, 1: s_new, 2: y_new, 3: g
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int b = col0 + k;
        colp[k] = x;
        colkind[k] = 3;
        if (b < MMX_M) {
            colkind[k] = (store && b == slot) ? 1 : 0;
            if (colkind[k] == 0) colp[k] = S + (size_t)b * n4;
        } else if (b < 2 * MMX_M) {
            colkind[k] = (store && b - MMX_M == slot) ? 2 : 0;
            if (colkind[k] == 0) colp[k] = Y + (size_t)(b - MMX_M) * n4;
        }
    }
    // the vector this group fetches for all four: x, xp, gp, and the gradient
    const float4 *const shared_src = cg == 0 ? x : cg == 1 ? xp : cg == 2 ? gp : (const float4 *)g;
    double acc[MMX_NROWS][4];
#pragma unroll
    for (int r = 0; r < MMX_NROWS; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[r][k] = 0.0;

    const int stride = gridDim.x * 256, base0 = blockIdx.x * 256;
    const int ntile = base0 < n4 ? (n4 - base0 + stride - 1) / stride : 0;
    // loads of tile t: issued one tile ahead of their use
    float4 ncv[4], nmine, nD;
    auto prefetch = [&](int t) {
        const int ic = min(base0 + t * stride + tv, n4 - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) ncv[k] = colp[k][ic];
        nmine = shared_src[ic]; // (group 3 with UNSORT: taken again below, after the merge)
        nD = d[ic];
    };
    // (the first tile's loads go out BEHIND the merge's gathers: loads return in order, and the gathers -- the second link of the
    //  merge's chain bead -> slot -> force -- would otherwise wait for six streaming loads that nobody needs before the history pass)
    bool fetched = false;
    if (!UNSORT && ntile > 0) {
        prefetch(0);
        fetched = true;
    }
    if (UNSORT) {
        // ---- the pair forces leave their cluster slots.  All the tiles of the workgroup at once: group q merges tiles q, q + 4, ...
        for (int t = cg; t < ntile; t += 4) {
            const int i = base0 + t * stride + tv;
            if (i < n4) {
                float4 G = g[i];
                // flat elements 4i .. 4i+3 = components of beads li0 and li0 + 1
                const int e0 = 4 * i, li0 = e0 / 3, r = e0 - 3 * li0;
                const int s0 = li0 < T.n_own ? T.slot_of[li0] : -1;
                const int s1 = li0 + 1 < T.n_own ? T.slot_of[li0 + 1] : -1;
                float f[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int q = r + c, nb = q >= 3 ? 1 : 0, comp = q - 3 * nb, sl = nb ? s1 : s0;
                    f[c] = 0.f;
                    if (sl >= 0) {
                        float *p = T.fsort + (size_t)comp * T.fstride + sl;
                        f[c] = *p;
                        *p = 0.f;
                    }
                }
                if (!fetched) {
                    prefetch(0);
                    fetched = true;
                }
                G.x -= f[0];
                G.y -= f[1];
                G.z -= f[2];
                G.w -= f[3];
                g[i] = G; // the gradient of this evaluation, complete (what the next trial move reads)
                if (t < kTailKeep) s_g[t][tv] = G;
            }
        }
        if (!fetched && ntile > 0) prefetch(0); // (a thread that merged nothing)
        if (ntile > kTailKeep) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // tiles beyond kTailKeep are read back from g by another wave
        __syncthreads();
    }
    STAGE_STAMP(blockIdx.x * 4 + 1);
    for (int t = 0; t < ntile; ++t) { // (block-uniform trip count: barriers inside)
        const int i = base0 + t * stride + tv;
        const bool in = i < n4;
        float4 cv[4], mine = nmine;
        const float4 D = nD;
#pragma unroll
        for (int k = 0; k < 4; ++k) cv[k] = ncv[k];
        if (UNSORT && last) mine = t < kTailKeep ? s_g[t][tv] : g[min(i, n4 - 1)];
        if (t + 1 < ntile) prefetch(t + 1);
        __syncthreads(); // the previous tile has been read
        s_v[cg][tv] = mine;
        __syncthreads();
        const float4 X = s_v[0][tv], XP = s_v[1][tv], GP = s_v[2][tv], G = s_v[3][tv];
        if (!in) continue; // (after the barriers)
        float4 sn = make_float4(X.x - XP.x, X.y - XP.y, X.z - XP.z, X.w - XP.w);
        float4 yn = make_float4(G.x - GP.x, G.y - GP.y, G.z - GP.z, G.w - GP.w);
        if (!store) {
            sn = make_float4(0.f, 0.f, 0.f, 0.f);
            yn = sn;
        }
        if (last) {
            e_gd += ((double)G.x * D.x + (double)G.y * D.y) + ((double)G.z * D.z + (double)G.w * D.w);
            e_xx += ((double)X.x * X.x + (double)X.y * X.y) + ((double)X.z * X.z + (double)X.w * X.w);
            if (store) {
                S[(size_t)slot * n4 + i] = sn;
                Y[(size_t)slot * n4 + i] = yn;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < ncol) {
                float4 v;
                if (colkind[k] == 0) v = cv[k];
                else if (colkind[k] == 1) v = sn;
                else if (colkind[k] == 2) v = yn;
                else v = G;
                acc[0][k] += (double)(sn.x * v.x + sn.y * v.y) + (double)(sn.z * v.z + sn.w * v.w);
                acc[1][k] += (double)(yn.x * v.x + yn.y * v.y) + (double)(yn.z * v.z + yn.w * v.w);
                acc[2][k] += (double)(G.x * v.x + G.y * v.y) + (double)(G.z * v.z + G.w * v.w);
            }
        }
    }
    STAGE_STAMP(blockIdx.x * 4 + 2);
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3; // wave of the group
    {
        // the twelve wave totals in one go (wave_sum12: the same bits as twelve wave_sum calls, a third of their instructions --
        // sixteen waves per CU fold here, and every workgroup's partials are on the critical path of the one that decides)
        static_assert(MMX_NROWS * 4 == 12, "wave_sum12");
        double flat[12], tot[3];
#pragma unroll
        for (int r = 0; r < MMX_NROWS; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) flat[r * 4 + k] = acc[r][k];
        wave_sum12(flat, tot);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) s_w[cg][(j + 3 * (lane >> 4)) * 4 + wave] = tot[j];
        }
    }
    __syncthreads();
    if (tv < MMX_NROWS * 4) {
        const int r = tv >> 2, k = tv & 3;
        if (k < ncol) {
            const double *q = s_w[cg] + tv * 4;
            rows_publish(rows, r * MMX_NBASIS + col0 + k, blockIdx.x, (q[0] + q[1]) + (q[2] + q[3]), T.epoch);
        }
    }
    __syncthreads();
    {
        const double s1 = wave_sum(e_gd), s2 = wave_sum(e_xx);
        if (last && lane == 0) {
            s_w[cg][wave] = s1;
            s_w[cg][4 + wave] = s2;
        }
        __syncthreads();
        if (last && tv < 2) {
            const double *q = s_w[cg] + 4 * tv;
            rows_publish(rows, MMX_ROW_GD + tv, blockIdx.x, (q[0] + q[1]) + (q[2] + q[3]), T.epoch);
        }
    }
    STAGE_STAMP(blockIdx.x * 4 + 3);
    // ---- the workgroup with the highest index (dispatched last) folds and decides: it polls the tagged partials of the others
    // (rows_poll); they are independent of it, so it cannot keep any of them from running
    if (blockIdx.x != gridDim.x - 1) return;
    STAGE_STAMP(4096);
    if (threadIdx.x == 0 && T.n3_queue) *T.n3_queue = 0;
    if (SOLO) decide_block<true>(A, part, (int)gridDim.x, rows, st, T.epoch, T.spin_limit);
    else reduce_all_block<true>(A, part, (int)gridDim.x, rows, st, T.epoch, T.spin_limit);
    STAGE_STAMP(4097);
}

// d = sum_a coef[a] * B_a (with xp <- x, gp <- g) is formed per bead by the next trial move: k_pack<.., DIR> in
// mmx_cells.hpp -- no stand-alone direction kernel.

__global__ __launch_bounds__(256) void k_copy4(int n4, const float4 *__restrict__ src, float4 *__restrict__ dst) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) dst[i] = src[i];
}

} // namespace mmx
