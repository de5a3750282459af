/*
 * mmx_cpu_fast.c -- a TUNED CPU evaluation of the same force field with the same L-BFGS: the second cpu_baseline of
 * bench.py ("what a CPU platform does"), next to the plain fp64 restatement of mmx_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY (see mmx_oracle.c): loaded by tests/ and by bench.py's cpu_baseline leg, never by the product.
 *
 * Why it exists.  The reference runs its minimization on OpenMM's CPU platform (PLATFORM = CPU, model.py:863-873, "Threads"
 * property): single precision, SIMD pair kernels over a cell/neighbour structure, all cores.  OpenMM is not installable here
 * (uv.lock:2462-2463), and the fp64 restatement -- pow()/exp() per pair in double, scalar, a 27-cell sweep per bead -- is a
 * checker, not a contender: the GPU : restatement ratio says nothing about "x times a CPU platform".  This file is the
 * credible stand-in: fp32 arithmetic, cell-sorted SoA positions, cells of edge cutoff / 2 (a 5 x 5 x 5 stencil: 27 % of the
 * swept candidates lie inside the cutoff instead of 15 %), the inner loop over a row's contiguous neighbour range
 * vectorised by the compiler (AVX-512 / AVX2 / baseline instances picked at run time by CPUID, a polynomial exp), OpenMP over cells,
 * energies in fp64, the bonded and per-bead terms parallel as well, and the very L-BFGS of orc_minimize with parallel
 * vector operations.  Every pair is evaluated from both sides (full shell: no scatter, no per-thread force buffers);
 * OpenMM's CPU kernels use Newton's third law on a neighbour list with per-thread buffers, which halves the arithmetic
 * and adds the list build and the buffer reduction -- the figure this file produces is reported with its pairs/s per
 * core so that it can be judged.  Only the default functional forms (config.py:269-312 defaults) are implemented; any
 * other returns -2.  Checked against the fp64 restatement by tests/test_oracle.py::test_fast_cpu_baseline_*.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "mmx_oracle.h"

#if defined(__x86_64__) && defined(__GNUC__)
#define FAST_X86 1
#else
#define FAST_X86 0
#endif

static double fnow(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

typedef struct {
    int n;
    float *sx, *sy, *sz, *fx, *fy, *fz; /* cell-sorted positions, pair forces in the same order */
    int32_t *sl, *sid, *cell_of, *start, *cur;
    int64_t ncell_cap;
    float *x32;        /* [3n] positions as float */
    float *bond, *ang; /* [3n] gradient shares of the bond / [6n] of the angle a bead starts */
    double pairs;      /* lane-pairs swept by the last evaluation (candidates, both directions) */
} fast_ws;

static void ws_free(fast_ws *w) {
    free(w->sx); free(w->sy); free(w->sz); free(w->fx); free(w->fy); free(w->fz);
    free(w->sl); free(w->sid); free(w->cell_of); free(w->start); free(w->cur); free(w->x32); free(w->bond); free(w->ang);
    memset(w, 0, sizeof(*w));
}
static int ws_alloc(fast_ws *w, int n) {
    memset(w, 0, sizeof(*w));
    w->n = n;
    const size_t nf = (size_t)n + 64;
    w->sx = malloc(sizeof(float) * nf); w->sy = malloc(sizeof(float) * nf); w->sz = malloc(sizeof(float) * nf);
    w->fx = malloc(sizeof(float) * nf); w->fy = malloc(sizeof(float) * nf); w->fz = malloc(sizeof(float) * nf);
    w->sl = malloc(sizeof(int32_t) * nf); w->sid = malloc(sizeof(int32_t) * nf); w->cell_of = malloc(sizeof(int32_t) * nf);
    w->x32 = malloc(sizeof(float) * 3 * nf); w->bond = malloc(sizeof(float) * 3 * nf); w->ang = malloc(sizeof(float) * 6 * nf);
    if (!w->sx || !w->sy || !w->sz || !w->fx || !w->fy || !w->fz || !w->sl || !w->sid || !w->cell_of || !w->x32 || !w->bond || !w->ang) {
        ws_free(w);
        return -1;
    }
    return 0;
}

/* the inner loops live in mmx_cpu_fast_sweep.c, compiled once per instruction set (see there) */
#define CELL_ARGS const float *restrict sx, const float *restrict sy, const float *restrict sz, const int32_t *restrict sl,     \
                  int b0, int b1, int nr, const int *restrict q0, const int *restrict q1, const float *restrict tab, float rc2,  \
                  float rs, float ev_c, float ev_p, float sigma, float g_c, float g_inv, float *restrict ofx,                     \
                  float *restrict ofy, float *restrict ofz, double *restrict e_ev, double *restrict e_g
#define CELL_DECL(SUF)                                                                                      \
    void orc_cell_ev6_g##SUF(CELL_ARGS); void orc_cell_ev6##SUF(CELL_ARGS); void orc_cell_g##SUF(CELL_ARGS); \
    void orc_cell_evp_g##SUF(CELL_ARGS); void orc_cell_evp##SUF(CELL_ARGS);
CELL_DECL(_512)
CELL_DECL(_256)
CELL_DECL(_base)
typedef void (*cell_fn)(CELL_ARGS);
static int simd_level(void) { /* 2: AVX-512, 1: AVX2 + FMA, 0: baseline */
#if FAST_X86
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl") &&
        __builtin_cpu_supports("avx512bw"))
        return 2;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return 1;
#endif
    return 0;
}
int orc_fast_simd_level(void) { return simd_level(); }

/* Pair terms: cells of edge >= cutoff / 2, stencil radius R = 2 cells (the edge doubles -- and R halves -- only for
 * systems too sparse for the cell budget). */
static int fast_pairs(const orc_system *s, fast_ws *w, const float *x, double *e_ev_out, double *e_g_out) {
    const int n = s->n;
    float rc = 0.f;
    if (s->use_ev) rc = (float)s->ev_cutoff > rc ? (float)s->ev_cutoff : rc;
    if (s->use_gauss) rc = (float)s->gauss_cutoff > rc ? (float)s->gauss_cutoff : rc;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int k = 0; k < 3; ++k) {
        float l = 3e38f, h = -3e38f;
#pragma omp parallel for reduction(min : l) reduction(max : h)
        for (int i = 0; i < n; ++i) {
            const float v = x[3 * i + k];
            l = v < l ? v : l;
            h = v > h ? v : h;
        }
        lo[k] = l;
        hi[k] = h;
    }
    float h = 0.5f * rc * 1.0001f;
    int R = 2;
    int64_t nc[3];
    for (;;) {
        for (int k = 0; k < 3; ++k) nc[k] = (int64_t)floorf((hi[k] - lo[k]) / h) + 1;
        if (nc[0] * nc[1] * nc[2] <= 8 * 1024 * 1024) break;
        h *= 2.f;
        R = R > 1 ? R / 2 : 1; /* h >= rc from the second doubling on: one cell of margin is enough */
    }
    const int64_t ncell = nc[0] * nc[1] * nc[2];
    if (ncell + 1 > w->ncell_cap) {
        free(w->start);
        free(w->cur);
        w->start = malloc(sizeof(int32_t) * (size_t)(ncell + 1));
        w->cur = malloc(sizeof(int32_t) * (size_t)(ncell + 1));
        w->ncell_cap = ncell + 1;
        if (!w->start || !w->cur) return -1;
    }
    memset(w->start, 0, sizeof(int32_t) * (size_t)(ncell + 1));
    const float inv_h = 1.0f / h;
#pragma omp parallel for
    for (int i = 0; i < n; ++i) {
        int64_t c[3];
        for (int k = 0; k < 3; ++k) {
            c[k] = (int64_t)floorf((x[3 * i + k] - lo[k]) * inv_h);
            c[k] = c[k] < 0 ? 0 : c[k] >= nc[k] ? nc[k] - 1 : c[k];
        }
        w->cell_of[i] = (int32_t)((c[2] * nc[1] + c[1]) * nc[0] + c[0]);
    }
    for (int i = 0; i < n; ++i) w->start[w->cell_of[i] + 1]++;
    for (int64_t c = 0; c < ncell; ++c) w->start[c + 1] += w->start[c];
    memcpy(w->cur, w->start, sizeof(int32_t) * (size_t)ncell);
    for (int i = 0; i < n; ++i) {
        const int q = w->cur[w->cell_of[i]]++;
        w->sid[q] = i;
    }
#pragma omp parallel for
    for (int q = 0; q < n; ++q) {
        const int i = w->sid[q];
        w->sx[q] = x[3 * i];
        w->sy[q] = x[3 * i + 1];
        w->sz[q] = x[3 * i + 2];
        w->sl[q] = s->labels ? s->labels[i] + 2 : 2;
    }
    float tab[25];
    for (int k = 0; k < 25; ++k) tab[k] = (float)s->gauss_table[k];
    const float rc2 = rc * rc, rs = (float)s->ev_rsmall, sigma = (float)s->ev_sigma, ev_c = (float)s->ev_eps,
                ev_p = (float)s->ev_power, g_c = (float)(-0.5 / (s->gauss_rc * s->gauss_rc)),
                g_inv = (float)(1.0 / (s->gauss_rc * s->gauss_rc));
    const int pow6 = s->ev_power == 6.0;
    const int lvl = simd_level();
#define PICK(NAME) (lvl == 2 ? NAME##_512 : lvl == 1 ? NAME##_256 : NAME##_base)
    const cell_fn cell = s->use_ev ? (pow6 ? (s->use_gauss ? PICK(orc_cell_ev6_g) : PICK(orc_cell_ev6))
                                           : (s->use_gauss ? PICK(orc_cell_evp_g) : PICK(orc_cell_evp)))
                                   : PICK(orc_cell_g);
#undef PICK
    double e_ev = 0.0, e_g = 0.0, swept = 0.0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : e_ev, e_g, swept)
    for (int64_t c = 0; c < ncell; ++c) {
        const int b0 = w->start[c], b1 = w->start[c + 1];
        if (b0 == b1) continue;
        const int64_t cx = c % nc[0], cy = (c / nc[0]) % nc[1], cz = c / (nc[0] * nc[1]);
        const int64_t x0 = cx - R < 0 ? 0 : cx - R, x1 = cx + R >= nc[0] ? nc[0] - 1 : cx + R;
        int q0[32], q1[32], nr = 0; /* (2 R + 1)^2 <= 25 neighbour ranges */
        for (int64_t zz = cz - R; zz <= cz + R; ++zz) {
            if (zz < 0 || zz >= nc[2]) continue;
            for (int64_t yy = cy - R; yy <= cy + R; ++yy) {
                if (yy < 0 || yy >= nc[1]) continue;
                const int64_t rowc = (zz * nc[1] + yy) * nc[0];
                q0[nr] = w->start[rowc + x0];
                q1[nr] = w->start[rowc + x1 + 1];
                swept += (double)(q1[nr] - q0[nr]) * (double)(b1 - b0);
                if (q1[nr] > q0[nr]) ++nr;
            }
        }
        double cev = 0.0, ceg = 0.0;
        cell(w->sx, w->sy, w->sz, w->sl, b0, b1, nr, q0, q1, tab, rc2, rs, ev_c, ev_p, sigma, g_c, g_inv, w->fx, w->fy, w->fz, &cev, &ceg);
        e_ev += 0.5 * cev;
        e_g += 0.5 * ceg;
    }
    w->pairs = swept;
    *e_ev_out = e_ev;
    *e_g_out = e_g;
    return 0;
}

static int fast_supported(const orc_system *s) {
    if (s->ev_form || s->cob_form || s->scb_form || s->lam_form || s->cf_form || s->loop_form) return 0;
    if (s->use_chb) return 0;
    if ((s->use_ev && !(s->ev_cutoff > 0.0)) || (s->use_gauss && !(s->gauss_cutoff > 0.0))) return 0; /* cell list only */
    if (s->use_ev && s->use_gauss && s->ev_cutoff != s->gauss_cutoff) return 0;
    return 1;
}

/* F[3n] (double, receives the total force), eterms[ORC_N_TERMS].  x: [3n] double (rounded to float inside). */
static int fast_eval(const orc_system *s, fast_ws *w, const double *xd, double *F, double *et) {
    const int n = s->n;
    float *x = w->x32;
#pragma omp parallel for
    for (int i = 0; i < 3 * n; ++i) x[i] = (float)xd[i];
    for (int t = 0; t < ORC_N_TERMS; ++t) et[t] = 0.0;
    const int timing = getenv("MMX_FAST_TIMING") != NULL;
    double tt0 = fnow();
    if (s->use_ev || s->use_gauss) {
        if (fast_pairs(s, w, x, &et[ORC_T_EV], &et[ORC_T_GAUSS]) != 0) return -1;
        if (timing) fprintf(stderr, "[fast] pairs %.4f s (simd level %d)\n", fnow() - tt0, simd_level());
        tt0 = fnow();
#pragma omp parallel for
        for (int q = 0; q < n; ++q) {
            const int i = w->sid[q];
            F[3 * i] = (double)w->fx[q];
            F[3 * i + 1] = (double)w->fy[q];
            F[3 * i + 2] = (double)w->fz[q];
        }
    } else {
        memset(F, 0, sizeof(double) * 3 * (size_t)n);
    }
    /* backbone: every bond and angle once, by the bead that starts it, into per-bead shares (pass 1); pass 2 gathers */
    const int bb = s->bb_flags && (s->use_bond || s->use_angle);
    if (bb) {
        double eb = 0.0, ea = 0.0;
        const float r0 = (float)s->bond_r0, kb = (float)s->bond_k, th0 = (float)s->angle_theta0, ka = (float)s->angle_k;
#pragma omp parallel for reduction(+ : eb, ea)
        for (int i = 0; i < n; ++i) {
            float *b = w->bond + 3 * (size_t)i, *a = w->ang + 6 * (size_t)i;
            b[0] = b[1] = b[2] = 0.f;
            for (int k = 0; k < 6; ++k) a[k] = 0.f;
            const int f = s->bb_flags[i];
            if (s->use_bond && (f & 1) && i + 1 < n) {
                const float dx = x[3 * i] - x[3 * i + 3], dy = x[3 * i + 1] - x[3 * i + 4], dz = x[3 * i + 2] - x[3 * i + 5];
                const float r = sqrtf(dx * dx + dy * dy + dz * dz), dr = r - r0;
                if (r > 0.f) {
                    const float fs = -kb * dr / r; /* force on i = fs * d */
                    b[0] = fs * dx; b[1] = fs * dy; b[2] = fs * dz;
                }
                eb += 0.5 * (double)kb * dr * dr;
            }
            if (s->use_angle && (f & 2) && i + 2 < n) {
                float av[3], bv[3], cv[3];
                for (int q = 0; q < 3; ++q) {
                    av[q] = x[3 * i + q] - x[3 * i + 3 + q];
                    bv[q] = x[3 * i + 6 + q] - x[3 * i + 3 + q];
                }
                cv[0] = av[1] * bv[2] - av[2] * bv[1];
                cv[1] = av[2] * bv[0] - av[0] * bv[2];
                cv[2] = av[0] * bv[1] - av[1] * bv[0];
                const float cn = sqrtf(cv[0] * cv[0] + cv[1] * cv[1] + cv[2] * cv[2]);
                const float dt = av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2];
                const float th = atan2f(cn, dt), rp = cn < 1e-6f ? 1e-6f : cn;
                const float aa = av[0] * av[0] + av[1] * av[1] + av[2] * av[2], bbq = bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2];
                const float dE = ka * (th - th0);
                if (aa > 0.f && bbq > 0.f) {
                    const float ta = -dE / (aa * rp), tc = -dE / (bbq * rp);
                    a[0] = ta * (av[1] * cv[2] - av[2] * cv[1]);
                    a[1] = ta * (av[2] * cv[0] - av[0] * cv[2]);
                    a[2] = ta * (av[0] * cv[1] - av[1] * cv[0]);
                    a[3] = tc * (cv[1] * bv[2] - cv[2] * bv[1]);
                    a[4] = tc * (cv[2] * bv[0] - cv[0] * bv[2]);
                    a[5] = tc * (cv[0] * bv[1] - cv[1] * bv[0]);
                }
                ea += 0.5 * (double)ka * (th - th0) * (th - th0);
            }
        }
#pragma omp parallel for
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < 3; ++q) {
                double f = (double)w->bond[3 * (size_t)i + q] + (double)w->ang[6 * (size_t)i + q];
                if (i >= 1) f += -(double)w->bond[3 * (size_t)(i - 1) + q] - (double)w->ang[6 * (size_t)(i - 1) + q] - (double)w->ang[6 * (size_t)(i - 1) + 3 + q];
                if (i >= 2) f += (double)w->ang[6 * (size_t)(i - 2) + 3 + q];
                F[3 * i + q] += f;
            }
        et[ORC_T_BOND] = eb;
        et[ORC_T_ANGLE] = ea;
    }
    if (timing) fprintf(stderr, "[fast] unsort + backbone %.4f s\n", fnow() - tt0);
    tt0 = fnow();
    for (int l = 0; l < s->n_loops; ++l) { /* a few thousand: serial */
        const int i = s->loop_m[l], j = s->loop_n[l];
        const float dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
        const float r = sqrtf(dx * dx + dy * dy + dz * dz), dr = r - (float)s->loop_r0[l];
        if (r > 0.f) {
            const float fs = -(float)s->loop_k * dr / r;
            F[3 * i] += fs * dx; F[3 * i + 1] += fs * dy; F[3 * i + 2] += fs * dz;
            F[3 * j] -= fs * dx; F[3 * j + 1] -= fs * dy; F[3 * j + 2] -= fs * dz;
        }
        et[ORC_T_LOOP] += 0.5 * s->loop_k * (double)dr * (double)dr;
    }
    if (s->use_container || s->use_lamina || s->use_central) {
        double ec = 0.0, el = 0.0, ef = 0.0;
        const float cx = (float)s->centre[0], cy = (float)s->centre[1], cz = (float)s->centre[2];
#pragma omp parallel for reduction(+ : ec, el, ef)
        for (int i = 0; i < n; ++i) {
            const float dx = x[3 * i] - cx, dy = x[3 * i + 1] - cy, dz = x[3 * i + 2] - cz;
            const float r = sqrtf(dx * dx + dy * dy + dz * dz);
            float dEdr = 0.f;
            if (s->use_container) {
                const float o = r - (float)s->sc_R2 > 0.f ? r - (float)s->sc_R2 : 0.f, in = (float)s->sc_R1 - r > 0.f ? (float)s->sc_R1 - r : 0.f;
                ec += s->sc_C * (double)(o * o + in * in);
                dEdr += 2.f * (float)s->sc_C * (o - in);
            }
            if (s->use_lamina && s->labels && s->labels[i] < 0) {
                const float wv = 3.14159265358979f / (float)(s->ibl_R2 - s->ibl_R1), u = wv * (r - (float)s->ibl_R1);
                const float sn = sinf(u), cs = cosf(u), s2 = sn * sn, s4 = s2 * s2;
                el += s->ibl_B * (double)(s4 * s4 - 1.f);
                dEdr += (float)s->ibl_B * 8.f * s4 * s2 * sn * cs * wv;
            }
            if (s->use_central && s->cf_w) {
                const float gw = (float)(s->cf_G * s->cf_w[i]), q = r - (float)s->cf_R1;
                ef += (double)(gw * q * q);
                dEdr += 2.f * gw * q;
            }
            if (r > 0.f) {
                const float f = dEdr / r;
                F[3 * i] -= f * dx;
                F[3 * i + 1] -= f * dy;
                F[3 * i + 2] -= f * dz;
            }
        }
        et[ORC_T_CONTAINER] = ec;
        et[ORC_T_LAMINA] = el;
        et[ORC_T_CENTRAL] = ef;
    }
    if (timing) fprintf(stderr, "[fast] loops + confinement %.4f s\n", fnow() - tt0);
    return 0;
}

int orc_fast_eval(const orc_system *s, const double *x, double *F, double *eterms, double *lane_pairs) {
    if (!fast_supported(s)) return -2;
    fast_ws w;
    if (ws_alloc(&w, s->n)) return -1;
    const int rc = fast_eval(s, &w, x, F, eterms);
    if (lane_pairs) *lane_pairs = w.pairs;
    ws_free(&w);
    return rc;
}

static double pdot(const double *a, const double *b, int64_t n) {
    double r = 0.0;
#pragma omp parallel for simd reduction(+ : r)
    for (int64_t i = 0; i < n; ++i) r += a[i] * b[i];
    return r;
}

/* orc_minimize (mmx_oracle.c) with the fast evaluation and parallel vector operations: same liblbfgs control flow. */
int orc_fast_minimize(const orc_system *s, double *x, double tolerance, int max_iterations, orc_min_stats *st, double *lane_pairs) {
    enum { M = 6, MAX_LS = 40 };
    if (!fast_supported(s)) return -2;
    const double ftol = 1e-4, wolfe = 0.9, min_step = 1e-20, max_step = 1e20;
    const int64_t n3 = 3 * (int64_t)s->n;
    const double t0 = fnow();
    fast_ws w;
    if (ws_alloc(&w, s->n)) return -1;
    double *buf = (double *)malloc(sizeof(double) * (size_t)n3 * (5 + 2 * M));
    if (!buf) {
        ws_free(&w);
        return -1;
    }
    double *g = buf, *F = g + n3, *d = F + n3, *xp = d + n3, *gp = xp + n3, *S = gp + n3, *Y = S + (int64_t)M * n3;
    double ys_h[M], alpha[M], et[ORC_N_TERMS], swept = 0.0;
    int nev = 0, status = 1, k = 1, end = 0, iters = 0;
#define FAST_FG(fout)                                                     \
    do {                                                                  \
        fast_eval(s, &w, x, F, et);                                       \
        swept += w.pairs;                                                 \
        _Pragma("omp parallel for simd") for (int64_t i_ = 0; i_ < n3; ++i_) g[i_] = -F[i_]; \
        fout = 0.0;                                                       \
        for (int t_ = 0; t_ < ORC_N_TERMS; ++t_) fout += et[t_];          \
        ++nev;                                                            \
    } while (0)
    double norm = pdot(x, x, n3) / (double)s->n;
    norm = norm < 1.0 ? 1.0 : sqrt(norm);
    const double epsilon = tolerance / norm;
    double fx;
    FAST_FG(fx);
    st->e_initial = fx;
#pragma omp parallel for simd
    for (int64_t i = 0; i < n3; ++i) d[i] = -g[i];
    double xnorm = sqrt(pdot(x, x, n3)), gnorm = sqrt(pdot(g, g, n3));
    if (xnorm < 1.0) xnorm = 1.0;
    if (gnorm / xnorm <= epsilon) {
        status = 0;
        goto done;
    }
    double step = 1.0 / sqrt(pdot(d, d, n3));
    for (;;) {
        memcpy(xp, x, sizeof(double) * (size_t)n3);
        memcpy(gp, g, sizeof(double) * (size_t)n3);
        int ls = 0, count = 0;
        {
            const double dec = 0.5, inc = 2.1;
            const double dginit = pdot(g, d, n3);
            if (step <= 0.0) ls = -1;
            else if (dginit > 0.0) ls = -2;
            else {
                const double finit = fx, dgtest = ftol * dginit;
                double width;
                for (;;) {
#pragma omp parallel for simd
                    for (int64_t i = 0; i < n3; ++i) x[i] = xp[i] + step * d[i];
                    FAST_FG(fx);
                    ++count;
                    if (fx > finit + step * dgtest) width = dec;
                    else {
                        const double dg = pdot(g, d, n3);
                        if (dg < wolfe * dginit) width = inc;
                        else if (dg > -wolfe * dginit) width = dec;
                        else { ls = count; break; }
                    }
                    if (step < min_step) { ls = -3; break; }
                    if (step > max_step) { ls = -4; break; }
                    if (MAX_LS <= count) { ls = -5; break; }
                    step *= width;
                }
            }
        }
        if (ls < 0) {
            memcpy(x, xp, sizeof(double) * (size_t)n3);
            memcpy(g, gp, sizeof(double) * (size_t)n3);
            status = ls;
            break;
        }
        ++iters;
        xnorm = sqrt(pdot(x, x, n3));
        gnorm = sqrt(pdot(g, g, n3));
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= epsilon) { status = 0; break; }
        if (max_iterations != 0 && max_iterations < k + 1) { status = 1; break; }
        double *sk = S + (int64_t)end * n3, *yk = Y + (int64_t)end * n3;
#pragma omp parallel for simd
        for (int64_t i = 0; i < n3; ++i) {
            sk[i] = x[i] - xp[i];
            yk[i] = g[i] - gp[i];
        }
        const double ys = pdot(yk, sk, n3), yy = pdot(yk, yk, n3);
        ys_h[end] = ys;
        const int bound = M <= k ? M : k;
        ++k;
        end = (end + 1) % M;
#pragma omp parallel for simd
        for (int64_t i = 0; i < n3; ++i) d[i] = -g[i];
        int j = end;
        for (int i = 0; i < bound; ++i) {
            j = (j + M - 1) % M;
            alpha[j] = pdot(S + (int64_t)j * n3, d, n3) / ys_h[j];
            const double *yj = Y + (int64_t)j * n3, aj = alpha[j];
#pragma omp parallel for simd
            for (int64_t q = 0; q < n3; ++q) d[q] -= aj * yj[q];
        }
        const double sc = ys / yy;
#pragma omp parallel for simd
        for (int64_t q = 0; q < n3; ++q) d[q] *= sc;
        for (int i = 0; i < bound; ++i) {
            const double beta = pdot(Y + (int64_t)j * n3, d, n3) / ys_h[j];
            const double *sj = S + (int64_t)j * n3, cf = alpha[j] - beta;
#pragma omp parallel for simd
            for (int64_t q = 0; q < n3; ++q) d[q] += cf * sj[q];
            j = (j + 1) % M;
        }
        step = 1.0;
    }
done:
    st->iterations = iters;
    st->evaluations = nev;
    st->status = status;
    st->e_final = fx;
    st->gnorm_final = gnorm;
    st->xnorm_final = xnorm;
    st->seconds = fnow() - t0;
    if (lane_pairs) *lane_pairs = swept;
    free(buf);
    ws_free(&w);
    return 0;
#undef FAST_FG
}
