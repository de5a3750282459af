/* mmx_cpu_fast_sweep.c -- the inner loop of the tuned CPU baseline (mmx_cpu_fast.c; TEST INFRASTRUCTURE ONLY), compiled
 * THREE times by oracle/Makefile with -DSWEEP_SUFFIX=_512 / _256 / _base and the matching -m flags: the shared object is
 * built in one container and runs on another machine's host cores, so mmx_cpu_fast.c picks an instance by CPUID at run
 * time.  (Per-function target attributes and target_clones were tried first: gcc 11 fixes the vector length of the loop
 * from the translation unit's baseline -- 128 bits -- and an AVX-512 host ran 4.3 x slower than with this.) */
#include <math.h>
#include <stdint.h>
#ifndef SWEEP_SUFFIX
#define SWEEP_SUFFIX _base
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

/* exp(x) for x <= 0 in plain arithmetic the compiler can vectorise (gcc 11 does not reach libmvec's vector expf from a
 * target_clones function): 2^(x log2 e) = 2^n * 2^f, f in [0, 1), 2^f by a degree-6 polynomial (relative error 2e-7, below
 * fp32 resolution of the sums it enters), 2^n through the exponent bits; anything below 2^-126 is zero. */
static inline float fast_exp(float x) {
    float y = x * 1.44269504088896341f;
    y = y < -126.f ? -126.f : y;
    const float n = floorf(y), f = y - n;
    float p = 1.535336188319500e-4f;
    p = p * f + 1.339887440266574e-3f;
    p = p * f + 9.618437357674640e-3f;
    p = p * f + 5.550332471162809e-2f;
    p = p * f + 2.402264791363012e-1f;
    p = p * f + 6.931472028550421e-1f;
    p = p * f + 1.0f;
    union { int32_t i; float f; } u;
    u.i = ((int32_t)n + 127) << 23;
    return y <= -126.f ? 0.f : p * u.f;
}

/* One i bead against the contiguous sorted range [q0, q1): returns through the accumulators.  The loop the compiler
 * vectorises (gather of the amplitude row by the neighbour's label, vector rsqrt/div/expf). */
static inline __attribute__((always_inline)) void sweep_body(const float *restrict sx, const float *restrict sy, const float *restrict sz, const int32_t *restrict sl,
                        int q0, int q1, int iq, float xi, float yi, float zi, const float *restrict row, float rc2, float rs,
                        float ev_c, float ev_p, float sigma, const int pow6, float g_c, float g_inv, const int use_ev, const int use_g,
                        float *restrict afx, float *restrict afy, float *restrict afz, float *restrict aev, float *restrict aeg) {
    (void)iq;
    float fx = 0.f, fy = 0.f, fz = 0.f, ev = 0.f, eg = 0.f;
    /* (no `#pragma omp simd`: gcc fixes the simd length of such a loop from the translation unit's baseline target -- 128
     * bits -- whatever the function's own target attribute says; -O3 -ffast-math vectorises the reductions by itself) */
    for (int q = q0; q < q1; ++q) {
        const float dx = xi - sx[q], dy = yi - sy[q], dz = zi - sz[q];
        const float r2 = dx * dx + dy * dy + dz * dz;
        const float in = r2 < rc2 ? 1.f : 0.f; /* (the bead itself is not in the range: the caller splits the range around it) */
        const float r2c = r2 > 1e-20f ? r2 : 1e-20f;
        const float rinv = 1.0f / sqrtf(r2c);
        const float r = r2c * rinv;
        float fs = 0.f;
        if (use_ev) {
            const float u = 1.0f / (r + rs);
            float E;
            if (pow6) {
                const float su = sigma * u, s2 = su * su;
                E = ev_c * s2 * s2 * s2;
            } else {
                E = ev_c * powf(sigma * u, ev_p);
            }
            ev += E * in;
            fs += ev_p * E * u * rinv;
        }
        if (use_g) {
            /* amplitude of (label of i, label of j): five selects instead of a gather (gcc 11 refuses the gather and with it
             * the whole loop) */
            const float g = row[sl[q]] * fast_exp(r2 * g_c);
            eg -= g * in;
            fs -= g * g_inv;
        }
        fs *= in;
        fx += fs * dx;
        fy += fs * dy;
        fz += fs * dz;
    }
    *afx += fx; *afy += fy; *afz += fz; *aev += ev; *aeg += eg;
}
/* ---- explicit SIMD for the instances the benchmark runs (EV with power 6 and / or the Gaussians): gcc 11's vectoriser gives
 * up on this loop (select next to integer operations, reciprocal square root), so the 512- and 256-bit builds spell it out.
 * Same arithmetic as sweep_body: rsqrt / rcp estimates refined by one Newton step (relative error ~1e-7), the amplitude of
 * (label of i, label of j) by a register permute of the i bead's table row, the cutoff as a 0 / 1 factor. */
#if defined(__AVX512F__) || defined(__AVX2__)
#include <immintrin.h>
#if defined(__AVX512F__)
typedef __m512 vf;
typedef __m512i vi;
#define VW 16
#define V1(x) _mm512_set1_ps(x)
#define VLD(p) _mm512_loadu_ps(p)
#define VLDI(p) _mm512_loadu_si512((const void *)(p))
#define VADD _mm512_add_ps
#define VSUB _mm512_sub_ps
#define VMUL _mm512_mul_ps
#define VFMA _mm512_fmadd_ps   /* a * b + c */
#define VFNMA _mm512_fnmadd_ps /* c - a * b */
#define VMAX _mm512_max_ps
#define VMIN _mm512_min_ps
#define VRSQRT(x) _mm512_rsqrt14_ps(x)
#define VRCP(x) _mm512_rcp14_ps(x)
#define VFLOOR(x) _mm512_roundscale_ps(x, _MM_FROUND_TO_NEG_INF | _MM_FROUND_NO_EXC)
#define VEXP2I(n) _mm512_castsi512_ps(_mm512_slli_epi32(_mm512_add_epi32(_mm512_cvtps_epi32(n), _mm512_set1_epi32(127)), 23))
#define VPERM(row, idx) _mm512_permutexvar_ps(idx, row)
#define VROW(p) _mm512_maskz_loadu_ps(0x1f, p)
#define VHSUM(v) _mm512_reduce_add_ps(v)
#else
typedef __m256 vf;
typedef __m256i vi;
#define VW 8
#define V1(x) _mm256_set1_ps(x)
#define VLD(p) _mm256_loadu_ps(p)
#define VLDI(p) _mm256_loadu_si256((const __m256i *)(p))
#define VADD _mm256_add_ps
#define VSUB _mm256_sub_ps
#define VMUL _mm256_mul_ps
#define VFMA _mm256_fmadd_ps
#define VFNMA _mm256_fnmadd_ps
#define VMAX _mm256_max_ps
#define VMIN _mm256_min_ps
#define VRSQRT(x) _mm256_rsqrt_ps(x)
#define VRCP(x) _mm256_rcp_ps(x)
#define VFLOOR(x) _mm256_floor_ps(x)
#define VEXP2I(n) _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_add_epi32(_mm256_cvtps_epi32(n), _mm256_set1_epi32(127)), 23))
#define VPERM(row, idx) _mm256_permutevar8x32_ps(row, idx)
static inline vf vrow8(const float *p) { /* 5 table entries + 3 zeros */
    float t[8] = {p[0], p[1], p[2], p[3], p[4], 0.f, 0.f, 0.f};
    return _mm256_loadu_ps(t);
}
#define VROW(p) vrow8(p)
static inline float vhsum8(__m256 v) {
    __m128 a = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
    a = _mm_add_ps(a, _mm_movehl_ps(a, a));
    a = _mm_add_ss(a, _mm_shuffle_ps(a, a, 1));
    return _mm_cvtss_f32(a);
}
#define VHSUM(v) vhsum8(v)
#endif
#define HAVE_SIMD_SWEEP 1
static inline __attribute__((always_inline)) void sweep_simd(const float *restrict sx, const float *restrict sy, const float *restrict sz,
                                                           const int32_t *restrict sl, int q0, int q1, float xi, float yi, float zi,
                                                           const float *restrict row, float rc2, float rs, float ev_c, float ev_p,
                                                           float sigma, float g_c, float g_inv, const int use_ev, const int use_g,
                                                           float *restrict afx, float *restrict afy, float *restrict afz,
                                                           float *restrict aev, float *restrict aeg) {
    const vf vxi = V1(xi), vyi = V1(yi), vzi = V1(zi), vrc2 = V1(rc2), vrs = V1(rs), one = V1(1.f), zero = V1(0.f);
    const vf vrow = VROW(row);
    vf fx = zero, fy = zero, fz = zero, ev = zero, eg = zero;
    int q = q0;
    for (; q + VW <= q1; q += VW) {
        const vf dx = VSUB(vxi, VLD(sx + q)), dy = VSUB(vyi, VLD(sy + q)), dz = VSUB(vzi, VLD(sz + q));
        const vf r2 = VFMA(dx, dx, VFMA(dy, dy, VMUL(dz, dz)));
        const vf in = VMIN(VMAX(VMUL(VSUB(vrc2, r2), V1(1e30f)), zero), one);
        const vf r2c = VMAX(r2, V1(1e-20f));
        vf rinv = VRSQRT(r2c);
        rinv = VMUL(rinv, VFNMA(VMUL(V1(0.5f), r2c), VMUL(rinv, rinv), V1(1.5f))); /* Newton: y (1.5 - 0.5 x y^2) */
        const vf r = VMUL(r2c, rinv);
        vf fs = zero;
        if (use_ev) {
            const vf den = VADD(r, vrs);
            vf u = VRCP(den);
            u = VMUL(u, VFNMA(den, u, V1(2.f))); /* Newton: u (2 - den u) */
            const vf su = VMUL(V1(sigma), u), s2 = VMUL(su, su);
            const vf E = VMUL(V1(ev_c), VMUL(VMUL(s2, s2), s2));
            ev = VFMA(E, in, ev);
            fs = VMUL(VMUL(V1(ev_p), E), VMUL(u, rinv));
        }
        if (use_g) {
            vf y = VMAX(VMUL(r2, V1(g_c * 1.44269504088896341f)), V1(-126.f));
            const vf n = VFLOOR(y), f = VSUB(y, n);
            vf p = V1(1.535336188319500e-4f);
            p = VFMA(p, f, V1(1.339887440266574e-3f));
            p = VFMA(p, f, V1(9.618437357674640e-3f));
            p = VFMA(p, f, V1(5.550332471162809e-2f));
            p = VFMA(p, f, V1(2.402264791363012e-1f));
            p = VFMA(p, f, V1(6.931472028550421e-1f));
            p = VFMA(p, f, one);
            const vf g = VMUL(VPERM(vrow, VLDI(sl + q)), VMUL(p, VEXP2I(n)));
            eg = VFNMA(g, in, eg);
            fs = VFNMA(g, V1(g_inv), fs);
        }
        fs = VMUL(fs, in);
        fx = VFMA(fs, dx, fx);
        fy = VFMA(fs, dy, fy);
        fz = VFMA(fs, dz, fz);
    }
    *afx += VHSUM(fx);
    *afy += VHSUM(fy);
    *afz += VHSUM(fz);
    *aev += VHSUM(ev);
    *aeg += VHSUM(eg);
    /* the remainder of the range: the scalar body (int labels) */
    float tfx = 0.f, tfy = 0.f, tfz = 0.f, tev = 0.f, teg = 0.f;
    for (; q < q1; ++q) {
        const float dx = xi - sx[q], dy = yi - sy[q], dz = zi - sz[q];
        const float r2 = dx * dx + dy * dy + dz * dz;
        const float in = r2 < rc2 ? 1.f : 0.f;
        const float r2c = r2 > 1e-20f ? r2 : 1e-20f;
        const float rinv = 1.0f / sqrtf(r2c), r = r2c * rinv;
        float fs = 0.f;
        if (use_ev) {
            const float u = 1.0f / (r + rs), su = sigma * u, s2 = su * su, E = ev_c * s2 * s2 * s2;
            tev += E * in;
            fs += ev_p * E * u * rinv;
        }
        if (use_g) {
            const float g = row[sl[q]] * fast_exp(r2 * g_c);
            teg -= g * in;
            fs -= g * g_inv;
        }
        fs *= in;
        tfx += fs * dx;
        tfy += fs * dy;
        tfz += fs * dz;
    }
    *afx += tfx; *afy += tfy; *afz += tfz; *aev += tev; *aeg += teg;
}
#else
#define HAVE_SIMD_SWEEP 0
#endif

/* One CELL: every bead [b0, b1) of the sorted arrays against the `nr` contiguous neighbour ranges [q0[k], q1[k]) of the
 * cell's stencil; writes the beads' pair forces, returns the cell's energy sums (each pair counted from both sides). */
#define CELL_ARGS const float *restrict sx, const float *restrict sy, const float *restrict sz, const int32_t *restrict sl,     \
                  int b0, int b1, int nr, const int *restrict q0, const int *restrict q1, const float *restrict tab, float rc2,  \
                  float rs, float ev_c, float ev_p, float sigma, float g_c, float g_inv, float *restrict ofx,                     \
                  float *restrict ofy, float *restrict ofz, double *restrict e_ev, double *restrict e_g
#if HAVE_SIMD_SWEEP
#define SWEEP_ONE(P6, EV, GA, A, B)                                                                            \
    do {                                                                                                       \
        if ((P6) || !(EV))                                                                                     \
            sweep_simd(sx, sy, sz, sl, A, B, xi, yi, zi, row, rc2, rs, ev_c, ev_p, sigma, g_c, g_inv, EV, GA, &fx, &fy, &fz, &ev, &eg); \
        else                                                                                                   \
            sweep_body(sx, sy, sz, sl, A, B, iq, xi, yi, zi, row, rc2, rs, ev_c, ev_p, sigma, P6, g_c, g_inv, EV, GA, &fx, &fy, &fz, &ev, &eg); \
    } while (0)
#else
#define SWEEP_ONE(P6, EV, GA, A, B) \
    sweep_body(sx, sy, sz, sl, A, B, iq, xi, yi, zi, row, rc2, rs, ev_c, ev_p, sigma, P6, g_c, g_inv, EV, GA, &fx, &fy, &fz, &ev, &eg)
#endif
#define CELL_BODY(P6, EV, GA)                                                                                  \
    double sev = 0.0, seg = 0.0;                                                                               \
    for (int iq = b0; iq < b1; ++iq) {                                                                         \
        const float xi = sx[iq], yi = sy[iq], zi = sz[iq];                                                     \
        const float *row = tab + 5 * sl[iq];                                                                   \
        float fx = 0.f, fy = 0.f, fz = 0.f, ev = 0.f, eg = 0.f;                                                \
        for (int k = 0; k < nr; ++k) {                                                                         \
            if (iq >= q0[k] && iq < q1[k]) { /* the bead's own row: around it */                               \
                SWEEP_ONE(P6, EV, GA, q0[k], iq);                                                              \
                SWEEP_ONE(P6, EV, GA, iq + 1, q1[k]);                                                          \
            } else                                                                                             \
                SWEEP_ONE(P6, EV, GA, q0[k], q1[k]);                                                           \
        }                                                                                                      \
        ofx[iq] = fx;                                                                                          \
        ofy[iq] = fy;                                                                                          \
        ofz[iq] = fz;                                                                                          \
        sev += (double)ev;                                                                                     \
        seg += (double)eg;                                                                                     \
    }                                                                                                          \
    *e_ev += sev;                                                                                              \
    *e_g += seg;
void CAT(orc_cell_ev6_g, SWEEP_SUFFIX)(CELL_ARGS) { CELL_BODY(1, 1, 1) }
void CAT(orc_cell_ev6, SWEEP_SUFFIX)(CELL_ARGS) { CELL_BODY(1, 1, 0) }
void CAT(orc_cell_g, SWEEP_SUFFIX)(CELL_ARGS) { CELL_BODY(0, 0, 1) }
void CAT(orc_cell_evp_g, SWEEP_SUFFIX)(CELL_ARGS) { CELL_BODY(0, 1, 1) } /* generic power: powf, scalar */
void CAT(orc_cell_evp, SWEEP_SUFFIX)(CELL_ARGS) { CELL_BODY(0, 1, 0) }
