/*
 * mmx_oracle.c -- fp64 CPU restatement of the MultiMM force field + OpenMM minimizer semantics.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's shared object.
 * The product path (multimm_amd/ + libmmx.so) never imports, links or calls it.
 *
 * PARITY UNPINNED: the reference (/root/reference, SFGLab/MultiMM v2.0.2) holds no golden
 * energies/forces/coordinates for this path (its tests assert file existence only,
 * tests/test_simulations.py:23,41,59-63,83,103-105) and its arithmetic lives in the third-party
 * dependency OpenMM 8.5.1 (uv.lock:2462-2463) which is not installed here.  This file therefore
 * restates (a) the energy expressions and index sets written in the reference's
 * src/multimm/model.py and (b) the published semantics of the OpenMM primitives those
 * expressions are handed to (HarmonicBondForce / HarmonicAngleForce conventions, NoCutoff /
 * CutoffNonPeriodic plain truncation, LocalEnergyMinimizer = liblbfgs L-BFGS with backtracking
 * strong-Wolfe line search).  It is pinned instead by hand-derivable known-answer tests and by
 * central finite differences of its own energy (tests/test_oracle_*.py).
 *
 * Units: nm, kJ/mol, rad.  Force on bead i is F_i = -dE/dx_i.
 *
 * Reference call sites followed (all under /root/reference/src/multimm/):
 *   EV power law            model.py:164-217  (sigma = LE_HARMONIC_BOND_R0, model.py:175)
 *   compartment Gaussians   model.py:219-294 (COB), 296-384 (SCB)
 *   spherical container     model.py:453-466
 *   B-lamina "sin" shell    model.py:468-507
 *   central force           model.py:552-623 (harmonic form :579-586)
 *   chromosomal blocks      model.py:386-451 (polynomial form :416-419)
 *   backbone bonds          model.py:625-636
 *   loop bonds              model.py:638-659
 *   angles                  model.py:708-720
 *   term order / plain sum  model.py:812-857
 *   minimizeEnergy()        model.py:886  -> OpenMM LocalEnergyMinimizer (liblbfgs)
 *   Hilbert start           initial_structure_tools.py:157-166 -> hilbertcurve 2.0.5 (uv.lock:1129-1130)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "mmx_oracle.h"

int orc_n_terms(void) { return ORC_N_TERMS; }
int orc_sizeof_system(void) { return (int)sizeof(orc_system); }
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* OpenMP threads of the calls that follow (bench.py: the CPU share of the box, not its hardware thread count) */
void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Hilbert curve: hilbertcurve 2.0.5 HilbertCurve(p, n).point_from_distance (Skilling 2004,
 * "Programming the Hilbert curve"), called at initial_structure_tools.py:158-161 with p=8, n=3.
 * out[h*n + i] = coordinate i of distance h.
 * ------------------------------------------------------------------------------------------ */
void orc_hilbert_points(int64_t n_points, int p, int n, int32_t *out) {
    for (int64_t h = 0; h < n_points; ++h) {
        uint32_t x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        /* _hilbert_integer_to_transpose: MSB-first bit string of length p*n; x[i] takes bits i, i+n, ... */
        for (int b = 0; b < p * n; ++b) {
            int bit = (int)((h >> (p * n - 1 - b)) & 1);
            x[b % n] = (x[b % n] << 1) | (uint32_t)bit;
        }
        uint32_t z = 2u << (p - 1);
        /* Gray decode by H ^ (H/2) */
        uint32_t t = x[n - 1] >> 1;
        for (int i = n - 1; i > 0; --i) x[i] ^= x[i - 1];
        x[0] ^= t;
        /* undo excess work */
        for (uint32_t q = 2; q != z; q <<= 1) {
            uint32_t pm = q - 1;
            for (int i = n - 1; i >= 0; --i) {
                if (x[i] & q) {
                    x[0] ^= pm;
                } else {
                    t = (x[0] ^ x[i]) & pm;
                    x[0] ^= t;
                    x[i] ^= t;
                }
            }
        }
        for (int i = 0; i < n; ++i) out[h * n + i] = (int32_t)x[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * Backbone masks from chr_ends, reproducing the reference's index quirks exactly:
 *   bond  (i,i+1)     for i in [0,N-2] unless i in chr_ends                      model.py:628-629
 *   angle (i,i+1,i+2) for i in [0,N-3] unless i in chr_ends or i in chr_ends-1   model.py:711-712
 * ------------------------------------------------------------------------------------------ */
void orc_backbone_flags(int32_t n, const int32_t *chr_ends, int32_t n_ends, uint8_t *flags) {
    for (int32_t i = 0; i < n; ++i) {
        int in_ends = 0, in_ends_m1 = 0;
        for (int32_t k = 0; k < n_ends; ++k) {
            if (chr_ends[k] == i) in_ends = 1;
            if (chr_ends[k] - 1 == i) in_ends_m1 = 1;
        }
        uint8_t f = 0;
        if (i <= n - 2 && !in_ends) f |= 1;
        if (i <= n - 3 && !in_ends && !in_ends_m1) f |= 2;
        flags[i] = f;
    }
}

/* ------------------------------------------------------------------------------------------
 * Pair terms.  d = x_i - x_j, r = |d|.
 *   EV   (model.py:199): E = eps*(sigma/(r+r_small))^p ; dE/dr = -p*E/(r+r_small)
 *   Gauss(model.py:246-250, 322-328): E = -Eab*exp(-r^2/(2 rc^2)) ; dE/dr = +Eab*(r/rc^2)*exp(.)
 * Returns fscale with F_i += fscale*d, F_j -= fscale*d.  Plain truncation at the per-term
 * cutoff (OpenMM CutoffNonPeriodic: pairs with r >= cutoff are skipped, no shift).
 * At r == 0 the direction is undefined: energy is counted, force is zero.
 * ------------------------------------------------------------------------------------------ */
/* One attraction term in the form `form` (0 gaussian, 1 yukawa, 2 theta) with amplitude A:
 *   gaussian  -A exp(-r^2/(2 rc^2))          dE/dr = +A r/rc^2 exp(.)
 *   yukawa    -A exp(-r/lambda)/r, lambda = r_comp (model.py:270,352)   dE/dr = A exp(-r/l) (1/(l r) + 1/r^2)
 *   theta     -A step(rc - r), step(0) = 1 [upstream: OpenMM step(x) = 0 if x < 0, 1 otherwise]; no force
 * Coincident beads (r == 0) get no yukawa term (the reference's expression is -inf there). */
static inline double comp_term(int form, double A, double rc, double r2, double r, double *fs) {
    if (A == 0.0) return 0.0;
    if (form == 0) {
        double inv = 1.0 / (rc * rc);
        double g = A * exp(-0.5 * r2 * inv);
        *fs -= g * inv;
        return -g;
    }
    if (form == 1) {
        if (!(r > 0.0)) return 0.0;
        double y = A * exp(-r / rc) / r;
        *fs -= y * (1.0 / rc + 1.0 / r) / r;
        return -y;
    }
    return r <= rc ? -A : 0.0;
}

/* `i_lower`: bead i has the lower index of the pair.  Only the COB yukawa amplitude needs it: the reference's
 * expression reads s1 twice (model.py:266-267), so the amplitude depends on ONE bead; OpenMM's Reference platform
 * evaluates pairs with particle 1 = the lower index, which is what is restated here. */
static inline double pair_terms(const orc_system *s, double r2, int li, int lj, int i_lower, double *e_ev, double *e_g) {
    double fs = 0.0;
    double r = sqrt(r2);
    if (s->use_ev && (s->ev_cutoff <= 0.0 || r < s->ev_cutoff)) {
        if (s->ev_form == 0) {
            double u = 1.0 / (r + s->ev_rsmall);
            double E = s->ev_eps * pow(s->ev_sigma * u, s->ev_power);
            *e_ev += E;
            if (r > 0.0) fs += s->ev_power * E * u / r;
        } else { /* gaussian_core: eps*exp(-r^2/(2 sigma^2)), model.py:205-209 */
            double inv = 1.0 / (s->ev_sigma * s->ev_sigma);
            double E = s->ev_eps * exp(-0.5 * r2 * inv);
            *e_ev += E;
            fs += E * inv;
        }
    }
    if (s->use_gauss && (s->gauss_cutoff <= 0.0 || r < s->gauss_cutoff)) {
        if (s->cob_form == 0 && s->scb_form == 0) {
            *e_g += comp_term(0, s->gauss_table[li * 5 + lj], s->gauss_rc, r2, r, &fs);
        } else {
            if (s->has_cob) {
                int l1 = i_lower ? li : lj;
                double A = s->cob_form == 1 ? s->tab_cob[l1 * 5 + l1] : s->tab_cob[li * 5 + lj];
                *e_g += comp_term(s->cob_form, A, s->gauss_rc, r2, r, &fs);
            }
            if (s->has_scb) *e_g += comp_term(s->scb_form, s->tab_scb[li * 5 + lj], s->gauss_rc, r2, r, &fs);
        }
    }
    return fs;
}

/* All pairs, each bead sums over every other bead (pairs visited twice, energies halved):
 * embarrassingly parallel and summation order independent of the thread count. */
static void nonbonded_allpairs(const orc_system *s, const double *x, double *F, double *eterms) {
    const int n = s->n;
    double e_ev = 0.0, e_g = 0.0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : e_ev, e_g)
    for (int i = 0; i < n; ++i) {
        const int li = s->labels ? s->labels[i] + 2 : 2;
        double fx = 0, fy = 0, fz = 0, ev = 0, eg = 0;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
            double r2 = dx * dx + dy * dy + dz * dz;
            const int lj = s->labels ? s->labels[j] + 2 : 2;
            double fs = pair_terms(s, r2, li, lj, i < j, &ev, &eg);
            fx += fs * dx;
            fy += fs * dy;
            fz += fs * dz;
        }
        F[3 * i] += fx;
        F[3 * i + 1] += fy;
        F[3 * i + 2] += fz;
        e_ev += 0.5 * ev;
        e_g += 0.5 * eg;
    }
    eterms[ORC_T_EV] += e_ev;
    eterms[ORC_T_GAUSS] += e_g;
}

/* Cell list with edge >= max cutoff, 27-cell stencil.  Same per-bead full-shell summation. */
static int nonbonded_cells(const orc_system *s, const double *x, double *F, double *eterms) {
    const int n = s->n;
    double rc = 0.0;
    if (s->use_ev) rc = s->ev_cutoff > rc ? s->ev_cutoff : rc;
    if (s->use_gauss) rc = s->gauss_cutoff > rc ? s->gauss_cutoff : rc;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            if (x[3 * i + k] < lo[k]) lo[k] = x[3 * i + k];
            if (x[3 * i + k] > hi[k]) hi[k] = x[3 * i + k];
        }
    double h = rc;
    int64_t nc[3];
    for (;;) {
        for (int k = 0; k < 3; ++k) nc[k] = (int64_t)floor((hi[k] - lo[k]) / h) + 1;
        if (nc[0] * nc[1] * nc[2] <= 16 * 1024 * 1024) break;
        h *= 2.0;
    }
    const int64_t ncell = nc[0] * nc[1] * nc[2];
    int32_t *cell_of = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *start = (int32_t *)calloc((size_t)ncell + 1, sizeof(int32_t));
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    if (!cell_of || !start || !order) {
        free(cell_of);
        free(start);
        free(order);
        return -1;
    }
    for (int i = 0; i < n; ++i) {
        int64_t c[3];
        for (int k = 0; k < 3; ++k) {
            c[k] = (int64_t)floor((x[3 * i + k] - lo[k]) / h);
            if (c[k] < 0) c[k] = 0;
            if (c[k] >= nc[k]) c[k] = nc[k] - 1;
        }
        cell_of[i] = (int32_t)((c[2] * nc[1] + c[1]) * nc[0] + c[0]);
        start[cell_of[i] + 1]++;
    }
    for (int64_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
    {
        int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncell);
        memcpy(cur, start, sizeof(int32_t) * (size_t)ncell);
        for (int i = 0; i < n; ++i) order[cur[cell_of[i]]++] = i; /* stable: ascending bead id per cell */
        free(cur);
    }
    double e_ev = 0.0, e_g = 0.0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : e_ev, e_g)
    for (int i = 0; i < n; ++i) {
        const int li = s->labels ? s->labels[i] + 2 : 2;
        int64_t ci = cell_of[i];
        int64_t cx = ci % nc[0], cy = (ci / nc[0]) % nc[1], cz = ci / (nc[0] * nc[1]);
        double fx = 0, fy = 0, fz = 0, ev = 0, eg = 0;
        for (int64_t zz = cz - 1; zz <= cz + 1; ++zz) {
            if (zz < 0 || zz >= nc[2]) continue;
            for (int64_t yy = cy - 1; yy <= cy + 1; ++yy) {
                if (yy < 0 || yy >= nc[1]) continue;
                int64_t x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= nc[0] ? nc[0] - 1 : cx + 1;
                int64_t row = (zz * nc[1] + yy) * nc[0];
                for (int32_t q = start[row + x0]; q < start[row + x1 + 1]; ++q) {
                    int j = order[q];
                    if (j == i) continue;
                    double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1],
                           dz = x[3 * i + 2] - x[3 * j + 2];
                    double r2 = dx * dx + dy * dy + dz * dz;
                    if (r2 >= rc * rc) continue;
                    const int lj = s->labels ? s->labels[j] + 2 : 2;
                    double fs = pair_terms(s, r2, li, lj, i < j, &ev, &eg);
                    fx += fs * dx;
                    fy += fs * dy;
                    fz += fs * dz;
                }
            }
        }
        F[3 * i] += fx;
        F[3 * i + 1] += fy;
        F[3 * i + 2] += fz;
        e_ev += 0.5 * ev;
        e_g += 0.5 * eg;
    }
    eterms[ORC_T_EV] += e_ev;
    eterms[ORC_T_GAUSS] += e_g;
    free(cell_of);
    free(start);
    free(order);
    return 0;
}

/* Chromosomal blocks (model.py:416-419): E = dE*(k_C r^4 - r^3 + r^2) for every pair of beads with the
 * same chrom_spin, no cutoff (the potential grows with r).  dE/dr = dE*(4 k_C r^3 - 3 r^2 + 2 r), so
 * F_i = -dE*(4 k_C r^2 - 3 r + 2) * d with d = x_i - x_j (no division by r). */
static void chromosomal_blocks(const orc_system *s, const double *x, double *F, double *eterms) {
    const int n = s->n;
    double e = 0.0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : e)
    for (int i = 0; i < n; ++i) {
        double fx = 0, fy = 0, fz = 0, ei = 0;
        const int ci = s->chrom_of[i];
        for (int j = 0; j < n; ++j) {
            if (j == i || s->chrom_of[j] != ci) continue;
            double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
            double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2);
            double fs;
            if (s->chb_form == 0) {
                ei += s->chb_de * r2 * (s->chb_kc * r2 - r + 1.0);
                fs = -s->chb_de * (4.0 * s->chb_kc * r2 - 3.0 * r + 2.0);
            } else if (s->chb_form == 1) { /* -dE exp(-k_C r^2), model.py:428-431 */
                double ex = s->chb_de * exp(-s->chb_kc * r2);
                ei -= ex;
                fs = -2.0 * s->chb_kc * ex;
            } else { /* -dE/(1 + k_C r^2), model.py:440-443 */
                double den = 1.0 / (1.0 + s->chb_kc * r2);
                ei -= s->chb_de * den;
                fs = -2.0 * s->chb_kc * s->chb_de * den * den;
            }
            fx += fs * dx;
            fy += fs * dy;
            fz += fs * dz;
        }
        F[3 * i] += fx;
        F[3 * i + 1] += fy;
        F[3 * i + 2] += fz;
        e += 0.5 * ei;
    }
    eterms[ORC_T_CHB] += e;
}

/* HarmonicBondForce: E = 1/2 k (r-r0)^2 (OpenMM convention).  F_i = -k (r-r0) d/r, d = x_i - x_j. */
static inline double harmonic_pair(const double *x, double *F, int i, int j, double r0, double k) {
    double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    double r = sqrt(dx * dx + dy * dy + dz * dz);
    double dr = r - r0;
    if (r > 0.0) {
        double fs = -k * dr / r;
        F[3 * i] += fs * dx;
        F[3 * i + 1] += fs * dy;
        F[3 * i + 2] += fs * dz;
        F[3 * j] -= fs * dx;
        F[3 * j + 1] -= fs * dy;
        F[3 * j + 2] -= fs * dz;
    }
    return 0.5 * k * dr * dr;
}

/* Loop restraint in the alternative forms of add_loops (CustomBondForce, no 1/2):
 *   1 fene_soft        k u^2/(1 + alpha u^2), alpha = 1/r0^2        model.py:664-682
 *   2 gaussian_tether  k (1 - exp(-u^2/sigma^2)), sigma = r0/2      model.py:685-701      (u = r - r0) */
static inline double loop_pair(int form, const double *x, double *F, int i, int j, double r0, double k) {
    if (form == 0) return harmonic_pair(x, F, i, j, r0, k);
    double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
    double r = sqrt(dx * dx + dy * dy + dz * dz);
    double u = r - r0, E, dEdu;
    if (form == 1) {
        double alpha = 1.0 / (r0 * r0), den = 1.0 / (1.0 + alpha * u * u);
        E = k * u * u * den;
        dEdu = 2.0 * k * u * den * den;
    } else {
        double sigma = 0.5 * r0, ex = exp(-u * u / (sigma * sigma));
        E = k * (1.0 - ex);
        dEdu = 2.0 * k * u / (sigma * sigma) * ex;
    }
    if (r > 0.0) {
        double fs = -dEdu / r;
        F[3 * i] += fs * dx;
        F[3 * i + 1] += fs * dy;
        F[3 * i + 2] += fs * dz;
        F[3 * j] -= fs * dx;
        F[3 * j + 1] -= fs * dy;
        F[3 * j + 2] -= fs * dz;
    }
    return E;
}

/* HarmonicAngleForce: E = 1/2 k (theta-theta0)^2, theta at the middle bead j of (i,j,k).
 * Force form of OpenMM's angle kernels: a = x_i-x_j, b = x_k-x_j, c = a x b, |c| clamped to >= 1e-6,
 *   F_i = -dEdtheta * (a x c)/(|a|^2 |c|),  F_k = -dEdtheta * (c x b)/(|b|^2 |c|),  F_j = -(F_i+F_k).
 * (signs fixed by the finite-difference test, tests/test_oracle.py) */
static inline double harmonic_angle(const double *x, double *F, int i, int j, int k, double th0, double kk) {
    double a[3], b[3], c[3];
    for (int q = 0; q < 3; ++q) {
        a[q] = x[3 * i + q] - x[3 * j + q];
        b[q] = x[3 * k + q] - x[3 * j + q];
    }
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
    double cn = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    double dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    double theta = atan2(cn, dot);
    double rp = cn < 1e-6 ? 1e-6 : cn;
    double aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    double bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    double dEdth = kk * (theta - th0);
    if (aa > 0.0 && bb > 0.0) {
        double ta = -dEdth / (aa * rp), tc = -dEdth / (bb * rp);
        double fi[3], fk[3];
        fi[0] = ta * (a[1] * c[2] - a[2] * c[1]);
        fi[1] = ta * (a[2] * c[0] - a[0] * c[2]);
        fi[2] = ta * (a[0] * c[1] - a[1] * c[0]);
        fk[0] = tc * (c[1] * b[2] - c[2] * b[1]);
        fk[1] = tc * (c[2] * b[0] - c[0] * b[2]);
        fk[2] = tc * (c[0] * b[1] - c[1] * b[0]);
        for (int q = 0; q < 3; ++q) {
            F[3 * i + q] += fi[q];
            F[3 * k + q] += fk[q];
            F[3 * j + q] -= fi[q] + fk[q];
        }
    }
    return 0.5 * kk * (theta - th0) * (theta - th0);
}

/* Total evaluation.  F[3n] (may be NULL) receives the total force, eterms[ORC_N_TERMS] the per-term
 * energies in the order of model.py:812-857 collapsed onto the ORC_T_* slots.  Returns 0 / -1. */
int orc_eval(const orc_system *s, const double *x, double *F_out, double *eterms) {
    const int n = s->n;
    double *F = F_out ? F_out : (double *)malloc(sizeof(double) * 3 * (size_t)n);
    if (!F) return -1;
    memset(F, 0, sizeof(double) * 3 * (size_t)n);
    for (int t = 0; t < ORC_N_TERMS; ++t) eterms[t] = 0.0;

    if (s->use_ev || s->use_gauss) {
        int allpairs = (s->use_ev && s->ev_cutoff <= 0.0) || (s->use_gauss && s->gauss_cutoff <= 0.0);
        if (allpairs)
            nonbonded_allpairs(s, x, F, eterms);
        else if (nonbonded_cells(s, x, F, eterms) != 0) {
            if (!F_out) free(F);
            return -1;
        }
    }
    if (s->use_chb && s->chrom_of) chromosomal_blocks(s, x, F, eterms);
    if (s->bb_flags) {
        if (s->use_bond)
            for (int i = 0; i + 1 < n; ++i)
                if (s->bb_flags[i] & 1) eterms[ORC_T_BOND] += harmonic_pair(x, F, i, i + 1, s->bond_r0, s->bond_k);
        if (s->use_angle)
            for (int i = 0; i + 2 < n; ++i)
                if (s->bb_flags[i] & 2)
                    eterms[ORC_T_ANGLE] += harmonic_angle(x, F, i, i + 1, i + 2, s->angle_theta0, s->angle_k);
    }
    for (int l = 0; l < s->n_loops; ++l)
        eterms[ORC_T_LOOP] += loop_pair(s->loop_form, x, F, s->loop_m[l], s->loop_n[l], s->loop_r0[l], s->loop_k);

    if (s->use_container || s->use_lamina || s->use_central) {
        for (int i = 0; i < n; ++i) {
            double d[3] = {x[3 * i] - s->centre[0], x[3 * i + 1] - s->centre[1], x[3 * i + 2] - s->centre[2]};
            double r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            double dEdr = 0.0;
            if (s->use_container) { /* model.py:454-456 */
                double o = r - s->sc_R2 > 0 ? r - s->sc_R2 : 0.0, in = s->sc_R1 - r > 0 ? s->sc_R1 - r : 0.0;
                eterms[ORC_T_CONTAINER] += s->sc_C * (o * o + in * in);
                dEdr += 2.0 * s->sc_C * (o - in);
            }
            if (s->use_lamina && s->labels && s->labels[i] < 0) {
                const double span = s->ibl_R2 - s->ibl_R1;
                if (s->lam_form == 0) { /* sin^8 shell, model.py:503-505 */
                    double w = M_PI / span;
                    double u = w * (r - s->ibl_R1);
                    double sn = sin(u), cs = cos(u);
                    double s2 = sn * sn, s4 = s2 * s2;
                    eterms[ORC_T_LAMINA] += s->ibl_B * (s4 * s4 - 1.0);
                    dEdr += s->ibl_B * 8.0 * s4 * s2 * sn * cs * w;
                } else if (s->lam_form == 1) { /* gaussian_shell, sigma = 0.1 (R2-R1), model.py:511-518 */
                    double sg = 0.1 * span, a = r - s->ibl_R1, b = r - s->ibl_R2;
                    double e1 = exp(-a * a / (2 * sg * sg)), e2 = exp(-b * b / (2 * sg * sg));
                    eterms[ORC_T_LAMINA] += -s->ibl_B * (e1 + e2);
                    dEdr += s->ibl_B * (a * e1 + b * e2) / (sg * sg);
                } else if (s->lam_form == 2) { /* harmonic_shell, r0 = (R1+R2)/2, model.py:521-528 */
                    double u = r - 0.5 * (s->ibl_R1 + s->ibl_R2);
                    eterms[ORC_T_LAMINA] += s->ibl_B * u * u;
                    dEdr += 2.0 * s->ibl_B * u;
                } else { /* logistic_shell, lambda = 0.05 (R2-R1), model.py:531-539 */
                    double lam = 0.05 * span;
                    double a = 1.0 / (1.0 + exp((r - s->ibl_R2) / lam)), b = 1.0 / (1.0 + exp(-(r - s->ibl_R1) / lam));
                    eterms[ORC_T_LAMINA] += -s->ibl_B * (a + b);
                    dEdr += -s->ibl_B * (b * (1.0 - b) - a * (1.0 - a)) / lam;
                }
            }
            if (s->use_central && s->cf_w) {
                const double gw = s->cf_G * s->cf_w[i];
                if (s->cf_form == 0) { /* harmonic, model.py:584-586 */
                    double q = r - s->cf_R1;
                    eterms[ORC_T_CENTRAL] += gw * q * q;
                    dEdr += 2.0 * gw * q;
                } else if (s->cf_form == 1) { /* gaussian, sigma = R1/2, model.py:591-599 */
                    double sg = 0.5 * s->cf_R1, e1 = exp(-r * r / (2 * sg * sg));
                    eterms[ORC_T_CENTRAL] += -gw * e1;
                    dEdr += gw * r / (sg * sg) * e1;
                } else { /* logistic, lambda = 0.2 R1, model.py:604-612 */
                    double lam = 0.2 * s->cf_R1, a = 1.0 / (1.0 + exp((r - s->cf_R1) / lam));
                    eterms[ORC_T_CENTRAL] += -gw * a;
                    dEdr += gw * a * (1.0 - a) / lam;
                }
            }
            if (r > 0.0)
                for (int q = 0; q < 3; ++q) F[3 * i + q] -= dEdr * d[q] / r;
        }
    }
    if (!F_out) free(F);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Minimizer: OpenMM LocalEnergyMinimizer::minimize(context, tolerance, maxIterations) as reached
 * from model.py:886, i.e. liblbfgs lbfgs() with m = 6, linesearch =
 * LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE, ftol = 1e-4, wolfe = 0.9, max_linesearch = 40,
 * min_step = 1e-20, max_step = 1e20, epsilon = tolerance / max(1, sqrt(mean_i |x_i|^2)).
 * No constraints in this system, so OpenMM's restraint outer loop runs once.
 * f = total potential energy, gradient = -F.
 * ------------------------------------------------------------------------------------------ */
static double dotn(const double *a, const double *b, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

static double eval_fg(const orc_system *s, const double *x, double *g, double *F, int *nev) {
    double et[ORC_N_TERMS];
    orc_eval(s, x, F, et);
    const int64_t n3 = 3 * (int64_t)s->n;
    for (int64_t i = 0; i < n3; ++i) g[i] = -F[i];
    double f = 0.0;
    for (int t = 0; t < ORC_N_TERMS; ++t) f += et[t];
    ++*nev;
    return f;
}

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

int orc_minimize(const orc_system *s, double *x, double tolerance, int max_iterations, orc_min_stats *st) {
    enum { M = 6, MAX_LS = 40 };
    const double ftol = 1e-4, wolfe = 0.9, min_step = 1e-20, max_step = 1e20;
    const int64_t n3 = 3 * (int64_t)s->n;
    double t0 = now_s();
    double *buf = (double *)malloc(sizeof(double) * (size_t)n3 * (5 + 2 * M));
    if (!buf) return -1;
    double *g = buf, *F = g + n3, *d = F + n3, *xp = d + n3, *gp = xp + n3;
    double *S = gp + n3, *Y = S + (int64_t)M * n3;
    double ys_h[M], alpha[M];
    int nev = 0, status = 1;

    double norm = dotn(x, x, n3) / (double)s->n;
    norm = norm < 1.0 ? 1.0 : sqrt(norm);
    const double epsilon = tolerance / norm;

    double fx = eval_fg(s, x, g, F, &nev);
    st->e_initial = fx;
    for (int64_t i = 0; i < n3; ++i) d[i] = -g[i];
    double xnorm = sqrt(dotn(x, x, n3)), gnorm = sqrt(dotn(g, g, n3));
    if (xnorm < 1.0) xnorm = 1.0;
    int k = 1, end = 0, iters = 0;
    if (gnorm / xnorm <= epsilon) {
        status = 0;
        goto done;
    }
    double step = 1.0 / sqrt(dotn(d, d, n3));
    for (;;) {
        memcpy(xp, x, sizeof(double) * (size_t)n3);
        memcpy(gp, g, sizeof(double) * (size_t)n3);
        /* line_search_backtracking (strong Wolfe) */
        int ls = 0, count = 0;
        {
            const double dec = 0.5, inc = 2.1;
            double dginit = dotn(g, d, n3);
            if (step <= 0.0) ls = -1;
            else if (dginit > 0.0) ls = -2; /* LBFGSERR_INCREASEGRADIENT */
            else {
                double finit = fx, dgtest = ftol * dginit, width;
                for (;;) {
                    for (int64_t i = 0; i < n3; ++i) x[i] = xp[i] + step * d[i];
                    fx = eval_fg(s, x, g, F, &nev);
                    ++count;
                    if (fx > finit + step * dgtest) {
                        width = dec;
                    } else {
                        double dg = dotn(g, d, n3);
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
        if (ls < 0) { /* revert to the previous point, as liblbfgs does */
            memcpy(x, xp, sizeof(double) * (size_t)n3);
            memcpy(g, gp, sizeof(double) * (size_t)n3);
            fx = eval_fg(s, x, g, F, &nev);
            --nev; /* bookkeeping re-evaluation, not part of the algorithm */
            status = ls;
            break;
        }
        ++iters;
        xnorm = sqrt(dotn(x, x, n3));
        gnorm = sqrt(dotn(g, g, n3));
        if (xnorm < 1.0) xnorm = 1.0;
        if (gnorm / xnorm <= epsilon) { status = 0; break; }
        if (max_iterations != 0 && max_iterations < k + 1) { status = 1; break; }
        double *sk = S + (int64_t)end * n3, *yk = Y + (int64_t)end * n3;
        for (int64_t i = 0; i < n3; ++i) {
            sk[i] = x[i] - xp[i];
            yk[i] = g[i] - gp[i];
        }
        double ys = dotn(yk, sk, n3), yy = dotn(yk, yk, n3);
        ys_h[end] = ys;
        int bound = M <= k ? M : k;
        ++k;
        end = (end + 1) % M;
        for (int64_t i = 0; i < n3; ++i) d[i] = -g[i];
        int j = end;
        for (int i = 0; i < bound; ++i) {
            j = (j + M - 1) % M;
            alpha[j] = dotn(S + (int64_t)j * n3, d, n3) / ys_h[j];
            const double *yj = Y + (int64_t)j * n3;
            for (int64_t q = 0; q < n3; ++q) d[q] -= alpha[j] * yj[q];
        }
        double sc = ys / yy;
        for (int64_t q = 0; q < n3; ++q) d[q] *= sc;
        for (int i = 0; i < bound; ++i) {
            double beta = dotn(Y + (int64_t)j * n3, d, n3) / ys_h[j];
            const double *sj = S + (int64_t)j * n3;
            for (int64_t q = 0; q < n3; ++q) d[q] += (alpha[j] - beta) * sj[q];
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
    st->seconds = now_s() - t0;
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Molecular dynamics: the integrators model.py:768-808 constructs, advanced by
 * simulation.step(n), model.py:931.  Restates OpenMM 8.5.1's leap-frog update rules for a system
 * without constraints [upstream: ReferenceStochasticDynamics / ReferenceVerletDynamics /
 * ReferenceBrownianDynamics]:
 *   langevin : v' = a v + (1-a)/gamma F/m + sqrt(kT (1-a^2)/m) N(0,1), a = exp(-gamma dt); x' = x + v' dt
 *   verlet   : v' = v + dt F/m;                                                            x' = x + v' dt
 *   brownian : x' = x + dt/(gamma m) F + sqrt(2 kT dt/(gamma m)) N(0,1);                   v' = (x'-x)/dt
 * Noise: Philox4x32-10 (Salmon et al., SC'11) with key = seed and counter = {bead, step_lo, step_hi,
 * stream}; three normals per block by Box-Muller on 24-bit uniforms.  OpenMM's own generator is not
 * reproduced: MD parity with OpenMM would be statistical only; the HIP path is compared with THIS
 * restatement, which uses the same generator.
 * ------------------------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_normal3(uint32_t bead, uint64_t step, uint32_t stream, uint64_t seed, double z[3]) {
    const uint32_t ctr[4] = {bead, (uint32_t)step, (uint32_t)(step >> 32), stream};
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    const double s = 1.0 / 16777216.0;
    const double u1 = ((double)(r[0] >> 8) + 0.5) * s, u2 = ((double)(r[1] >> 8) + 0.5) * s;
    const double u3 = ((double)(r[2] >> 8) + 0.5) * s, u4 = ((double)(r[3] >> 8) + 0.5) * s;
    const double ra = sqrt(-2.0 * log(u1)), rb = sqrt(-2.0 * log(u3));
    z[0] = ra * cos(2.0 * M_PI * u2);
    z[1] = ra * sin(2.0 * M_PI * u2);
    z[2] = rb * cos(2.0 * M_PI * u4);
}

#define ORC_KB 0.008314462618 /* kJ/(mol K), model.py:967 */

/* context.setVelocitiesToTemperature(T, seed), model.py:878 (stream 1, step 0). */
void orc_md_velocities(int32_t n, double temperature, double mass, uint64_t seed, double *v) {
    const double sigma = sqrt(ORC_KB * temperature / mass);
    for (int i = 0; i < n; ++i) {
        double z[3];
        orc_normal3((uint32_t)i, 0, 1, seed, z);
        for (int q = 0; q < 3; ++q) v[3 * i + q] = sigma * z[q];
    }
}

typedef struct {
    int64_t step_count;
    double potential, kinetic, temperature;
    double eterms[ORC_N_TERMS];
} orc_md_stats;

/* kind: 0 langevin, 1 verlet, 2 brownian.  Advances (x, v) by n_steps starting at step index step0 and
 * reports the energies at the final point the way OpenMM does (kinetic energy of velocities shifted by
 * half a step for the leap-frog integrators, unshifted for brownian).  Returns 0 / -1. */
/* kind 3 = mm.amd.AMDIntegrator(dt, alpha, E) (model.py:794-800; OpenMM app amd.py [upstream]): a CustomIntegrator
 *   v <- v + dt f'/m,  f' = f ((1 - modify) + modify (alpha / (alpha + E - energy))^2),  modify = step(E - energy)
 *   x <- x + dt v
 * with `energy` the potential energy of the current positions; its kinetic energy is the CustomIntegrator default
 * m v^2 / 2 (no half-step shift). */
static int md_step_impl(const orc_system *s, int kind, double dt, double temperature, double friction, double mass,
                        double amd_alpha, double amd_e, uint64_t seed, int64_t step0, int32_t n_steps, double *x,
                        double *v, orc_md_stats *st) {
    const int n = s->n;
    double *F = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    if (!F) return -1;
    double et[ORC_N_TERMS];
    const double kT = ORC_KB * temperature;
    const double a = exp(-friction * dt);
    double fscale, noise;
    if (kind == 0) {
        fscale = (friction > 0.0 ? (1.0 - a) / friction : dt) / mass;
        noise = sqrt(kT * (1.0 - a * a) / mass);
    } else if (kind == 1 || kind == 3) {
        fscale = dt / mass;
        noise = 0.0;
    } else {
        fscale = dt / (friction * mass);
        noise = sqrt(2.0 * kT * dt / (friction * mass));
    }
    if (orc_eval(s, x, F, et) != 0) { free(F); return -1; }
    for (int32_t k = 0; k < n_steps; ++k) {
        const uint64_t step = (uint64_t)(step0 + k);
        double boost = 1.0;
        if (kind == 3) {
            double u = 0.0;
            for (int t = 0; t < ORC_N_TERMS; ++t) u += et[t];
            if (amd_e - u >= 0.0) {
                const double r = amd_alpha / (amd_alpha + amd_e - u);
                boost = r * r;
            }
        }
        for (int i = 0; i < n; ++i) {
            double z[3] = {0.0, 0.0, 0.0};
            if (kind == 0 || kind == 2) orc_normal3((uint32_t)i, step, 0, seed, z);
            for (int q = 0; q < 3; ++q) {
                const int64_t c = 3 * (int64_t)i + q;
                if (kind == 0) {
                    v[c] = a * v[c] + fscale * F[c] + noise * z[q];
                    x[c] += v[c] * dt;
                } else if (kind == 1) {
                    v[c] += fscale * F[c];
                    x[c] += v[c] * dt;
                } else if (kind == 3) {
                    v[c] += fscale * boost * F[c];
                    x[c] += v[c] * dt;
                } else {
                    const double dx = fscale * F[c] + noise * z[q];
                    x[c] += dx;
                    v[c] = dx / dt;
                }
            }
        }
        if (orc_eval(s, x, F, et) != 0) { free(F); return -1; }
    }
    if (st) {
        const double shift = (kind == 2 || kind == 3) ? 0.0 : 0.5 * dt;
        double ke = 0.0, pot = 0.0;
        for (int64_t c = 0; c < 3 * (int64_t)n; ++c) {
            const double w = v[c] + shift * F[c] / mass;
            ke += w * w;
        }
        ke *= 0.5 * mass;
        for (int t = 0; t < ORC_N_TERMS; ++t) {
            st->eterms[t] = et[t];
            pot += et[t];
        }
        st->step_count = step0 + n_steps;
        st->potential = pot;
        st->kinetic = ke;
        st->temperature = 2.0 * ke / (3.0 * (double)n * ORC_KB);
    }
    free(F);
    return 0;
}

int orc_md_step(const orc_system *s, int kind, double dt, double temperature, double friction, double mass,
                uint64_t seed, int64_t step0, int32_t n_steps, double *x, double *v, orc_md_stats *st) {
    if (kind < 0 || kind > 2) return -1;
    return md_step_impl(s, kind, dt, temperature, friction, mass, 0.0, 0.0, seed, step0, n_steps, x, v, st);
}

int orc_md_step_amd(const orc_system *s, double dt, double mass, double amd_alpha, double amd_e, int64_t step0,
                    int32_t n_steps, double *x, double *v, orc_md_stats *st) {
    return md_step_impl(s, 3, dt, 0.0, 0.0, mass, amd_alpha, amd_e, 0, step0, n_steps, x, v, st);
}
