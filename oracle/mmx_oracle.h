/* mmx_oracle.h -- the structures shared by the two CPU sources under oracle/ (TEST INFRASTRUCTURE ONLY, see mmx_oracle.c):
 * mmx_oracle.c   the fp64 restatement (the checker of every parity test)
 * mmx_cpu_fast.c a tuned fp32 / SIMD / OpenMP evaluation of the same force field with the same L-BFGS: what a CPU platform
 *                does -- bench.py's second cpu_baseline entry, itself checked against the fp64 restatement. */
#pragma once
#include <stdint.h>

enum {
    ORC_T_EV = 0,
    ORC_T_GAUSS = 1,
    ORC_T_BOND = 2,
    ORC_T_ANGLE = 3,
    ORC_T_LOOP = 4,
    ORC_T_CONTAINER = 5,
    ORC_T_LAMINA = 6,
    ORC_T_CENTRAL = 7,
    ORC_T_CHB = 8,
    ORC_N_TERMS = 9
};

/* Mirrored field-for-field by oracle/oracle.py:OrcSystem (ctypes.Structure). */
typedef struct {
    int32_t n;
    const int8_t *labels;    /* [n] in {-2..2}, may be NULL (all 0) */
    const uint8_t *bb_flags; /* [n] bit0: bond (i,i+1) present; bit1: angle (i,i+1,i+2) present */
    int32_t use_bond, use_angle;
    double bond_r0, bond_k, angle_theta0, angle_k;
    int32_t n_loops;
    const int32_t *loop_m;
    const int32_t *loop_n;
    const double *loop_r0;
    double loop_k;
    int32_t use_ev;
    double ev_eps, ev_sigma, ev_rsmall, ev_power, ev_cutoff; /* cutoff <= 0: NoCutoff (all pairs) */
    int32_t use_gauss;
    double gauss_table[25]; /* amplitude E(s_i+2, s_j+2) >= 0; E_pair = -E*exp(-r^2/(2 rc^2)) */
    double gauss_rc, gauss_cutoff;
    int32_t use_container;
    double sc_C, sc_R1, sc_R2;
    int32_t use_lamina;
    double ibl_B, ibl_R1, ibl_R2;
    int32_t use_central;
    double cf_G, cf_R1;
    const double *cf_w; /* [n] chrom_strength */
    double centre[3];   /* mass_center, model.py:759 */
    int32_t use_chb;
    double chb_kc, chb_de;
    const int32_t *chrom_of; /* [n] chrom_spin: pairs interact iff equal (model.py:416-419) */
    /* alternative functional forms (the *_FORCE_TYPE keys, config.py:269-312); 0 = the default form */
    int32_t ev_form;              /* 1 gaussian_core                          model.py:205-209 */
    int32_t has_cob, has_scb;     /* which tables below are live (their sum is gauss_table) */
    int32_t cob_form, scb_form;   /* 1 yukawa, 2 theta                        model.py:262-288, 340-377 */
    double tab_cob[25], tab_scb[25];
    int32_t chb_form;             /* 1 gaussian, 2 saturating                 model.py:424-443 */
    int32_t lam_form;             /* 1 gaussian_shell, 2 harmonic_shell, 3 logistic_shell   model.py:508-539 */
    int32_t cf_form;              /* 1 gaussian, 2 logistic                   model.py:588-612 */
    int32_t loop_form;            /* 1 fene_soft, 2 gaussian_tether           model.py:662-701 */
} orc_system;

typedef struct {
    int32_t iterations;  /* accepted L-BFGS iterations (liblbfgs "progress" calls) */
    int32_t evaluations; /* energy+force evaluations */
    int32_t status;      /* 0 converged, 1 max iterations, <0 liblbfgs-style line-search error */
    double e_initial, e_final, gnorm_final, xnorm_final, seconds;
} orc_min_stats;
