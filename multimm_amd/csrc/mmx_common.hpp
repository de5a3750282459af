// mmx_common.hpp -- shared device/host structures and wave-level helpers (gfx950, wave64).
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MMX_M 6                    // L-BFGS history (liblbfgs default m, used by OpenMM)
#define MMX_NBASIS (2 * MMX_M + 1) // basis {S_0..S_5, Y_0..Y_5, g}
#define MMX_NROWS 3                // Gram rows recomputed per accepted iteration: s_new, y_new, g
// k_history's block partials: the MMX_NROWS x MMX_NBASIS Gram entries, then the two line-search reductions that are
// not Gram entries (g.g is): g.d and x.x.
#define MMX_ROW_GD (MMX_NROWS * MMX_NBASIS)
#define MMX_ROW_XX (MMX_NROWS * MMX_NBASIS + 1)
#define MMX_NROWSUM (MMX_NROWS * MMX_NBASIS + 2)

namespace mmx {

constexpr int kPartStride = 16384; // max per-slot block partials
constexpr int kWave = 64;

// Partial-sum slots (double, one value per block) written by the force kernels.
enum PartSlot {
    P_EV = 0, P_GAUSS, P_BOND, P_ANGLE, P_LOOP, P_CONT, P_LAM, P_CENT, P_CHB, // == MMX_T_* order
    P_GD, P_GG, P_XX,                                                        // g.d, g.g, x.x
    P_NSLOTS
};

// Minimizer phases (device-side state machine, one transition per evaluation).
// PH_HALT (decomposed runs): an evaluation found the ghost lists out of date; nothing was decided, every kernel of the
// evaluations still in the stream returns at once (phase >= PH_DONE), the host rebuilds the lists and repeats it.
enum Phase { PH_IDLE = 0, PH_INIT = 1, PH_LINESEARCH = 2, PH_DONE = 3, PH_HALT = 4 };

struct GridParams {
    float ox, oy, oz; // origin (bbox min)
    float h, inv_h;   // cell edge >= max cutoff
    int nx, ny, nz, ncells;
};

// Ownership of a decomposed run.  The beads are cut into SEGMENTS of kSeg consecutive beads (one backbone
// tile of k_backbone each, so the implicit bonds and angles stay inside a segment or reach two beads into its
// neighbours); a rank owns a set of segments, its L-BFGS vectors hold them in ascending segment order: local index
// li = (local segment) * kSeg + offset.  At the start a rank owns a contiguous run of segments (seg_own == nullptr:
// beads [lo, lo + n)); mmx_minimize re-assigns the segments to the ranks by recursive bisection of their centroids
// while the structure deforms (dd_reassign, mmx_engine.hpp), after which the two tables translate.
constexpr int kSeg = 62;
struct Own {
    int lo, n;            // identity mapping: owned beads [lo, lo + n); with tables: n = owned beads, lo unused
    int nseg;             // entries of seg_local (segments of the whole padded system)
    const int *seg_own;   // [owned segments] global segment id per local segment (nullptr: identity)
    const int *seg_local; // [nseg] local segment index of a global segment, -1: not owned
    __device__ __forceinline__ int bead(int li) const { // global bead id of local index li
        if (!seg_own) return lo + li;
        const int s = li / kSeg;
        return seg_own[s] * kSeg + (li - s * kSeg);
    }
    __device__ __forceinline__ int local(int b) const { // local index of global bead b, -1: not owned (or no bead: b < 0)
        if (!seg_local) {
            const int l = b - lo;
            return (unsigned)l < (unsigned)n ? l : -1;
        }
        const unsigned s = (unsigned)b / (unsigned)kSeg;
        if (s >= (unsigned)nseg) return -1;
        const int ls = seg_local[s];
        const int l = ls * kSeg + (b - (int)s * kSeg);
        return ls >= 0 && l < n ? l : -1;
    }
    __device__ __forceinline__ bool owns(int b) const { return local(b) >= 0; }
};

// Force-field constants, passed by value as kernel arguments.
struct FFParams {
    int n;                 // beads of the whole system
    int n_all;             // entries of pos4 (n padded to world * slice in a multi-GPU run)
    int own_lo, n_own;     // this handle owns n_own beads: forces / L-BFGS state are local (which ones: own())
    int nseg;              // decomposed runs after a re-assignment: the ownership tables (see Own), else nullptr
    const int *seg_own, *seg_local;
    __device__ __forceinline__ Own own() const { return Own{own_lo, n_own, nseg, seg_own, seg_local}; }
    int use_ev, ev_pmode; // pmode: 6 / 3 integer fast paths, 0 generic pow
    float ev_eps, ev_sigma, ev_rs, ev_power, ev_rc2; // rc2 = +inf when NoCutoff
    int use_gauss;
    float g_c2;      // -log2(e) / (2 rc^2)
    float g_inv_rc2; // 1 / rc^2
    float g_rc2;     // cutoff^2 (+inf when NoCutoff)
    float rc2max;    // max of the two cutoffs, squared
    float table[25]; // amplitude E(s_i+2, s_j+2)
    int use_bond, use_angle;
    float bond_r0, bond_k, ang_th0, ang_k;
    float loop_k;
    int use_container, use_lamina, use_central;
    float sc_C, sc_R1, sc_R2;
    float ibl_B, ibl_R1, ibl_R2;
    float cf_G, cf_R1;
    float cx, cy, cz;
    int use_chb;
    float chb_kc, chb_de;
};

// Alternative functional forms (model.py:173,229,305,395,479,557,648).  0 = the default form.
// Kept out of FFParams so that the default pair kernel's argument block (and its code) is unaffected.
struct FormParams {
    int generic_pairs;      // 1: some pair term uses a non-default form -> the FORMS instances of the pair kernels
    int ev_form;            // 1 gaussian_core
    float ev_gc2, ev_inv_s2; // gaussian core: -log2(e)/(2 sigma^2), 1/sigma^2
    int has_cob, has_scb, cob_form, scb_form; // 1 yukawa, 2 theta
    float tab_cob[25], tab_scb[25];           // separate amplitude tables (the default path uses their sum)
    float cob_a[5];         // COB yukawa: amplitude by the label of ONE bead (the expression reads s1 twice)
    float g_rcomp;          // r_comp: theta contact radius and yukawa screening length
    float g_yuk;            // -log2(e)/r_comp
    int lam_form;           // 1 gaussian_shell, 2 harmonic_shell, 3 logistic_shell
    int cf_form;            // 1 gaussian, 2 logistic
    int loop_form;          // 1 fene_soft, 2 gaussian_tether
    int chb_form;           // 1 gaussian, 2 saturating
};

// MinState::kernel_error bits
enum KernelError { KERR_N3_SPIN = 1,   // k_nb_n3: a wave waited for a unit / flush that never came (protocol bug)
                   KERR_N3_ITEMS = 2,  // k_nb_n3's item list is too short for this cell build
                   KERR_ORDER_DD = 4,  // decomposed rank: a cell too large for the in-cell sort would mix owned beads and ghosts
                   KERR_BUILD_WAIT = 8, // k_build: a stage waited for an earlier one that never finished (protocol bug)
                   KERR_BOUNDS = 16,   // a cell-build offset would have left its array (stale counters, corrupt state): nothing was written
                   KERR_TAIL_WAIT = 32 }; // k_tail: a workgroup's partial sums never arrived at the one that folds them (protocol bug)

// Device-resident minimizer state; mirrored to pinned host memory when polled.
struct MinState {
    int phase;
    int accepted;   // 1: the evaluation just controlled ended an iteration -> history/direction kernels run
    int store_hist; // 1: store (s,y) of the accepted step in slot `end`
    int status;
    int iters, evals, ls_count, k, end, bound, max_iters;
    int n_items;    // non-bonded work items of the last cell build
    int ncells, max_per_cell;
    int nan_seen;
    int n_clusters; // 8-bead clusters of the last cell build
    int n_clusters_own; // ... of them clusters of owned beads, listed FIRST when the build keeps the ghosts' clusters in a region
                        // of their own (decomposed ranks running the half-shell kernel: cell_scan_block, `split`); else = n_clusters
    int order_fallbacks; // cells too large for the in-LDS sort since the state was pushed (arrival order kept)
    int n_big;           // cells of > 64 beads in the last cell build (sorted by a whole block each)
    int n3_items;        // work items of the half-shell pair kernel (k_n3_items, after every cell scan)
    int n3_queue;        // ... and the head of their queue (persistent workgroups pull from it)
    int dd_stale;        // decomposed runs: an owned bead has moved more than half the skin since the ghost lists were built
    int halt_phase;      // phase to return to after PH_HALT
    int halt_reason;     // why (bit 0: a list went stale, bit 1: a list outgrew its message, bit 2: the kept cell structure went
                         // stale) -- bits 0-1 from the all-reduced flags, so the
                         // same on every rank; kept until the host has read it (the evaluations behind a halt zero the sums)
    int kernel_error;    // a force kernel could not do its work (KERR_*): the evaluation is void, the controller ends the
                         // call with status MMX_MIN_KERNEL instead of deciding on partial sums
    int dd_overflow;     // decomposed runs: a ghost list built on the stream outgrew its message (capacity known to the host)
    int cell_stale;      // kept cell structure (mmx_engine.hpp, "cell_reuse"): a bead has moved more than half the skin since it was
                         // binned -- the evaluation is void (PH_HALT, halt_reason bit 2) and is repeated after a full build
    int ncells_set[2];   // direct build (mmx_build.hpp): cells of the grid each of the two counter sets was last used with (what the
                         // build that zeroes the set for the next pack has to cover)
    unsigned dd_excess_bits; // decomposed ranks, direct build: how far (nm, float bits) an owned bead lies outside the build's grid box
                             // shrunk by the cutoff -- ghosts farther than that outside the grid box cannot be within the cutoff of
                             // any owned bead and are not binned (k_dd_unpack_count); reset by the build
    unsigned disp2_bits; // largest squared displacement of a bead from where it was binned, as float bits (non-negative floats
                         // order like unsigned ints), over the evaluations since the host last cleared it
    double fx;      // energy at the last accepted point
    double ftrial;  // energy of the last evaluation
    double finit, dginit, step, epsilon;
    double tolerance; // minimizer: epsilon = tolerance / max(1, rms |x_i|) is formed on the device from the first evaluation's x.x
    double n_total;   // ... beads of the whole system
    double f0;        // energy of the first evaluation of the call
    double gnorm, xnorm;
    double cell_edge;
    double eterms[9];       // per-term energies of the last evaluation
    double eterms_acc[9];   // ... at the last accepted point
    double ys[MMX_M];
    double gram[MMX_NBASIS * MMX_NBASIS];
    double coef[MMX_NBASIS];
    double sums[16];             // folded slot sums (all-reduced across ranks in a multi-GPU run)
    double rowsum[MMX_NROWSUM];  // folded k_history rows (same; directly behind sums: ONE all-reduce covers both)
    double dd_move;              // decomposed ranks: (this evaluation's largest squared trial move of an owned bead, nm^2)^8, summed over the
                                 // ranks by the same all-reduce: its 8th root is within 1.3 x of the maximum over the ranks, and the same number on every rank
    double dd_move2_max;         // ... its maximum over the evaluations since the host last cleared it: what the polls size the number of
                                 // evaluations a set of ghost lists may serve by (option dd_adaptive)
    unsigned dd_move2_bits;      // (this rank, this evaluation: float bits, k_pack)
    unsigned dd_pad_;
};
static_assert(offsetof(MinState, dd_move) == offsetof(MinState, rowsum) + MMX_NROWSUM * sizeof(double), "dd_move rides behind rowsum in the all-reduce");
static_assert(offsetof(MinState, rowsum) == offsetof(MinState, sums) + 16 * sizeof(double), "sums and rowsum are reduced as one array");

#ifdef MMX_STAGE_TIMING
// timing build (10 ns ticks): [workgroup * 4 + {0 start, 1 merge done, 2 loop done, 3 published}] of k_tail, [4096 ..] its folding
// workgroup; [4200 ..] k_build_direct: +0..3 workgroup 0 (start, grid, prefix, done), +4..7 the first order workgroup that takes
// small cells (start, grid, prefix, done), +8..11 the first one that takes large cells, +12..15 item workgroup 1
__device__ unsigned long long g_stage_t[8192];
// ... and of the direct builds (k_build_direct, k_build_direct_dd): [workgroup * 3 + {0 start, 1 row prefixes done, 2 end}], workgroups < 2730
__device__ unsigned long long g_stage_b[8192];
__device__ int g_stage_c[8192]; // [workgroup * 2 + {0 first cell of wave 0, 1 its population}]
#define BUILD_STAMP(k)                                                                                \
    do {                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 2730) g_stage_b[blockIdx.x * 3 + (k)] = wall_clock64();   \
    } while (0)
#define STAGE_STAMP(idx)                                                                              \
    do {                                                                                               \
        if (threadIdx.x == 0) g_stage_t[(idx)] = wall_clock64();                                       \
    } while (0)
#else
#define STAGE_STAMP(idx) do {} while (0)
#define BUILD_STAMP(k) do {} while (0)
#endif
// ---- wave helpers ---------------------------------------------------------------------------
// The xor butterfly (offsets 32, 16, 8, 4, 2, 1: every lane ends up with the wave's total, association fixed) WITHOUT the LDS
// crossbar: __shfl_xor is a ds_bpermute per 32 bits and step -- ~100 cycles each, a dependent chain of six (twelve for a double)
// per value, and the epilogue of a kernel that folds a dozen accumulators spent 4-8 us in them.  Here: lanes i <-> i ^ 32 and
// i <-> i ^ 16 by gfx950's v_permlane32_swap / v_permlane16_swap (a copy of the value is swapped half-wise / row-wise with the
// value itself: one of the two then holds the partner's value in every lane), i ^ 8 and i ^ 4 by DPP row_ror:8 / row_ror:4 (a
// rotation by 4 pairs lane i with i + 4 mod 16, whose value equals that of i ^ 4 once the i ^ 8 step has made the row 8-periodic),
// i ^ 2 and i ^ 1 by DPP quad_perm.  Same operand pairs in the same order as the shuffle version (addition, min and max commute):
// the same bits in every lane (scripts/ubench/wave_sum_check.hip).
// (after a swap of v with a copy of itself, the two results hold {own value, partner's value} in one order or the other: OP of
//  the two is OP(own, partner) whichever lane asks)
#define MMX_SWAP_STEP_I(OP, SWAP, v)                                                                        \
    do {                                                                                                    \
        const auto r_ = SWAP((v), (v), false, false);                                                       \
        (v) = OP((int)r_[0], (int)r_[1]); /* (the builtin returns unsigned) */                              \
    } while (0)
#define MMX_SWAP_STEP_F(OP, SWAP, v)                                                                        \
    do {                                                                                                    \
        const auto r_ = SWAP(__float_as_int(v), __float_as_int(v), false, false);                           \
        (v) = OP(__int_as_float(r_[0]), __int_as_float(r_[1]));                                             \
    } while (0)
template <int CTRL>
__device__ __forceinline__ int dpp_partner(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_partner(float v) { return __int_as_float(dpp_partner<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp_partner(double v) {
    return __hiloint2double(dpp_partner<CTRL>(__double2hiint(v)), dpp_partner<CTRL>(__double2loint(v)));
}
// Value of lane i ^ J (J a power of two) without the LDS crossbar.  32 / 16: permlane swaps (see above; the lane's half / row
// decides which of the two results holds the partner), 8: DPP row_ror:8, 4: row_half_mirror (i -> i ^ 7 within 8 lanes) followed by
// quad_perm:[3,2,1,0] (i -> i ^ 3), 2 / 1: quad_perm.  Whole waves must call.
template <int J>
__device__ __forceinline__ int lane_xor(int v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "power of two below 64");
    if (J == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
    }
    if (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? (int)r[0] : (int)r[1];
    }
    if (J == 8) return dpp_partner<0x128>(v);
    if (J == 4) return dpp_partner<0x1b>(dpp_partner<0x141>(v)); // row_half_mirror, then quad_perm:[3,2,1,0]
    if (J == 2) return dpp_partner<0x4e>(v);
    return dpp_partner<0xb1>(v);
}
template <int J>
__device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor<J>(__float_as_int(v))); }
template <int J>
__device__ __forceinline__ unsigned long long lane_xor(unsigned long long v) {
    const unsigned lo = (unsigned)lane_xor<J>((int)(unsigned)(v & 0xffffffffull)), hi = (unsigned)lane_xor<J>((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// ... with the distance as a run-time value (wave-uniform): the switch folds away where the caller's loop is unrolled
template <class T>
__device__ __forceinline__ T lane_xor_rt(T v, int j) {
    switch (j) {
    case 32: return lane_xor<32>(v);
    case 16: return lane_xor<16>(v);
    case 8: return lane_xor<8>(v);
    case 4: return lane_xor<4>(v);
    case 2: return lane_xor<2>(v);
    default: return lane_xor<1>(v);
    }
}
__device__ __forceinline__ float mmx_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float wave_sum(float v) {
    MMX_SWAP_STEP_F(mmx_addf, __builtin_amdgcn_permlane32_swap, v);
    MMX_SWAP_STEP_F(mmx_addf, __builtin_amdgcn_permlane16_swap, v);
    v += dpp_partner<0x128>(v); // row_ror:8
    v += dpp_partner<0x124>(v); // row_ror:4
    v += dpp_partner<0x4e>(v);  // quad_perm:[2,3,0,1]
    v += dpp_partner<0xb1>(v);  // quad_perm:[1,0,3,2]
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    {
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
    {
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
    v += dpp_partner<0x128>(v);
    v += dpp_partner<0x124>(v);
    v += dpp_partner<0x4e>(v);
    v += dpp_partner<0xb1>(v);
    return v;
}
// Twelve wave totals for the price of three and a half: the first two butterfly steps are done PAIRWISE -- v_permlane32_swap of
// two different registers leaves {x[0:31], y[0:31]} and {x[32:63], y[32:63]}, whose sum is x's step in the lower half of the wave
// and y's in the upper one; v_permlane16_swap does the same with rows -- so twelve values shrink to six, then to three registers
// that hold four values each (one per row of 16 lanes); the steps within a row follow as in wave_sum.  Every addition has the
// operands wave_sum gives it (in one order or the other): the same bits.  out[j], lanes 16 r .. 16 r + 15: total of v[j + 3 r].
__device__ __forceinline__ double swap32_add(double x, double y) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap16_add(double x, double y) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void wave_sum12(const double (&v)[12], double (&out)[3]) {
    double h[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) h[k] = swap32_add(v[k], v[k + 6]); // lower half: v[k], upper half: v[k + 6]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double q = swap16_add(h[j], h[j + 3]); // rows 0..3: v[j], v[j + 3], v[j + 6], v[j + 9]
        q += dpp_partner<0x128>(q);
        q += dpp_partner<0x124>(q);
        q += dpp_partner<0x4e>(q);
        q += dpp_partner<0xb1>(q);
        out[j] = q;
    }
}
// Wave total by DPP (no LDS crossbar): butterfly inside each row of 16 lanes (quad_perm xor 1, xor 2, row_ror 4, 8),
// then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (gfx9 DPP controls).  The total is valid in
// lanes 48..63; the result is broadcast from lane 63 through a scalar register.  Fixed order => deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
    return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add<0xb1, 0xf>(v);  // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);  // quad_perm:[2,3,0,1]
    v = dpp_add<0x124, 0xf>(v); // row_ror:4
    v = dpp_add<0x128, 0xf>(v); // row_ror:8
    v = dpp_add<0x142, 0xa>(v); // row_bcast:15 -> rows 1, 3
    v = dpp_add<0x143, 0xc>(v); // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min(float v) {
    MMX_SWAP_STEP_F(fminf, __builtin_amdgcn_permlane32_swap, v);
    MMX_SWAP_STEP_F(fminf, __builtin_amdgcn_permlane16_swap, v);
    v = fminf(v, dpp_partner<0x128>(v));
    v = fminf(v, dpp_partner<0x124>(v));
    v = fminf(v, dpp_partner<0x4e>(v));
    v = fminf(v, dpp_partner<0xb1>(v));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    MMX_SWAP_STEP_F(fmaxf, __builtin_amdgcn_permlane32_swap, v);
    MMX_SWAP_STEP_F(fmaxf, __builtin_amdgcn_permlane16_swap, v);
    v = fmaxf(v, dpp_partner<0x128>(v));
    v = fmaxf(v, dpp_partner<0x124>(v));
    v = fmaxf(v, dpp_partner<0x4e>(v));
    v = fmaxf(v, dpp_partner<0xb1>(v));
    return v;
}
__device__ __forceinline__ int mmx_addi(int a, int b) { return a + b; }
__device__ __forceinline__ int wave_sum_i(int v) {
    MMX_SWAP_STEP_I(mmx_addi, __builtin_amdgcn_permlane32_swap, v);
    MMX_SWAP_STEP_I(mmx_addi, __builtin_amdgcn_permlane16_swap, v);
    v += dpp_partner<0x128>(v);
    v += dpp_partner<0x124>(v);
    v += dpp_partner<0x4e>(v);
    v += dpp_partner<0xb1>(v);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
    MMX_SWAP_STEP_I(max, __builtin_amdgcn_permlane32_swap, v);
    MMX_SWAP_STEP_I(max, __builtin_amdgcn_permlane16_swap, v);
    v = max(v, dpp_partner<0x128>(v));
    v = max(v, dpp_partner<0x124>(v));
    v = max(v, dpp_partner<0x4e>(v));
    v = max(v, dpp_partner<0xb1>(v));
    return v;
}

// Orders LDS traffic between lanes of ONE wave (no s_barrier needed: a wave's DS ops retire in order).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Orders LDS traffic between the WAVES of a workgroup around a flag in LDS that other waves poll: release before the
// flag is raised, acquire after it has been seen.  Workgroup scope, LDS only ("local": global loads and atomics in
// flight are not waited for) -- on gfx950 an s_waitcnt lgkmcnt(0), the same instruction the wavefront-scope fence
// emits, so the stronger scope costs nothing; what it adds is the guarantee the protocol relies on.
__device__ __forceinline__ void wg_lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void wg_lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }

// Block-wide sum of one double per thread; result valid on thread 0.  Fixed order => deterministic.
template <int BLOCK>
__device__ __forceinline__ double block_sum(double v, double *lds /* [BLOCK/64] */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) r += lds[w];
    }
    return r;
}

// Grid derived from a bounding box; every thread that calls it gets identical values.
__device__ __forceinline__ GridParams grid_from_box(float lox, float loy, float loz, float hix, float hiy,
                                                     float hiz, float hmin, int maxcells) {
    GridParams G;
    G.ox = lox;
    G.oy = loy;
    G.oz = loz;
    float ex = hix - lox, ey = hiy - loy, ez = hiz - loz;
    if (!(ex >= 0.f) || !(ex < 1e30f)) { ex = 0.f; G.ox = 0.f; } // empty / non-finite box: one cell
    if (!(ey >= 0.f) || !(ey < 1e30f)) { ey = 0.f; G.oy = 0.f; }
    if (!(ez >= 0.f) || !(ez < 1e30f)) { ez = 0.f; G.oz = 0.f; }
    float h = hmin;
    G.nx = G.ny = G.nz = 1;
    for (int it = 0; it < 200; ++it) {
        const float fx = floorf(ex / h) + 1.f, fy = floorf(ey / h) + 1.f, fz = floorf(ez / h) + 1.f;
        if (fx * fy * fz <= (float)maxcells) {
            G.nx = (int)fx;
            G.ny = (int)fy;
            G.nz = (int)fz;
            break;
        }
        h *= 1.25f;
    }
    G.h = h;
    G.inv_h = 1.0f / h;
    G.ncells = G.nx * G.ny * G.nz;
    return G;
}

// ---- coarse grid of a decomposed run's need-maps (mmx_dd.hpp; here because k_pack marks the occupancy on the way) ----
constexpr int kDDGridN = 64; // coarse cells per axis (x = the bit of a 64-bit word)
// Coarse grid of the need-maps (device; written by k_dd_grid at a synchronous rebuild, read by the rebuilds on the stream)
struct DDGrid {
    float ox, oy, oz, inv_edge, edge;
    int radius; // cells a map is grown by: ceil(reach / edge)
};

// coarse cell of a position: word index (z, y) and x bit; indices wrap, so any finite position has a cell
__device__ __forceinline__ void dd_cell(const DDGrid &G, float px, float py, float pz, int &word, int &bit) {
    const int cx = (int)floorf((px - G.ox) * G.inv_edge) & (kDDGridN - 1);
    const int cy = (int)floorf((py - G.oy) * G.inv_edge) & (kDDGridN - 1);
    const int cz = (int)floorf((pz - G.oz) * G.inv_edge) & (kDDGridN - 1);
    word = cz * kDDGridN + cy;
    bit = cx;
}

// occ |= the coarse cell of this lane's position.  Whole waves call; consecutive beads mostly share a cell: lanes of a wave with
// the same cell issue one atomic.
__device__ __forceinline__ void dd_mark(const DDGrid &G, unsigned long long *__restrict__ occ, bool act, float px, float py, float pz) {
    int word = 0, bit = 0;
    if (act) dd_cell(G, px, py, pz, word, bit);
    const int key = word * kDDGridN + bit;
    unsigned long long pending = __ballot(act);
    const int lane = threadIdx.x & 63;
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int k0 = __shfl(key, leader, 64);
        const unsigned long long same = __ballot(act && key == k0);
        if (lane == leader) atomicOr(&occ[word], 1ull << bit);
        pending &= ~same;
    }
}

__device__ __forceinline__ int cell_coord(float p, float o, float inv_h, int n) {
    int c = (int)floorf((p - o) * inv_h);
    return min(max(c, 0), n - 1);
}

} // namespace mmx
