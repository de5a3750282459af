// mmx_engine.hpp -- what the C ABI calls into: derived parameters, kernel selection, and the launch sequences of one
// evaluation (pack -> [cell scan + bonded pass] -> fill -> order -> pair kernel -> history -> decide), collectives of a
// decomposed run, profiling events.  Host code of libmmx.so; included by mmx_api.hip only.
#pragma once
#include "mmx_handle.hpp"

namespace {

#define HIPCHK(h, expr)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                                  \
            return MMX_ERR_HIP;                                                                            \
        }                                                                                                  \
    } while (0)

int fail(mmx_handle h, int code, const std::string &msg) {
    if (h) h->err = msg;
    return code;
}

template <class T>
hipError_t dalloc(T **p, size_t count) {
    hipError_t e = hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) e = hipMemset(*p, 0, std::max<size_t>(count, 1) * sizeof(T));
    // hipMemset on device memory is enqueued on the NULL stream and may return before it has run; the handle's stream is
    // non-blocking (it does not wait for the null stream), so a buffer allocated in the middle of a call -- the migration
    // staging area of dd_reassign -- could be zeroed AFTER the first copy into it (seen: 12 MB at 400 000 beads)
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

bool all_pairs(const mmx_handle_s *h) {
    return (h->P.use_ev && h->ev_cut <= 0.f) || (h->P.use_gauss && h->g_cut <= 0.f);
}
bool has_nb(const mmx_handle_s *h) { return h->P.use_ev || h->P.use_gauss; }

// cells of edge factor x cutoff: the measurement option, else what the polls chose (cell_edge_auto)
float edge_factor(const mmx_handle_s *h) {
    if (h->cell_edge_scale > 1.f) return h->cell_edge_scale;
    return h->cell_edge_auto ? std::max(h->edge_auto, 1.f) : 1.f;
}
float hmin_of(const mmx_handle_s *h) {
    float rc = 0.f;
    if (h->P.use_ev) rc = std::max(rc, h->ev_cut);
    if (h->P.use_gauss) rc = std::max(rc, h->g_cut);
    return rc * 1.001f * edge_factor(h);
}

// OPT template bits of the cluster-kernel instance that nb_variant selects (see launch_nb_cells_p).
int nb_launch_opt(const mmx_handle_s *h) {
    const int opt = ((h->nb_variant & 32) ? 0 : 2) | ((h->nb_variant & 64) ? 0 : 4) | ((h->nb_variant & 128) ? 0 : 8) |
                    ((h->nb_variant & 256) ? 16 : 0) | ((h->nb_variant & 512) ? 32 : 0) |
                    ((h->nb_variant & 1024) ? 64 : 0) | ((h->nb_variant & 2048) ? 128 : 0);
    if (h->nb_force_plain) return 10; // clamp mask + per-bead cull, per-bead energies: not LEAN
    if (opt == 142) return 142;
    switch (opt & 127) {
    case 14: case 30: case 46: case 22: case 78: case 62: case 6: case 12: case 10: return opt & 127;
    default: return 0;
    }
}

// Derived constants refreshed before every launch sequence.
void refresh_params(mmx_handle_s *h) {
    FFParams &P = h->P;
    P.n = h->n;
    P.n_all = h->n_all;
    P.own_lo = h->own_lo;
    P.n_own = h->n_own;
    P.nseg = h->nseg;
    P.seg_own = h->seg_owner.empty() ? nullptr : h->d_seg_own;     // nullptr while the ownership is the initial contiguous one
    P.seg_local = h->seg_owner.empty() ? nullptr : h->d_seg_local;
    const float inf = std::numeric_limits<float>::infinity();
    P.ev_rc2 = (P.use_ev && h->ev_cut > 0.f) ? h->ev_cut * h->ev_cut : inf;
    P.g_rc2 = (P.use_gauss && h->g_cut > 0.f) ? h->g_cut * h->g_cut : inf;
    if (all_pairs(h)) {
        P.rc2max = inf;
    } else {
        float rc = 0.f;
        if (P.use_ev) rc = std::max(rc, h->ev_cut);
        if (P.use_gauss) rc = std::max(rc, h->g_cut);
        P.rc2max = rc * rc;
    }
    P.ev_pmode = (P.ev_power == 6.0f) ? 6 : (P.ev_power == 3.0f) ? 3 : 0;
    P.use_gauss = (h->has_cob || h->has_scb) ? 1 : 0;
    for (int i = 0; i < 25; ++i) P.table[i] = (h->has_cob ? h->tab_cob[i] : 0.f) + (h->has_scb ? h->tab_scb[i] : 0.f);
    P.g_inv_rc2 = 1.0f / (h->g_rc * h->g_rc);
    P.g_c2 = (float)(-1.4426950408889634 / (2.0 * (double)h->g_rc * (double)h->g_rc));
    // alternative functional forms
    FormParams &Q = h->Q;
    Q.ev_form = h->forms[MMX_SEL_EV];
    Q.ev_gc2 = (float)(-1.4426950408889634 / (2.0 * (double)P.ev_sigma * (double)P.ev_sigma));
    Q.ev_inv_s2 = 1.0f / (P.ev_sigma * P.ev_sigma);
    Q.has_cob = h->has_cob ? 1 : 0;
    Q.has_scb = h->has_scb ? 1 : 0;
    Q.cob_form = h->forms[MMX_SEL_COB];
    Q.scb_form = h->forms[MMX_SEL_SCB];
    for (int i = 0; i < 25; ++i) {
        Q.tab_cob[i] = h->has_cob ? h->tab_cob[i] : 0.f;
        Q.tab_scb[i] = h->has_scb ? h->tab_scb[i] : 0.f;
    }
    for (int l = 0; l < 5; ++l) Q.cob_a[l] = Q.tab_cob[l * 5 + l];
    Q.g_rcomp = h->g_rc;
    Q.g_yuk = (float)(-1.4426950408889634 / (double)h->g_rc);
    Q.lam_form = h->forms[MMX_SEL_LAMINA];
    Q.cf_form = h->forms[MMX_SEL_CENTRAL];
    Q.loop_form = h->forms[MMX_SEL_LOOPS];
    Q.chb_form = h->forms[MMX_SEL_CHB];
    Q.generic_pairs = ((P.use_ev && Q.ev_form != 0) || (h->has_cob && Q.cob_form != 0) ||
                       (h->has_scb && Q.scb_form != 0)) ? 1 : 0;
    // the kernel's LEAN condition: clamp mask, one cutoff, merged energies, no rank-2 / no-sweep variant
    h->nb_force_plain = false;
    const int lo = nb_launch_opt(h);
    const bool lean = !Q.generic_pairs && h->nb_variant != 1 && !all_pairs(h) &&
                      (!(P.use_ev && P.use_gauss) || P.ev_rc2 == P.g_rc2) && (lo & 2) && (lo & 4) && !(lo & 1) && !(lo & 16);
    h->nb_lean = lean;
    // the clamp mask needs cutoff^2 well below 1e6 nm^2: anything beyond falls back to a non-lean instance
    h->nb_force_plain = lean && !(P.rc2max < 1e5f);
    if (h->nb_force_plain) h->nb_lean = false;
}

// choice of the pair kernel: see use_n3
constexpr double kWideCellsBelow = 32.0; // beads per cutoff-sized cell under which the grid switches to cells kWideCellFactor wider
constexpr float kWideCellFactor = 1.12f;
constexpr int kReuseMax = 16;            // evaluations a kept cell structure serves at most
constexpr int kWideCellsFromBeads = 20000;
constexpr double kN3MinBeadsPerCell = 20.0;
constexpr int kN3MinBeads = 70000, kN3MinBeadsGauss = 55000; // (round 5: the unsort launch is gone, the crossover moved down -- see use_n3)

// Beads in this handle's cell list: its owned beads and, on a decomposed rank, the ghost slots of its halo.
int local_beads(const mmx_handle_s *h) {
    if (h->world == 1 && h->n_own == h->n) return h->n;
    const bool comm = h->comm != nullptr || h->lcomm != nullptr;
    return comm && h->dd_halo && h->dd_lists_valid ? h->n_own + h->dd_nghost : h->n_all;
}

// The half-shell kernel (k_nb_n3) runs when the lean pair loop applies and the caller did not ask for bitwise
// reproducibility -- on single-domain handles and on the ranks of a decomposed run alike (there with its DD instance:
// ghosts in clusters of their own, ghost-ghost pairs culled).  nb_variant bit 4096 forces it on (deterministic or
// not), bit 8192 forces it off (A/B timing).
bool use_n3(const mmx_handle_s *h) {
    if (!h->nb_lean || h->n3_cap <= 0 || !h->fsort || !h->n3_items) return false;
    if (h->nb_variant & 8192) return false;
    if (h->nb_variant & 4096) return true;
    if (h->deterministic || (h->nb_variant & 0xffff & ~(4096 | 8192)) != 0) return false;
    // Which kernel is faster depends on the size of the system (scripts/kernel_choice.py: minimizations from the lattice
    // with either kernel forced, iterations/s half shell against full shell):
    //   beads      first 200 iterations (150 -> 50 beads per grid cell)    1500-3000 iterations (-> 25 per cell)
    //   5 000           - 5 %                                                  -15 %
    //   50 000          - 4 %                                                  -12 %
    //   80 000          + 4 %                                                  - 1 %
    //   110 000         + 3 %                                                  + 2 %
    //   200 000         + 5 %                                                  + 4 %
    //   1 000 000       + 7 %                                                  + 7 %
    // (the persistent workgroups of the half-shell kernel want several work items each; its path costs one small launch
    // more.)  Below 20 beads per cell nothing was measured: the full-shell kernel, which needs no atomics, stays there;
    // the last poll's cell count decides.  Both kernels compute the same forces to rounding.
    // Round 5 (k_tail took the unsort launch and its pass over g away; iterations/s over the first 200 iterations, half shell against
    // full shell): EV + compartment Gaussians 55 000 beads +5.8 %, 62 000 +5.7 %, 70 000 +9.5 %, 85 000 +9.8 %; EV alone 50 000 -2 %,
    // 55 000 +2.8 %, 62 000 -2.3 % (noise level): from 55 000 beads with the Gaussians, from 70 000 without.
    const int nl = local_beads(h);
    if (nl < (h->P.use_gauss ? kN3MinBeadsGauss : kN3MinBeads)) return false;
    return h->last_ncells <= 0 || (double)nl >= kN3MinBeadsPerCell * (double)h->last_ncells;
}

int grid_beads(int n) { return std::min((n + 255) / 256, 1024); }

// ---- ownership (decomposed runs: segments of kSeg beads, see Own in mmx_common.hpp) ----------------------------
Own own_of(const mmx_handle_s *h) {
    const bool tables = !h->seg_owner.empty(); // (the device tables may be allocated ahead of their first use: dd_reassign)
    return Own{h->own_lo, h->n_own, h->nseg, tables ? h->d_seg_own : nullptr, tables ? h->d_seg_local : nullptr};
}
// rank that owns global bead b
int owner_of(const mmx_handle_s *h, int b) {
    if (h->world == 1) return 0;
    return h->seg_owner.empty() ? b / h->slice : h->seg_owner[(size_t)(b / kSeg)];
}
// local index of global bead b on this handle, -1: not owned
int local_of(const mmx_handle_s *h, int b) {
    if (h->seg_owner.empty()) {
        const int l = b - h->own_lo;
        return l >= 0 && l < h->n_own ? l : -1;
    }
    const int s = b / kSeg;
    if (b < 0 || s >= (int)h->seg_owner.size() || h->seg_owner[s] != h->rank) return -1;
    const int l = h->seg_lidx[s] * kSeg + (b - s * kSeg);
    return l < h->n_own ? l : -1;
}
// global bead of local index li
int bead_of(const mmx_handle_s *h, int li) {
    if (h->seg_owner.empty()) return h->own_lo + li;
    return h->my_segs[(size_t)(li / kSeg)] * kSeg + li % kSeg;
}

// ---- profiling helpers ----------------------------------------------------------------------
bool prof_begin(mmx_handle_s *h, int slot, EventPair &ep) {
    h->launches[slot]++;
    if (h->profile <= 0 || h->capturing) return false;
    // minimizer: whole evaluations are sampled (every profile-th one; the others may be graph replays); elsewhere
    // every profile-th launch of the slot
    if (h->prof_eval >= 0 ? !(h->prof_eval == 1 || (h->prof_eval == 2 && slot == MMX_K_NONBONDED))
                          : (h->launches[slot] - 1) % h->profile != 0) return false;
    if (h->ev_pool.empty()) return false;
    ep = h->ev_pool.back();
    h->ev_pool.pop_back();
    ep.slot = slot;
    (void)hipEventRecord(ep.a, h->stream);
    return true;
}
void prof_end(mmx_handle_s *h, bool on, EventPair &ep) {
    if (!on) return;
    (void)hipEventRecord(ep.b, h->stream);
    h->ev_used.push_back(ep);
}
// Collectives of a decomposed run, timed in the sampled evaluations of a minimization (slot = kCollBase + which)
enum { kCollBase = 64, kCollNeedmap = 0, kCollHalo = 1, kCollAllreduce = 2, kCollOther = 3 };
bool coll_prof_begin(mmx_handle_s *h, int which, EventPair &ep) {
    if (h->profile <= 0 || h->capturing || h->prof_eval != 1 || h->ev_pool.empty()) return false;
    ep = h->ev_pool.back();
    h->ev_pool.pop_back();
    ep.slot = kCollBase + which;
    (void)hipEventRecord(ep.a, h->stream);
    return true;
}
void prof_collect(mmx_handle_s *h, mmx_stats *out) {
    for (auto &ep : h->ev_used) {
        float ms = 0.f;
        if (ep.slot >= kCollBase) {
            if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
                h->coll_ns[ep.slot - kCollBase] += (double)ms * 1e6;
                h->coll_samples[ep.slot - kCollBase] += 1;
            }
        } else if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess && out) {
            out->kernel_ns[ep.slot] += (double)ms * 1e6;
            out->kernel_samples[ep.slot] += 1;
        }
        h->ev_pool.push_back(ep);
    }
    h->ev_used.clear();
}

// ---- launch sequences -----------------------------------------------------------------------
// LDS force window of k_nb_n3: `want` clusters when the runtime grants the dynamic LDS beyond the default 64 KB per
// workgroup (gfx950 has 160 KB per CU), else what fits in 64 KB next to the kernel's ~35 KB of static LDS.
int n3_configure(int want) {
    const int bytes = (int)n3_lds_bytes(want);
    bool ok = true;
#define N3ATTR(PM, EV, GA, NE)                                                                              \
    ok = ok && hipFuncSetAttribute((const void *)k_nb_n3<PM, EV, GA, NE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   bytes) == hipSuccess &&                                                  \
         hipFuncSetAttribute((const void *)k_nb_n3<PM, EV, GA, NE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                             bytes) == hipSuccess
#define N3ATTR_PM(PM)                                                                                       \
    N3ATTR(PM, true, true, false); N3ATTR(PM, true, true, true); N3ATTR(PM, true, false, false);            \
    N3ATTR(PM, true, false, true); N3ATTR(PM, false, true, false); N3ATTR(PM, false, true, true)
    N3ATTR_PM(6);
    N3ATTR_PM(3);
    N3ATTR_PM(0);
#undef N3ATTR_PM
#undef N3ATTR
    if (!ok) (void)hipGetLastError();
    return ok ? want : 0; // refused: the full-shell kernel stays in charge
}

int n3_grid(const mmx_handle_s *h) { return std::max(1, h->n_cus); } // one persistent workgroup per CU

template <int PMODE>
void launch_nb_n3_p(mmx_handle_s *h, int grid) {
    const FFParams &P = h->P;
    const int cap = h->n3_cap;
    const size_t lds = n3_lds_bytes(cap);
    // the tail of the item queue is taken in shares (k_nb_n3, stage_unit): by default one item per workgroup in two shares
    // each, then one item per four workgroups in four shares (scripts/n3_tail_ab.py, gw_200k at the lattice / after 400 /
    // 2000 iterations, final kernel: 190 / 175 / 157 us without, 182 / 168 / 149 with; half or 1.5 x the items, four
    // shares in the first tier, no second tier: within 1-2 % of that).  For the A/B, nb_variant bits 24-27: items per
    // workgroup in the first tier (x 1/2), bits 28-29: log2(shares) of it, bit 30: no tail at all
    const unsigned tcfg = ((unsigned)h->nb_variant >> 24) & 127u;
    const bool tail = !(tcfg & 64u);
    const int tail_items = !tail ? 0 : (tcfg & 15u) ? (int)(tcfg & 15u) * grid / 2 : grid;
    const int tail_sh = !tail ? 0 : (tcfg & 15u) ? (int)((tcfg >> 4) & 3u) : 1;
    const int tail2_items = tail ? grid / 4 : 0, tail2_sh = tail ? 2 : 0;
    const int spin = (h->inject_fault & 1) ? 0 : kN3SpinLimit; // option "inject_fault" bit 0: every wait of the kernel fails
    // the cell list holds ghosts: the DD instance (nb_variant bit 16384 forces it on a single domain: same results, for A/B timing)
    const bool dd = h->world > 1 || h->n_own != h->n || (h->nb_variant & 16384);
#define N3L(EV, GA, NE, DDI)                                                                                \
    hipLaunchKernelGGL((k_nb_n3<PMODE, EV, GA, NE, DDI>), dim3(grid), dim3(kN3Threads), lds, h->stream, P,  \
                       h->spos4, h->cl_lo, h->n3_items, h->st, h->fsort, h->fstride, h->part,               \
                       cap, (h->nb_variant >> 16) & 255, tail_items, tail_sh, tail2_items, tail2_sh, spin)
#define N3(EV, GA)                                                                                          \
    do {                                                                                                    \
        if (h->nb_skip_energy) {                                                                            \
            if (dd) N3L(EV, GA, true, true);                                                                \
            else N3L(EV, GA, true, false);                                                                  \
        } else {                                                                                            \
            if (dd) N3L(EV, GA, false, true);                                                               \
            else N3L(EV, GA, false, false);                                                                 \
        }                                                                                                   \
    } while (0)
    if (P.use_ev && P.use_gauss) N3(true, true);
    else if (P.use_ev) N3(true, false);
    else N3(false, true);
#undef N3
#undef N3L
}

// What has to follow the half-shell kernel: forces from cluster-slot order into the gradient (outside the "nonbonded"
// timing bracket of enqueue_eval, so that the slot's HIP-event time is the pair kernel's own, as rocprofv3 reports it).
void launch_nb_finish(mmx_handle_s *h) {
    if (!h->n3_build) return;
    const int cl = h->last_clusters > 0 ? h->last_clusters : h->n_all / 8 + 4096;
    const int gu = std::max(64, std::min((cl * 8 + 255) / 256, 2048));
    hipLaunchKernelGGL(k_nb_n3_unsort, dim3(gu), dim3(256), 0, h->stream, h->sbead, h->fsort, h->fstride, h->g, h->st,
                       own_of(h));
}

template <int PMODE>
void launch_nb_cells_p(mmx_handle_s *h, int grid) {
    const FFParams &P = h->P;
    if (h->n3_build) { // the kernel the last cell build prepared for (its work items exist): not re-decided here
        h->n3_launches++;
        launch_nb_n3_p<PMODE>(h, grid);
        return;
    }
#define NBJ(PM, EV, GA, SC, OPT)                                                                            \
    hipLaunchKernelGGL((k_nb_clusters_j<PM, EV, GA, SC, OPT>), dim3(grid), dim3(256), 0, h->stream, P,      \
                       h->spos4, h->cl_lo, h->cl_hi, h->cstart, h->gcur, h->st, h->g, h->part)
#define NBC(EV, GA)                                                                                         \
    do {                                                                                                    \
        if (h->Q.generic_pairs) { /* non-default functional forms: one generic instance per term combination */ \
            hipLaunchKernelGGL((k_nb_clusters_j<0, EV, GA, false, 14, true>), dim3(grid), dim3(256), 0,       \
                               h->stream, P, h->spos4, h->cl_lo, h->cl_hi, h->cstart, h->gcur, h->st, h->g,  \
                               h->part, h->formp);                                                     \
        } else if (h->nb_variant == 1)                                                                      \
            hipLaunchKernelGGL((k_nb_cells<PMODE, EV, GA>), dim3(grid), dim3(192), 0, h->stream, P, h->pos4, \
                               h->perm, h->start, h->items, h->gcur, h->st, h->g, h->part);                 \
        else if (!(EV && GA) || P.ev_rc2 == P.g_rc2) {                                                      \
            /* default: cutoff by v_fma clamp + one energy accumulator pair per lane + per-bead cull;       \
               nb_variant bits 32/64/128 switch these off one by one (A/B timing) */                        \
            const int opt = nb_launch_opt(h);                                                               \
            if (opt == 142) { NBJ(PMODE, EV, GA, true, 142); break; }                                       \
            if (opt == 14 && h->nb_skip_energy) { NBJ(PMODE, EV, GA, true, 14 | 512); break; }              \
            switch (opt) { /* A/B and diagnosis instances keep the plain block -> cluster mapping */        \
            case 14: NBJ(PMODE, EV, GA, true, 14); break;                                                   \
            case 30: NBJ(PMODE, EV, GA, true, 30); break;                                                   \
            case 46: NBJ(PMODE, EV, GA, true, 46); break;                                                   \
            case 22: NBJ(PMODE, EV, GA, true, 22); break;                                                   \
            case 78: NBJ(PMODE, EV, GA, true, 78); break;                                                   \
            case 62: NBJ(PMODE, EV, GA, true, 62); break;                                                   \
            case 6: NBJ(PMODE, EV, GA, true, 6); break;                                                     \
            case 12: NBJ(PMODE, EV, GA, true, 12); break;                                                   \
            case 10: NBJ(PMODE, EV, GA, true, 10); break;                                                   \
            default: NBJ(PMODE, EV, GA, true, 0); break;                                                    \
            }                                                                                               \
        } else                                                                                              \
            NBJ(PMODE, EV, GA, false, 0);                                                                   \
    } while (0)
    if (P.use_ev && P.use_gauss) NBC(true, true);
    else if (P.use_ev) NBC(true, false);
    else NBC(false, true);
#undef NBC
#undef NBJ
}

template <int PMODE>
void launch_nb_allpairs_p(mmx_handle_s *h, int tiles_per_slice) {
    const FFParams &P = h->P;
    dim3 b(256), gdim((h->n + 255) / 256, h->ap_slices);
#define NBA(EV, GA)                                                                                         \
    do {                                                                                                    \
        const bool nocut = (!(EV) || std::isinf(P.ev_rc2)) && (!(GA) || std::isinf(P.g_rc2));               \
        if (h->Q.generic_pairs)                                                                             \
            hipLaunchKernelGGL((k_nb_allpairs<0, EV, GA, true>), gdim, b, 0, h->stream, P, h->pos4,          \
                               tiles_per_slice, h->fpart, h->epart, h->st, h->formp);                       \
        else if (nocut) /* the pure NoCutoff case (the reference's semantics): no masks in the pair loop */  \
            hipLaunchKernelGGL((k_nb_allpairs_lean<PMODE, EV, GA>), gdim, b, 0, h->stream, P, h->pos4,       \
                               tiles_per_slice, h->fpart, h->epart, h->st);                                 \
        else                                                                                                \
            hipLaunchKernelGGL((k_nb_allpairs<PMODE, EV, GA>), gdim, b, 0, h->stream, P, h->pos4,            \
                               tiles_per_slice, h->fpart, h->epart, h->st);                                 \
    } while (0)
    if (P.use_ev && P.use_gauss) NBA(true, true);
    else if (P.use_ev) NBA(true, false);
    else NBA(false, true);
#undef NBA
}

int nb_grid(const mmx_handle_s *h) {
    if (h->n3_build) return n3_grid(h);
    if (h->nb_variant == 1) { // v1: one block per {cell, 64-bead chunk}
        int items = h->last_items > 0 ? h->last_items : (h->n + kChunk - 1) / kChunk + 1024;
        int g = items + items / 4 + 64;
        return std::max(256, std::min(g, kPartStride));
    }
    // cluster kernel: 4 clusters (waves) per block, grid-stride beyond the estimate
    int cl = h->last_clusters > 0 ? h->last_clusters : h->n_all / 8 + 4096;
    int g = (cl + cl / 8) / 4 + 64;
    return (std::max(256, std::min(g, kPartStride)) + 7) & ~7; // multiple of 8: whole rounds over the XCDs
}

// Fills pos4 of every bead from the host-set global positions (multi-GPU: beads of other ranks are
// needed as ghosts before the first all-gather of a call).
__global__ __launch_bounds__(256) void k_fill_pos4_all(int n, int n_all, const float *__restrict__ xg,
                                                       const int8_t *__restrict__ labels, float4 *__restrict__ pos4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_all) return;
    if (i < n)
        pos4[i] = make_float4(xg[3 * i], xg[3 * i + 1], xg[3 * i + 2], __int_as_float((i << 3) | ((int)labels[i] + 2)));
    else
        pos4[i] = make_float4(3e18f, 3e18f, 3e18f, __int_as_float(-8 + 2)); // padding of the last slice
}

// pos4 of the owned beads from x, whatever the minimizer's phase (after a reverted line search: the state is DONE)
__global__ __launch_bounds__(256) void k_repack_own(int n_own, const Own own, const float *__restrict__ x,
                                                    const int8_t *__restrict__ labels, float4 *__restrict__ pos4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_own) return;
    const int bead = own.bead(i);
    pos4[bead] = make_float4(x[3 * i], x[3 * i + 1], x[3 * i + 2], __int_as_float((bead << 3) | ((int)labels[bead] + 2)));
}

enum PackMode { PACK_PLAIN = 0, PACK_MOVE = 1, PACK_MD = 2 };

DirArgs dir_args(const mmx_handle_s *h) { return DirArgs{h->g, h->gp, h->S, h->Y, (size_t)h->n4 * 4}; }


bool has_comm(const mmx_handle_s *h) { return h->comm != nullptr || h->lcomm != nullptr; }

__global__ void k_local_sum(int world, int count, double *const *__restrict__ boxes, double *__restrict__ out) {
    const int i = threadIdx.x;
    if (i >= count) return;
    double r = 0.0;
    for (int q = 0; q < world; ++q) r += boxes[q][i]; // rank order: every rank gets the same bits
    out[i] = r;
}

// Loopback collectives.  Protocol per call (p = parity of the call number): [stage own data] -> record ready[r][p]
// -> host barrier -> wait ready[q][p] of every rank, read their data -> record done[r][p] -> host barrier -> wait
// done[q][p] of every rank (nobody may overwrite what another rank is still reading).
bool local_begin(mmx_handle_s *h, int &p) {
    LocalComm &L = *h->lcomm;
    p = (int)(h->coll_seq++ & 1);
    (void)hipEventRecord(L.ready[h->rank * 2 + p], h->stream);
    if (!L.barrier()) return false;
    for (int q = 0; q < L.world; ++q)
        if (q != h->rank) (void)hipStreamWaitEvent(h->stream, L.ready[q * 2 + p], 0);
    return true;
}
bool local_end(mmx_handle_s *h, int p) {
    LocalComm &L = *h->lcomm;
    (void)hipEventRecord(L.done[h->rank * 2 + p], h->stream);
    if (!L.barrier()) return false;
    for (int q = 0; q < L.world; ++q)
        if (q != h->rank) (void)hipStreamWaitEvent(h->stream, L.done[q * 2 + p], 0);
    return true;
}

// Every RCCL call's result is kept: a failed collective leaves stale ghosts or unreduced sums behind, so the next
// poll (pull_state) returns MMX_ERR_RCCL instead of carrying on to a wrong result.
void rccl_check(mmx_handle_s *h, ncclResult_t r, const char *what) {
    if (r == ncclSuccess || h->coll_failed) return;
    h->coll_failed = true;
    h->coll_error = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error");
}

// In-place all-gather of the position slices (ghost beads of every term).
void coll_allgather_pos4(mmx_handle_s *h) {
    if (h->comm) {
        rccl_check(h, g_rccl.AllGather(h->pos4 + (size_t)h->rank * h->slice, h->pos4, (size_t)h->slice * 4, ncclFloat,
                                       h->comm, h->stream), "ncclAllGather");
    } else if (h->lcomm && !h->coll_failed) {
        LocalComm &L = *h->lcomm;
        int p;
        if (!local_begin(h, p)) { h->coll_failed = true; return; }
        for (int q = 0; q < L.world; ++q)
            if (q != h->rank)
                (void)hipMemcpyAsync(h->pos4 + (size_t)q * h->slice, L.h[q]->pos4 + (size_t)q * h->slice,
                                     sizeof(float4) * (size_t)h->slice, hipMemcpyDeviceToDevice, h->stream);
        if (!local_end(h, p)) h->coll_failed = true;
    }
}

// In-place fp64 sum of `count` (<= 64) doubles over the ranks.
void coll_allreduce(mmx_handle_s *h, double *buf, int count) {
    if (h->comm) {
        rccl_check(h, g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, h->comm, h->stream), "ncclAllReduce");
    } else if (h->lcomm && !h->coll_failed) {
        LocalComm &L = *h->lcomm;
        const int p0 = (int)(h->coll_seq & 1);
        (void)hipMemcpyAsync(L.mailbox[h->rank * 2 + p0], buf, sizeof(double) * count, hipMemcpyDeviceToDevice, h->stream);
        int p;
        if (!local_begin(h, p)) { h->coll_failed = true; return; }
        hipLaunchKernelGGL(k_local_sum, dim3(1), dim3(64), 0, h->stream, L.world, count, h->lbox[p], buf);
        if (!local_end(h, p)) h->coll_failed = true;
    }
}

// ---- ghost-bead halo (mmx_dd.hpp) -------------------------------------------------------------------------------
// The chromosomal-block term reads EVERY bead of the owned beads' chromosomes, with no cutoff (k_chb): there is no halo
// for it.  A decomposed run with that term keeps the all-gather of every position per evaluation.
bool use_halo(const mmx_handle_s *h) {
    return h->dd_halo && has_comm(h) && h->world > 1 && !(h->P.use_chb && h->chrom_of);
}

// In-place all-gather of `bytes` per rank at buf + rank * bytes; `peer(q)` = the same buffer of rank q's handle
// (loopback communicator only).
template <class Peer>
void coll_allgather_small(mmx_handle_s *h, void *buf, size_t bytes, Peer peer) {
    if (h->comm) {
        rccl_check(h, g_rccl.AllGather((const char *)buf + (size_t)h->rank * bytes, buf, bytes, ncclChar, h->comm, h->stream),
                   "ncclAllGather");
    } else if (h->lcomm && !h->coll_failed) {
        LocalComm &L = *h->lcomm;
        int p;
        if (!local_begin(h, p)) { h->coll_failed = true; return; }
        for (int q = 0; q < L.world; ++q)
            if (q != h->rank)
                (void)hipMemcpyAsync((char *)buf + (size_t)q * bytes, (const char *)peer(L.h[q]) + (size_t)q * bytes, bytes,
                                     hipMemcpyDeviceToDevice, h->stream);
        if (!local_end(h, p)) h->coll_failed = true;
    }
}

// The halo exchange of one evaluation: one message of host-known capacity per pair of ranks that share any ghost.
void coll_halo_exchange(mmx_handle_s *h) {
    const size_t S = (size_t)h->slice;
    if (h->comm) {
        rccl_check(h, g_rccl.GroupStart(), "ncclGroupStart");
        for (int q = 0; q < h->world; ++q) {
            if (q == h->rank) continue;
            if (h->dd_scap.cap[q] > 0)
                rccl_check(h, g_rccl.Send(h->dd_sendbuf + q * S, (size_t)h->dd_scap.cap[q] * 4, ncclFloat, q, h->comm, h->stream), "ncclSend");
            if (h->dd_rcap.cap[q] > 0)
                rccl_check(h, g_rccl.Recv(h->dd_recvbuf + q * S, (size_t)h->dd_rcap.cap[q] * 4, ncclFloat, q, h->comm, h->stream), "ncclRecv");
        }
        rccl_check(h, g_rccl.GroupEnd(), "ncclGroupEnd");
    } else if (h->lcomm && !h->coll_failed) {
        LocalComm &L = *h->lcomm;
        int p;
        if (!local_begin(h, p)) { h->coll_failed = true; return; }
        for (int q = 0; q < L.world; ++q)
            if (q != h->rank && h->dd_rcap.cap[q] > 0) // what rank q packed for me
                (void)hipMemcpyAsync(h->dd_recvbuf + q * S, L.h[q]->dd_sendbuf + (size_t)h->rank * S,
                                     sizeof(float4) * (size_t)h->dd_rcap.cap[q], hipMemcpyDeviceToDevice, h->stream);
        if (!local_end(h, p)) h->coll_failed = true;
    }
    h->dd_exchanges++;
    for (int q = 0; q < h->world; ++q) h->dd_bytes_sent += (long long)h->dd_scap.cap[q] * 16;
}

int dd_K(const mmx_handle_s *h);

int dd_alloc(mmx_handle_s *h) {
    if (h->dd_boxes) return MMX_OK;
    const size_t W = (size_t)h->world, S = (size_t)h->slice;
    HIPCHK(h, dalloc(&h->dd_boxes, W * 6));
    HIPCHK(h, dalloc(&h->dd_grid, (size_t)1));
    HIPCHK(h, dalloc(&h->dd_occ, (size_t)kDDWords));
    HIPCHK(h, dalloc(&h->dd_maps, W * kDDPayload));
    HIPCHK(h, dalloc(&h->dd_static, (size_t)std::max(h->slice, 1)));
    HIPCHK(h, dalloc(&h->dd_send_ids, W * S));
    HIPCHK(h, dalloc(&h->dd_send_cnt, W));
    HIPCHK(h, dalloc(&h->dd_cntmat, W * W));
    HIPCHK(h, hipHostMalloc((void **)&h->dd_cnt_host, sizeof(int) * W * W, hipHostMallocDefault));
    std::memset(h->dd_cnt_host, 0, sizeof(int) * W * W);
    HIPCHK(h, dalloc(&h->dd_ghost_ids, W * S));
    HIPCHK(h, dalloc(&h->dd_sendbuf, W * S));
    HIPCHK(h, dalloc(&h->dd_recvbuf, W * S));
    HIPCHK(h, dalloc(&h->dd_xref, (size_t)3 * std::max(h->slice, 1)));
    return MMX_OK;
}

// Ranks that always need owned bead b: the owners of b-2 .. b+2 (backbone bonds and angles reach two beads across a
// slice end) and of its loop partners (mmx_set_loops).
int dd_upload_static(mmx_handle_s *h) {
    std::vector<unsigned long long> m((size_t)std::max(h->n_own, 1), 0ull);
    for (int i = 0; i < h->n_own; ++i) {
        const int b = bead_of(h, i);
        unsigned long long bits = (size_t)i < h->dd_loop_mask.size() ? h->dd_loop_mask[i] : 0ull;
        for (int d = -2; d <= 2; ++d) {
            const int o = b + d;
            if (o < 0 || o >= h->n) continue;
            const int r = owner_of(h, o);
            if (r != h->rank) bits |= 1ull << r;
        }
        m[i] = bits;
    }
    HIPCHK(h, hipMemcpyAsync(h->dd_static, m.data(), sizeof(unsigned long long) * m.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return MMX_OK;
}

// Capacity of a message for a list of `cnt` entries: room for cnt / dd_slack_div + 512 more before the list outgrows
// it (dd_slack_div: 4 at the start of a minimization, 8 after eight quiet polls, halved -- down to 1 -- whenever a list
// did outgrow its message: the collapse phase of a minimization from the lattice triples some lists within 30 evaluations).
int dd_capacity(const mmx_handle_s *h, int cnt) {
    if (h->inject_fault & 4) return std::max(cnt, 0); // tests: no slack at all -- any growth of a list halts the evaluation
    // (an empty list still gets a message of 512 entries = 8 KB: two slices that come into contact during a run -- the
    // collapse phase re-arranges who touches whom -- would otherwise void an evaluation for their first shared ghost)
    return std::min(h->slice, std::max(cnt, 0) + std::max(cnt, 0) / h->dd_slack_div + 512);
}

// Message capacities from the world x world matrix of list lengths (row = sender).  `fresh`: every capacity is set from
// the length (a synchronous rebuild: the lists are these very ones); otherwise only where a list has used up half of its
// slack or would fit a message of half the size.  Both ends of a message evaluate the same rule on the same numbers.
bool dd_set_capacities(mmx_handle_s *h, const int *mat, bool fresh) {
    const int W = h->world;
    bool changed = false;
    auto upd = [&](int &cap, int cnt) {
        int want = cap;
        if (fresh || cnt + cnt / (2 * h->dd_slack_div) > cap || (cap > 2048 && dd_capacity(h, cnt) < cap / 2)) want = dd_capacity(h, cnt);
        if (want != cap) {
            cap = want;
            changed = true;
        }
    };
    for (int q = 0; q < W; ++q) {
        if (q == h->rank) {
            h->dd_scap.cap[q] = h->dd_rcap.cap[q] = 0;
            continue;
        }
        upd(h->dd_scap.cap[q], mat[W * h->rank + q]);
        upd(h->dd_rcap.cap[q], mat[W * q + h->rank]);
    }
    h->dd_nghost = 0;
    h->dd_off.off[0] = 0;
    for (int q = 0; q < W; ++q) {
        h->dd_nghost += h->dd_rcap.cap[q];
        h->dd_off.off[q + 1] = h->dd_nghost;
    }
    return changed;
}

// Rebuild of the ghost lists from the owned positions as they are (a pack has just run on the stream: x, pos4 of the
// owned beads and bbox_part are current).  On the stream, no host round trip: need-map of the owned beads, all-gather of
// the maps (32 KB per rank), one send list per destination -- checked against the capacity of its message.
// `sync` (start of every API call, repeat of a halted evaluation): before that the coarse grid is laid over the
// all-gathered owned boxes, the lists are unbounded, and afterwards the matrix of list lengths is all-gathered and read
// by the host: fresh capacities.
int dd_rebuild(mmx_handle_s *h, bool sync, bool occ_done = false) { // occ_done: the pack has marked the occupancy already
    const int gb = std::max((h->n_own + 255) / 256, 1);
    const size_t W = (size_t)h->world;
    const float reach = hmin_of(h) / (1.001f * edge_factor(h)) + (dd_K(h) > 1 ? h->dd_skin_cur : 0.f);
    if (sync) {
        hipLaunchKernelGGL(k_dd_bbox, dim3(1), dim3(256), 0, h->stream, h->bbox_part, (h->n_own + 255) / 256,
                           h->dd_boxes + 6 * h->rank);
        coll_allgather_small(h, h->dd_boxes, 6 * sizeof(float), [](mmx_handle_s *o) { return (void *)o->dd_boxes; });
        hipLaunchKernelGGL(k_dd_grid, dim3(1), dim3(64), 0, h->stream, h->dd_boxes, h->world, reach, h->dd_grid);
    }
    if (!occ_done) {
        h->dd_occ_clean = false;
        HIPCHK(h, hipMemsetAsync(h->dd_occ, 0, sizeof(unsigned long long) * kDDWords, h->stream));
        hipLaunchKernelGGL(k_dd_occupancy, dim3(gb), dim3(256), 0, h->stream, h->n_own, h->x, h->dd_grid, h->dd_occ, h->st);
    }
    // rebuilds on the stream: the dilation zeroes the list lengths after it has read them, the list kernel the occupancy words after
    // the dilation has: two memset launches less per evaluation (the words are zero whenever a pack is about to mark them: h->dd_occ_clean)
    hipLaunchKernelGGL(k_dd_dilate, dim3(kDDWords / 256 + 1), dim3(256), 0, h->stream, h->dd_occ, h->dd_grid,
                       h->dd_maps + (size_t)h->rank * kDDPayload, h->dd_send_cnt, h->world, h->st, sync ? 0 : 1);
    {
        EventPair cep{};
        const bool con = coll_prof_begin(h, kCollNeedmap, cep);
        coll_allgather_small(h, h->dd_maps, sizeof(unsigned long long) * kDDPayload,
                             [](mmx_handle_s *o) { return (void *)o->dd_maps; });
        prof_end(h, con, cep);
    }
    if (sync) HIPCHK(h, hipMemsetAsync(h->dd_send_cnt, 0, sizeof(int) * W, h->stream));
    DDCaps caps = h->dd_scap;
    if (sync)
        for (int q = 0; q < h->world; ++q) caps.cap[q] = h->slice;
    hipLaunchKernelGGL(k_dd_build_lists, dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->rank, h->world, h->x,
                       h->dd_grid, h->dd_maps, h->dd_static, h->dd_send_ids, h->slice, h->dd_send_cnt, caps, h->st,
                       sync ? nullptr : h->dd_cntmat, // (on the stream: + the lengths of the lists in use until now, of every
                                                      // rank: what the next poll sizes the messages by)
                       h->dd_occ);
    h->dd_occ_clean = true;
    if (dd_K(h) > 1 && !h->dd_ref_in_pack) // the lists start a new life: reference positions of the displacement test.  (st->dd_stale is NOT
                         // cleared here: the minimizer halts in the very evaluation that raises it, and an MD call must still
                         // see at its next poll that one of its steps ran on stale lists)
        HIPCHK(h, hipMemcpyAsync(h->dd_xref, h->x, sizeof(float) * 3 * (size_t)h->n_own, hipMemcpyDeviceToDevice, h->stream));
    h->dd_since = 0;
    h->dd_redecompositions++;
    if (!sync) return MMX_OK;
    HIPCHK(h, hipMemcpyAsync(h->dd_cntmat + W * h->rank, h->dd_send_cnt, sizeof(int) * W, hipMemcpyDeviceToDevice, h->stream));
    coll_allgather_small(h, h->dd_cntmat, W * sizeof(int), [](mmx_handle_s *o) { return (void *)o->dd_cntmat; });
    HIPCHK(h, hipMemcpyAsync(h->dd_cnt_host, h->dd_cntmat, sizeof(int) * W * W, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemsetAsync(&h->st->dd_stale, 0, sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(&h->st->dd_overflow, 0, sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(h->cell_of, 0xff, sizeof(int) * (size_t)h->n_all, h->stream)); // beads that are no longer ghosts
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->coll_failed) return fail(h, MMX_ERR_RCCL, !h->coll_error.empty() ? h->coll_error : "loopback collective timed out");
    (void)dd_set_capacities(h, h->dd_cnt_host, true);
    h->dd_lists_valid = true;
    h->dd_sync_rebuilds++;
    return MMX_OK;
}

// Evaluations a set of ghost lists serves: the option dd_rebuild_every, or -- option dd_adaptive -- what the polls derive from the
// trial moves they read back (1 while the structure collapses: exact lists, no skin; up to dd_rebuild_every once it has settled)
int dd_K(const mmx_handle_s *h) { return h->dd_adaptive ? std::max(1, std::min(h->dd_k_cur, std::max(h->dd_every, 1))) : h->dd_every; }
// Which rebuild the next evaluation of a running call gets: 2 = on the stream (every dd_K-th evaluation), 0 = none.
int dd_schedule(mmx_handle_s *h) {
    if (!use_halo(h) || !h->dd_lists_valid) return 0;
    return ++h->dd_since >= dd_K(h) ? 2 : 0;
}

void enqueue_bonded(mmx_handle_s *h, CtlArgs &A, bool in_scan);

// The counter set of parity `par` must be zero before a pack counts into it: it is, unless the last direct build used this very
// parity (scan-based builds in between flip the parity without touching the sets)
void direct_claim_set(mmx_handle_s *h, const int par) {
    if (h->dset_dirty[par]) {
        const size_t cset = (size_t)h->maxcells + 1;
        (void)hipMemsetAsync(h->dcount + (size_t)par * cset, 0, sizeof(int) * cset, h->stream);
        (void)hipMemsetAsync(h->drows + (size_t)par * kDirectRowSet, 0, sizeof(int) * kDirectRowSet, h->stream);
        if (h->dcount_g) (void)hipMemsetAsync(h->dcount_g + (size_t)par * cset, 0, sizeof(int) * cset, h->stream);
    }
    h->dset_dirty[par] = true;      // this build's counts stay in it ...
    h->dset_dirty[par ^ 1] = false; // ... and it zeroes the other one
}

// Decomposed ranks, direct build: what arrived from the peers goes to pos4 AND is counted into this parity's ghost set
void dd_ghost_count(mmx_handle_s *h, const dim3 gq, const bool split) { // split: a two-launch build follows (cells are large by their ghosts alone)
    const int par = h->build_idx & 1;
    const size_t cset = (size_t)h->maxcells + 1;
    int *const drcl = h->drows + (size_t)par * kDirectRowSet;
    const GhostCount C{h->grid + par, h->cell_of, h->rank_in_cell, h->dcount_g + (size_t)par * cset, h->dcount + (size_t)par * cset,
                       drcl + 2 * kDirectMaxRows, drcl + (split ? 3 : 1) * kDirectMaxRows, h->slotkeys, h->slot_cap, h->slot_cells,
                       split ? 1 : 0};
    hipLaunchKernelGGL(k_dd_unpack_count, gq, dim3(256), 0, h->stream, h->dd_recvbuf, h->dd_off, h->slice, h->pos4, h->dd_ghost_ids,
                       h->n_all, C, h->st);
}

int local_beads(const mmx_handle_s *h);
bool use_n3(const mmx_handle_s *h);
// phase 0: the whole build in one launch; 1 / 2: the owned beads' share (nothing in it waits for a peer) and the ghosts' share +
// work items + bonded pass (k_build_direct_dd<.., PHASE>; half-shell kernel's split list only)
void launch_build_direct_dd(mmx_handle_s *h, CtlArgs &bonded, const int gb, const int phase = 0) {
    const float hm = hmin_of(h);
    const int par = h->build_idx & 1;
    GridParams *cur = h->grid + par, *next = h->grid + (par ^ 1);
    h->grid_factor[par ^ 1] = edge_factor(h);
    h->grid_factor[par] = edge_factor(h);
    if (phase != 2) h->n3_build = use_n3(h);
    const int nvb = grid_beads(h->n_own);
    const int nib = (h->n3_build && phase != 1) ? kN3ItemBlocks : 1;
    const int vi = phase ? 1 + phase : h->n3_build ? 1 : 0;
    if (h->direct_dd_slots[vi] <= 0) {
        int per_cu = 0;
        hipError_t oe = hipErrorUnknown;
        switch (vi) {
        case 0: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct_dd<kDirectDDCap, false, 0>, 256, 0); break;
        case 1: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct_dd<kDirectDDCap, true, 0>, 256, 0); break;
        case 2: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct_dd<kDirectDDCap, true, 1>, 256, 0); break;
        default: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct_dd<kDirectDDCap, true, 2>, 256, 0); break;
        }
        if (oe != hipSuccess || per_cu <= 0) {
            (void)hipGetLastError();
            per_cu = 3;
        }
        h->direct_dd_slots[vi] = per_cu * std::max(h->n_cus, 1);
    }
    const int nbr = phase == 1 ? 0 : (nvb + 1) / 2;
    int go = std::max(256, std::min(2048, h->direct_dd_slots[vi] - nib - nbr));
    if (phase == 1 && h->dd_overlap_go > 0) go = std::max(64, std::min(go, h->dd_overlap_go));
    const size_t cset = (size_t)h->maxcells + 1;
    const bool bb_on = h->flags && (h->P.use_bond || h->P.use_angle);
    const bool loops_on = h->n_rows > 0 && h->lstart;
    DirectArgs D{};
    D.grid = cur;
    D.grid_next = next;
    D.parity = par;
    D.bbox_part = h->bbox_part;
    D.nblk_bbox = gb;
    D.hmin = hm;
    D.maxcells = h->maxcells;
    D.count = h->dcount + (size_t)par * cset;
    D.rowcl = h->drows + (size_t)par * kDirectRowSet;
    D.rowbig = D.rowcl + (phase == 2 ? 3 : 1) * kDirectMaxRows; // (phase 2: the cells of many ghosts)
    D.count_zero = h->dcount + (size_t)(par ^ 1) * cset;
    D.rowcl_zero = h->drows + (size_t)(par ^ 1) * kDirectRowSet;
    D.rowbig_zero = D.rowcl_zero + kDirectMaxRows;
    D.keys = h->slotkeys;
    D.slot_cap = h->slot_cap;
    D.slot_cells = h->slot_cells;
    D.pos4 = h->pos4;
    D.spos4 = h->spos4;
    D.cl_lo = h->cl_lo;
    D.cstart = h->cstart;
    D.sbead = h->sbead;
    D.slot_of = h->slot_of;
    D.cap_clusters = h->n_all;
    D.n_beads = h->n_all;
    D.n3_items = h->n3_items;
    D.n3_max_items = (h->inject_fault & 2) ? 1 : h->n3_max_items;
    D.n3_flags = (h->n3_long_items == 2 ? 5 : h->n3_long_items >= 0 ? h->n3_long_items : local_beads(h) >= kN3LongItemsFrom ? 1 : 0) |
                 (h->n3_pass_records ? 0 : 2) | (h->n3_slice_cap << 8);
    D.n_items_blocks = nib;
    D.n_bonded_blocks = nbr;
    D.n_order = go;
    DirectDD X{};
    X.count_g = h->dcount_g + (size_t)par * cset;
    X.rowclg = D.rowcl + 2 * kDirectMaxRows;
    X.count_g_zero = h->dcount_g + (size_t)(par ^ 1) * cset;
    X.rowclg_zero = D.rowcl_zero + 2 * kDirectMaxRows;
    X.rowbigg_zero = D.rowcl_zero + 3 * kDirectMaxRows;
    X.istart = h->istart;
    X.expand = hm / edge_factor(h); // (grown by the cutoff, whatever the cell edge)
    X.own = own_of(h);
    const BondedArgs BA{bb_on ? h->flags : nullptr, loops_on ? h->lstart : nullptr, h->partner, h->loop_r0, h->cf_w, h->g, h->part,
                        h->Q.loop_form, h->Q.lam_form, h->Q.cf_form, nvb};
    const dim3 gd(nib + nbr + go);
    if (phase == 1) hipLaunchKernelGGL((k_build_direct_dd<kDirectDDCap, true, 1>), gd, dim3(256), 0, h->stream, D, X, h->st, h->P, BA);
    else if (phase == 2) hipLaunchKernelGGL((k_build_direct_dd<kDirectDDCap, true, 2>), gd, dim3(256), 0, h->stream, D, X, h->st, h->P, BA);
    else if (h->n3_build) hipLaunchKernelGGL((k_build_direct_dd<kDirectDDCap, true, 0>), gd, dim3(256), 0, h->stream, D, X, h->st, h->P, BA);
    else hipLaunchKernelGGL((k_build_direct_dd<kDirectDDCap, false, 0>), gd, dim3(256), 0, h->stream, D, X, h->st, h->P, BA);
    if (phase == 1) return; // (the second launch closes the build)
    enqueue_bonded(h, bonded, true);
    h->gcur = cur;
    h->build_idx++;
    h->grid_ready = true;
    h->last_build_direct = true;
    h->last_direct_parity = par;
    h->direct_builds++;
}

// Pack (+ trial move / integrator step), then the cell build.  With `bonded` set, the bonded terms of the evaluation
// -- which only need pos4 -- are enqueued with it: inside the launch of the cell scan ("overlap_bonded", default), or
// right behind the pack.
// redecomp (decomposed runs with a halo): 0 = the ghost lists stay, 1 = synchronous rebuild, 2 = rebuild on the stream.
void enqueue_build(mmx_handle_s *h, int mode, bool init = false, CtlArgs *bonded = nullptr, int redecomp = 0) {
    const int gb = (h->n_own + 255) / 256;  // blocks over owned beads (k_pack, bbox partials)
    const int ga = (h->n_all + 255) / 256;  // blocks over every bead of pos4
    const bool dd = h->world > 1 || h->n_own != h->n;
    const bool fuse_count = !dd && !init && has_nb(h) && !all_pairs(h);
    // decomposed ranks, ghost lists rebuilt on the stream (redecomp 2): the occupancy of the need-map is marked by the pack, which
    // has the new positions in registers (one launch and one pass over x less per evaluation)
    const bool occ_in_pack = dd && redecomp == 2 && use_halo(h) && !h->dd_frozen && h->dd_occ && h->dd_grid &&
                             (mode == PACK_MOVE || mode == PACK_PLAIN);
    if (occ_in_pack && !h->dd_occ_clean) (void)hipMemsetAsync(h->dd_occ, 0, sizeof(unsigned long long) * kDDWords, h->stream);
    if (occ_in_pack) h->dd_occ_clean = false; // (marked by this pack; the rebuild that follows zeroes it again)
    const DDGrid *const ddg = occ_in_pack ? h->dd_grid : nullptr;
    unsigned long long *const ddo = occ_in_pack ? h->dd_occ : nullptr;
    // decomposed ranks, lists kept over dd_every > 1 evaluations: the trial move itself checks the owned beads against where they
    // were when the lists were built (and records that place when this evaluation rebuilds them)
    const bool ref_in_pack = dd && dd_K(h) > 1 && use_halo(h) && !h->dd_frozen && h->dd_xref && mode == PACK_MOVE &&
                             (redecomp != 0 || h->dd_lists_valid);
    const float dd_half = 0.5f * h->dd_skin_cur;
    const int track = (dd && h->dd_adaptive && use_halo(h) && mode == PACK_MOVE) ? 1 : 0;
    const RefArgs RD = ref_in_pack ? RefArgs{h->dd_xref, redecomp ? 3 : 1, dd_half * dd_half, track, 1} : RefArgs{nullptr, 0, 0.f, track, 0};
    h->dd_ref_in_pack = ref_in_pack;
    // kept cell structure (see mmx_handle_s::cell_reuse): trial moves of a single-domain minimization only
    // (and only on grids wider than the cutoff: without a skin there is nothing to keep, and the reference costs 24 B / bead)
    const bool tracked = fuse_count && mode == PACK_MOVE && h->cell_reuse && !h->capturing && !h->use_graph && h->cell_xref &&
                         h->grid_factor[h->build_idx & 1] > 1.f;
    const float rcut = hmin_of(h) / (1.001f * edge_factor(h));
    const float half_skin = 0.5f * (h->struct_factor - 1.f) * rcut;
    const bool reuse = tracked && h->struct_valid && h->reuse_K > 1 && h->struct_evals < h->reuse_K && half_skin > 0.f &&
                       h->last_clusters > 0;
    if (reuse) {
        // (option inject_fault bit 3, tests: a skin of nothing -- every evaluation on a kept structure finds it stale)
        const RefArgs R{h->cell_xref, 1, (h->inject_fault & 8) ? 1e-16f : half_skin * half_skin};
        hipLaunchKernelGGL((k_pack<true, false, true>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x,
                           h->xp, h->d, h->labels, h->pos4, h->bbox_part, h->st, (const GridParams *)nullptr,
                           (int *)nullptr, (int *)nullptr, (int *)nullptr, dir_args(h), R, h->st);
        if (bonded) enqueue_bonded(h, *bonded, false);
        const int gu = std::max(64, std::min((h->last_clusters * 8 + 255) / 256, 2048));
        hipLaunchKernelGGL(k_refresh_clusters, dim3(gu), dim3(256), 0, h->stream, h->sbead, h->pos4, h->spos4, h->cl_lo, h->st);
        h->struct_evals++;
        h->cell_reuses++;
        return; // gcur, n3_build, the work items and the cluster list are those of the structure's full build
    }
    if (!tracked) h->struct_valid = false; // whatever is built below is not tracked by cell_xref
    h->slots_now = false;
    bool direct = false;
    // decomposed ranks: the direct build over owned beads and ghosts (mmx_build.hpp, k_build_direct_dd) -- on the grid the
    // previous build laid out, with the ghost lists in place (or about to be rebuilt on the way: redecomp)
    const bool direct_dd = dd && !init && h->world > 1 && use_halo(h) && (h->dd_lists_valid || redecomp == 1) &&
                           h->fused_build && h->direct_ok && h->dcount_g && bonded && h->fused_bonded && h->overlap_bonded &&
                           has_nb(h) && !all_pairs(h) && h->grid_ready && h->cell_slots && h->slotkeys && h->slot_cap > 0 &&
                           !h->capturing && (mode == PACK_MOVE || mode == PACK_PLAIN);
    if (mode == PACK_MD) { // integrator step fused with the pack (forces of the current positions are in g)
        MdParams M = h->md;
        M.step_lo = (uint32_t)h->md_step;
        M.step_hi = (uint32_t)(h->md_step >> 32);
#define MDP(K)                                                                                              \
    do {                                                                                                    \
        if (fuse_count)                                                                                     \
            hipLaunchKernelGGL((k_md_pack<K, true>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x, \
                               h->xlo, h->v, h->g, h->labels, h->pos4, h->bbox_part, M, &h->st->ftrial,      \
                               h->grid + (h->build_idx & 1), h->cell_of, h->rank_in_cell, h->count);        \
        else                                                                                                \
            hipLaunchKernelGGL((k_md_pack<K>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x,   \
                               h->xlo, h->v, h->g, h->labels, h->pos4, h->bbox_part, M, &h->st->ftrial);     \
    } while (0)
        if (h->md_kind == MD_LANGEVIN) MDP(MD_LANGEVIN);
        else if (h->md_kind == MD_VERLET) MDP(MD_VERLET);
        else if (h->md_kind == MD_AMD) MDP(MD_AMD);
        else MDP(MD_BROWNIAN);
#undef MDP
    } else if (fuse_count) { // single GPU, cell list in use, grid already known: pack + cell count in one launch
        GridParams *cur = h->grid + (h->build_idx & 1);
        if (mode == PACK_MOVE) { // trial move of the minimizer: also forms the new direction after an accepted step
            // (tracked: cell_xref <- the positions this build bins; the displacement from the previous structure's reference
            //  is recorded on the way -- what the host sizes reuse_K by)
            const RefArgs R{h->cell_xref, tracked ? (h->struct_valid ? 2 : 3) : 0, 0.f};
            // sort keys straight into the slot table: no k_cell_fill below
            h->slots_now = h->cell_slots && h->slotkeys && h->slot_cap > 0 && !h->capturing && !h->use_graph;
            // the direct build (mmx_build.hpp): the pack also keeps the per-row totals, in the counter set of this build's parity
            direct = h->slots_now && h->fused_build && h->direct_ok && h->dcount && bonded && h->fused_bonded && h->overlap_bonded;
            const int par = h->build_idx & 1;
            if (direct) direct_claim_set(h, par);
            int *const dcnt = h->dcount + (size_t)par * ((size_t)h->maxcells + 1);
            int *const drcl = h->drows + (size_t)par * kDirectRowSet, *const drbig = drcl + kDirectMaxRows;
            h->direct_key32 = direct && h->n <= (1 << 20) && h->key32;
            const SlotArgs T{h->slots_now ? h->slotkeys : nullptr, h->slot_cap, h->slot_cells, direct ? drcl : nullptr,
                             direct ? drbig : nullptr, (h->inject_fault & 64) ? 1 : 0, kDirectMaxRows, -1.f, h->direct_key32 ? 1 : 0};
            hipLaunchKernelGGL((k_pack<true, true, true>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x,
                               h->xp, h->d, h->labels, h->pos4, h->bbox_part, h->st, cur, h->cell_of, h->rank_in_cell,
                               direct ? dcnt : h->count, dir_args(h), R, h->st, T);
            if (tracked) {
                h->struct_valid = true;
                h->struct_evals = 1;
                h->struct_factor = h->grid_factor[h->build_idx & 1];
                h->cell_builds++;
            }
        }
        else
            hipLaunchKernelGGL((k_pack<false, true>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x, h->xp,
                               h->d, h->labels, h->pos4, h->bbox_part, h->st, cur, h->cell_of, h->rank_in_cell, h->count);
    } else if (direct_dd) { // decomposed rank, direct build: the pack counts the OWNED beads into this parity's set (keys into the slot table)
        const int par = h->build_idx & 1;
        direct_claim_set(h, par);
        int *const dcnt = h->dcount + (size_t)par * ((size_t)h->maxcells + 1);
        int *const drcl = h->drows + (size_t)par * kDirectRowSet;
        const SlotArgs T{h->slotkeys, h->slot_cap, h->slot_cells, drcl, drcl + kDirectMaxRows, (h->inject_fault & 64) ? 1 : 0, kDirectDDRows,
                         hmin_of(h) / edge_factor(h)};
        GridParams *cur = h->grid + par;
        if (mode == PACK_MOVE)
            hipLaunchKernelGGL((k_pack<true, true, true>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x,
                               h->xp, h->d, h->labels, h->pos4, h->bbox_part, h->st, cur, h->cell_of, h->rank_in_cell, dcnt,
                               dir_args(h), RD, h->st, T, ddg, ddo);
        else
            hipLaunchKernelGGL((k_pack<false, true, false>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x,
                               h->xp, h->d, h->labels, h->pos4, h->bbox_part, h->st, cur, h->cell_of, h->rank_in_cell, dcnt,
                               DirArgs{}, RefArgs{nullptr, 0, 0.f}, h->st, T, ddg, ddo);
    } else if (mode == PACK_MOVE)
        hipLaunchKernelGGL((k_pack<true, false, true>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x,
                           h->xp, h->d, h->labels, h->pos4, h->bbox_part, h->st, (const GridParams *)nullptr,
                           (int *)nullptr, (int *)nullptr, (int *)nullptr, dir_args(h), RD,
                           (ref_in_pack || track) ? h->st : (MinState *)nullptr, SlotArgs{nullptr, 0, 0}, ddg, ddo);
    else
        hipLaunchKernelGGL((k_pack<false>), dim3(gb), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x, h->xp, h->d,
                           h->labels, h->pos4, h->bbox_part, h->st, (const GridParams *)nullptr, (int *)nullptr,
                           (int *)nullptr, (int *)nullptr, DirArgs{}, RefArgs{nullptr, 0, 0.f}, (MinState *)nullptr,
                           SlotArgs{nullptr, 0, 0}, ddg, ddo);
    if (direct) { // ONE launch: bonded pass || in-cell order || work items, every workgroup finding its offsets by itself
        const float hm = hmin_of(h);
        const int par = h->build_idx & 1;
        GridParams *cur = h->grid + par, *next = h->grid + (par ^ 1);
        h->grid_factor[par ^ 1] = edge_factor(h); // the grid this build lays out for the next one
        h->n3_build = use_n3(h);
        const bool small_cells = h->last_max_per_cell > 0 && h->last_max_per_cell <= 640;
        const int nvb = grid_beads(h->n_own);
        const int nib = h->n3_build ? kN3ItemBlocks : 1; // (workgroup 0 publishes the totals and the next grid either way)
        // ONE round of workgroups: every workgroup of the launch pays the row-prefix prologue (~2 us), so the launch is sized to what
        // is resident at once -- the bonded pass in half as many workgroups as it has virtual blocks, the rest of the slots order
        // cells (a wave per cell: four cells in flight per workgroup)
        const int vi = (small_cells ? 0 : 2) + (h->n3_build ? 1 : 0);
        if (h->direct_slots[vi] <= 0) { // (the 32-bit-key instances use no more registers than their 64-bit twins)
            int per_cu = 0;
            hipError_t oe = hipErrorUnknown;
            switch (vi) {
            case 0: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct<kChunk, 1024, false, false>, 256, 0); break;
            case 1: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct<kChunk, 1024, true, false>, 256, 0); break;
            case 2: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct<kChunk, 4096, false, false>, 256, 0); break;
            default: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_build_direct<kChunk, 4096, true, false>, 256, 0); break;
            }
            if (oe != hipSuccess || per_cu <= 0) {
                (void)hipGetLastError();
                per_cu = 3;
            }
            h->direct_slots[vi] = per_cu * std::max(h->n_cus, 1);
        }
        const int nbr = (nvb + 1) / 2;
        // (large grids: a wave should not walk more than ~3 cells one after the other -- the prologue then amortises over a
        //  second round of workgroups: gw_1m, 39 000 cells)
        const int go = std::max(std::max(256, std::min(2048, h->direct_slots[vi] - nib - nbr)), std::min(4096, h->last_ncells / 12));
        const size_t cset = (size_t)h->maxcells + 1;
        const bool bb_on = h->flags && (h->P.use_bond || h->P.use_angle);
        const bool loops_on = h->n_rows > 0 && h->lstart;
        DirectArgs D{};
        D.grid = cur;
        D.grid_next = next;
        D.parity = par;
        D.bbox_part = h->bbox_part;
        D.nblk_bbox = gb;
        D.hmin = hm;
        D.maxcells = h->maxcells;
        D.count = h->dcount + (size_t)par * cset;
        D.rowcl = h->drows + (size_t)par * kDirectRowSet;
        D.rowbig = D.rowcl + kDirectMaxRows;
        D.count_zero = h->dcount + (size_t)(par ^ 1) * cset;
        D.rowcl_zero = h->drows + (size_t)(par ^ 1) * kDirectRowSet;
        D.rowbig_zero = D.rowcl_zero + kDirectMaxRows;
        D.keys = h->slotkeys;
        D.slot_cap = h->slot_cap;
        D.slot_cells = h->slot_cells;
        D.pos4 = h->pos4;
        D.spos4 = h->spos4;
        D.cl_lo = h->cl_lo;
        D.cstart = h->cstart;
        D.sbead = h->sbead;
        D.slot_of = h->slot_of;
        D.cap_clusters = h->n_all;
        D.n_beads = h->n;
        D.n3_items = h->n3_items;
        D.n3_max_items = (h->inject_fault & 2) ? 1 : h->n3_max_items;
        D.n3_flags = (h->n3_long_items == 2 ? 5 : h->n3_long_items >= 0 ? h->n3_long_items : local_beads(h) >= kN3LongItemsFrom ? 1 : 0) |
                     (h->n3_pass_records ? 0 : 2) | (h->n3_slice_cap << 8);
        D.n_items_blocks = nib;
        D.n_bonded_blocks = nbr;
        D.n_order = go;
        const BondedArgs BA{bb_on ? h->flags : nullptr, loops_on ? h->lstart : nullptr, h->partner, h->loop_r0, h->cf_w, h->g, h->part,
                            h->Q.loop_form, h->Q.lam_form, h->Q.cf_form, nvb};
        const dim3 gd(nib + nbr + go);
#define BUILD_DIRECT(CAPV, N3V)                                                                                      \
    do {                                                                                                              \
        if (h->direct_key32)                                                                                          \
            hipLaunchKernelGGL((k_build_direct<kChunk, CAPV, N3V, true>), gd, dim3(256), 0, h->stream, D, h->st, h->P, BA); \
        else                                                                                                          \
            hipLaunchKernelGGL((k_build_direct<kChunk, CAPV, N3V, false>), gd, dim3(256), 0, h->stream, D, h->st, h->P, BA); \
    } while (0)
        if (small_cells) {
            if (h->n3_build) BUILD_DIRECT(1024, true);
            else BUILD_DIRECT(1024, false);
        } else {
            if (h->n3_build) BUILD_DIRECT(4096, true);
            else BUILD_DIRECT(4096, false);
        }
#undef BUILD_DIRECT
        enqueue_bonded(h, *bonded, true); // (bookkeeping of the pass that rode along + the chromosomal blocks)
        h->gcur = cur;
        h->build_idx++;
        h->grid_ready = true;
        h->last_build_direct = true;
        h->last_direct_parity = par;
        h->direct_builds++;
        return;
    }
    h->last_build_direct = false;
    // Decomposed ranks, the halo BESIDE the owned beads' share of the build (option dd_overlap): everything between the pack and the
    // arrival of the ghosts -- list kernels, need-map all-gather, message pack, send / recv, ghost count -- goes to the handle's
    // second stream, fenced by two events; this stream meanwhile lays out the next grid and sorts the owned beads into their clusters
    // (k_build_direct_dd<.., 1>), waits for the ghosts, and finishes with their clusters, the work items and the bonded pass (<.., 2>).
    // Not for synchronous rebuilds (the host waits in them anyway) and not for the full-shell kernel's interleaved list (a cell's
    // owned clusters are placed behind the ghosts' counts there).
    const bool overlap = direct_dd && h->dd_overlap && h->stream2 && h->ev_pack && h->ev_halo && redecomp != 1 && h->dd_lists_valid &&
                         use_n3(h);
    hipStream_t const main_stream = h->stream;
    if (overlap) {
        (void)hipEventRecord(h->ev_pack, main_stream);
        if (!(h->dd_overlap & 2)) launch_build_direct_dd(h, *bonded, gb, 1);
        (void)hipStreamWaitEvent(h->stream2, h->ev_pack, 0);
        h->stream = h->stream2; // (every launch and collective below is enqueued through h->stream)
        h->dd_overlapped++;
    }
    // dd_frozen (measurement: scripts/dd_projection.py): no collective is issued -- the ghost lists and the ghost positions
    // last received stay, so one rank's kernels can be timed alone on exactly the beads it holds in a real run
    if (redecomp && use_halo(h) && !h->dd_frozen) { // fresh ghost lists from the positions the pack has just written
        const int rc = dd_rebuild(h, redecomp == 1, occ_in_pack);
        if (rc != MMX_OK && h->dd_rc == MMX_OK) h->dd_rc = rc;
    }
    const bool halo = use_halo(h) && h->dd_lists_valid;
    if (halo && h->dd_frozen) {
        // nothing to exchange: the ghosts of the last exchange are binned again (below); an overlapped evaluation also runs the
        // list kernels of a rebuild on the stream, so that what is timed is what a real evaluation puts beside the owned build
        if (overlap) {
            hipLaunchKernelGGL(k_dd_dilate, dim3(kDDWords / 256 + 1), dim3(256), 0, h->stream, h->dd_occ, h->dd_grid,
                               h->dd_maps + (size_t)h->rank * kDDPayload, h->dd_send_cnt, h->world, h->st, 1);
            hipLaunchKernelGGL(k_dd_build_lists, dim3(std::max(gb, 1)), dim3(256), 0, h->stream, h->n_own, own_of(h), h->rank, h->world,
                               h->x, h->dd_grid, h->dd_maps, h->dd_static, h->dd_send_ids, h->slice, h->dd_send_cnt, h->dd_scap, h->st,
                               h->dd_cntmat, h->dd_occ);
            int mx = 1;
            for (int q = 0; q < h->world; ++q) mx = std::max(mx, std::max(h->dd_scap.cap[q], h->dd_rcap.cap[q]));
            hipLaunchKernelGGL(k_dd_pack, dim3(std::min((mx + 255) / 256, 256), h->world), dim3(256), 0, h->stream, h->dd_send_ids,
                               h->dd_send_cnt, h->slice, h->pos4, h->dd_sendbuf, h->dd_scap, h->st);
        }
    } else if (halo) { // ghosts for pairs, bonds, loops: the listed beads only (mmx_dd.hpp)
        if (dd_K(h) > 1 && !redecomp && !h->dd_ref_in_pack) { // lists older than this evaluation: still within the skin?
            const float half = 0.5f * h->dd_skin_cur;
            hipLaunchKernelGGL(k_dd_displacement, dim3(std::max(gb, 1)), dim3(256), 0, h->stream, h->n_own, h->x, h->dd_xref,
                               half * half, h->st);
        }
        int mx = 1;
        for (int q = 0; q < h->world; ++q) mx = std::max(mx, std::max(h->dd_scap.cap[q], h->dd_rcap.cap[q]));
        const dim3 gq(std::min((mx + 255) / 256, 256), h->world);
        hipLaunchKernelGGL(k_dd_pack, gq, dim3(256), 0, h->stream, h->dd_send_ids, h->dd_send_cnt, h->slice, h->pos4,
                           h->dd_sendbuf, h->dd_scap, h->st);
        {
            EventPair cep{};
            const bool con = coll_prof_begin(h, kCollHalo, cep);
            coll_halo_exchange(h);
            prof_end(h, con, cep);
        }
        if (direct_dd) {
            dd_ghost_count(h, gq, overlap);
        } else
            hipLaunchKernelGGL(k_dd_unpack, gq, dim3(256), 0, h->stream, h->dd_recvbuf, h->dd_off, h->slice, h->pos4,
                               h->dd_ghost_ids, h->n_all, h->st);
    } else if (has_comm(h) && !h->dd_frozen) // every rank contributes its slice of pos4 (in place): ghosts for pairs, bonds, loops
        coll_allgather_pos4(h);
    if (direct_dd) {
        if (halo && h->dd_frozen) { // (measurement: the ghosts of the last exchange are counted again, nothing is exchanged)
            int mx = 1;
            for (int q = 0; q < h->world; ++q) mx = std::max(mx, h->dd_rcap.cap[q]);
            dd_ghost_count(h, dim3(std::min((mx + 255) / 256, 256), h->world), overlap);
        }
        if (overlap) { // the ghosts are in place and counted: back to the evaluation's own stream
            (void)hipEventRecord(h->ev_halo, h->stream);
            h->stream = main_stream;
            if (h->dd_overlap & 2) launch_build_direct_dd(h, *bonded, gb, 1); // (A/B: the halo's launches enqueued first)
            (void)hipStreamWaitEvent(main_stream, h->ev_halo, 0);
        }
        launch_build_direct_dd(h, *bonded, gb, overlap ? 2 : 0);
        return;
    }
    const bool in_scan = bonded && h->fused_bonded && h->overlap_bonded && has_nb(h) && !all_pairs(h);
    if (bonded && !in_scan) enqueue_bonded(h, *bonded, false);
    if (has_nb(h) && !all_pairs(h)) {
        const float hm = hmin_of(h);
        GridParams *cur = h->grid + (h->build_idx & 1), *next = h->grid + ((h->build_idx + 1) & 1);
        h->grid_factor[(h->build_idx + 1) & 1] = edge_factor(h); // the grid this build's scan lays out for the next one
        if (init || dd) h->grid_factor[h->build_idx & 1] = edge_factor(h);
        if (init || dd) // multi-GPU: exact box of the owned beads grown by the cutoff, every build
            hipLaunchKernelGGL(k_grid_init, dim3(1), dim3(256), 0, h->stream, h->bbox_part, gb, hm, h->maxcells,
                               dd ? hm / edge_factor(h) : 0.f, cur, h->st); // (grown by the cutoff, whatever the cell edge)
        const int gl = (h->n_own + h->dd_nghost + 255) / 256; // owned beads + listed ghosts
        if (halo)
            hipLaunchKernelGGL(k_cell_count_dd, dim3(std::max(gl, 1)), dim3(256), 0, h->stream, h->n_own, own_of(h),
                               h->dd_nghost, h->dd_ghost_ids, h->pos4, cur, h->cell_of, h->rank_in_cell, h->count,
                               h->count_own, h->st);
        else if (!fuse_count)
            hipLaunchKernelGGL(k_cell_count, dim3(ga), dim3(256), 0, h->stream, h->n_all, own_of(h), h->pos4,
                               cur, h->cell_of, h->rank_in_cell, h->count, h->st, h->count_own);
        // (count_own: decomposed handles only -- a cell's owned beads and its ghosts form separate clusters)
        h->n3_build = use_n3(h); // latched per build: the pair kernel that follows must be the one whose work items exist
        // decomposed ranks on the half-shell kernel: the ghosts' clusters in a region of their own behind the owned ones
        const int split = (dd && h->n3_build && h->count_own && h->dd_split) ? 1 : 0;
        const ScanArgs sa{h->bbox_part, gb, hm, h->maxcells, h->count, h->start, h->istart, h->cstart, h->biglist, cur, next,
                          h->count_own, dd ? hm / edge_factor(h) : 0.f, split};
        if (in_scan) { // block 0 scans, the others are the bonded pass (four virtual 256-thread blocks each)
            const int nvb = grid_beads(h->n_own);
            const bool bb_on = h->flags && (h->P.use_bond || h->P.use_angle);
            const bool loops_on = h->n_rows > 0 && h->lstart;
            hipLaunchKernelGGL((k_scan_bonded<kChunk>), dim3(1 + (nvb + 3) / 4), dim3(1024), 0, h->stream, sa, h->st,
                               h->P, h->pos4, bb_on ? h->flags : nullptr, loops_on ? h->lstart : nullptr, h->partner,
                               h->loop_r0, h->cf_w, h->g, h->part, h->Q.loop_form, h->Q.lam_form, h->Q.cf_form, nvb);
            enqueue_bonded(h, *bonded, true);
        } else {
            hipLaunchKernelGGL((k_cell_scan<kChunk>), dim3(1), dim3(1024), 0, h->stream, sa, h->st);
        }
        if (halo)
            hipLaunchKernelGGL(k_cell_fill_dd, dim3(std::max(gl, 1)), dim3(256), 0, h->stream, h->n_own, own_of(h), h->dd_nghost,
                               h->dd_ghost_ids, h->cell_of, h->rank_in_cell, h->start, h->perm, h->okeys, h->pos4, cur, h->st);
        else if (!h->slots_now)
            hipLaunchKernelGGL(k_cell_fill, dim3(ga), dim3(256), 0, h->stream, h->n_all, h->cell_of, h->rank_in_cell,
                               h->start, h->perm, h->okeys, h->pos4, cur, own_of(h), h->st);
        const unsigned long long *keys = h->slots_now ? h->slotkeys : h->okeys;
        const int scap = h->slots_now ? h->slot_cap : 0, scells = h->slots_now ? h->slot_cells : 0;
        // in-LDS sort capacity from the largest cell of the last poll (60 % headroom), see k_cell_order; the work items
        // of the half-shell pair kernel are built by extra workgroups of the same launch (k_order_items)
        // (decomposed ranks always take the 4096-bead instance: a cell that outgrows the sort is an error there, see cell_order_block)
        const bool small_cells = !dd && h->last_max_per_cell > 0 && h->last_max_per_cell <= 640;
        const int go = small_cells ? 2048 : 1024;
        if (h->n3_build) {
            if (small_cells)
                hipLaunchKernelGGL((k_order_items<kChunk, 1024>), dim3(go + kN3ItemBlocks), dim3(256), 0, h->stream, go, cur,
                                   h->start, h->istart, h->count, h->perm, h->items, h->cstart, h->pos4, h->spos4, h->cl_lo,
                                   h->cl_hi, own_of(h), keys, h->biglist, h->n3_items,
                                   (h->inject_fault & 2) ? 1 : h->n3_max_items, h->st, h->count_own, h->sbead,
                                   (h->n3_long_items == 2 ? 5 : h->n3_long_items >= 0 ? h->n3_long_items : local_beads(h) >= kN3LongItemsFrom ? 1 : 0) | (h->n3_pass_records ? 0 : 2) | (h->n3_slice_cap << 8), scap, scells, split, h->slot_of);
            else
                hipLaunchKernelGGL((k_order_items<kChunk, 4096>), dim3(go + kN3ItemBlocks), dim3(256), 0, h->stream, go, cur,
                                   h->start, h->istart, h->count, h->perm, h->items, h->cstart, h->pos4, h->spos4, h->cl_lo,
                                   h->cl_hi, own_of(h), keys, h->biglist, h->n3_items,
                                   (h->inject_fault & 2) ? 1 : h->n3_max_items, h->st, h->count_own, h->sbead,
                                   (h->n3_long_items == 2 ? 5 : h->n3_long_items >= 0 ? h->n3_long_items : local_beads(h) >= kN3LongItemsFrom ? 1 : 0) | (h->n3_pass_records ? 0 : 2) | (h->n3_slice_cap << 8), scap, scells, split, h->slot_of);
        } else if (small_cells)
            hipLaunchKernelGGL((k_cell_order<kChunk, 1024>), dim3(go), dim3(256), 0, h->stream, cur, h->start,
                               h->istart, h->count, h->perm, h->items, h->cstart, h->pos4, h->spos4, h->cl_lo, h->cl_hi,
                               own_of(h), keys, h->biglist, h->st, h->count_own, h->sbead, scap, scells, h->slot_of);
        else
            hipLaunchKernelGGL((k_cell_order<kChunk, 4096>), dim3(go), dim3(256), 0, h->stream, cur, h->start,
                               h->istart, h->count, h->perm, h->items, h->cstart, h->pos4, h->spos4, h->cl_lo, h->cl_hi,
                               own_of(h), keys, h->biglist, h->st, h->count_own, h->sbead, scap, scells, h->slot_of);
        h->gcur = cur;
        h->build_idx++;
        h->grid_ready = true;
    }
}

// Bonded terms + confinement (+ chromosomal blocks): the FIRST writers of the gradient in an evaluation (they read
// pos4 only); the pair kernel adds its forces afterwards.  in_scan: the fused bonded pass already went out inside
// k_scan_bonded, only its bookkeeping and the chromosomal blocks are left.
void enqueue_bonded(mmx_handle_s *h, CtlArgs &A, bool in_scan) {
    EventPair ep{};
    const int gb = grid_beads(h->n_own);
    const int gs = gb; // blocks of the confinement pass (= of every term when the passes are fused)
    const bool bb_on = h->flags && (h->P.use_bond || h->P.use_angle);
    const bool loops_on = h->n_rows > 0 && h->lstart;
    bool on;
    if (in_scan) {
        h->launches[MMX_K_CONFINE]++;
        A.nblk[P_BOND] = A.nblk[P_ANGLE] = A.nblk[P_LOOP] = gb;
    } else if (h->fused_bonded) {
        // backbone + loops + confinement in one pass; booked in the "confine" timing slot
        on = prof_begin(h, MMX_K_CONFINE, ep);
        hipLaunchKernelGGL(k_bonded_fused, dim3(gb), dim3(256), 0, h->stream, h->P, h->pos4, bb_on ? h->flags : nullptr,
                           loops_on ? h->lstart : nullptr, h->partner, h->loop_r0, h->cf_w, h->g, h->part, h->st,
                           h->Q.loop_form, h->Q.lam_form, h->Q.cf_form);
        prof_end(h, on, ep);
        A.nblk[P_BOND] = A.nblk[P_ANGLE] = A.nblk[P_LOOP] = gb;
    } else {
        // k_backbone is a dependent load chain per 62-bead tile: 8 waves per SIMD (2 048 blocks) instead of 4, and it STORES
        // the gradient (first writer, no memset before it): 167 -> 130 -> ... us at 16 M beads.  (k_confine is slower on 2 048.)
        const int gk = std::min((h->n_own + 255) / 256, 2048);
        if (!bb_on) (void)hipMemsetAsync(h->g, 0, sizeof(float) * 4 * (size_t)h->n4, h->stream);
        if (bb_on) {
            on = prof_begin(h, MMX_K_BACKBONE, ep);
            hipLaunchKernelGGL(k_backbone, dim3(gk), dim3(256), 0, h->stream, h->P, h->pos4, h->flags, h->g, h->part,
                               h->st, 1);
            prof_end(h, on, ep);
            A.nblk[P_BOND] = A.nblk[P_ANGLE] = gk;
        }
        if (h->n_rows > 0) {
            const int gl = std::min((h->n_rows + 255) / 256, 1024);
            on = prof_begin(h, MMX_K_LOOPS, ep);
            hipLaunchKernelGGL(k_loops, dim3(gl), dim3(256), 0, h->stream, h->P, h->n_rows, h->pos4, h->row_bead,
                               h->row_start, h->partner, h->loop_r0, h->g, h->part, h->st, h->Q.loop_form);
            prof_end(h, on, ep);
            A.nblk[P_LOOP] = gl;
        }
        on = prof_begin(h, MMX_K_CONFINE, ep);
        hipLaunchKernelGGL(k_confine, dim3(gs), dim3(256), 0, h->stream, h->P, h->pos4, h->cf_w, h->g, h->part, h->st,
                           h->Q.lam_form, h->Q.cf_form);
        prof_end(h, on, ep);
    }
    A.nblk[P_CONT] = A.nblk[P_LAM] = A.nblk[P_CENT] = gs;
    if (h->P.use_chb && h->chrom_of) {
        const int gc = (h->n_own + 255) / 256; // one block per 256 owned beads (no grid-stride: LDS tiling)
        on = prof_begin(h, MMX_K_CHB, ep);
#define CHB(F)                                                                                              \
    hipLaunchKernelGGL((k_chb<F>), dim3(gc), dim3(256), 0, h->stream, h->P, h->pos4, h->chrom_of, h->chrom_lo,    \
                       h->chrom_hi, h->g, h->part, h->st)
        if (h->Q.chb_form == 0) CHB(0);
        else if (h->Q.chb_form == 1) CHB(1);
        else CHB(2);
#undef CHB
        prof_end(h, on, ep);
        A.nblk[P_CHB] = std::min(gc, kPartStride);
    }
}

// What follows the forces of an evaluation.
enum { FOLD_NONE = 0, // nothing (MD steps whose energies nobody reads)
       FOLD_PLAIN,    // fold the energies (mmx_compute, MD reports)
       FOLD_MIN };    // minimizer: history pass, then energies + line search + direction coefficients

// One full energy+gradient evaluation: pack/move, [bonded terms || cell build], pair kernel, fold.
void enqueue_eval(mmx_handle_s *h, int mode, int fold, int redecomp = 0) {
    EventPair ep{};
    CtlArgs A{};
    bool on = prof_begin(h, MMX_K_CELL_BUILD, ep);
    enqueue_build(h, mode, false, &A, redecomp);
    prof_end(h, on, ep);
    on = prof_begin(h, MMX_K_NONBONDED, ep);
    if (!has_nb(h)) {
        // the bonded pass wrote the whole gradient
    } else if (all_pairs(h)) {
        const int tiles = (h->n + 255) / 256;
        const int tps = (tiles + h->ap_slices - 1) / h->ap_slices;
        switch (h->P.ev_pmode) {
        case 6: launch_nb_allpairs_p<6>(h, tps); break;
        case 3: launch_nb_allpairs_p<3>(h, tps); break;
        default: launch_nb_allpairs_p<0>(h, tps); break;
        }
        const int gf = grid_beads(h->n);
        hipLaunchKernelGGL(k_nb_allpairs_fold, dim3(gf), dim3(256), 0, h->stream, h->n, h->ap_slices, h->fpart,
                           h->epart, h->g, h->part, h->st);
        A.nblk[P_EV] = A.nblk[P_GAUSS] = gf;
    } else {
        const int gn = nb_grid(h);
        switch (h->P.ev_pmode) {
        case 6: launch_nb_cells_p<6>(h, gn); break;
        case 3: launch_nb_cells_p<3>(h, gn); break;
        default: launch_nb_cells_p<0>(h, gn); break;
        }
        A.nblk[P_EV] = A.nblk[P_GAUSS] = gn;
    }
    prof_end(h, on, ep);
    const bool solo = !has_comm(h) || h->dd_frozen; // no all-reduce: this handle's sums are all there is
    const int gh = std::min((h->n4 + 255) / 256, 256); // x kHistGroups column groups
    if (fold == FOLD_MIN && h->fused_tail) {
        // the minimizer's tail in ONE launch (k_tail): the half-shell kernel's forces out of their cluster slots, the history
        // pass, and -- by the workgroup that finishes last -- the fold + decision
        const bool unsort = has_nb(h) && !all_pairs(h) && h->n3_build;
        if (++h->tail_epoch == 0u) h->tail_epoch = 1u;
        const TailArgs T{h->slot_of, h->fsort, h->fstride, h->n_own, unsort ? &h->st->n3_queue : nullptr, h->tail_epoch,
                         (h->inject_fault & 32) ? -1 : kTailSpinLimit}; // option inject_fault bit 5: the fold gives up before its first poll
        on = prof_begin(h, MMX_K_LBFGS, ep);
#define TAIL(UNS, SOL)                                                                                        \
    hipLaunchKernelGGL((k_tail<UNS, SOL>), dim3(gh), dim3(1024), 0, h->stream, h->n4, (const float4 *)h->x,     \
                       (const float4 *)h->xp, (float4 *)h->g, (const float4 *)h->gp, (const float4 *)h->d,  \
                       (float4 *)h->S, (float4 *)h->Y, h->rows, h->st, T, A, h->part)
        if (unsort) {
            if (solo) TAIL(true, true);
            else TAIL(true, false);
        } else {
            if (solo) TAIL(false, true);
            else TAIL(false, false);
        }
#undef TAIL
        prof_end(h, on, ep);
        if (!solo) { // energies, Gram rows, g.d, x.x of all ranks: ONE fp64 all-reduce per evaluation
            on = prof_begin(h, MMX_K_REDUCE, ep);
            EventPair cep{};
            const bool con = coll_prof_begin(h, kCollAllreduce, cep);
            coll_allreduce(h, h->st->sums, 16 + MMX_NROWSUM + 1);
            prof_end(h, con, cep);
            hipLaunchKernelGGL(k_decide_reduced, dim3(1), dim3(64), 0, h->stream, h->st);
            prof_end(h, on, ep);
        }
        return;
    }
    if (has_nb(h) && !all_pairs(h)) launch_nb_finish(h);
    if (fold == FOLD_NONE) return;

    if (fold == FOLD_MIN) {
        // (s, y) of the step this evaluation would accept, its Gram rows, g.d and x.x: before the decision, so that
        // decision and direction coefficients are ONE launch (and, multi-GPU, one all-reduce)
        on = prof_begin(h, MMX_K_LBFGS, ep);
        hipLaunchKernelGGL(k_history, dim3(gh, kHistGroups), dim3(256), 0, h->stream, h->n4, (const float4 *)h->x,
                           (const float4 *)h->xp, (const float4 *)h->g, (const float4 *)h->gp, (const float4 *)h->d,
                           (float4 *)h->S, (float4 *)h->Y, h->rows, h->st);
        prof_end(h, on, ep);
    }
    on = prof_begin(h, MMX_K_REDUCE, ep);
    if (fold == FOLD_MIN) {
        if (solo) {
            hipLaunchKernelGGL(k_decide, dim3(1), dim3(kCtlThreads), 0, h->stream, A, h->part, gh, h->rows, h->st);
        } else { // energies, Gram rows, g.d, x.x of all ranks: ONE fp64 all-reduce of 57 doubles per evaluation
            hipLaunchKernelGGL(k_reduce_all, dim3(1), dim3(kCtlThreads), 0, h->stream, A, h->part, gh, h->rows, h->st);
            EventPair cep{};
            const bool con = coll_prof_begin(h, kCollAllreduce, cep);
            coll_allreduce(h, h->st->sums, 16 + MMX_NROWSUM + 1);
            prof_end(h, con, cep);
            hipLaunchKernelGGL(k_decide_reduced, dim3(1), dim3(64), 0, h->stream, h->st);
        }
        // d = sum_a coef[a] B_a, xp <- x, gp <- g: done per bead by the next trial move (k_pack<.., DIR>)
    } else if (solo) {
        hipLaunchKernelGGL(k_controller, dim3(1), dim3(kCtlThreads), 0, h->stream, A, h->part, h->st);
    } else { // energies of all ranks: one fp64 all-reduce of 16 doubles
        hipLaunchKernelGGL(k_reduce_slots, dim3(1), dim3(kCtlThreads), 0, h->stream, A, h->part, h->st);
        coll_allreduce(h, h->st->sums, 16);
        hipLaunchKernelGGL(k_controller_decide, dim3(1), dim3(64), 0, h->stream, h->st);
    }
    prof_end(h, on, ep);
}

// ---- hipGraph replay of the minimizer's evaluation -------------------------------------------------------------
// The launch sequence of a trial evaluation is fixed and device-driven (enqueue_eval: what the kernels do is decided by
// MinState on the device), so TWO consecutive evaluations (the cell grid ping-pongs between two buffers) are captured
// once per mmx_minimize call and replayed.  The key holds every host-side quantity that shapes the launches; when a
// poll changes one of them the graph is captured again.
struct GraphKey {
    int items, clusters, order_cap, n3, parity, wide;
    bool operator==(const GraphKey &o) const {
        return items == o.items && clusters == o.clusters && order_cap == o.order_cap && n3 == o.n3 && parity == o.parity &&
               wide == o.wide;
    }
};
GraphKey graph_key(const mmx_handle_s *h) {
    return GraphKey{h->last_items, h->last_clusters, (h->last_max_per_cell > 0 && h->last_max_per_cell <= 640) ? 1 : 0,
                    use_n3(h) ? 1 : 0, h->build_idx & 1, edge_factor(h) > 1.f ? 1 : 0};
}
void graph_drop(mmx_handle_s *h) {
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->gexec = nullptr;
    h->graph = nullptr;
}
// Captures two trial evaluations.  Returns false (and leaves the direct path in charge) when the runtime refuses.
bool graph_capture(mmx_handle_s *h) {
    graph_drop(h);
    const int idx0 = h->build_idx;
    GridParams *const gcur0 = h->gcur;
    int64_t saved[MMX_N_KERNELS];
    const long long n3_saved = h->n3_launches;
    for (int k = 0; k < MMX_N_KERNELS; ++k) saved[k] = h->launches[k];
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    h->capturing = true;
    for (int k = 0; k < h->graph_evals; ++k) enqueue_eval(h, PACK_MOVE, FOLD_MIN);
    h->capturing = false;
    const hipError_t e = hipStreamEndCapture(h->stream, &h->graph);
    for (int k = 0; k < MMX_N_KERNELS; ++k) { // what one replay adds to the launch counters
        h->glaunches[k] = h->launches[k] - saved[k];
        h->launches[k] = saved[k];
    }
    h->gn3_launches = h->n3_launches - n3_saved;
    h->n3_launches = n3_saved;
    h->build_idx = idx0;
    h->gcur = gcur0;
    if (e != hipSuccess || !h->graph || hipGraphInstantiate(&h->gexec, h->graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        graph_drop(h);
        return false;
    }
    h->gkey_parity = idx0 & 1;
    return true;
}
bool graph_replay(mmx_handle_s *h) {
    if (hipGraphLaunch(h->gexec, h->stream) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    h->build_idx += h->graph_evals;
    h->gcur = h->grid + ((h->build_idx - 1) & 1);
    for (int k = 0; k < MMX_N_KERNELS; ++k) h->launches[k] += h->glaunches[k];
    h->n3_launches += h->gn3_launches;
    return true;
}

int ensure_slots(mmx_handle_s *h, bool grow);

int push_state(mmx_handle_s *h) {
    HIPCHK(h, hipMemcpyAsync(h->st, h->st_host, sizeof(MinState), hipMemcpyHostToDevice, h->stream));
    return MMX_OK;
}
// Both counter sets of the direct build back to zero (start of a call, repeat of a halted evaluation: the evaluations the device
// skipped have advanced the host's build parity, so which set the next pack counts into is not the one the last build zeroed)
int direct_reset(mmx_handle_s *h) {
    if (!h->dcount) return MMX_OK;
    HIPCHK(h, hipMemsetAsync(h->dcount, 0, sizeof(int) * 2 * ((size_t)h->maxcells + 1), h->stream));
    HIPCHK(h, hipMemsetAsync(h->drows, 0, sizeof(int) * 2 * kDirectRowSet, h->stream));
    if (h->dcount_g) HIPCHK(h, hipMemsetAsync(h->dcount_g, 0, sizeof(int) * 2 * ((size_t)h->maxcells + 1), h->stream));
    h->last_build_direct = false;
    h->dset_dirty[0] = h->dset_dirty[1] = false;
    return MMX_OK;
}

int pull_state(mmx_handle_s *h) {
    if (h->last_build_direct) // the fullest cell of the last build (the scan-based build publishes it itself)
        hipLaunchKernelGGL(k_poll_stats, dim3(1), dim3(256), 0, h->stream,
                           h->dcount + (size_t)h->last_direct_parity * ((size_t)h->maxcells + 1), h->gcur, h->st,
                           (h->world > 1 && h->dcount_g) ? h->dcount_g + (size_t)h->last_direct_parity * ((size_t)h->maxcells + 1) : nullptr);
    HIPCHK(h, hipMemcpyAsync(h->st_host, h->st, sizeof(MinState), hipMemcpyDeviceToHost, h->stream));
    const bool halo = use_halo(h) && h->dd_lists_valid && h->dd_cnt_host;
    if (halo)
        HIPCHK(h, hipMemcpyAsync(h->dd_cnt_host, h->dd_cntmat, sizeof(int) * (size_t)h->world * h->world,
                                 hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // decomposed runs: messages follow the lists (every rank sees the same matrix at the same poll: same capacities)
    if (halo && !h->coll_failed && dd_set_capacities(h, h->dd_cnt_host, false)) h->dd_cap_updates++;
    if (h->st_host->n_items > 0) h->last_items = h->st_host->n_items;
    if (h->st_host->n_clusters > 0) h->last_clusters = h->st_host->n_clusters;
    if (h->st_host->max_per_cell > 0) h->last_max_per_cell = h->st_host->max_per_cell;
    if (h->st_host->ncells > 0) {
        h->last_ncells = h->st_host->ncells;
        // beads per cutoff-sized cell of the build just read back (the grid in force may already be the wider one)
        const double f = edge_factor(h);
        const double per_cell = (double)local_beads(h) / ((double)h->last_ncells * f * f * f);
        // wider cells once the structure has thinned out.  Measured (scripts/wide_cells_ab.py, iterations 1000-2000): a fixed
        // 1.12 gives chr1_50k 11 140 -> 11 970 it/s, gw_200k 4 350 -> 4 415; cells sized for ~40 beads each (up to 1.6 x the
        // cutoff) are no better (11 870 / 4 400) and cost region_5k 1.4 %, where 1.12 is neutral: small systems keep the cutoff
        // (single-domain handles that keep cell structures take a wider skin: the structure then serves more evaluations)
        // -- measured, iterations 1000-2000 against a build per evaluation: gw_200k (half-shell kernel) +4.5 % at 1.3, -0.6 % at
        // 1.45; chr1_50k (full-shell kernel, indifferent to the cell size) +15 % at 1.3, +15.6 % at 1.45
        const float rf = h->reuse_factor > 0.f ? h->reuse_factor : use_n3(h) ? 1.3f : 1.45f;
        const float wide = (h->cell_reuse && h->cell_xref) ? std::max(rf, kWideCellFactor) : kWideCellFactor;
        h->edge_auto = (per_cell < (h->wide_below > 0.0 ? h->wide_below : kWideCellsBelow) && local_beads(h) >= kWideCellsFromBeads) ? wide : 1.f;
        // MD keeps no cell structure (every step is a full build): there the wider edge only pays through beads per cell, and the
        // fixed factors above overshoot on a structure that is still dense (chr1_50k after 200 iterations, 26 beads per
        // cutoff-sized cell: 64.7 us per step at the cutoff, 65.8 at 1.12 x, 70.2 at 1.45 x; after 1 500 iterations, 7 per cell:
        // 65.5 against 52.6 at 1.45 x -- scripts/md_options_ab.py).  Under MD the edge aims at ~30 beads per cell.
        if (h->md_active && h->edge_auto > 1.f)
            h->edge_auto = std::min(std::max((float)std::cbrt(30.0 / std::max(per_cell, 1.0)), 1.f), 1.45f);
    }
    {
        const int rc_slots = ensure_slots(h, false);
        if (rc_slots) return rc_slots;
    }
    // kept cell structure: how many evaluations may a structure serve?  disp2_bits = the largest displacement of a bead from
    // where its structure binned it, over the evaluations since the last poll (structures of up to reuse_K evaluations);
    // a structure holds while that stays below half the skin of the grid in force.  70 % of the room is used, the figure
    // at most doubles per poll, and a stale halt halves it (mmx_minimize).
    if (h->cell_reuse && h->cell_xref && h->st_host->disp2_bits != 0u) {
        float d2;
        std::memcpy(&d2, &h->st_host->disp2_bits, 4);
        const float per_eval = std::sqrt(d2) / (float)std::max(1, h->reuse_K);
        const float rcut = hmin_of(h) / (1.001f * edge_factor(h));
        const float allowed = 0.5f * (edge_factor(h) - 1.f) * rcut;
        int K = per_eval > 0.f ? (int)(0.7f * allowed / per_eval) : kReuseMax;
        K = std::max(1, std::min(K, kReuseMax));
        h->reuse_K = std::min(K, std::max(2 * h->reuse_K, 2));
        HIPCHK(h, hipMemsetAsync(&h->st->disp2_bits, 0, sizeof(unsigned), h->stream));
    }
    // decomposed ranks, option dd_adaptive: how many evaluations may a set of ghost lists serve?  dd_move2_max = the largest squared
    // trial move of an owned bead on any rank (all-reduced: the same number everywhere) since the last poll; lists hold while the
    // moves since their build stay below half the skin: 70 % of that room is used, the figure at most doubles per poll, a stale
    // halt halves it (mmx_minimize).  1 = lists rebuilt before every evaluation, exact, no skin (the collapse from the lattice).
    if (halo && h->dd_adaptive && !h->md_active) { // (MD steps are not trial moves: exact lists there, mmx_md_step)
        const double m = std::sqrt(std::pow(std::max(h->st_host->dd_move2_max, 0.0), 1.0 / 8.0)); // (dd_move: 8th power of the squared move)
        h->dd_move_seen = m;
        int K = m > 0.0 ? (int)(0.7 * 0.5 * (double)h->dd_skin_cur / m) : std::max(h->dd_every, 1);
        K = std::max(1, std::min(K, std::max(h->dd_every, 1)));
        K = std::min(K, std::max(2 * h->dd_k_cur, 2));
        if ((K > 1) != (h->dd_k_cur > 1)) h->dd_since = 1 << 20; // lists with / without a skin from the next evaluation on: rebuild
        h->dd_k_cur = K;
        HIPCHK(h, hipMemsetAsync(&h->st->dd_move2_max, 0, sizeof(double), h->stream));
    }
    if (h->comm && g_rccl.CommGetAsyncError && !h->coll_failed) { // errors RCCL found after the call returned
        ncclResult_t ar = ncclSuccess;
        if (g_rccl.CommGetAsyncError(h->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
            rccl_check(h, ar, "asynchronous RCCL error");
    }
    if (h->coll_failed)
        return fail(h, MMX_ERR_RCCL, !h->coll_error.empty() ? h->coll_error
                                         : "loopback collective timed out: every rank of the group must make the same "
                                           "call from its own thread");
    return MMX_OK;
}

// After a poll: did a force kernel of the evaluations just read back report that it could not do its work?  Such an
// evaluation is void -- partial force sums are finite, so nothing else would notice -- and the call ends in
// MMX_ERR_STATE.  (Decomposed runs: the controller spreads the flag of any rank through the all-reduce.)
int kernel_error_rc(mmx_handle_s *h) {
    const int ke = h->st_host->kernel_error;
    if (!ke) return MMX_OK;
    std::string what;
    if (ke & KERR_N3_SPIN) what += " k_nb_n3: a wave waited for a work unit or a window flush that never came;";
    if (ke & KERR_N3_ITEMS) what += " k_nb_n3: the work-item list is too short for this cell build;";
    if (ke & KERR_ORDER_DD) what += " cell build of a decomposed rank: a cell holds more than 4096 beads, owned beads and ghosts cannot be kept in separate clusters;";
    if (ke & KERR_BUILD_WAIT) what += " cell build: a stage of the launch waited for an earlier one that never finished;";
    if (ke & KERR_BOUNDS) what += " cell build: an offset derived from the cell counters would have left its array, nothing was written there;";
    if (ke & KERR_TAIL_WAIT) what += " k_tail: the partial sums of a workgroup never arrived at the one that folds them;";
    if ((ke & 0xff) == 0) what += " reported by another rank;";
    return fail(h, MMX_ERR_STATE, "a force kernel could not do its work, the evaluation is void:" + what +
                                  " forces and energies of this call must not be used");
}

// Slot table of the trial moves: ONE allocation of kSlotsPerBead key slots per bead (256 B / bead), made when the first poll
// knows a cell count, and never again -- hipFree / hipMalloc inside a call cost a twentieth of the driver's 20-iteration
// window when the table followed the polls (measured: 3 058 against 3 228 iterations/s).  What follows the polls is how the
// buffer is CUT: rows of 2 x the fullest cell (+ 32, rounded up to 64 slots), as many rows as fit; rows are only valid for the build
// that wrote them, so a new cut needs no clearing.  A state the buffer cannot hold with 25 % of spare rows keeps the counting
// sort's fill until a later poll.  Called with the stream idle.  grow: after an evaluation that did not fit its cut.
constexpr int kSlotsPerBead = 32;
int ensure_slots(mmx_handle_s *h, bool grow) {
    if (!h->cell_slots || h->last_ncells <= 0 || h->last_max_per_cell <= 0) return MMX_OK;
    if (!h->slotkeys) {
        // (decomposed ranks: a slice of owned beads and about as many ghosts at most)
        h->slot_total = (size_t)kSlotsPerBead * (size_t)std::max(h->world > 1 ? 2 * h->slice : h->n_all, 4096);
        if (h->slot_total * 8 > ((size_t)1 << 33)) { // (16.7 M beads: 4.3 GB -- fine on this part; beyond: the fill stays)
            h->cell_slots = 0;
            return MMX_OK;
        }
        HIPCHK(h, dalloc(&h->slotkeys, h->slot_total));
    }
    // (the fullest cell of the last poll was counted on the grid of the last build; the grids of the builds to come may be wider
    //  -- the poll that has just run switches the wide cells on and off: beads per cell follow the cube of the edge)
    const float f_last = std::max(h->grid_factor[(h->build_idx + 1) & 1], 1.f);
    const float f_next = std::max(std::max(h->grid_factor[h->build_idx & 1], edge_factor(h)), f_last);
    const double widen = (double)(f_next / f_last) * (f_next / f_last) * (f_next / f_last);
    int cap = (((int)(2.0 * widen * h->last_max_per_cell) + 32 + 63) / 64) * 64; // (whole 512-byte lines per row)
    if ((h->inject_fault & 16) && !grow) cap = 64; // tests: rows that every crowded cell outgrows -- halt, larger rows, repeat
    if (grow) cap = std::max(cap, 2 * std::max(h->slot_cap, 32)); // the fullest cell outgrew its row (or the grid its rows: below)
    const size_t rows = h->slot_total / (size_t)cap;
    // (... and narrower: cells then multiply by the cube of the ratio)
    const float f_min = std::max(std::min(std::min(h->grid_factor[h->build_idx & 1], edge_factor(h)), f_last), 1.f);
    const double narrow = (double)(f_last / f_min) * (f_last / f_min) * (f_last / f_min);
    const size_t need = (size_t)(1.25 * narrow * h->last_ncells) + 512 + (grow ? (size_t)h->slot_cells / 2 : 0);
    if (rows < need || rows > (size_t)0x7fffffff) { // does not fit: no slot table until a poll finds a state that does
        h->slot_cap = h->slot_cells = 0;
        return MMX_OK;
    }
    h->slot_cap = cap;
    h->slot_cells = (int)rows;
    return MMX_OK;
}

int ensure_allpairs_scratch(mmx_handle_s *h) {
    if (!all_pairs(h) || !has_nb(h)) return MMX_OK;
    const int tiles = (h->n + 255) / 256;
    int slices = std::max(1, std::min(64, (2048 + tiles - 1) / tiles));
    slices = std::min(slices, tiles);
    if (slices != h->ap_slices || !h->fpart) {
        if (h->fpart) (void)hipFree(h->fpart);
        if (h->epart) (void)hipFree(h->epart);
        h->fpart = nullptr;
        h->epart = nullptr;
        HIPCHK(h, dalloc(&h->fpart, (size_t)slices * h->n));
        HIPCHK(h, dalloc(&h->epart, (size_t)slices * h->n));
        h->ap_slices = slices;
    }
    return MMX_OK;
}

// ---- loops: per-rank CSR from the loops as given (mmx_set_loops; again after a re-assignment of the segments) ------
int rebuild_loops(mmx_handle_s *h) {
    const int n_loops = (int)h->loop_m.size();
    const int *m = h->loop_m.data(), *n = h->loop_n.data();
    const float *r0 = h->loop_r0v.data();
    // CSR over beads that carry a loop end; entries of a bead keep loop order (fixed summation order).
    std::vector<int> deg((size_t)h->n, 0);
    for (int l = 0; l < n_loops; ++l) {
        deg[m[l]]++;
        deg[n[l]]++;
    }
    std::vector<int> row_of((size_t)h->n, -1), row_bead, row_start;
    int ne = 0;
    for (int li = 0; li < h->n_own; ++li) { // rows of the beads this handle owns, in local order
        const int b = bead_of(h, li);
        if (deg[b]) {
            row_of[b] = (int)row_bead.size();
            row_bead.push_back(b);
            row_start.push_back(ne);
            ne += deg[b];
        }
    }
    row_start.push_back(ne);
    std::vector<int> fill(row_start.begin(), row_start.end()), partner((size_t)ne);
    std::vector<float> er0((size_t)ne);
    for (int l = 0; l < n_loops; ++l) {
        if (row_of[m[l]] >= 0) {
            const int q = fill[row_of[m[l]]]++;
            partner[q] = n[l];
            er0[q] = r0[l];
        }
        if (row_of[n[l]] >= 0) {
            const int q = fill[row_of[n[l]]]++;
            partner[q] = m[l];
            er0[q] = r0[l];
        }
    }
    // decomposed runs: the owner of a loop end always needs the other end (static part of the ghost lists)
    h->dd_loop_mask.assign((size_t)std::max(h->n_own, 1), 0ull);
    if (h->world > 1 && h->world <= kDDMaxWorld)
        for (int l = 0; l < n_loops; ++l) {
            const int rm = owner_of(h, m[l]), rn = owner_of(h, n[l]);
            if (rm == rn) continue;
            if (rm == h->rank) h->dd_loop_mask[local_of(h, m[l])] |= 1ull << rn;
            if (rn == h->rank) h->dd_loop_mask[local_of(h, n[l])] |= 1ull << rm;
        }
    h->dd_static_dirty = true;
    for (void *p : {(void *)h->row_bead, (void *)h->row_start, (void *)h->partner, (void *)h->loop_r0, (void *)h->lstart})
        if (p) (void)hipFree(p);
    h->row_bead = h->row_start = h->partner = h->lstart = nullptr;
    h->loop_r0 = nullptr;
    h->n_rows = (int)row_bead.size();
    h->n_loops = n_loops;
    if (h->n_rows > 0) {
        HIPCHK(h, dalloc(&h->row_bead, row_bead.size()));
        HIPCHK(h, dalloc(&h->row_start, row_start.size()));
        HIPCHK(h, dalloc(&h->partner, partner.size()));
        HIPCHK(h, dalloc(&h->loop_r0, er0.size()));
        HIPCHK(h, hipMemcpy(h->row_bead, row_bead.data(), row_bead.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->row_start, row_start.data(), row_start.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->partner, partner.data(), partner.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->loop_r0, er0.data(), er0.size() * sizeof(float), hipMemcpyHostToDevice));
        // the same entries addressed per owned bead (rows are in local order): offsets for the fused bonded kernel
        std::vector<int> lstart((size_t)h->n_own + 1, 0);
        for (int li = 0; li < h->n_own; ++li) lstart[(size_t)li + 1] = lstart[li] + deg[bead_of(h, li)];
        HIPCHK(h, dalloc(&h->lstart, lstart.size()));
        HIPCHK(h, hipMemcpy(h->lstart, lstart.data(), lstart.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return MMX_OK;
}

// ---- spatial re-assignment of the ownership (decomposed runs with a halo; SURVEY 8e "re-decompose") ------------------
// Index ranges are compact bricks only at the Hilbert start: once the globule has been pushed into the container's shell a
// range is a bent slab and a rank carries more ghosts than beads (rounds 1-3: 79 k - 217 k ghosts per 125 000 owned beads
// on 8 ranks of gw_1m).  So the 62-bead segments are re-assigned while the minimization runs: every rank reduces its
// segments to centroids, the centroids are all-gathered (16 B per segment), every rank runs the same recursive
// coordinate bisection on the same numbers (longest axis of the centroids' box, cut in proportion to the ranks on either
// side, never more than seg_per segments per rank) and -- when at least 2 % of the segments would change hands -- the
// per-bead vectors of the optimizer (x, xp, g, gp, d, the 12 history vectors; v and xlo when MD is configured) migrate
// with their segments: each is all-gathered block-wise into a staging area (3 * slice floats per rank) and every rank
// picks the segments it now owns.  L-BFGS is invariant under a permutation of the beads (every reduction is a sum over
// beads: only the rounding order changes), so the optimisation carries on as if nothing had happened; what follows is a
// synchronous rebuild of the ghost lists.  Offline comparison of ownership rules on dumped states of gw_1m
// (scripts/dd_ownership_offline.py, 8 ranks, the need-map rule of mmx_dd.hpp): index ranges 134 k ghosts per rank on
// average (max 215 k); segments of 62 / 248 / 992 beads by this bisection 69 k / 78 k / 84 k (max 74 k / 90 k / 119 k);
// single beads 42 k (the bound: chain segments of neighbouring regions interpenetrate).
__global__ __launch_bounds__(256) void k_seg_centroids(int n_own, const float *__restrict__ x, float4 *__restrict__ out) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63; // one wave per local segment
    const int li = t * kSeg + lane;
    const bool act = lane < kSeg && li < n_own;
    float sx = act ? x[3 * li] : 0.f, sy = act ? x[3 * li + 1] : 0.f, sz = act ? x[3 * li + 2] : 0.f, c = act ? 1.f : 0.f;
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    sz = wave_sum(sz);
    c = wave_sum(c);
    if (lane == 0 && t * kSeg < n_own) out[t] = make_float4(sx / c, sy / c, sz / c, c);
}
// vec (local order of the NEW assignment) <- the staging area: new local segment t comes from block src[t] (in segments)
__global__ __launch_bounds__(256) void k_migrate_vec(int n_own_new, int n_floats_cap, const int *__restrict__ src,
                                                     const float *__restrict__ mig, float *__restrict__ vec) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n_floats_cap; e += gridDim.x * 256) {
        float v = 0.f; // beyond the owned beads the vectors hold zeros (they enter every dot product)
        if (e < 3 * n_own_new) {
            const int li = e / 3, t = li / kSeg;
            v = mig[(size_t)3 * ((size_t)src[t] * kSeg + (li - t * kSeg)) + (e - 3 * li)];
        }
        vec[e] = v;
    }
}

// every rank's `vec` (3 * n_own floats in local order) -> h->mig[q][3 * slice] on every rank
int coll_allgather_vec(mmx_handle_s *h, const float *vec) {
    const size_t blk = (size_t)3 * h->slice;
    if (!h->mig) HIPCHK(h, dalloc(&h->mig, blk * h->world));
    HIPCHK(h, hipMemcpyAsync(h->mig + blk * h->rank, vec, sizeof(float) * 3 * (size_t)h->n_own, hipMemcpyDeviceToDevice, h->stream));
    coll_allgather_small(h, h->mig, blk * sizeof(float), [](mmx_handle_s *o) { return (void *)o->mig; });
    return MMX_OK;
}

// Balanced recursive coordinate bisection: segments `ids` (centroids c) -> ranks [r0, r1), at most `cap` each.
void rcb_assign(const std::vector<float4> &c, std::vector<int> &ids, int lo, int hi, int r0, int r1, int cap, std::vector<int> &owner) {
    const int nr = r1 - r0, len = hi - lo;
    if (nr == 1) {
        for (int k = lo; k < hi; ++k) owner[ids[k]] = r0;
        return;
    }
    float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
    for (int k = lo; k < hi; ++k) {
        const float4 &p = c[ids[k]];
        mn[0] = std::min(mn[0], p.x); mx[0] = std::max(mx[0], p.x);
        mn[1] = std::min(mn[1], p.y); mx[1] = std::max(mx[1], p.y);
        mn[2] = std::min(mn[2], p.z); mx[2] = std::max(mx[2], p.z);
    }
    int ax = 0;
    if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
    if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
    const int nl = nr / 2;
    long long cut = ((long long)len * nl + nr / 2) / nr;                 // in proportion to the ranks on either side
    cut = std::max<long long>(cut, (long long)len - (long long)(nr - nl) * cap); // ... and within what either side can hold
    cut = std::min<long long>(cut, (long long)nl * cap);
    auto key = [&](int s) { return ax == 0 ? c[s].x : ax == 1 ? c[s].y : c[s].z; };
    // (ties by segment id: every rank sorts the same numbers the same way)
    std::nth_element(ids.begin() + lo, ids.begin() + lo + cut, ids.begin() + hi,
                     [&](int a, int b) { const float ka = key(a), kb = key(b); return ka < kb || (ka == kb && a < b); });
    rcb_assign(c, ids, lo, lo + (int)cut, r0, r0 + nl, cap, owner);
    rcb_assign(c, ids, lo + (int)cut, hi, r0 + nl, r1, cap, owner);
}

// Collective: every rank calls it at the same point of the same call (a poll of mmx_minimize: the stream is idle).
// changed: the ownership is different now (the ghost lists are invalid, the next evaluation must rebuild them synchronously).
int dd_reassign(mmx_handle_s *h, bool &changed, double min_moved_fraction = 0.02) {
    changed = false;
    if (!use_halo(h) || h->seg_per <= 0) return MMX_OK;
    h->dd_reassign_attempts++;
    const int W = h->world, SP = h->seg_per, NS = h->nseg;
    // (every allocation of the call before its first collective: a rank that cannot allocate fails here, where its peers are not
    //  yet waiting for it inside one)
    if (!h->seg_cent) {
        HIPCHK(h, dalloc(&h->seg_cent, (size_t)W * SP));
        HIPCHK(h, hipHostMalloc((void **)&h->seg_cent_host, sizeof(float4) * (size_t)W * SP, hipHostMallocDefault));
        HIPCHK(h, dalloc(&h->d_mig_src, (size_t)SP));
    }
    if (!h->mig) HIPCHK(h, dalloc(&h->mig, (size_t)3 * h->slice * h->world));
    if (!h->d_seg_own) {
        HIPCHK(h, dalloc(&h->d_seg_own, (size_t)SP));
        HIPCHK(h, dalloc(&h->d_seg_local, (size_t)NS));
    }
    // 1. centroids of the owned segments -> everybody
    HIPCHK(h, hipMemsetAsync(h->seg_cent + (size_t)h->rank * SP, 0, sizeof(float4) * SP, h->stream));
    const int nsl = (h->n_own + kSeg - 1) / kSeg;
    if (nsl > 0)
        hipLaunchKernelGGL(k_seg_centroids, dim3((nsl + 3) / 4), dim3(256), 0, h->stream, h->n_own, h->x, h->seg_cent + (size_t)h->rank * SP);
    coll_allgather_small(h, h->seg_cent, sizeof(float4) * SP, [](mmx_handle_s *o) { return (void *)o->seg_cent; });
    HIPCHK(h, hipMemcpyAsync(h->seg_cent_host, h->seg_cent, sizeof(float4) * (size_t)W * SP, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->coll_failed) return fail(h, MMX_ERR_RCCL, !h->coll_error.empty() ? h->coll_error : "loopback collective timed out");
    // 2. the current assignment as tables (identity until the first re-assignment), centroid per global segment
    const int nseg_real = (h->n + kSeg - 1) / kSeg;
    std::vector<int> old_owner((size_t)NS, -1), old_lidx((size_t)NS, -1);
    if (h->seg_owner.empty()) {
        for (int s = 0; s < nseg_real; ++s) {
            old_owner[s] = s / SP;
            old_lidx[s] = s % SP;
        }
    } else {
        old_owner = h->seg_owner;
        old_lidx = h->seg_lidx;
    }
    std::vector<float4> cent((size_t)NS);
    std::vector<int> ids;
    ids.reserve((size_t)nseg_real);
    for (int s = 0; s < nseg_real; ++s) {
        cent[s] = h->seg_cent_host[(size_t)old_owner[s] * SP + old_lidx[s]];
        if (!(cent[s].w > 0.f) || !std::isfinite(cent[s].x) || !std::isfinite(cent[s].y) || !std::isfinite(cent[s].z))
            return MMX_OK; // a non-finite state: leave it to the controller
        ids.push_back(s);
    }
    // 3. bisection (the same on every rank) and how much it would move
    std::vector<int> new_owner((size_t)NS, -1);
    rcb_assign(cent, ids, 0, nseg_real, 0, W, SP, new_owner);
    // The bisection numbers its regions arbitrarily: relabel them so that a region goes to the rank that already owns most
    // of it (greedy by overlap, largest first) -- what then changes hands is what has to.
    {
        std::vector<long long> ov((size_t)W * W, 0);
        for (int s = 0; s < nseg_real; ++s) ov[(size_t)new_owner[s] * W + old_owner[s]]++;
        std::vector<int> map((size_t)W, -1);
        std::vector<char> taken((size_t)W, 0);
        for (int round = 0; round < W; ++round) {
            long long best = -1;
            int bi = -1, bj = -1;
            for (int i = 0; i < W; ++i) {
                if (map[i] >= 0) continue;
                for (int j = 0; j < W; ++j)
                    if (!taken[j] && ov[(size_t)i * W + j] > best) {
                        best = ov[(size_t)i * W + j];
                        bi = i;
                        bj = j;
                    }
            }
            map[bi] = bj;
            taken[bj] = 1;
        }
        for (int s = 0; s < nseg_real; ++s) new_owner[s] = map[new_owner[s]];
    }
    int moved = 0;
    for (int s = 0; s < nseg_real; ++s) moved += new_owner[s] != old_owner[s];
    if (moved == 0 || (double)moved < min_moved_fraction * (double)nseg_real) return MMX_OK;
    // the segment that holds the last beads (it may be partial) must stay LAST in its owner's local order: it is, the
    // local order is ascending in the segment id
    std::vector<int> new_lidx((size_t)NS, -1), cnt((size_t)W, 0), mine;
    for (int s = 0; s < nseg_real; ++s) {
        new_lidx[s] = cnt[new_owner[s]]++;
        if (new_owner[s] == h->rank) mine.push_back(s);
    }
    for (int q = 0; q < W; ++q)
        if (cnt[q] > SP || cnt[q] == 0) return MMX_OK; // (cannot happen: the cuts respect the capacity; a rank is never left empty-handed on purpose)
    const int n_own_new = (int)mine.size() * kSeg - (mine.back() == nseg_real - 1 ? nseg_real * kSeg - h->n : 0);
    // 4. migration of the per-bead vectors
    std::vector<int> src(mine.size());
    for (size_t t = 0; t < mine.size(); ++t) src[t] = old_owner[mine[t]] * SP + old_lidx[mine[t]];
    HIPCHK(h, hipMemcpyAsync(h->d_mig_src, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream)); // (src is pageable)
    const int cap_floats = h->n4 * 4;
    const size_t nv = (size_t)h->n4 * 4;
    std::vector<float *> vecs = {h->x, h->xp, h->g, h->gp, h->d};
    for (int a = 0; a < MMX_M; ++a) {
        vecs.push_back(h->S + a * nv);
        vecs.push_back(h->Y + a * nv);
    }
    if (h->v) vecs.push_back(h->v);
    if (h->xlo) vecs.push_back(h->xlo);
    const int gm = std::min((cap_floats + 255) / 256, 1024);
    // From the first migrated vector on, an error leaves some vectors in the new order under the old tables: the handle is
    // poisoned (positions must be set again, the lists rebuilt) instead of carrying on to a wrong result.
    auto poison = [&](int code) {
        h->have_pos = false;
        h->dd_lists_valid = false;
        h->grid_ready = false;
        return code;
    };
    for (float *vec : vecs) {
        int rc = coll_allgather_vec(h, vec); // (blocks in the OLD local order: n_own is still the old count)
        if (rc) return poison(rc);
        hipLaunchKernelGGL(k_migrate_vec, dim3(gm), dim3(256), 0, h->stream, n_own_new, cap_floats, h->d_mig_src, h->mig, vec);
    }
    // 5. the new tables
    h->seg_owner = new_owner;
    h->seg_lidx = new_lidx;
    h->my_segs = mine;
    h->n_own = n_own_new;
    h->own_lo = mine.front() * kSeg;
    std::vector<int> seg_local((size_t)NS, -1);
    for (size_t t = 0; t < mine.size(); ++t) seg_local[mine[t]] = (int)t;
    {
        hipError_t e1 = hipMemcpyAsync(h->d_seg_own, mine.data(), sizeof(int) * mine.size(), hipMemcpyHostToDevice, h->stream);
        if (e1 == hipSuccess) e1 = hipMemcpyAsync(h->d_seg_local, seg_local.data(), sizeof(int) * (size_t)NS, hipMemcpyHostToDevice, h->stream);
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(h->stream);
        if (e1 != hipSuccess) {
            h->err = std::string("dd_reassign: ") + hipGetErrorString(e1);
            return poison(MMX_ERR_HIP);
        }
    }
    if (h->coll_failed) return poison(fail(h, MMX_ERR_RCCL, !h->coll_error.empty() ? h->coll_error : "loopback collective timed out"));
    int rc = rebuild_loops(h);
    if (rc) return poison(rc);
    if ((rc = dd_upload_static(h))) return poison(rc);
    h->dd_static_dirty = false;
    refresh_params(h);
    h->dd_lists_valid = false;
    h->md_forces_valid = false;
    h->dd_reassignments++;
    h->dd_segments_moved += moved;
    changed = true;
    return MMX_OK;
}

// First build of a call: learn the work-item count so the pair kernel's grid is sized to it.
int prime_items(mmx_handle_s *h) {
    if (!has_nb(h) || all_pairs(h)) return MMX_OK;
    enqueue_build(h, PACK_PLAIN, true, nullptr, 1); // decomposed runs: the ghost lists of this call are built here
    if (h->dd_rc != MMX_OK) return h->dd_rc;
    return pull_state(h);
}

int prepare(mmx_handle_s *h) {
    if (!h->have_pos) return fail(h, MMX_ERR_STATE, "positions not set (mmx_set_positions)");
    HIPCHK(h, hipSetDevice(h->device));
    refresh_params(h);
    h->struct_valid = false; // a kept cell structure does not outlive the call that built it
    if (h->world > 1 && has_nb(h) && (all_pairs(h) || h->nb_variant == 1))
        return fail(h, MMX_ERR_STATE, "multi-GPU runs need a pair cutoff and the cluster kernel (nb_variant 0)");
    // A minimization with the halo may have re-assigned the 62-bead segments to the ranks (dd_reassign): the vectors then hold
    // scattered segments.  The paths without a halo (option dd_halo = 0; chromosomal blocks, which switch it off) all-gather
    // contiguous slices of pos4 and walk contiguous bead ranges (k_chb): on a re-assigned ownership they would compute wrong
    // forces without any sign of it.
    if (h->world > 1 && !h->seg_owner.empty() && !use_halo(h))
        return fail(h, MMX_ERR_STATE, "the ownership of this decomposed system was re-assigned by an earlier minimization (scattered "
                                      "62-bead segments); evaluations without the ghost-bead halo (dd_halo = 0, chromosomal blocks) "
                                      "need the initial contiguous slices: create the handles anew");
    if (has_nb(h) && !all_pairs(h) && h->n_all > (1 << 24)) // the cluster kernel addresses spos4 with 32-bit offsets
        return fail(h, MMX_ERR_BAD_ARG, "the cell-list pair kernel supports up to 2^24 beads (the largest Hilbert start the "
                                        "reference can build)");
    if (h->Q.generic_pairs && h->nb_variant == 1 && has_nb(h) && !all_pairs(h))
        return fail(h, MMX_ERR_STATE, "nb_variant 1 only implements the default functional forms");
    int rc = ensure_allpairs_scratch(h);
    if (rc) return rc;
    if (h->Q.generic_pairs) {
        if (!h->formp) HIPCHK(h, dalloc(&h->formp, (size_t)1));
        HIPCHK(h, hipMemcpyAsync(h->formp, &h->Q, sizeof(FormParams), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream)); // h->Q is pageable host memory that may change afterwards
    }
    h->dd_rc = MMX_OK;
    {   // (evaluations the device skipped at the end of the last call -- the minimizer was done -- advanced the host's build parity:
        //  which counter set the next pack finds clean is not known any more)
        const int rc_d = direct_reset(h);
        if (rc_d) return rc_d;
    }
    if (h->fsort_dirty && h->fsort) { // see mmx_handle_s::fsort_dirty
        HIPCHK(h, hipMemsetAsync(h->fsort, 0, sizeof(float) * 3 * (size_t)h->fstride, h->stream));
        h->fsort_dirty = false;
    }
    if (use_halo(h)) {
        if (h->world > kDDMaxWorld) return fail(h, MMX_ERR_BAD_ARG, "the ghost-bead halo supports up to 64 ranks");
        if ((rc = dd_alloc(h))) return rc;
        if (h->dd_static_dirty) {
            if ((rc = dd_upload_static(h))) return rc;
            h->dd_static_dirty = false;
        }
    }
    if (h->xg && h->pos4_dirty) { // ghosts of the first evaluation come from the host-set global positions
        hipLaunchKernelGGL(k_fill_pos4_all, dim3((h->n_all + 255) / 256), dim3(256), 0, h->stream, h->n, h->n_all,
                           h->xg, h->labels, h->pos4);
        h->pos4_dirty = false;
    }
    return MMX_OK;
}

} // namespace
