// mmx_api.hip -- the C ABI of libmmx.so (include/mmx.h): argument checks, error policy, and the call sequences of
// mmx_compute / mmx_minimize / mmx_md_step.  The handle lives in mmx_handle.hpp, the launch sequences in mmx_engine.hpp.
// gfx950 only; there is no CPU path in this library.
#include "mmx_engine.hpp"

// =================================================================================================
// No C++ exception may cross the C ABI: every entry point is a function-try-block ending in MMX_CATCH.
#define MMX_CATCH(H)                                                                                       \
    catch (const std::bad_alloc &) { return fail((H), MMX_ERR_NOMEM, "out of host memory"); }              \
    catch (const std::exception &e) { return fail((H), MMX_ERR_STATE, std::string("internal error: ") + e.what()); } \
    catch (...) { return fail((H), MMX_ERR_STATE, "internal error"); }

extern "C" {

int mmx_abi_version(void) { return 1; }

const char *mmx_last_error(mmx_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int create_impl(int32_t n_beads, int32_t rank, int32_t world, int32_t device_id, mmx_handle *out) try {
    if (!out) return MMX_ERR_BAD_ARG;
    *out = nullptr;
    if (n_beads < 1 || n_beads > (1 << 28)) {
        g_create_error = "n_beads out of range";
        return MMX_ERR_BAD_ARG;
    }
    if (world < 1 || rank < 0 || rank >= world || world > n_beads) {
        g_create_error = "rank/world out of range";
        return MMX_ERR_BAD_ARG;
    }
    // decomposed runs: ownership in segments of kSeg beads, seg_per of them per rank
    const int nseg_real = (n_beads + kSeg - 1) / kSeg;
    const int seg_per = world > 1 ? (nseg_real + world - 1) / world : 0;
    const int slice0 = world > 1 ? seg_per * kSeg : n_beads;
    if ((long long)(world - 1) * slice0 >= n_beads) { // the last ranks would own nothing
        g_create_error = "too many ranks for this system: with slices of 62 * ceil(ceil(n_beads / 62) / world) beads some rank would own no bead";
        return MMX_ERR_BAD_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "no HIP device visible (libmmx has no CPU path)";
        return MMX_ERR_HIP;
    }
    if (device_id < 0 || device_id >= ndev) {
        g_create_error = "device_id out of range";
        return MMX_ERR_BAD_ARG;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        g_create_error = "hipGetDeviceProperties failed";
        return MMX_ERR_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", libmmx is built for gfx950 (MI355X) only";
        return MMX_ERR_HIP;
    }
    mmx_handle_s *h = new (std::nothrow) mmx_handle_s();
    if (!h) return MMX_ERR_BAD_ARG;
    h->n = n_beads;
    h->rank = rank;
    h->world = world;
    h->slice = slice0;                              // equal slices (ncclAllGather), the last one is padded
    h->seg_per = seg_per;
    h->nseg = seg_per * world;
    h->n_all = h->slice * world;
    h->own_lo = rank * h->slice;
    h->n_own = std::max(0, std::min(n_beads, h->own_lo + h->slice) - h->own_lo);
    // decomposed handles: vectors sized for a full slice (a re-assignment may hand this rank up to seg_per segments);
    // what lies beyond the owned beads is zero and stays zero
    h->n4 = (3 * (world > 1 ? h->slice : h->n_own) + 3) / 4;
    h->device = device_id;
    h->max_items = (h->n_all + kChunk - 1) / kChunk + std::min(h->n_all, h->maxcells) + 64;
    auto boot = [&]() -> int {
        HIPCHK(h, hipSetDevice(device_id));
        HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        const size_t nv = (size_t)h->n4 * 4;
        HIPCHK(h, dalloc(&h->x, nv));
        HIPCHK(h, dalloc(&h->xp, nv));
        HIPCHK(h, dalloc(&h->g, nv));
        HIPCHK(h, dalloc(&h->gp, nv));
        HIPCHK(h, dalloc(&h->d, nv));
        if (world == 1) HIPCHK(h, dalloc(&h->cell_xref, nv)); // kept cell structure: where its beads were binned (cell_reuse)
        HIPCHK(h, dalloc(&h->S, nv * MMX_M));
        HIPCHK(h, dalloc(&h->Y, nv * MMX_M));
        HIPCHK(h, dalloc(&h->pos4, (size_t)h->n_all));
        HIPCHK(h, dalloc(&h->labels, (size_t)h->n_all));
        HIPCHK(h, dalloc(&h->cell_of, (size_t)h->n_all));
        HIPCHK(h, dalloc(&h->perm, (size_t)h->n_all));
        if (h->world > 1) HIPCHK(h, dalloc(&h->xg, (size_t)3 * h->n_all));
        HIPCHK(h, dalloc(&h->count, (size_t)h->maxcells + 1));
        if (h->world > 1) HIPCHK(h, dalloc(&h->count_own, (size_t)h->maxcells + 1));
        HIPCHK(h, dalloc(&h->rank_in_cell, (size_t)h->n_all));
        HIPCHK(h, dalloc(&h->start, (size_t)h->maxcells + 1));
        HIPCHK(h, dalloc(&h->istart, (size_t)h->maxcells + 1));
        HIPCHK(h, dalloc(&h->items, (size_t)h->max_items));
        HIPCHK(h, dalloc(&h->cstart, (size_t)h->maxcells + 1));
        HIPCHK(h, dalloc(&h->okeys, (size_t)h->n_all));
        HIPCHK(h, dalloc(&h->biglist, (size_t)h->maxcells + 1));
        // clusters hold >= 1 bead; a rank only bins its owned beads and the ghosts inside its grid, but the
        // bound that needs no host knowledge is "every bead": n_all clusters
        HIPCHK(h, dalloc(&h->spos4, ((size_t)h->n_all + 1) * 8)); // + one all-padding cluster at index n_all
        {
            float4 farc[8];
            for (auto &f : farc) {
                f.x = f.y = f.z = -1e18f;
                const int w = -8 + 2;
                std::memcpy(&f.w, &w, 4);
            }
            HIPCHK(h, hipMemcpy(h->spos4 + (size_t)h->n_all * 8, farc, sizeof(farc), hipMemcpyHostToDevice));
        }
        // half-shell pair kernel: force per cluster slot and its LDS window.  Above 64 KB of LDS per workgroup the runtime
        // wants to be told; a device that refuses keeps the full-shell kernel.
        {
            h->fstride = (h->n_all + 1) * 8;
            HIPCHK(h, dalloc(&h->fsort, (size_t)3 * h->fstride));
            HIPCHK(h, dalloc(&h->sbead, (size_t)h->fstride));
            HIPCHK(h, dalloc(&h->slot_of, (size_t)std::max(world > 1 ? h->slice : h->n_own, 1)));
            // direct build: two sets (build parity) of cell populations and row totals; decomposed ranks: + the ghosts' populations and
            // the rows' ghost-cluster totals
            HIPCHK(h, dalloc(&h->dcount, (size_t)2 * ((size_t)h->maxcells + 1)));
            HIPCHK(h, dalloc(&h->drows, (size_t)2 * kDirectRowSet));
            if (world > 1) {
                HIPCHK(h, dalloc(&h->dcount_g, (size_t)2 * ((size_t)h->maxcells + 1)));
                // the halo's own stream (enqueue_build, dd_overlap); without it the decomposed evaluation stays on the one stream
                if (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_pack, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev_halo, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    h->stream2 = nullptr; // (option dd_overlap then stays off)
                }
            }
            h->n3_cap = n3_configure(kN3MaxCap);
            // a run starts at a dense cell, at a segment start or every 16 clusters: never more than cells + clusters / 16 (<= n_all /
            // 128 + cells / 16) runs; a run is one record per window pass over its candidates: the rest of n_all / 16 is theirs
            // (a list that overflows all the same voids the evaluation: KERR_N3_ITEMS)
            h->n3_max_items = std::min(h->maxcells, h->n_all) + h->n_all / 16 + 64;
            HIPCHK(h, dalloc(&h->n3_items, (size_t)h->n3_max_items));
            h->n_cus = prop.multiProcessorCount;
        }
        HIPCHK(h, dalloc(&h->cl_lo, (size_t)h->n_all * 2)); // interleaved {lo, hi} box records
        HIPCHK(h, dalloc(&h->cl_hi, (size_t)1));             // (kept as a kernel argument, unused)
        HIPCHK(h, dalloc(&h->grid, 2));
        HIPCHK(h, dalloc(&h->bbox_part, (size_t)6 * ((std::max(h->n_own, world > 1 ? h->slice : 0) + 255) / 256 + 1)));
        HIPCHK(h, dalloc(&h->part, (size_t)P_NSLOTS * kPartStride));
        HIPCHK(h, dalloc(&h->rows, (size_t)MMX_NROWSUM * kPartStride));
        HIPCHK(h, dalloc(&h->st, 1));
        HIPCHK(h, hipHostMalloc((void **)&h->st_host, sizeof(MinState), hipHostMallocDefault));
        std::memset(h->st_host, 0, sizeof(MinState));
        for (int i = 0; i < 256; ++i) {
            EventPair ep{};
            HIPCHK(h, hipEventCreate(&ep.a));
            HIPCHK(h, hipEventCreate(&ep.b));
            h->ev_pool.push_back(ep);
        }
        return MMX_OK;
    };
    int rc = boot();
    if (rc != MMX_OK) {
        g_create_error = h->err;
        mmx_destroy(h);
        return rc;
    }
    h->P.bond_r0 = 0.1f;
    *out = h;
    return MMX_OK;
} MMX_CATCH(nullptr)

int mmx_create(int32_t n_beads, int32_t device_id, mmx_handle *out) { return create_impl(n_beads, 0, 1, device_id, out); }

int mmx_create_dd(int32_t n_beads, int32_t rank, int32_t world, int32_t device_id, mmx_handle *out) try {
    return create_impl(n_beads, rank, world, device_id, out);
} MMX_CATCH(nullptr)

int mmx_dd_info(mmx_handle h, int32_t *own_lo, int32_t *n_own, int32_t *rank, int32_t *world) try {
    if (!h) return MMX_ERR_BAD_ARG;
    if (own_lo) *own_lo = h->n_own > 0 ? bead_of(h, 0) : h->own_lo;
    if (n_own) *n_own = h->n_own;
    if (rank) *rank = h->rank;
    if (world) *world = h->world;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_dd_owned_beads(mmx_handle h, int32_t *bead_ids) try {
    if (!h || !bead_ids) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    for (int li = 0; li < h->n_own; ++li) bead_ids[li] = bead_of(h, li);
    return MMX_OK;
} MMX_CATCH(h)

int mmx_comm_unique_id(uint8_t *id128) try {
    if (!id128) return MMX_ERR_BAD_ARG;
    if (!load_rccl(g_create_error)) return MMX_ERR_RCCL;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) {
        g_create_error = "ncclGetUniqueId failed";
        return MMX_ERR_RCCL;
    }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, 128);
    return MMX_OK;
} MMX_CATCH(nullptr)

int mmx_comm_init(mmx_handle h, const uint8_t *id128) try {
    if (!h || !id128) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    if (h->comm) return fail(h, MMX_ERR_STATE, "communicator already initialised");
    if (!load_rccl(h->err)) return MMX_ERR_RCCL;
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    const ncclResult_t r = g_rccl.CommInitRank(&h->comm, h->world, id, h->rank);
    if (r != ncclSuccess) {
        h->comm = nullptr;
        return fail(h, MMX_ERR_RCCL, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
    }
    return MMX_OK;
} MMX_CATCH(h)

int mmx_comm_init_local(mmx_handle *handles, int32_t world) try {
    if (!handles || world < 1) return MMX_ERR_BAD_ARG;
    for (int r = 0; r < world; ++r) {
        mmx_handle_s *h = handles[r];
        if (!h) return MMX_ERR_BAD_ARG;
        if (h->world != world || h->rank != r || h->n != handles[0]->n || h->device != handles[0]->device)
            return fail(h, MMX_ERR_BAD_ARG, "handles must be ranks 0..world-1 of one system on one device");
        if (has_comm(h)) return fail(h, MMX_ERR_STATE, "communicator already initialised");
    }
    mmx_handle_s *h0 = handles[0];
    HIPCHK(h0, hipSetDevice(h0->device));
    auto L = std::make_shared<LocalComm>();
    L->world = world;
    L->h.assign(handles, handles + world);
    for (int i = 0; i < 2 * world; ++i) {
        hipEvent_t a, b;
        HIPCHK(h0, hipEventCreateWithFlags(&a, hipEventDisableTiming));
        L->ready.push_back(a);
        HIPCHK(h0, hipEventCreateWithFlags(&b, hipEventDisableTiming));
        L->done.push_back(b);
        double *m = nullptr;
        HIPCHK(h0, dalloc(&m, (size_t)64));
        L->mailbox.push_back(m);
    }
    for (int r = 0; r < world; ++r) {
        mmx_handle_s *h = handles[r];
        for (int p = 0; p < 2; ++p) {
            std::vector<double *> boxes(world);
            for (int q = 0; q < world; ++q) boxes[q] = L->mailbox[q * 2 + p];
            HIPCHK(h, dalloc(&h->lbox[p], (size_t)world));
            HIPCHK(h, hipMemcpy(h->lbox[p], boxes.data(), sizeof(double *) * world, hipMemcpyHostToDevice));
        }
        h->lcomm = L;
        h->coll_seq = 0;
        h->coll_failed = false;
    }
    return MMX_OK;
} MMX_CATCH(nullptr)

int mmx_destroy(mmx_handle h) try {
    if (!h) return MMX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    graph_drop(h);
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    if (h->xg) (void)hipFree(h->xg);
    for (void *p : {(void *)h->v, (void *)h->xlo, (void *)h->ke_part, (void *)h->ke_out, (void *)h->formp, (void *)h->lbox[0],
                    (void *)h->lbox[1], (void *)h->okeys, (void *)h->lstart, (void *)h->biglist, (void *)h->fsort, (void *)h->n3_items, (void *)h->dd_boxes,
                    (void *)h->dd_static, (void *)h->dd_send_ids, (void *)h->dd_send_cnt, (void *)h->dd_cntmat,
                    (void *)h->dd_ghost_ids, (void *)h->dd_sendbuf, (void *)h->dd_recvbuf, (void *)h->dd_xref,
                    (void *)h->dd_grid, (void *)h->dd_occ, (void *)h->dd_maps, (void *)h->count_own, (void *)h->sbead, (void *)h->slot_of, (void *)h->dcount, (void *)h->drows, (void *)h->dcount_g,
                    (void *)h->d_seg_own, (void *)h->d_seg_local, (void *)h->mig, (void *)h->seg_cent, (void *)h->d_mig_src,
                    (void *)h->md_snap, (void *)h->cell_xref, (void *)h->slotkeys})
        if (p) (void)hipFree(p);
    if (h->dd_cnt_host) (void)hipHostFree(h->dd_cnt_host);
    if (h->seg_cent_host) (void)hipHostFree(h->seg_cent_host);
    void *bufs[] = {h->x,     h->xp,     h->g,      h->gp,    h->d,      h->S,        h->Y,         h->pos4,
                    h->labels, h->flags,  h->cf_w,   h->cell_of, h->count, h->rank_in_cell, h->start, h->istart,
                    h->perm,  h->items,  h->grid,   h->bbox_part, h->part,   h->rows,     h->st,        h->row_bead,
                    h->row_start, h->partner, h->loop_r0, h->fpart, h->epart, h->cstart, h->spos4, h->cl_lo, h->cl_hi,
                    h->chrom_of, h->chrom_lo, h->chrom_hi};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->st_host) (void)hipHostFree(h->st_host);
    for (auto &ep : h->ev_pool) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    for (auto &ep : h->ev_used) {
        (void)hipEventDestroy(ep.a);
        (void)hipEventDestroy(ep.b);
    }
    if (h->ev_pack) (void)hipEventDestroy(h->ev_pack);
    if (h->ev_halo) (void)hipEventDestroy(h->ev_halo);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MMX_OK;
} MMX_CATCH(nullptr)

int mmx_set_positions(mmx_handle h, const float *xyz) try {
    if (!h || !xyz) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    HIPCHK(h, hipSetDevice(h->device));
    for (size_t i = 0; i < (size_t)3 * h->n; ++i)
        if (!std::isfinite(xyz[i])) return fail(h, MMX_ERR_BAD_ARG, "non-finite position");
    // the L-BFGS point holds the owned beads only; xyz is always the WHOLE system [N,3]
    std::vector<float> own_x; // (must outlive the asynchronous copy: synchronised below)
    if (h->seg_owner.empty()) {
        HIPCHK(h, hipMemcpyAsync(h->x, xyz + (size_t)3 * h->own_lo, sizeof(float) * 3 * (size_t)h->n_own,
                                 hipMemcpyHostToDevice, h->stream));
    } else { // the owned segments, in local order
        own_x.resize((size_t)3 * h->n_own);
        for (int li = 0; li < h->n_own; ++li) std::memcpy(&own_x[(size_t)3 * li], xyz + (size_t)3 * bead_of(h, li), 3 * sizeof(float));
        HIPCHK(h, hipMemcpyAsync(h->x, own_x.data(), sizeof(float) * own_x.size(), hipMemcpyHostToDevice, h->stream));
    }
    if (h->xg) {
        HIPCHK(h, hipMemcpyAsync(h->xg, xyz, sizeof(float) * 3 * (size_t)h->n, hipMemcpyHostToDevice, h->stream));
        h->pos4_dirty = true;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_pos = true;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_get_positions(mmx_handle h, float *xyz) try {
    if (!h || !xyz) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    if (!h->have_pos) return fail(h, MMX_ERR_STATE, "positions not set");
    HIPCHK(h, hipSetDevice(h->device));
    if (h->world == 1) {
        HIPCHK(h, hipMemcpyAsync(xyz, h->x, sizeof(float) * 3 * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return MMX_OK;
    }
    // multi-GPU: every rank holds every bead's position in pos4 (all-gathered at the last evaluation);
    // the owned slice is taken from the L-BFGS point itself
    int rc = prepare(h);
    if (rc) return rc;
    if (has_comm(h) && !h->seg_owner.empty()) {
        // re-assigned ownership: every rank's x (local order) is all-gathered into the staging area and put back in
        // bead order with the ownership tables (identical on every rank)
        if ((rc = coll_allgather_vec(h, h->x))) return rc;
        std::vector<float> all((size_t)3 * h->slice * h->world);
        HIPCHK(h, hipMemcpyAsync(all.data(), h->mig, sizeof(float) * all.size(), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->coll_failed) return fail(h, MMX_ERR_RCCL, !h->coll_error.empty() ? h->coll_error : "loopback collective timed out");
        for (int b = 0; b < h->n; ++b) {
            const int sg = b / kSeg, q = h->seg_owner[sg];
            const size_t src = (size_t)3 * ((size_t)q * h->slice + (size_t)h->seg_lidx[sg] * kSeg + (b - sg * kSeg));
            std::memcpy(xyz + (size_t)3 * b, &all[src], 3 * sizeof(float));
        }
        return MMX_OK;
    }
    if (has_comm(h)) { // a collective: every rank calls it (between calls a rank only holds its own beads and its halo)
        hipLaunchKernelGGL(k_repack_own, dim3((h->n_own + 255) / 256), dim3(256), 0, h->stream, h->n_own, own_of(h), h->x,
                           h->labels, h->pos4);
        coll_allgather_pos4(h);
        h->dd_lists_valid = false;
    }
    std::vector<float4> p4((size_t)h->n);
    HIPCHK(h, hipMemcpyAsync(p4.data(), h->pos4, sizeof(float4) * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < h->n; ++i) {
        xyz[3 * i] = p4[i].x;
        xyz[3 * i + 1] = p4[i].y;
        xyz[3 * i + 2] = p4[i].z;
    }
    HIPCHK(h, hipMemcpy(xyz + (size_t)3 * h->own_lo, h->x, sizeof(float) * 3 * (size_t)h->n_own, hipMemcpyDeviceToHost));
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_labels(mmx_handle h, const int8_t *s) try {
    if (!h || !s) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    for (int i = 0; i < h->n; ++i)
        if (s[i] < -2 || s[i] > 2) return fail(h, MMX_ERR_BAD_ARG, "label outside {-2..2}");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(h->labels, s, (size_t)h->n, hipMemcpyHostToDevice));
    if (h->xg) h->pos4_dirty = true;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_backbone_masks(mmx_handle h, const uint8_t *flags, float bond_r0, float bond_k, float angle_theta0,
                           float angle_k, int32_t use_bond, int32_t use_angle) try {
    if (!h || !flags) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    const int n = h->n;
    for (int i = 0; i < n; ++i) {
        if ((flags[i] & 1) && i + 1 >= n) return fail(h, MMX_ERR_BAD_ARG, "bond flag on the last bead");
        if ((flags[i] & 2) && i + 2 >= n) return fail(h, MMX_ERR_BAD_ARG, "angle flag on the last two beads");
    }
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->flags) HIPCHK(h, dalloc(&h->flags, (size_t)n));
    HIPCHK(h, hipMemcpy(h->flags, flags, (size_t)n, hipMemcpyHostToDevice));
    h->P.bond_r0 = bond_r0;
    h->P.bond_k = bond_k;
    h->P.ang_th0 = angle_theta0;
    h->P.ang_k = angle_k;
    h->P.use_bond = use_bond ? 1 : 0;
    h->P.use_angle = use_angle ? 1 : 0;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_backbone(mmx_handle h, const int32_t *chr_ends, int32_t n_ends, float bond_r0, float bond_k,
                     float angle_theta0, float angle_k, int32_t use_bond, int32_t use_angle) try {
    if (!h || (!chr_ends && n_ends > 0) || n_ends < 0) return fail(h, MMX_ERR_BAD_ARG, "bad chr_ends");
    const int n = h->n;
    // model.py:629 "i not in chr_ends" ; model.py:712 "(i not in chr_ends) and (i not in chr_ends - 1)"
    std::vector<uint8_t> in_ends((size_t)n + 2, 0), in_ends_m1((size_t)n + 2, 0);
    for (int k = 0; k < n_ends; ++k) {
        const int e = chr_ends[k];
        if (e < 0 || e > n) return fail(h, MMX_ERR_BAD_ARG, "chr_ends entry outside [0, N]");
        in_ends[e] = 1;
        if (e >= 1) in_ends_m1[e - 1] = 1;
    }
    std::vector<uint8_t> flags((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        uint8_t f = 0;
        if (i <= n - 2 && !in_ends[i]) f |= 1;
        if (i <= n - 3 && !in_ends[i] && !in_ends_m1[i]) f |= 2;
        flags[i] = f;
    }
    return mmx_set_backbone_masks(h, flags.data(), bond_r0, bond_k, angle_theta0, angle_k, use_bond, use_angle);
} MMX_CATCH(h)

int mmx_set_loops(mmx_handle h, const int32_t *m, const int32_t *n, const float *r0, int32_t n_loops, float k_loop) try {
    if (!h || n_loops < 0 || (n_loops > 0 && (!m || !n || !r0))) return fail(h, MMX_ERR_BAD_ARG, "bad loop arrays");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    HIPCHK(h, hipSetDevice(h->device));
    for (int l = 0; l < n_loops; ++l) {
        if (m[l] < 0 || m[l] >= h->n || n[l] < 0 || n[l] >= h->n || m[l] == n[l])
            return fail(h, MMX_ERR_BAD_ARG, "loop anchor out of range or degenerate");
        if (!std::isfinite(r0[l])) return fail(h, MMX_ERR_BAD_ARG, "non-finite loop rest length");
    }
    h->loop_m.assign(m, m + n_loops);
    h->loop_n.assign(n, n + n_loops);
    h->loop_r0v.assign(r0, r0 + n_loops);
    h->P.loop_k = k_loop;
    return rebuild_loops(h);
} MMX_CATCH(h)

int mmx_set_excluded_volume(mmx_handle h, float eps, float sigma, float r_small, float power, float cutoff_nm) try {
    if (!h) return MMX_ERR_BAD_ARG;
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    if (!(sigma > 0.f) || !(r_small >= 0.f) || !std::isfinite(eps) || !std::isfinite(power) || !(power > 0.f))
        return fail(h, MMX_ERR_BAD_ARG, "bad excluded-volume parameters (sigma > 0, r_small >= 0, power > 0)");
    // eps == 0 is a term that contributes nothing: it is left out (the pair kernels factor eps out of the pair loop)
    h->P.use_ev = eps != 0.f ? 1 : 0;
    h->P.ev_eps = eps;
    h->P.ev_sigma = sigma;
    h->P.ev_rs = r_small;
    h->P.ev_power = power;
    h->ev_cut = cutoff_nm;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_compartments(mmx_handle h, int32_t mode, const float *E, float rc, float cutoff_nm) try {
    if (!h || !E || !(rc > 0.f)) return fail(h, MMX_ERR_BAD_ARG, "bad compartment parameters");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    auto idx = [](int si, int sj) { return (si + 2) * 5 + (sj + 2); };
    if (mode == MMX_COMP_COB) { // model.py:246-250: A = {1,2}, B = {-1,-2}
        std::memset(h->tab_cob, 0, sizeof(h->tab_cob));
        for (int a : {1, 2})
            for (int b : {1, 2}) h->tab_cob[idx(a, b)] = E[0];
        for (int a : {-1, -2})
            for (int b : {-1, -2}) h->tab_cob[idx(a, b)] = E[1];
        h->has_cob = true;
    } else if (mode == MMX_COMP_SCB) { // model.py:322-328: Ea1 (2,2), Ea2 (1,1), Eb1 (-1,-1), Eb2 (-2,-2)
        std::memset(h->tab_scb, 0, sizeof(h->tab_scb));
        h->tab_scb[idx(2, 2)] = E[0];
        h->tab_scb[idx(1, 1)] = E[1];
        h->tab_scb[idx(-1, -1)] = E[2];
        h->tab_scb[idx(-2, -2)] = E[3];
        h->has_scb = true;
    } else {
        return fail(h, MMX_ERR_BAD_ARG, "unknown compartment mode");
    }
    h->g_rc = rc;
    h->g_cut = cutoff_nm;
    h->P.use_gauss = 1;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_container(mmx_handle h, float C, float R1, float R2, const float centre[3]) try {
    if (!h || !centre) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    h->P.use_container = 1;
    h->P.sc_C = C;
    h->P.sc_R1 = R1;
    h->P.sc_R2 = R2;
    h->P.cx = centre[0];
    h->P.cy = centre[1];
    h->P.cz = centre[2];
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_lamina(mmx_handle h, float B, float R1, float R2, const float centre[3]) try {
    if (!h || !centre || !(R2 != R1)) return fail(h, MMX_ERR_BAD_ARG, "bad lamina parameters");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    h->P.use_lamina = 1;
    h->P.ibl_B = B;
    h->P.ibl_R1 = R1;
    h->P.ibl_R2 = R2;
    h->P.cx = centre[0];
    h->P.cy = centre[1];
    h->P.cz = centre[2];
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_central(mmx_handle h, float G, float R1, const float centre[3], const float *w) try {
    if (!h || !centre || !w) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->cf_w) HIPCHK(h, dalloc(&h->cf_w, (size_t)h->n));
    HIPCHK(h, hipMemcpy(h->cf_w, w, sizeof(float) * (size_t)h->n, hipMemcpyHostToDevice));
    h->P.use_central = 1;
    h->P.cf_G = G;
    h->P.cf_R1 = R1;
    h->P.cx = centre[0];
    h->P.cy = centre[1];
    h->P.cz = centre[2];
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_chromosomal_blocks(mmx_handle h, float k_C, float dE, const int32_t *chrom) try {
    if (!h || !chrom) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    if ((h->n + 255) / 256 > kPartStride) return fail(h, MMX_ERR_BAD_ARG, "too many beads for the chromosomal-block kernel");
    HIPCHK(h, hipSetDevice(h->device));
    // chromosome ids must form contiguous runs; remap them to 0..K-1 in order of appearance
    std::vector<int> id((size_t)h->n), lo, hi;
    int k = -1;
    std::vector<int> seen;
    for (int i = 0; i < h->n; ++i) {
        if (i == 0 || chrom[i] != chrom[i - 1]) {
            for (int v : seen)
                if (v == chrom[i]) return fail(h, MMX_ERR_BAD_ARG, "beads of one chromosome must be contiguous");
            seen.push_back(chrom[i]);
            ++k;
            lo.push_back(i);
            hi.push_back(i);
        }
        id[i] = k;
        hi[k] = i + 1;
    }
    for (void *p : {(void *)h->chrom_of, (void *)h->chrom_lo, (void *)h->chrom_hi})
        if (p) (void)hipFree(p);
    h->chrom_of = h->chrom_lo = h->chrom_hi = nullptr;
    HIPCHK(h, dalloc(&h->chrom_of, (size_t)h->n));
    HIPCHK(h, dalloc(&h->chrom_lo, lo.size()));
    HIPCHK(h, dalloc(&h->chrom_hi, hi.size()));
    HIPCHK(h, hipMemcpy(h->chrom_of, id.data(), sizeof(int) * id.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->chrom_lo, lo.data(), sizeof(int) * lo.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->chrom_hi, hi.data(), sizeof(int) * hi.size(), hipMemcpyHostToDevice));
    h->P.use_chb = 1;
    h->P.chb_kc = k_C;
    h->P.chb_de = dE;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_functional_form(mmx_handle h, int32_t selector, int32_t form) try {
    if (!h) return MMX_ERR_BAD_ARG;
    h->md_forces_valid = false;
    static const int n_forms[MMX_N_SELECTORS] = {2, 3, 3, 3, 4, 3, 3}; // EV, COB, SCB, CHB, LAMINA, CENTRAL, LOOPS
    if (selector < 0 || selector >= MMX_N_SELECTORS) return fail(h, MMX_ERR_BAD_ARG, "unknown term selector");
    if (form < 0 || form >= n_forms[selector]) return fail(h, MMX_ERR_BAD_ARG, "unknown functional form for this term");
    h->forms[selector] = form;
    return MMX_OK;
} MMX_CATCH(h)

int mmx_disable_term(mmx_handle h, int32_t term) try {
    if (!h) return MMX_ERR_BAD_ARG;
    h->md_forces_valid = h->grid_ready = false; // cached forces (MD) and the cell grid of the last build are stale now
    switch (term) {
    case MMX_T_EV: h->P.use_ev = 0; break;
    case MMX_T_GAUSS: h->has_cob = h->has_scb = false; h->P.use_gauss = 0; break;
    case MMX_T_BOND: h->P.use_bond = 0; break;
    case MMX_T_ANGLE: h->P.use_angle = 0; break;
    case MMX_T_LOOP: h->n_rows = 0; h->n_loops = 0; h->dd_loop_mask.clear(); h->dd_static_dirty = true; break;
    case MMX_T_CONTAINER: h->P.use_container = 0; break;
    case MMX_T_LAMINA: h->P.use_lamina = 0; break;
    case MMX_T_CENTRAL: h->P.use_central = 0; break;
    case MMX_T_CHB: h->P.use_chb = 0; break;
    default: return fail(h, MMX_ERR_BAD_ARG, "unknown term");
    }
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_option(mmx_handle h, const char *key, double value) try {
    if (!h || !key) return MMX_ERR_BAD_ARG;
    h->md_forces_valid = false; // forces cached for the MD integrator are stale now (no option changes the cell grid)
    const std::string k(key);
    if (k == "deterministic") h->deterministic = value != 0.0;
    else if (k == "profile") h->profile = (int)value;
    else if (k == "profile_nb") h->profile_nb = std::max(0, (int)value);
    else if (k == "poll_interval") h->poll_interval = std::max(1, (int)value);
    else if (k == "nb_variant") h->nb_variant = (int)value;
    else if (k == "fused_bonded") h->fused_bonded = value != 0.0;
    else if (k == "overlap_bonded") h->overlap_bonded = value != 0.0;
    else if (k == "fused_tail") {
        h->fused_tail = value != 0.0;
        // (k_tail's partials are tagged words, k_history's plain doubles, in the same buffer: neither may take the other's for its own)
        HIPCHK(h, hipMemsetAsync(h->rows, 0, sizeof(double) * (size_t)MMX_NROWSUM * kPartStride, h->stream));
    }
    else if (k == "fused_build") h->fused_build = value != 0.0;
    else if (k == "key32") h->key32 = value != 0.0;
    else if (k == "use_graph") h->use_graph = value != 0.0;
    else if (k == "dd_halo") h->dd_halo = value != 0.0;
    else if (k == "dd_skin") {
        if (!(value > 0.0)) return fail(h, MMX_ERR_BAD_ARG, "dd_skin must be positive");
        h->dd_skin = h->dd_skin_cur = (float)value;
    }
    else if (k == "dd_rebuild_every") {
        if (!(value >= 1.0)) return fail(h, MMX_ERR_BAD_ARG, "dd_rebuild_every must be >= 1");
        h->dd_every = (int)value;
        h->dd_adaptive = 0; // a fixed lifetime was asked for (option dd_adaptive, set afterwards, turns the choice by the polls back on)
        h->dd_lists_valid = false;
    }
    else if (k == "dd_adaptive") {
        h->dd_adaptive = value != 0.0;
        h->dd_k_cur = 1;
        h->dd_lists_valid = false;
    }
    else if (k == "dd_overlap") h->dd_overlap = (value != 0.0 && h->stream2 && h->ev_pack && h->ev_halo) ? (int)value : 0;
    else if (k == "dd_overlap_go") h->dd_overlap_go = (int)value;
    else if (k == "graph_evals") h->graph_evals = std::max(2, 2 * ((int)value / 2));
    else if (k == "inject_fault") h->inject_fault = (int)value;
    else if (k == "n3_long_items") h->n3_long_items = value < 0.0 ? -1 : value == 2.0 ? 2 : value != 0.0;
    else if (k == "dd_freeze") h->dd_frozen = value != 0.0;
    else if (k == "cell_slots") h->cell_slots = value != 0.0;
    else if (k == "dd_split") h->dd_split = value != 0.0;
    else if (k == "n3_pass_records") h->n3_pass_records = value != 0.0;
    else if (k == "n3_slice_cap") h->n3_slice_cap = std::max(0, std::min((int)value, (int)kN3MaxCap));
    else if (k == "cell_reuse") { h->cell_reuse = value != 0.0; h->reuse_K = 1; h->struct_valid = false; }
    else if (k == "cell_reuse_factor") h->reuse_factor = value > 0.0 ? (float)std::max(1.0, value) : 0.f;
    else if (k == "cell_wide_below") h->wide_below = std::max(0.0, value);
    else if (k == "cell_edge_auto") { h->cell_edge_auto = value != 0.0; h->edge_auto = 1.f; }
    else if (k == "cell_edge_scale") { h->cell_edge_scale = (float)std::max(1.0, value); h->grid_ready = false; }
    else if (k == "dd_spatial") h->dd_spatial = value != 0.0;
    else if (k == "dd_reassign_first") h->dd_reassign_first = std::max(1, (int)value);
    else if (k == "dd_reassign_max") h->dd_reassign_max = std::max(1, (int)value);
    else return fail(h, MMX_ERR_BAD_ARG, "unknown option " + k);
    return MMX_OK;
} MMX_CATCH(h)

int mmx_get_option(mmx_handle h, const char *key, double *value) try {
    if (!h || !key || !value) return MMX_ERR_BAD_ARG;
    const std::string k(key);
    if (k == "deterministic") *value = h->deterministic;
    else if (k == "profile") *value = h->profile;
    else if (k == "profile_nb") *value = h->profile_nb;
    else if (k == "poll_interval") *value = h->poll_interval;
    else if (k == "nb_variant") *value = h->nb_variant;
    else if (k == "fused_bonded") *value = h->fused_bonded;
    else if (k == "overlap_bonded") *value = h->overlap_bonded;
    else if (k == "fused_tail") *value = h->fused_tail;
    else if (k == "fused_build") *value = h->fused_build;
    else if (k == "key32") *value = h->key32;
    else if (k == "use_graph") *value = h->use_graph;
    else if (k == "inject_fault") *value = h->inject_fault;
    else if (k == "n3_launches") *value = (double)h->n3_launches;   // read-only: how often the half-shell kernel ran
    else if (k == "dd_halo") *value = h->dd_halo;
    else if (k == "dd_skin") *value = h->dd_skin;
    else if (k == "dd_skin_now") *value = h->dd_skin_cur;
    else if (k == "dd_rebuild_every") *value = h->dd_every;
    else if (k == "dd_adaptive") *value = h->dd_adaptive;
    else if (k == "dd_overlap") *value = h->dd_overlap;
    else if (k == "dd_overlapped") *value = (double)h->dd_overlapped; // read-only: evaluations whose halo ran beside the owned build
    else if (k == "dd_move_seen") *value = h->dd_move_seen;
    else if (k == "dd_lists_serve") *value = dd_K(h); // read-only: evaluations per set of ghost lists in force
    else if (k == "dd_halts") *value = (double)h->dd_halts;
    else if (k == "dd_capacity_updates") *value = (double)h->dd_cap_updates;
    else if (k == "dd_sync_rebuilds") *value = (double)h->dd_sync_rebuilds;
    else if (k == "dd_ghost_slots") *value = h->dd_nghost;
    else if (k == "dd_ghosts") { // read-only statistics of a decomposed run: ghosts listed for this rank at the last poll
        double g = 0.0;
        if (h->dd_cnt_host)
            for (int q = 0; q < h->world; ++q)
                if (q != h->rank) g += h->dd_cnt_host[(size_t)h->world * q + h->rank];
        *value = g;
    }
    else if (k == "dd_redecompositions") *value = (double)h->dd_redecompositions;
    else if (k == "dd_exchanges") *value = (double)h->dd_exchanges;
    else if (k == "dd_bytes_sent") *value = (double)h->dd_bytes_sent;
    else if (k == "n3_long_items") *value = h->n3_long_items;
    else if (k == "dd_us_needmap_allgather" || k == "dd_us_halo_exchange" || k == "dd_us_allreduce") {
        // mean HIP-event time (us) of that collective over the sampled evaluations since the handle was created; 0: no sample
        const int w = k == "dd_us_needmap_allgather" ? kCollNeedmap : k == "dd_us_halo_exchange" ? kCollHalo : kCollAllreduce;
        *value = h->coll_samples[w] ? h->coll_ns[w] / (double)h->coll_samples[w] / 1e3 : 0.0;
    }
    else if (k == "dd_collective_samples") *value = (double)h->coll_samples[kCollAllreduce];
    else if (k == "cell_slots") *value = h->cell_slots;
    else if (k == "dd_split") *value = h->dd_split;
    else if (k == "n3_pass_records") *value = h->n3_pass_records;
    else if (k == "n3_slice_cap") *value = h->n3_slice_cap;
    else if (k == "cell_slot_halts") *value = (double)h->slot_halts;
    else if (k == "cell_reuse") *value = h->cell_reuse;
    else if (k == "direct_builds") *value = (double)h->direct_builds; // read-only: full builds that went through k_build_direct
    else if (k == "cell_builds") *value = (double)h->cell_builds;     // read-only: tracked full builds / evaluations on a kept
    else if (k == "cell_reuses") *value = (double)h->cell_reuses;     // structure / evaluations voided because it had gone stale
    else if (k == "cell_stale_halts") *value = (double)h->cell_stale_halts;
    else if (k == "cell_reuse_K") *value = h->reuse_K;
    else if (k == "md_step") *value = (double)h->md_step; // read-only: MD steps integrated (and not taken back) so far
    else if (k == "dd_spatial") *value = h->dd_spatial;
    else if (k == "dd_reassign_first") *value = h->dd_reassign_first;
    else if (k == "dd_reassign_max") *value = h->dd_reassign_max;
    else if (k == "dd_reassignments") *value = (double)h->dd_reassignments;
    else if (k == "dd_reassign_attempts") *value = (double)h->dd_reassign_attempts;
    else if (k == "dd_segments_moved") *value = (double)h->dd_segments_moved;
    else if (k == "n_clusters") *value = h->st_host ? h->st_host->n_clusters : 0;   // read-only: the last cell build
    else if (k == "n_cells") *value = h->st_host ? h->st_host->ncells : 0;
    else if (k == "max_per_cell") *value = h->st_host ? h->st_host->max_per_cell : 0;
    else if (k == "cell_edge") *value = h->st_host ? h->st_host->cell_edge : 0.0;
    else if (k == "slot_cap") *value = h->slot_cap;       // read-only: rows of the slot table as cut at the last poll
    else if (k == "slot_cells") *value = h->slot_cells;
    else if (k == "kernel_error") *value = h->st_host ? h->st_host->kernel_error : 0;
    else if (k == "n3_items") *value = h->st_host ? h->st_host->n3_items : 0;
    else if (k == "order_fallbacks") *value = h->st_host ? h->st_host->order_fallbacks : 0; // read-only diagnostic
    else return fail(h, MMX_ERR_BAD_ARG, "unknown option " + k);
    return MMX_OK;
} MMX_CATCH(h)

int mmx_compute(mmx_handle h, float *forces_out, double *energy_terms_out) try {
    if (!h) return MMX_ERR_BAD_ARG;
    h->md_forces_valid = false;
    int rc = prepare(h);
    if (rc) return rc;
    std::memset(h->st_host, 0, sizeof(MinState));
    h->st_host->phase = PH_IDLE;
    if ((rc = push_state(h))) return rc;
    if ((rc = prime_items(h))) return rc;
    enqueue_eval(h, PACK_PLAIN, FOLD_PLAIN);
    if ((rc = pull_state(h))) return rc;
    HIPCHK(h, hipGetLastError());
    prof_collect(h, nullptr);
    if ((rc = kernel_error_rc(h))) return rc;
    if (energy_terms_out)
        for (int t = 0; t < MMX_N_TERMS; ++t) energy_terms_out[t] = h->st_host->eterms[t];
    if (forces_out) {
        std::vector<float> g((size_t)3 * h->n_own); // forces of the owned beads [n_own,3]
        HIPCHK(h, hipMemcpy(g.data(), h->g, sizeof(float) * g.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < g.size(); ++i) forces_out[i] = -g[i];
    }
    const double f = h->st_host->ftrial;
    if (!(f - f == 0.0)) return fail(h, MMX_ERR_NAN, "non-finite energy");
    return MMX_OK;
} MMX_CATCH(h)

int mmx_minimize(mmx_handle h, double tolerance, int32_t max_iters, mmx_stats *out) try {
    if (!h || max_iters < 0 || !(tolerance >= 0.0)) return fail(h, MMX_ERR_BAD_ARG, "bad minimize arguments");
    h->md_forces_valid = false;
    h->md_active = false;
    // A minimization from the lattice adds 10-25 % of ghosts within its first ten iterations: decomposed runs start
    // with roomier messages (1/4) and short intervals between the polls that resize them (4, 8, 16, ... evaluations)
    h->dd_slack_div = 4;
    int rc = prepare(h);
    if (rc) return rc;
    h->fsort_dirty = true; // until this call has ended in order
    h->direct_ok = true;
    if ((rc = direct_reset(h))) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    mmx_stats local;
    std::memset(&local, 0, sizeof(local));
    for (int k = 0; k < MMX_N_KERNELS; ++k) h->launches[k] = 0;

    // A build before the first evaluation (one launch sequence + one host round trip) teaches the host the cluster and
    // item counts of this state and, on a decomposed run, builds the ghost lists.  A single-domain handle that has
    // built cells since anything about the system last changed skips it: the grid of its last build is a valid (if
    // slightly stale) grid for these positions -- k_cell_count's clamping argument -- and the counts are known.
    if (!h->grid_ready || h->world > 1 || h->n_own != h->n || h->last_clusters <= 0) {
        std::memset(h->st_host, 0, sizeof(MinState));
        h->st_host->phase = PH_IDLE;
        if ((rc = push_state(h))) return rc;
        if ((rc = prime_items(h))) return rc;
    }

    std::memset(h->st_host, 0, sizeof(MinState));
    h->st_host->phase = PH_INIT;
    h->st_host->k = 1;
    h->st_host->max_iters = max_iters;
    h->st_host->tolerance = tolerance; // epsilon = tolerance / max(1, rms |x_i|): formed on the device (controller_decide)
    h->st_host->n_total = (double)h->n;
    h->st_host->n_items = h->last_items > 0 ? h->last_items : 0;
    if ((rc = push_state(h))) return rc;
    const size_t nv = (size_t)h->n4 * 4;
    HIPCHK(h, hipMemsetAsync(h->S, 0, sizeof(float) * nv * MMX_M, h->stream));
    HIPCHK(h, hipMemsetAsync(h->Y, 0, sizeof(float) * nv * MMX_M, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d, 0, sizeof(float) * nv, h->stream));

    h->dd_skin_cur = h->dd_skin;
    h->dd_k_cur = 1; // (dd_adaptive: exact lists until the polls have seen how far the trial moves go)
    // every exit from here on goes through leave(): per-evaluation profiling back to per-slot sampling, graph dropped
    auto leave = [&](int code) {
        h->prof_eval = -1;
        graph_drop(h);
        return code;
    };
    h->prof_eval = 1; // the first evaluation is a profiling sample when profiling is on
    enqueue_eval(h, PACK_PLAIN, FOLD_MIN);
    // (no poll here: the trial evaluations go out behind it at once; MinState::f0 keeps the initial energy)
    // Trial evaluations: pairs of them are replayed from a hipGraph (single-GPU runs; option use_graph), except the
    // every profile-th evaluation, which goes out launch by launch with HIP events around every kernel slot.
    const bool graphs = h->use_graph && !has_comm(h) && h->world == 1;
    bool have_graph = false;
    GraphKey gkey{};
    long long eval_no = 1;
    int ramp = use_halo(h) ? 4 : h->poll_interval, quiet_polls = 0;
    // decomposed runs: the segments are re-assigned to the ranks while the structure deforms (dd_reassign) -- attempted at
    // polls, first after dd_reassign_first evaluations, then at doubling intervals; an attempt that changed the ownership
    // is followed by a synchronous rebuild of the ghost lists
    int reassign_interval = h->dd_reassign_first;
    long long next_reassign = (h->dd_spatial && use_halo(h)) ? reassign_interval : std::numeric_limits<long long>::max();
    bool sync_lists = false;
    while (h->st_host->phase != PH_DONE) {
        int batch = std::min(h->poll_interval, ramp);
        ramp = std::min(2 * ramp, std::max(h->poll_interval, 1));
        if (max_iters > 0) batch = std::max(1, std::min(batch, max_iters - h->st_host->iters));
        if (graphs && (!have_graph || !(gkey == graph_key(h)))) {
            gkey = graph_key(h);
            have_graph = graph_capture(h);
        }
        for (int b = 0; b < batch;) {
            const bool s0 = h->profile > 0 && eval_no % h->profile == 0;
            bool sampled = false; // a profiling sample inside the span a replay would cover?
            for (int k = 0; k < h->graph_evals; ++k) sampled = sampled || (h->profile > 0 && (eval_no + k) % h->profile == 0);
            if (have_graph && !sampled && b + h->graph_evals <= batch && (h->build_idx & 1) == h->gkey_parity && graph_replay(h)) {
                b += h->graph_evals;
                eval_no += h->graph_evals;
                continue;
            }
            h->prof_eval = s0 ? 1 : (h->profile > 0 && h->profile_nb > 0 && eval_no % h->profile_nb == 0) ? 2 : 0;
            enqueue_eval(h, PACK_MOVE, FOLD_MIN, sync_lists ? 1 : dd_schedule(h));
            sync_lists = false;
            if (h->dd_rc != MMX_OK) return leave(h->dd_rc);
            ++b;
            ++eval_no;
        }
        if ((rc = pull_state(h))) return leave(rc);
        if (h->st_host->phase == PH_HALT) quiet_polls = 0;
        else if (++quiet_polls >= 8 && h->dd_slack_div < 8) { // the lists have settled: tighter messages again
            h->dd_slack_div *= 2;
            quiet_polls = 0;
        }
        while (h->st_host->phase == PH_HALT) {
            // decomposed run: an evaluation found a ghost list out of date (a bead beyond half the skin, dd_every > 1) or
            // longer than its message, and decided nothing.  New lists and capacities at the trial point (the pack of the
            // repeated evaluation writes it again), then the evaluation itself.
            h->dd_halts++;
            if (std::getenv("MMX_DD_DEBUG")) { // which list outgrew which message
                std::vector<int> cnt(h->world);
                (void)hipMemcpy(cnt.data(), h->dd_send_cnt, sizeof(int) * h->world, hipMemcpyDeviceToHost);
                std::string line = "[dd] rank " + std::to_string(h->rank) + " halt at eval " + std::to_string(h->st_host->evals) +
                                   " reason=" + std::to_string(h->st_host->halt_reason) + " slack_div=" + std::to_string(h->dd_slack_div) + ":";
                for (int q = 0; q < h->world; ++q)
                    if (q != h->rank) line += " " + std::to_string(cnt[q]) + "/" + std::to_string(h->dd_scap.cap[q]);
                std::fprintf(stderr, "%s\n", line.c_str());
            }
            // (both policies read the reason k_decide_reduced recorded from the ALL-REDUCED flags: every rank must arrive at the
            // same capacities and the same skin)
            if (h->st_host->halt_reason & 8) { // the slot table was too small for this state: a larger one for the repeat
                h->slot_halts++;
                h->struct_valid = false;
                h->st_host->cell_stale = 0;
                if ((rc = ensure_slots(h, true))) return leave(rc);
            }
            if (h->st_host->halt_reason & 16) { // the grid has outgrown the direct build: the scan-based one for the rest of the call
                h->direct_ok = false;
                h->struct_valid = false;
                h->st_host->cell_stale = 0;
            }
            if ((rc = direct_reset(h))) return leave(rc);
            if (h->st_host->halt_reason & 4) { // the kept cell structure went stale: build anew for the repeat, keep structures for half as long
                h->cell_stale_halts++;
                h->struct_valid = false;
                h->reuse_K = std::max(1, h->reuse_K / 2);
                h->st_host->cell_stale = 0;
            }
            if (h->st_host->halt_reason & 2) h->dd_slack_div = std::max(1, h->dd_slack_div / 2); // lists grow faster than assumed
            if (h->st_host->halt_reason & 1) { // the skin did not last: a wider one -- or, option dd_adaptive, fewer evaluations per set of lists
                if (h->dd_adaptive) h->dd_k_cur = std::max(1, h->dd_k_cur / 2);
                else h->dd_skin_cur = std::min(2.f * h->dd_skin_cur, 0.8f);
            }
            ramp = 4; // what follows a halt is polled (and its messages resized) at short intervals again
            h->st_host->phase = h->st_host->halt_phase;
            h->st_host->dd_stale = 0;
            h->st_host->dd_overflow = 0;
            h->st_host->halt_reason = 0;
            if ((rc = push_state(h))) return leave(rc);
            h->prof_eval = 0;
            // (a void evaluation may have skipped cells -- a slot table too small --, whose cluster slots k_tail then never
            //  visited: whatever the pair kernel left there must not leak into the repeat)
            if (h->fsort && h->fused_tail) HIPCHK(h, hipMemsetAsync(h->fsort, 0, sizeof(float) * 3 * (size_t)h->fstride, h->stream));
            enqueue_eval(h, PACK_MOVE, FOLD_MIN, 1);
            if (h->dd_rc != MMX_OK) return leave(h->dd_rc);
            if ((rc = pull_state(h))) return leave(rc);
        }
        if (eval_no >= next_reassign && h->st_host->phase != PH_DONE) {
            bool changed = false;
            if ((rc = dd_reassign(h, changed))) return leave(rc);
            reassign_interval = std::min(2 * reassign_interval, std::max(h->dd_reassign_max, 1));
            next_reassign = eval_no + reassign_interval;
            if (changed) {
                sync_lists = true;
                ramp = 4; // new lists: their messages are sized at short intervals again
            }
        }
        if ((int)h->ev_used.size() > 200) prof_collect(h, &local);
    }
    h->prof_eval = -1;
    graph_drop(h);
    HIPCHK(h, hipGetLastError());
    const MinState &s = *h->st_host;
    if (s.status <= MMX_MIN_LS_MIN_STEP && s.evals > 1) {
        // line search failed (-3..-6): liblbfgs reverts to the last accepted point
        const int g4 = std::min((h->n4 + 255) / 256, 1024);
        hipLaunchKernelGGL(k_copy4, dim3(g4), dim3(256), 0, h->stream, h->n4, (const float4 *)h->xp, (float4 *)h->x);
        hipLaunchKernelGGL(k_copy4, dim3(g4), dim3(256), 0, h->stream, h->n4, (const float4 *)h->gp, (float4 *)h->g);
        if (h->world > 1) { // every rank reverted its own slice: what the others hold of it (pos4) is the rejected trial point
            hipLaunchKernelGGL(k_repack_own, dim3((h->n_own + 255) / 256), dim3(256), 0, h->stream, h->n_own, own_of(h),
                               h->x, h->labels, h->pos4);
            if (has_comm(h) && h->seg_owner.empty()) coll_allgather_pos4(h); // (re-assigned ownership: only ever with the halo,
                                                                             //  whose next call starts from a synchronous rebuild)
            h->dd_lists_valid = false;
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    prof_collect(h, &local);
    local.iterations = s.iters;
    local.evaluations = s.evals;
    local.status = s.status;
    local.n_beads = h->n;
    local.e_initial = s.f0;
    local.e_final = s.fx;
    local.gnorm_final = s.gnorm;
    local.xnorm_final = s.xnorm;
    local.rms_force = s.gnorm / std::sqrt((double)h->n);
    for (int t = 0; t < MMX_N_TERMS; ++t) local.energy_terms[t] = s.eterms_acc[t];
    for (int k = 0; k < MMX_N_KERNELS; ++k) local.kernel_launches[k] = h->launches[k];
    local.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out) *out = local;
    if (s.status == MMX_MIN_KERNEL || s.kernel_error) return kernel_error_rc(h);
    h->fsort_dirty = false;
    if (s.status == MMX_MIN_NAN) return fail(h, MMX_ERR_NAN, "non-finite energy during minimization");
    return MMX_OK;
} MMX_CATCH(h)

// ---- molecular dynamics ---------------------------------------------------------------
static const double kBoltz = 0.008314462618; // kJ/(mol K), the constant of model.py:967

static void md_refresh(mmx_handle_s *h) {
    MdParams &M = h->md;
    const double dt = h->md_dt, kT = kBoltz * h->md_temp, m = h->md_mass, gam = h->md_friction;
    M.dt = (float)dt;
    M.inv_dt = (float)(1.0 / dt);
    M.key0 = (uint32_t)h->md_seed;
    M.key1 = (uint32_t)(h->md_seed >> 32);
    if (h->md_kind == MD_LANGEVIN) {
        const double a = std::exp(-gam * dt);
        M.vscale = (float)a;
        M.fscale = (float)((gam > 0.0 ? (1.0 - a) / gam : dt) / m);
        M.noise = (float)std::sqrt(kT * (1.0 - a * a) / m);
    } else if (h->md_kind == MD_VERLET || h->md_kind == MD_AMD) {
        M.vscale = 1.f;
        M.fscale = (float)(dt / m);
        M.noise = 0.f;
    } else {
        M.vscale = 0.f;
        M.fscale = (float)(dt / (gam * m));
        M.noise = (float)std::sqrt(2.0 * kT * dt / (gam * m));
    }
    M.amd_alpha = h->amd_alpha;
    M.amd_e = h->amd_e;
}

int mmx_md_configure(mmx_handle h, int32_t integrator, double dt_ps, double temperature_K, double friction_per_ps,
                     double mass_amu, uint64_t seed) try {
    if (!h) return MMX_ERR_BAD_ARG;
    if (integrator < MMX_INT_LANGEVIN || integrator > MMX_INT_AMD)
        return fail(h, MMX_ERR_BAD_ARG, "integrator must be MMX_INT_LANGEVIN, MMX_INT_VERLET, MMX_INT_BROWNIAN or MMX_INT_AMD");
    if (!(dt_ps > 0.0) || !(mass_amu > 0.0) || !(temperature_K >= 0.0) || !(friction_per_ps >= 0.0) ||
        (integrator == MMX_INT_BROWNIAN && !(friction_per_ps > 0.0)))
        return fail(h, MMX_ERR_BAD_ARG, "bad integrator parameters");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t nv = (size_t)h->n4 * 4;
    if (!h->v) HIPCHK(h, dalloc(&h->v, nv));
    if (!h->xlo) HIPCHK(h, dalloc(&h->xlo, nv));
    if (!h->ke_part) HIPCHK(h, dalloc(&h->ke_part, (size_t)1024));
    if (!h->ke_out) HIPCHK(h, dalloc(&h->ke_out, (size_t)2));
    h->md_kind = integrator;
    h->md_dt = dt_ps;
    h->md_temp = temperature_K;
    h->md_friction = friction_per_ps;
    h->md_mass = mass_amu;
    h->md_seed = seed;
    h->md_step = 0;
    h->md_configured = true;
    h->md_forces_valid = false;
    md_refresh(h);
    return MMX_OK;
} MMX_CATCH(h)

int mmx_md_set_amd(mmx_handle h, double alpha_kj_per_mol, double e_kj_per_mol) try {
    if (!h) return MMX_ERR_BAD_ARG;
    if (!(alpha_kj_per_mol > 0.0) || !(e_kj_per_mol - e_kj_per_mol == 0.0))
        return fail(h, MMX_ERR_BAD_ARG, "aMD needs alpha > 0 and a finite E");
    h->amd_alpha = alpha_kj_per_mol;
    h->amd_e = e_kj_per_mol;
    md_refresh(h);
    return MMX_OK;
} MMX_CATCH(h)

int mmx_md_set_velocities_to_temperature(mmx_handle h, double temperature_K, uint64_t seed) try {
    if (!h || !(temperature_K >= 0.0)) return fail(h, MMX_ERR_BAD_ARG, "bad temperature");
    if (!h->md_configured) return fail(h, MMX_ERR_STATE, "mmx_md_configure first");
    HIPCHK(h, hipSetDevice(h->device));
    const float sigma = (float)std::sqrt(kBoltz * temperature_K / h->md_mass);
    hipLaunchKernelGGL(k_md_init_velocities, dim3((h->n_own + 255) / 256), dim3(256), 0, h->stream, h->n_own,
                       own_of(h), sigma, (uint32_t)seed, (uint32_t)(seed >> 32), h->v);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return MMX_OK;
} MMX_CATCH(h)

int mmx_set_velocities(mmx_handle h, const float *vel) try {
    if (!h || !vel) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    if (!h->md_configured) return fail(h, MMX_ERR_STATE, "mmx_md_configure first");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<float> own_v((size_t)3 * std::max(h->n_own, 1));
    for (int li = 0; li < h->n_own; ++li) std::memcpy(&own_v[(size_t)3 * li], vel + (size_t)3 * bead_of(h, li), 3 * sizeof(float));
    HIPCHK(h, hipMemcpy(h->v, own_v.data(), sizeof(float) * 3 * (size_t)h->n_own, hipMemcpyHostToDevice));
    return MMX_OK;
} MMX_CATCH(h)

int mmx_get_velocities(mmx_handle h, float *vel) try {
    if (!h || !vel) return fail(h, MMX_ERR_BAD_ARG, "null argument");
    if (!h->md_configured) return fail(h, MMX_ERR_STATE, "mmx_md_configure first");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<float> own_v((size_t)3 * std::max(h->n_own, 1));
    HIPCHK(h, hipMemcpy(own_v.data(), h->v, sizeof(float) * 3 * (size_t)h->n_own, hipMemcpyDeviceToHost));
    for (int li = 0; li < h->n_own; ++li) std::memcpy(vel + (size_t)3 * bead_of(h, li), &own_v[(size_t)3 * li], 3 * sizeof(float));
    return MMX_OK;
} MMX_CATCH(h)

int mmx_md_step(mmx_handle h, int32_t n_steps, mmx_md_stats *out) try {
    if (!h || n_steps < 0) return fail(h, MMX_ERR_BAD_ARG, "bad step count");
    if (!h->md_configured) return fail(h, MMX_ERR_STATE, "mmx_md_configure first");
    h->md_active = true;
    if (h->dd_adaptive && h->dd_k_cur != 1) { // (the polls of a minimization size the lists' lifetime by its trial moves: MD steps get exact lists)
        h->dd_k_cur = 1;
        h->dd_lists_valid = false;
    }
    int rc = prepare(h);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    // The forces of the current positions are cached between calls (md_forces_valid), and so must be their potential
    // energy: the aMD integrator reads it (st->ftrial) in the first pack of this call, before any evaluation.
    const bool keep = h->md_forces_valid;
    const double ftrial_prev = h->st_host->ftrial;
    double eterms_prev[9];
    for (int t = 0; t < 9; ++t) eterms_prev[t] = h->st_host->eterms[t];
    std::memset(h->st_host, 0, sizeof(MinState));
    h->st_host->phase = PH_IDLE;
    if (keep) {
        h->st_host->ftrial = ftrial_prev;
        for (int t = 0; t < 9; ++t) h->st_host->eterms[t] = eterms_prev[t];
    }
    if ((rc = push_state(h))) return rc;
    if (!h->md_forces_valid) {
        // positions / parameters changed: low-order position bits restart at zero, forces are recomputed
        HIPCHK(h, hipMemsetAsync(h->xlo, 0, sizeof(float) * 4 * (size_t)h->n4, h->stream));
        if ((rc = prime_items(h))) return rc;
        enqueue_eval(h, PACK_PLAIN, FOLD_PLAIN);
        h->md_forces_valid = true;
    }
    const int poll_every = 8 * std::max(1, h->poll_interval);
    // Decomposed runs: a ghost list that goes out of date (a bead beyond half the skin, dd_every > 1) or outgrows its message
    // is only seen at a poll, after positions and velocities have been integrated with forces that lacked ghosts -- on
    // every rank, through the stale ghosts.  Those steps are VOID: x, v, xlo and the step counter are kept as of the last
    // poll that found the lists in order and are put back before the error is returned, so the retry the message asks for
    // repeats exactly those steps (the noise of a bead at a step is a function of (seed, bead, step) only).
    const bool snap = use_halo(h);
    const size_t nvf = (size_t)h->n4 * 4;
    auto md_snapshot = [&]() -> int {
        if (!h->md_snap) HIPCHK(h, dalloc(&h->md_snap, 3 * nvf));
        HIPCHK(h, hipMemcpyAsync(h->md_snap, h->x, sizeof(float) * nvf, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->md_snap + nvf, h->v, sizeof(float) * nvf, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->md_snap + 2 * nvf, h->xlo, sizeof(float) * nvf, hipMemcpyDeviceToDevice, h->stream));
        h->md_snap_step = h->md_step;
        return MMX_OK;
    };
    auto md_rollback = [&]() -> int {
        HIPCHK(h, hipMemcpyAsync(h->x, h->md_snap, sizeof(float) * nvf, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->v, h->md_snap + nvf, sizeof(float) * nvf, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->xlo, h->md_snap + 2 * nvf, sizeof(float) * nvf, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->md_step = h->md_snap_step;
        h->md_forces_valid = false;
        h->dd_lists_valid = false;
        h->md_sync_next = true; // the step that failed is retried with lists and capacities made for its own positions
        return MMX_OK;
    };
    if (snap && (rc = md_snapshot())) return rc;
    for (int s = 0; s < n_steps; ++s) {
        // pair energies are only needed where they are read: at the last step (report) and at the polls (NaN check)
        // aMD reads the potential energy of the current positions at every step
        const bool report = h->md_kind == MD_AMD || s + 1 == n_steps || (s + 1) % poll_every == 0;
        h->nb_skip_energy = !report;
        // decomposed runs: a call that has no ghost lists builds them synchronously at its first step; afterwards they are
        // rebuilt on the stream like the minimizer's (every dd_rebuild_every-th step; checked at the polls all the same)
        const int redecomp = !use_halo(h) ? 0 : (!h->dd_lists_valid || h->md_sync_next) ? 1 : dd_schedule(h);
        h->md_sync_next = false;
        enqueue_eval(h, PACK_MD, report ? FOLD_PLAIN : FOLD_NONE, redecomp);
        h->nb_skip_energy = false;
        h->md_step++;
        if (h->dd_rc != MMX_OK) return h->dd_rc;
        if ((s + 1) % poll_every == 0) { // bound the queue depth; learn the cluster count
            if ((rc = pull_state(h))) return rc;
            if ((rc = kernel_error_rc(h))) {
                h->md_forces_valid = false;
                return rc;
            }
            if (h->st_host->sums[kSumStale] > 0.5 || h->st_host->sums[kSumOverflow] > 0.5) { // all-reduced: every rank sees it
                if (snap && (rc = md_rollback())) return rc;
                h->md_forces_valid = false;
                h->dd_lists_valid = false;
                return fail(h, MMX_ERR_STATE, "a ghost list of the decomposed run went out of date during MD (a bead moved more "
                                              "than half of dd_skin between two rebuilds, or a list outgrew its message): "
                                              "the steps since the last poll are void and have been taken back; call again "
                                              "for the steps not yet reported (the lists are rebuilt)");
            }
            const double f = h->st_host->ftrial;
            if (!(f - f == 0.0)) {
                h->md_forces_valid = false;
                return fail(h, MMX_ERR_NAN, "non-finite energy during MD (step too large?)");
            }
            if (snap && (rc = md_snapshot())) return rc; // the lists were in order up to here
        }
    }
    // kinetic energy with the half-step shift OpenMM applies to leap-frog velocities (none for brownian)
    const int gk = std::min((h->n_own + 255) / 256, 1024);
    const double shift = (h->md_kind == MD_BROWNIAN || h->md_kind == MD_AMD) ? 0.0 : 0.5 * h->md_dt;
    hipLaunchKernelGGL(k_md_kinetic, dim3(gk), dim3(256), 0, h->stream, h->n_own, h->v, h->g, (float)(shift / h->md_mass),
                       0.5 * h->md_mass, h->ke_part);
    hipLaunchKernelGGL(k_md_kinetic_fold, dim3(1), dim3(256), 0, h->stream, gk, h->ke_part, h->ke_out);
    if (has_comm(h)) coll_allreduce(h, h->ke_out, 1);
    if ((rc = pull_state(h))) return rc;
    if ((rc = kernel_error_rc(h))) {
        h->md_forces_valid = false;
        return rc;
    }
    if (h->st_host->sums[kSumStale] > 0.5 || h->st_host->sums[kSumOverflow] > 0.5) {
        if (snap && (rc = md_rollback())) return rc;
        h->md_forces_valid = false;
        h->dd_lists_valid = false;
        return fail(h, MMX_ERR_STATE, "a ghost list of the decomposed run went out of date during MD: the steps since the last "
                                      "poll are void and have been taken back; call again for the steps not yet reported "
                                      "(the lists are rebuilt)");
    }
    double ke = 0.0;
    HIPCHK(h, hipMemcpy(&ke, h->ke_out, sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(h, hipGetLastError());
    prof_collect(h, nullptr);
    const double f = h->st_host->ftrial;
    if (!(f - f == 0.0) || !(ke - ke == 0.0)) {
        h->md_forces_valid = false;
        return fail(h, MMX_ERR_NAN, "non-finite energy during MD (step too large?)");
    }
    if (out) {
        std::memset(out, 0, sizeof(*out));
        out->step_count = (int64_t)h->md_step;
        out->n_steps = n_steps;
        out->integrator = h->md_kind;
        out->potential = f;
        out->kinetic = ke;
        out->temperature = 2.0 * ke / (3.0 * (double)h->n * kBoltz);
        for (int t = 0; t < MMX_N_TERMS; ++t) out->energy_terms[t] = h->st_host->eterms[t];
        out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return MMX_OK;
} MMX_CATCH(h)

int mmx_time_kernel(mmx_handle h, int32_t kernel, int32_t reps, double *mean_us, double *algorithmic_bytes) try {
    if (!h || reps < 1 || !mean_us) return fail(h, MMX_ERR_BAD_ARG, "bad arguments");
    h->md_forces_valid = false;
    if ((kernel < MMX_K_CELL_BUILD || kernel > MMX_K_CONFINE) && kernel != MMX_K_FORCES && kernel != MMX_K_DD_LISTS)
        return fail(h, MMX_ERR_BAD_ARG, "mmx_time_kernel covers slots 0..4 and MMX_K_FORCES; L-BFGS slots are timed live (option profile)");
    int rc = prepare(h);
    if (rc) return rc;
    std::memset(h->st_host, 0, sizeof(MinState));
    h->st_host->phase = PH_IDLE;
    if ((rc = push_state(h))) return rc;
    if ((rc = prime_items(h))) return rc;
    enqueue_eval(h, PACK_PLAIN, FOLD_PLAIN); // warm: builds cells, fills every buffer the slot reads
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    const int gb = grid_beads(h->n_own);
    const int saved_profile = h->profile;
    h->profile = 0;
    double bytes = 0.0;
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; ++r) {
        switch (kernel) {
        case MMX_K_CELL_BUILD:
            enqueue_build(h, PACK_PLAIN);
            bytes = 56.0 * h->n;
            break;
        case MMX_K_NONBONDED:
            if (!has_nb(h)) break;
            if (all_pairs(h)) {
                const int tiles = (h->n + 255) / 256;
                const int tps = (tiles + h->ap_slices - 1) / h->ap_slices;
                switch (h->P.ev_pmode) {
                case 6: launch_nb_allpairs_p<6>(h, tps); break;
                case 3: launch_nb_allpairs_p<3>(h, tps); break;
                default: launch_nb_allpairs_p<0>(h, tps); break;
                }
                hipLaunchKernelGGL(k_nb_allpairs_fold, dim3(gb), dim3(256), 0, h->stream, h->n, h->ap_slices,
                                   h->fpart, h->epart, h->g, h->part, h->st);
            } else {
                const int gn = nb_grid(h);
                switch (h->P.ev_pmode) {
                case 6: launch_nb_cells_p<6>(h, gn); break;
                case 3: launch_nb_cells_p<3>(h, gn); break;
                default: launch_nb_cells_p<0>(h, gn); break;
                }
                launch_nb_finish(h);
            }
            bytes = 32.0 * h->n;
            break;
        case MMX_K_BACKBONE:
            if (!h->flags) break;
            hipLaunchKernelGGL(k_backbone, dim3(std::min((h->n_own + 255) / 256, 2048)), dim3(256), 0, h->stream, h->P,
                               h->pos4, h->flags, h->g, h->part, h->st, 1);
            bytes = 25.0 * h->n;
            break;
        case MMX_K_LOOPS:
            if (h->n_rows <= 0) break;
            hipLaunchKernelGGL(k_loops, dim3(std::min((h->n_rows + 255) / 256, 1024)), dim3(256), 0, h->stream, h->P,
                               h->n_rows, h->pos4, h->row_bead, h->row_start, h->partner, h->loop_r0, h->g, h->part,
                               h->st, h->Q.loop_form);
            bytes = 64.0 * h->n_loops;
            break;
        case MMX_K_CONFINE:
            hipLaunchKernelGGL(k_confine, dim3(gb), dim3(256), 0, h->stream, h->P, h->pos4, h->cf_w, h->g, h->part,
                               h->st, h->Q.lam_form, h->Q.cf_form);
            bytes = 25.0 * h->n;
            break;
        case MMX_K_DD_LISTS: { // the halo's own kernels of one evaluation (dd_rebuild on the stream + message pack / unpack)
            if (!use_halo(h) || !h->dd_lists_valid) break;
            const int gbl = std::max((h->n_own + 255) / 256, 1);
            // (the occupancy map itself is marked by the evaluation's k_pack: enqueue_build, occ_in_pack)
            // (as dd_rebuild enqueues it on the stream: the two kernels zero the list lengths / the occupancy words themselves)
            hipLaunchKernelGGL(k_dd_dilate, dim3(kDDWords / 256 + 1), dim3(256), 0, h->stream, h->dd_occ, h->dd_grid,
                               h->dd_maps + (size_t)h->rank * kDDPayload, h->dd_send_cnt, h->world, h->st, 1);
            hipLaunchKernelGGL(k_dd_build_lists, dim3(gbl), dim3(256), 0, h->stream, h->n_own, own_of(h), h->rank, h->world, h->x,
                               h->dd_grid, h->dd_maps, h->dd_static, h->dd_send_ids, h->slice, h->dd_send_cnt, h->dd_scap, h->st,
                               h->dd_cntmat, h->dd_occ);
            int mx = 1;
            for (int q = 0; q < h->world; ++q) mx = std::max(mx, std::max(h->dd_scap.cap[q], h->dd_rcap.cap[q]));
            const dim3 gq(std::min((mx + 255) / 256, 256), h->world);
            hipLaunchKernelGGL(k_dd_pack, gq, dim3(256), 0, h->stream, h->dd_send_ids, h->dd_send_cnt, h->slice, h->pos4,
                               h->dd_sendbuf, h->dd_scap, h->st);
            // (direct build: the unpack is k_dd_unpack_count, part of the force evaluation's launch sequence -- MMX_K_FORCES times it)
            if (!(h->fused_build && h->direct_ok && h->dcount_g && h->cell_slots && h->slotkeys && h->slot_cap > 0))
                hipLaunchKernelGGL(k_dd_unpack, gq, dim3(256), 0, h->stream, h->dd_recvbuf, h->dd_off, h->slice, h->pos4,
                                   h->dd_ghost_ids, h->n_all, h->st);
            bytes = 16.0 * (double)h->dd_nghost;
            break;
        }
        case MMX_K_FORCES: // the force evaluation as the minimizer launches it
            enqueue_eval(h, PACK_PLAIN, FOLD_NONE);
            bytes = (56.0 + 32.0 + 25.0 + 25.0) * h->n + 64.0 * h->n_loops;
            break;
        }
    }
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    h->profile = saved_profile;
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(h, hipGetLastError());
    *mean_us = (double)ms * 1e3 / reps;
    if (algorithmic_bytes) *algorithmic_bytes = bytes;
    return MMX_OK;
} MMX_CATCH(h)

// Census of the pair work at the current positions (diagnostics for DESIGN.md / bench): one thread
// per bead walks its 27-cell stencil.
__global__ __launch_bounds__(256) static void k_census(int n, const float4 *__restrict__ pos4,
                                                       const int *__restrict__ perm, const int *__restrict__ start,
                                                       const int *__restrict__ cell_of,
                                                       const GridParams *__restrict__ grid, float rc2,
                                                       double *__restrict__ out /* [2] */) {
    __shared__ double s_w[4];
    const GridParams G = *grid;
    double cand = 0.0, within = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 p = pos4[i];
        const int c = cell_of[i];
        if (c < 0) continue; // not binned on this rank
        const int cx = c % G.nx, cy = (c / G.nx) % G.ny, cz = c / (G.nx * G.ny);
        for (int zz = max(cz - 1, 0); zz <= min(cz + 1, G.nz - 1); ++zz)
            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, G.ny - 1); ++yy) {
                const int row = (zz * G.ny + yy) * G.nx;
                const int rs = start[row + max(cx - 1, 0)], re = start[row + min(cx + 1, G.nx - 1) + 1];
                cand += (double)(re - rs);
                for (int q = rs; q < re; ++q) {
                    const float4 o = pos4[perm[q]];
                    const float dx = p.x - o.x, dy = p.y - o.y, dz = p.z - o.z;
                    if (dx * dx + dy * dy + dz * dz < rc2) within += 1.0;
                }
            }
    }
    const double a = block_sum<256>(cand, s_w);
    const double b = block_sum<256>(within, s_w);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], a);
        atomicAdd(&out[1], b);
    }
}

// Tile census of the cluster-pair kernel: repeats its box-box cull and counts candidate / accepted tiles.
__global__ __launch_bounds__(256) static void k_tile_census(int ncl, const float4 *__restrict__ cl_lo,
                                                            const float4 *__restrict__ cl_hi,
                                                            const int *__restrict__ cstart,
                                                            const float4 *__restrict__ spos4,
                                                            const GridParams *__restrict__ grid, float rc2,
                                                            double *__restrict__ out /* [3] */) {
    const GridParams G = *grid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double cand = 0.0, acc = 0.0, beads = 0.0;
    for (int icl = blockIdx.x * 4 + wave; icl < ncl; icl += gridDim.x * 4) {
        const float4 lo_i = cl_lo[2 * icl], hi_i = cl_lo[2 * icl + 1];
        const int c = __float_as_int(lo_i.w);
        const int cx = c % G.nx, cy = (c / G.nx) % G.ny, cz = c / (G.nx * G.ny);
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.nx - 1);
        for (int zz = max(cz - 1, 0); zz <= min(cz + 1, G.nz - 1); ++zz)
            for (int yy = max(cy - 1, 0); yy <= min(cy + 1, G.ny - 1); ++yy) {
                const int row = (zz * G.ny + yy) * G.nx;
                const int c0 = cstart[row + x0], c1 = cstart[row + x1 + 1];
                for (int jc = c0 + lane; jc < c1; jc += 64) {
                    const float4 lo_j = cl_lo[2 * jc], hi_j = cl_lo[2 * jc + 1];
                    const float dx = fmaxf(fmaxf(lo_j.x - hi_i.x, lo_i.x - hi_j.x), 0.f);
                    const float dy = fmaxf(fmaxf(lo_j.y - hi_i.y, lo_i.y - hi_j.y), 0.f);
                    const float dz = fmaxf(fmaxf(lo_j.z - hi_i.z, lo_i.z - hi_j.z), 0.f);
                    cand += 1.0;
                    if (dx * dx + dy * dy + dz * dz < rc2) {
                        acc += 1.0;
                        // second-level cull: beads of the accepted j-cluster within the cutoff of the i box
                        for (int k = 0; k < 8; ++k) {
                            const float4 q = spos4[(size_t)jc * 8 + k];
                            const float bx = fmaxf(fmaxf(lo_i.x - q.x, q.x - hi_i.x), 0.f);
                            const float by = fmaxf(fmaxf(lo_i.y - q.y, q.y - hi_i.y), 0.f);
                            const float bz = fmaxf(fmaxf(lo_i.z - q.z, q.z - hi_i.z), 0.f);
                            if (bx * bx + by * by + bz * bz < rc2) beads += 1.0;
                        }
                    }
                }
            }
    }
    cand = wave_sum(cand);
    acc = wave_sum(acc);
    beads = wave_sum(beads);
    if (lane == 0) {
        atomicAdd(&out[0], cand);
        atomicAdd(&out[1], acc);
        atomicAdd(&out[2], beads);
    }
}

int mmx_cluster_census(mmx_handle h, int64_t *n_clusters, double *tiles_candidate, double *tiles_accepted,
                       double *beads_swept) try {
    if (!h) return MMX_ERR_BAD_ARG;
    int rc = prepare(h);
    if (rc) return rc;
    if (!has_nb(h) || all_pairs(h)) return fail(h, MMX_ERR_STATE, "census needs a cutoff (cell-list mode)");
    std::memset(h->st_host, 0, sizeof(MinState));
    h->st_host->phase = PH_IDLE;
    if ((rc = push_state(h))) return rc;
    if ((rc = prime_items(h))) return rc;
    double *dout = nullptr;
    HIPCHK(h, dalloc(&dout, 3));
    const int ncl = h->st_host->n_clusters;
    hipLaunchKernelGGL(k_tile_census, dim3(1024), dim3(256), 0, h->stream, ncl, h->cl_lo, h->cl_hi, h->cstart,
                       h->spos4, h->gcur, h->P.rc2max, dout);
    double res[3] = {0, 0, 0};
    HIPCHK(h, hipMemcpyAsync(res, dout, sizeof(res), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(dout);
    if (n_clusters) *n_clusters = ncl;
    if (tiles_candidate) *tiles_candidate = res[0];
    if (tiles_accepted) *tiles_accepted = res[1];
    if (beads_swept) *beads_swept = res[2];
    return MMX_OK;
} MMX_CATCH(h)

int mmx_nb_census(mmx_handle h, int64_t *n_cells, int32_t *max_per_cell, double *cell_edge, double *pair_candidates,
                  double *pairs_within_cutoff) try {
    if (!h) return MMX_ERR_BAD_ARG;
    int rc = prepare(h);
    if (rc) return rc;
    if (!has_nb(h) || all_pairs(h)) return fail(h, MMX_ERR_STATE, "census needs a cutoff (cell-list mode)");
    std::memset(h->st_host, 0, sizeof(MinState));
    h->st_host->phase = PH_IDLE;
    if ((rc = push_state(h))) return rc;
    if ((rc = prime_items(h))) return rc;
    double *dout = nullptr;
    HIPCHK(h, dalloc(&dout, 2));
    hipLaunchKernelGGL(k_census, dim3(grid_beads(h->n)), dim3(256), 0, h->stream, h->n, h->pos4, h->perm, h->start,
                       h->cell_of, h->gcur, h->P.rc2max, dout);
    double res[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(res, dout, sizeof(res), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(dout);
    if (n_cells) *n_cells = h->st_host->ncells;
    if (max_per_cell) *max_per_cell = h->st_host->max_per_cell;
    if (cell_edge) *cell_edge = h->st_host->cell_edge;
    if (pair_candidates) *pair_candidates = res[0];
    if (pairs_within_cutoff) *pairs_within_cutoff = res[1] - (double)h->n; // minus self pairs
    return MMX_OK;
} MMX_CATCH(h)

} // extern "C"

#ifdef MMX_STAGE_TIMING
extern "C" int mmx_debug_build_cells(int *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mmx::g_stage_c), sizeof(int) * 8192);
}
extern "C" int mmx_debug_build_times(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mmx::g_stage_b), sizeof(unsigned long long) * 8192);
}
extern "C" int mmx_debug_stage_times(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mmx::g_stage_t), sizeof(unsigned long long) * 8192);
}
#endif
#ifdef MMX_N3_TIMING
extern "C" int mmx_debug_n3_times(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mmx::g_n3_t), sizeof(unsigned long long) * 512 * 20);
}
extern "C" int mmx_debug_n3_trace(unsigned *events, unsigned *counts) {
    int rc = (int)hipMemcpyFromSymbol(events, HIP_SYMBOL(mmx::g_n3_v), sizeof(unsigned) * mmx::kN3TraceBlocks * mmx::kN3TraceEvents * 8);
    if (!rc) rc = (int)hipMemcpyFromSymbol(counts, HIP_SYMBOL(mmx::g_n3_vn), sizeof(unsigned) * mmx::kN3TraceBlocks);
    return rc;
}
extern "C" int mmx_debug_n3_waits(unsigned *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mmx::g_n3_w), sizeof(unsigned) * 512 * 16 * 8);
}
extern "C" int mmx_debug_n3_counters(unsigned long long *out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mmx::g_n3_c), sizeof(unsigned long long) * 16);
    if (!rc && reset) {
        const unsigned long long z[16] = {};
        rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(mmx::g_n3_c), z, sizeof(z));
    }
    return rc;
}
#endif
