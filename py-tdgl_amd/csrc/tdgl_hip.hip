// libtdgl_hip: MI355X-native TDGL time-stepping core behind a C ABI (include/tdgl_hip.h).
// Single translation unit: kernels (kernels.inc), Poisson solver (poisson.inc), the
// per-step driver (run.inc).

#include "tdgl_internal.h"

static thread_local std::string g_last_error;

#define TDGL_FAIL(ctx, code, ...)                                   \
    do {                                                            \
        char _buf[512];                                             \
        snprintf(_buf, sizeof(_buf), __VA_ARGS__);                  \
        if (ctx) (ctx)->err = _buf;                                 \
        g_last_error = _buf;                                        \
        return (code);                                              \
    } while (0)

#define HIP_TRY(ctx, expr)                                                               \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            TDGL_FAIL(ctx, TDGL_ERR_HIP, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), \
                      __FILE__, __LINE__, #expr);                                        \
    } while (0)

#define TDGL_TRY(expr)                 \
    do {                               \
        int _s = (expr);               \
        if (_s != TDGL_OK) return _s;  \
    } while (0)

#include "kernels.inc"

using namespace tdgl;

static inline int grid_for(int64_t n, int block = BLOCK) { return (int)((n + block - 1) / block); }

// ---------------------------------------------------------------------------------------
// SELL construction from CSR (host)
static int build_sell_pattern(tdgl_ctx *ctx, int64_t n_rows, const int32_t *indptr,
                              const int32_t *indices, SellPattern &pat,
                              std::vector<int64_t> *slot_of_nnz, bool want16 = false, bool want_off16 = false) {
    pat.n_rows = n_rows;
    pat.n_pad = round_up(std::max<int64_t>(n_rows, 1), WAVE);
    pat.n_slices = (int32_t)(pat.n_pad / WAVE);
    std::vector<int32_t> off(pat.n_slices + 1, 0);
    for (int s = 0; s < pat.n_slices; ++s) {
        int w = 0;
        for (int64_t r = (int64_t)s * WAVE; r < std::min<int64_t>(n_rows, (int64_t)(s + 1) * WAVE); ++r)
            w = std::max(w, indptr[r + 1] - indptr[r]);
        off[s + 1] = off[s] + w;
    }
    pat.n_slots = (int64_t)off[pat.n_slices] * WAVE;
    std::vector<int32_t> cols(pat.n_slots);
    if (slot_of_nnz) slot_of_nnz->assign(indptr[n_rows], -1);
    for (int s = 0; s < pat.n_slices; ++s) {
        const int w = off[s + 1] - off[s];
        for (int lane = 0; lane < WAVE; ++lane) {
            const int64_t r = (int64_t)s * WAVE + lane;
            const int deg = (r < n_rows) ? indptr[r + 1] - indptr[r] : 0;
            for (int k = 0; k < w; ++k) {
                const int64_t slot = ((int64_t)off[s] + k) * WAVE + lane;
                if (k < deg) {
                    cols[slot] = indices[indptr[r] + k];
                    if (slot_of_nnz) (*slot_of_nnz)[indptr[r] + k] = slot;
                } else {
                    // padding (value 0): repeat the row's first column -- always a valid
                    // column, also for rectangular operators, and already in cache
                    cols[slot] = deg > 0 ? indices[indptr[r]] : 0;
                }
            }
        }
    }
    HIP_TRY(ctx, pat.slice_off.upload(off));
    HIP_TRY(ctx, pat.cols.upload(cols));
    pat.use16 = false;
    if (want16) {
        std::vector<int16_t> d16(pat.n_slots);
        bool fits = true;
        for (int sl = 0; sl < pat.n_slices && fits; ++sl)
            for (int k = off[sl]; k < off[sl + 1] && fits; ++k)
                for (int lane = 0; lane < WAVE; ++lane) {
                    const int64_t slot = (int64_t)k * WAVE + lane, row = (int64_t)sl * WAVE + lane;
                    // padded lanes past n_rows point at column 0: give them delta 0 (own row)
                    const int64_t delta = (row < n_rows) ? (int64_t)cols[slot] - row : 0;
                    if (delta < -32768 || delta > 32767) {
                        fits = false;
                        break;
                    }
                    d16[slot] = (int16_t)delta;
                }
        if (fits) {
            HIP_TRY(ctx, pat.cols16.upload(d16));
            pat.use16 = true;
        }
    }
    pat.use_off16 = false;
    if (want_off16) {
        std::vector<uint16_t> o16(std::max<int64_t>(pat.n_slots, 1), 0);
        std::vector<int32_t> sbase(std::max<int32_t>(pat.n_slices, 1), 0);
        bool fits = true;
        for (int sl = 0; sl < pat.n_slices && fits; ++sl) {
            const int64_t rows_here = std::min<int64_t>(WAVE, n_rows - (int64_t)sl * WAVE);  // (padded lanes hold column 0)
            int32_t lo = INT32_MAX, hi = 0;
            for (int k = off[sl]; k < off[sl + 1]; ++k)
                for (int lane = 0; lane < rows_here; ++lane) {
                    lo = std::min(lo, cols[(int64_t)k * WAVE + lane]);
                    hi = std::max(hi, cols[(int64_t)k * WAVE + lane]);
                }
            if (lo == INT32_MAX) lo = 0;
            if ((int64_t)hi - lo > 65535) {
                fits = false;
                break;
            }
            sbase[sl] = lo;
            for (int k = off[sl]; k < off[sl + 1]; ++k)
                for (int lane = 0; lane < WAVE; ++lane)
                    o16[(int64_t)k * WAVE + lane] = lane < rows_here ? (uint16_t)(cols[(int64_t)k * WAVE + lane] - lo) : 0;
        }
        if (fits) {
            HIP_TRY(ctx, pat.offs16.upload(o16));
            HIP_TRY(ctx, pat.slice_base.upload(sbase));
            pat.use_off16 = true;
        }
    }
    return TDGL_OK;
}

static int build_sell_f64(tdgl_ctx *ctx, int64_t n_rows, const int32_t *indptr,
                          const int32_t *indices, const double *data, SellF64 &A, bool want16 = false,
                          bool want_off16 = false) {
    std::vector<int64_t> slot;
    TDGL_TRY(build_sell_pattern(ctx, n_rows, indptr, indices, A.pat, &slot, want16, want_off16));
    std::vector<double> vals(A.pat.n_slots, 0.0);
    for (int64_t k = 0; k < indptr[n_rows]; ++k) vals[slot[k]] = data[k];
    HIP_TRY(ctx, A.vals.upload(vals));
    return TDGL_OK;
}

// ---------------------------------------------------------------------------------------
extern "C" int tdgl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *tdgl_version(void) { return "tdgl_hip 0.1.0 (gfx950)"; }

extern "C" const char *tdgl_last_error(const tdgl_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

static void ipc_free(tdgl_ctx *ctx);  // ipc.inc

extern "C" void tdgl_destroy(tdgl_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto *lv : ctx->levels) delete lv;
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    ipc_free(ctx);
    if (ctx->h_sendbuf) (void)hipHostFree(ctx->h_sendbuf);
    if (ctx->h_recvbuf) (void)hipHostFree(ctx->h_recvbuf);
    if (ctx->h_deep_send) (void)hipHostFree(ctx->h_deep_send);
    if (ctx->h_deep_recv) (void)hipHostFree(ctx->h_deep_recv);
    if (ctx->h_status) (void)hipHostFree(ctx->h_status);
    if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
    if (ctx->h_rec) (void)hipHostFree(ctx->h_rec);
    if (ctx->h_probe_out) (void)hipHostFree(ctx->h_probe_out);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_pack) (void)hipEventDestroy(ctx->ev_pack);
    if (ctx->ev_halo) (void)hipEventDestroy(ctx->ev_halo);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    for (auto &pr : ctx->prof_pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (auto &pr : ctx->prof2_pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (auto &pr : ctx->prof3_pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (hipEvent_t e : ctx->prof_pool) (void)hipEventDestroy(e);
    hipStream_t s = ctx->stream;
    delete ctx;  // frees the DevBufs
    if (s) (void)hipStreamDestroy(s);
}

extern "C" int tdgl_synchronize(tdgl_ctx *ctx) {
    if (!ctx) return TDGL_ERR_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TDGL_OK;
}

static int create_impl(tdgl_ctx *ctx, const tdgl_mesh_desc *d) {
    const int64_t n = d->n_sites, m = d->n_edges, nb = d->n_boundary_edges;
    ctx->n = n;
    ctx->m = m;
    ctx->nb = nb;
    ctx->n_pad = round_up(n, WAVE);
    ctx->m_pad = round_up(std::max<int64_t>(m, 1), WAVE);
    ctx->n_own = (d->n_owned > 0) ? d->n_owned : n;
    ctx->n_global = n;
    if (ctx->n_own > n) TDGL_FAIL(ctx, TDGL_ERR_ARG, "n_owned exceeds n_sites");
    if (ctx->n_own < n && d->site_perm) TDGL_FAIL(ctx, TDGL_ERR_ARG, "site_perm must be NULL when n_owned < n_sites");
    ctx->u = d->u;
    ctx->gamma = d->gamma;
    ctx->fix_psi = d->fix_psi != 0;

    // ---- site permutation ----------------------------------------------------------
    ctx->perm.resize(n);
    ctx->iperm.assign(n, -1);
    for (int64_t i = 0; i < n; ++i) ctx->perm[i] = d->site_perm ? d->site_perm[i] : (int32_t)i;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t r = ctx->perm[i];
        if (r < 0 || r >= n || ctx->iperm[r] != -1)
            TDGL_FAIL(ctx, TDGL_ERR_ARG, "site_perm is not a permutation of 0..n_sites-1");
        ctx->iperm[r] = (int32_t)i;
    }

    // ---- edges in internal numbering, sorted for gather locality ---------------------
    std::vector<int32_t> p0(m), p1(m);
    for (int64_t e = 0; e < m; ++e) {
        const int32_t i = d->edges[2 * e], j = d->edges[2 * e + 1];
        if (i < 0 || i >= n || j < 0 || j >= n || i == j)
            TDGL_FAIL(ctx, TDGL_ERR_ARG, "edge %lld has invalid sites (%d, %d)", (long long)e, i, j);
        p0[e] = ctx->iperm[i];
        p1[e] = ctx->iperm[j];
    }
    ctx->edge_perm.resize(m);
    std::iota(ctx->edge_perm.begin(), ctx->edge_perm.end(), 0);
    // (one-process-per-GPU mode: edges that touch a ghost site go last, so that the edges between
    // owned sites form a prefix the edge kernel can process while ghost values travel)
    std::sort(ctx->edge_perm.begin(), ctx->edge_perm.end(), [&](int32_t a, int32_t b) {
        const int32_t alo = std::min(p0[a], p1[a]), ahi = std::max(p0[a], p1[a]);
        const int32_t blo = std::min(p0[b], p1[b]), bhi = std::max(p0[b], p1[b]);
        const bool ag = ahi >= ctx->n_own, bg = bhi >= ctx->n_own;
        if (ag != bg) return bg;
        return alo != blo ? alo < blo : (ahi != bhi ? ahi < bhi : a < b);
    });
    ctx->m_int = 0;
    for (int64_t e = 0; e < m; ++e) ctx->m_int += std::max(p0[e], p1[e]) < ctx->n_own ? 1 : 0;
    ctx->edge_iperm.resize(m);
    for (int64_t k = 0; k < m; ++k) ctx->edge_iperm[ctx->edge_perm[k]] = (int32_t)k;

    std::vector<int32_t> e0(ctx->m_pad, 0), e1(ctx->m_pad, 0);
    std::vector<double> inv_len(ctx->m_pad, 0.0), dx(ctx->m_pad, 0.0), dy(ctx->m_pad, 0.0), w(m);
    for (int64_t k = 0; k < m; ++k) {
        const int32_t e = ctx->edge_perm[k];
        e0[k] = p0[e];
        e1[k] = p1[e];
        if (!(d->edge_lengths[e] > 0.0) || !(d->dual_edge_lengths[e] >= 0.0))
            TDGL_FAIL(ctx, TDGL_ERR_ARG, "edge %d has non-positive length or negative dual length", e);
        inv_len[k] = 1 / d->edge_lengths[e];                       // operators.py:271
        w[k] = d->dual_edge_lengths[e] / d->edge_lengths[e];       // operators.py:272
        dx[k] = d->directions[2 * e];
        dy[k] = d->directions[2 * e + 1];
    }

    // ---- site graph (CSR, neighbours sorted) -> SELL pattern ------------------------
    std::vector<int32_t> deg(n + 1, 0);
    for (int64_t k = 0; k < m; ++k) {
        deg[e0[k] + 1]++;
        deg[e1[k] + 1]++;
    }
    std::vector<int32_t> indptr(n + 1, 0);
    for (int64_t i = 0; i < n; ++i) indptr[i + 1] = indptr[i] + deg[i + 1];
    std::vector<int32_t> nbr(2 * m), code(2 * m), fill(indptr.begin(), indptr.end() - 1);
    for (int64_t k = 0; k < m; ++k) {  // edges sorted by min site -> rows fill in order
        nbr[fill[e0[k]]] = e1[k];
        code[fill[e0[k]]++] = (int32_t)(k << 1);        // row = tail: U_e
        nbr[fill[e1[k]]] = e0[k];
        code[fill[e1[k]]++] = (int32_t)(k << 1) | 1;    // row = head: conj(U_e)
    }
    for (int64_t i = 0; i < n; ++i) {  // sort each row by neighbour index
        const int b = indptr[i], e = indptr[i + 1];
        std::vector<std::pair<int32_t, int32_t>> row(e - b);
        for (int k = b; k < e; ++k) row[k - b] = {nbr[k], code[k]};
        std::sort(row.begin(), row.end());
        for (int k = b; k < e; ++k) {
            nbr[k] = row[k - b].first;
            code[k] = row[k - b].second;
        }
    }
    // rows exist for owned sites only; ghost sites appear as columns
    const int64_t n_rows = ctx->n_own;
    std::vector<int64_t> slot;
    TDGL_TRY(build_sell_pattern(ctx, n_rows, indptr.data(), nbr.data(), ctx->lap_pat, &slot, /*want16=*/true));
    const int64_t n_slots = ctx->lap_pat.n_slots;
    std::vector<int32_t> slot_edge(n_slots, -1);
    std::vector<double> slot_w(n_slots, 0.0), diag(ctx->n_pad, 0.0), area(ctx->n_pad, 0.0);
    for (int64_t i = 0; i < n; ++i) {
        const double a = d->areas[ctx->perm[i]];
        if (!(a > 0.0)) TDGL_FAIL(ctx, TDGL_ERR_ARG, "site %d has non-positive area", ctx->perm[i]);
        area[i] = a;
        if (i >= n_rows) continue;
        double dsum = 0.0;
        for (int k = indptr[i]; k < indptr[i + 1]; ++k) {
            const double wk = w[code[k] >> 1];
            slot_edge[slot[k]] = code[k];
            slot_w[slot[k]] = wk / a;
            dsum += -wk / a;  // operators.py:166-167
        }
        diag[i] = dsum;
    }
    std::vector<uint8_t> fixed(ctx->n_pad, 0);
    if (d->fix_psi)
        for (int64_t k = 0; k < d->n_fixed; ++k) {
            const int32_t s = d->fixed_sites[k];
            if (s < 0 || s >= n) TDGL_FAIL(ctx, TDGL_ERR_ARG, "fixed site %d out of range", s);
            fixed[ctx->iperm[s]] = 1;
        }

    HIP_TRY(ctx, ctx->lap_slot_edge.upload(slot_edge));
    HIP_TRY(ctx, ctx->lap_slot_w.upload(slot_w));
    HIP_TRY(ctx, ctx->lap_vals.alloc(n_slots));
    HIP_TRY(ctx, ctx->lap_diag.upload(diag));
    HIP_TRY(ctx, ctx->fixed_mask.upload(fixed));
    HIP_TRY(ctx, ctx->area.upload(area));
    HIP_TRY(ctx, ctx->e0.upload(e0));
    HIP_TRY(ctx, ctx->e1.upload(e1));
    HIP_TRY(ctx, ctx->e_inv_len.upload(inv_len));
    HIP_TRY(ctx, ctx->e_dirx.upload(dx));
    HIP_TRY(ctx, ctx->e_diry.upload(dy));
    HIP_TRY(ctx, ctx->e_U.alloc(ctx->m_pad));
    HIP_TRY(ctx, ctx->e_A.alloc(2 * ctx->m_pad));
    HIP_TRY(ctx, ctx->e_Aprev.alloc(2 * ctx->m_pad));
    HIP_TRY(ctx, ctx->e_dAdt.alloc(ctx->m_pad));

    // ---- Neumann boundary term (operators.py:188-230) -------------------------------
    std::vector<int32_t> s0(std::max<int64_t>(nb, 1), 0), s1(std::max<int64_t>(nb, 1), 0);
    std::vector<double> c0(std::max<int64_t>(nb, 1), 0.0), c1(std::max<int64_t>(nb, 1), 0.0);
    for (int64_t k = 0; k < nb; ++k) {
        const int32_t e = d->boundary_edge_indices[k];
        if (e < 0 || e >= m) TDGL_FAIL(ctx, TDGL_ERR_ARG, "boundary edge %d out of range", e);
        const int32_t i = d->edges[2 * e], j = d->edges[2 * e + 1];
        s0[k] = ctx->iperm[i];
        s1[k] = ctx->iperm[j];
        c0[k] = d->edge_lengths[e] / (2 * d->areas[i]);
        c1[k] = d->edge_lengths[e] / (2 * d->areas[j]);
    }
    HIP_TRY(ctx, ctx->b_s0.upload(s0));
    HIP_TRY(ctx, ctx->b_s1.upload(s1));
    HIP_TRY(ctx, ctx->b_c0.upload(c0));
    HIP_TRY(ctx, ctx->b_c1.upload(c1));
    HIP_TRY(ctx, ctx->b_mu.alloc(std::max<int64_t>(nb, 1)));
    HIP_TRY(ctx, ctx->cvec.alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->ceff.alloc(ctx->n_pad));

    // ---- state ------------------------------------------------------------------------
    HIP_TRY(ctx, ctx->psi[0].alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->psi[1].alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->lap[0].alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->lap[1].alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->mu.alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->eps.alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->bvec.alloc(ctx->n_pad));
    HIP_TRY(ctx, ctx->js.alloc(ctx->m_pad));
    HIP_TRY(ctx, ctx->jn.alloc(ctx->m_pad));
    HIP_TRY(ctx, ctx->d_status.alloc(1));
    // The status block is published into device memory and copied to this pinned block at every host
    // synchronisation.
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_status), sizeof(StepStatus)));
    memset(ctx->h_status, 0, sizeof(StepStatus));
    // (the three environment switches of the library: what the test-suite needs to reach a code path -- DESIGN.md section 5)
    ctx->run_ahead_disabled = getenv("TDGL_NO_RUN_AHEAD") != nullptr;
    if (const char *e = getenv("TDGL_PCG_PREDICT")) ctx->pcg_predict_from_guess = strcmp(e, "last") != 0;
    ctx->status_dev = ctx->d_status.p;
    HIP_TRY(ctx, ctx->scal.alloc(S_COUNT));
    HIP_TRY(ctx, ctx->d_ctl.alloc(1));
    HIP_TRY(ctx, ctx->d_rec.alloc(RA_BATCH_MAX));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_ctl), sizeof(StepCtl)));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_rec), RA_BATCH_MAX * sizeof(StepRec)));
    HIP_TRY(ctx, hipEventCreate(&ctx->ev0));
    HIP_TRY(ctx, hipEventCreate(&ctx->ev1));
    ctx->psi_blocks = (int)std::min<int64_t>(std::max<int64_t>(grid_for(ctx->n_own), 1), 2048);
    {
        const int64_t tiles = (ctx->n_own + BLOCK - 1) / BLOCK;
        ctx->npart = (int)std::min<int64_t>(NB, std::max<int64_t>(1, (tiles + XCDS - 1) / XCDS) * XCDS);
        if (ctx->n_own < ctx->n) ctx->npart = NB;
    }
    HIP_TRY(ctx, ctx->psi_dmax_part.alloc(ctx->psi_blocks));
    HIP_TRY(ctx, ctx->psi_fail_part.alloc(ctx->psi_blocks));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_pack, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
    return TDGL_OK;
}

extern "C" int tdgl_create(tdgl_ctx **out, const tdgl_mesh_desc *d, int device_id) {
    if (!out || !d) TDGL_FAIL((tdgl_ctx *)nullptr, TDGL_ERR_ARG, "tdgl_create: null argument");
    *out = nullptr;
    if (d->n_sites <= 0 || d->n_edges <= 0 || d->n_boundary_edges < 0 || !d->edges || !d->areas ||
        !d->edge_lengths || !d->dual_edge_lengths || !d->directions ||
        (d->n_boundary_edges > 0 && !d->boundary_edge_indices) || (d->n_fixed > 0 && !d->fixed_sites))
        TDGL_FAIL((tdgl_ctx *)nullptr, TDGL_ERR_ARG, "tdgl_create: incomplete mesh description");
    if (d->n_sites >= (1ll << 30) || d->n_edges >= (1ll << 30))
        TDGL_FAIL((tdgl_ctx *)nullptr, TDGL_ERR_ARG, "tdgl_create: mesh too large for int32 indices");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        TDGL_FAIL((tdgl_ctx *)nullptr, TDGL_ERR_HIP,
                  "tdgl_create: no HIP device available (%s); this library has no CPU fallback",
                  hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        TDGL_FAIL((tdgl_ctx *)nullptr, TDGL_ERR_ARG, "tdgl_create: device %d not in [0, %d)", device_id, ndev);
    tdgl_ctx *ctx = new tdgl_ctx();
    ctx->device = device_id;
    int st = TDGL_OK;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        g_last_error = "tdgl_create: cannot select device / create stream";
        st = TDGL_ERR_HIP;
    } else {
        st = create_impl(ctx, d);
    }
    if (st != TDGL_OK) {
        g_last_error = ctx->err.empty() ? g_last_error : ctx->err;
        tdgl_destroy(ctx);
        return st;
    }
    *out = ctx;
    return TDGL_OK;
}

// ---------------------------------------------------------------------------------------
// host <-> device with the site / edge permutation applied
template <class T>
static int upload_sites(tdgl_ctx *ctx, const T *ref, DevBuf<T> &dst) {
    std::vector<T> tmp(ctx->n_pad, T{});
    for (int64_t i = 0; i < ctx->n; ++i) tmp[i] = ref[ctx->perm[i]];
    HIP_TRY(ctx, hipMemcpyAsync(dst.p, tmp.data(), ctx->n_pad * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TDGL_OK;
}

template <class T>
static int download_sites(tdgl_ctx *ctx, const T *src, T *ref) {
    std::vector<T> tmp(ctx->n);
    HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), src, ctx->n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < ctx->n; ++i) ref[ctx->perm[i]] = tmp[i];
    return TDGL_OK;
}

static int download_edges(tdgl_ctx *ctx, const double *src, double *ref) {
    std::vector<double> tmp(ctx->m);
    HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), src, ctx->m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t k = 0; k < ctx->m; ++k) ref[ctx->edge_perm[k]] = tmp[k];
    return TDGL_OK;
}

#define CTX_GUARD(ctx)                              \
    do {                                            \
        if (!(ctx)) return TDGL_ERR_ARG;            \
        HIP_TRY(ctx, hipSetDevice((ctx)->device));  \
    } while (0)

// ---------------------------------------------------------------------------------------
// Tile ranges of the SELL kernels whose rows are the owned sites.  part 0: all tiles; 1: the
// leading ghost-free tiles (safe to run while a halo exchange is in flight); 2: the rest.
static inline void tile_range(const tdgl_ctx *ctx, int n_slices, int part, int *tile_base, int *slice_end,
                              int *tiles) {
    const int all = (n_slices + BLOCK / WAVE - 1) / (BLOCK / WAVE);
    const int ni = std::min(ctx->int_tiles, all);
    if (part == 1) {
        *tile_base = 0, *tiles = ni, *slice_end = ni * (BLOCK / WAVE);
    } else if (part == 2) {
        *tile_base = ni, *tiles = all - ni, *slice_end = n_slices;
    } else {
        *tile_base = 0, *tiles = all, *slice_end = n_slices;
    }
}

static void launch_psi_laplacian(tdgl_ctx *ctx, bool rhs, const double2 *psi, double2 *lap, int part = 0) {
    int tile_base, slice_end, tiles;
    tile_range(ctx, ctx->lap_pat.n_slices, part, &tile_base, &slice_end, &tiles);
    if (tiles <= 0) return;
    const int per_xcd = (tiles + XCDS - 1) / XCDS, grid = per_xcd * XCDS;
    const SellPattern &pat = ctx->lap_pat;
    const double *c = rhs ? ctx->ceff.p : ctx->cvec.p;
#define TDGL_K1(RHS, IT, COLS)                                                                              \
    hipLaunchKernelGGL((k_psi_laplacian<RHS, IT>), dim3(grid), dim3(BLOCK), 0, ctx->stream, slice_end, per_xcd, \
                       tile_base, pat.n_rows, pat.slice_off.p, COLS, ctx->lap_vals.p, ctx->lap_diag.p,     \
                       ctx->fixed_mask.p, psi, lap, ctx->area.p, c, ctx->bvec.p)
    if (pat.use16) {
        if (rhs) TDGL_K1(true, int16_t, pat.cols16.p); else TDGL_K1(false, int16_t, pat.cols16.p);
    } else {
        if (rhs) TDGL_K1(true, int32_t, pat.cols.p); else TDGL_K1(false, int32_t, pat.cols.p);
    }
#undef TDGL_K1
}

static void launch_psi_update(tdgl_ctx *ctx, const double2 *psi, const double *mu, const double2 *lap,
                              double dt, double2 *psi_new, double *abs_sq, const double *abs_sq_in = nullptr) {
    hipLaunchKernelGGL(k_psi_update, dim3(ctx->psi_blocks), dim3(BLOCK), 0, ctx->stream, ctx->n_own, psi, mu,
                       ctx->eps.p, lap, dt, ctx->u, ctx->gamma, psi_new, abs_sq, ctx->psi_dmax_part.p,
                       ctx->psi_fail_part.p, abs_sq_in);
    ctx->psi_status_pending = true;
}

// ... with the edge currents of the step just accepted in the same launch (k_psi_update_with_currents)
static void launch_psi_update_with_currents(tdgl_ctx *ctx, const double2 *psi, const double *mu, const double2 *lap,
                                            double dt, double2 *psi_new) {
    hipLaunchKernelGGL(k_psi_update_with_currents, dim3(ctx->psi_blocks + grid_for(ctx->m)), dim3(BLOCK), 0, ctx->stream,
                       ctx->psi_blocks, ctx->n_own, psi, mu, ctx->eps.p, lap, dt, ctx->u, ctx->gamma, psi_new,
                       ctx->psi_dmax_part.p, ctx->psi_fail_part.p, ctx->m, ctx->e0.p, ctx->e1.p, ctx->e_inv_len.p,
                       ctx->e_U.p, ctx->js.p, ctx->jn.p);
    ctx->psi_status_pending = true;
}

// one attempt of the run-ahead loop: psi update (+ the owed edge currents) ...
static void launch_ra_psi(tdgl_ctx *ctx, bool with_currents) {
    const int64_t m = with_currents ? ctx->m : 0;
    hipLaunchKernelGGL(k_ra_psi_update, dim3(ctx->psi_blocks + (m > 0 ? grid_for(m) : 0)), dim3(BLOCK), 0, ctx->stream,
                       ctx->psi_blocks, ctx->n_own, ctx->psi[0].p, ctx->psi[1].p, (const double2 *)ctx->lap[0].p,
                       (const double2 *)ctx->lap[1].p, (const double *)ctx->mu.p, (const double *)ctx->eps.p, ctx->u, ctx->gamma,
                       ctx->psi_dmax_part.p, ctx->psi_fail_part.p, m, ctx->e0.p, ctx->e1.p, ctx->e_inv_len.p, ctx->e_U.p,
                       ctx->js.p, ctx->jn.p, ctx->d_ctl.p);
}

// ... then L psi' and the right-hand side
static void launch_ra_laplacian(tdgl_ctx *ctx) {
    const SellPattern &pat = ctx->lap_pat;
    const int tiles = (pat.n_slices + BLOCK / WAVE - 1) / (BLOCK / WAVE);
    const int per_xcd = (tiles + XCDS - 1) / XCDS, grid = per_xcd * XCDS;
#define TDGL_K1RA(IT, COLS)                                                                                          \
    hipLaunchKernelGGL((k_psi_laplacian_ra<IT>), dim3(grid), dim3(BLOCK), 0, ctx->stream, pat.n_slices, per_xcd, pat.n_rows, \
                       pat.slice_off.p, COLS, ctx->lap_vals.p, ctx->lap_diag.p, ctx->fixed_mask.p,                   \
                       (const double2 *)ctx->psi[0].p, (const double2 *)ctx->psi[1].p, ctx->lap[0].p, ctx->lap[1].p, \
                       ctx->area.p, (const double *)ctx->ceff.p, ctx->bvec.p, (const StepCtl *)ctx->d_ctl.p)
    if (pat.use16) TDGL_K1RA(int16_t, pat.cols16.p); else TDGL_K1RA(int32_t, pat.cols.p);
#undef TDGL_K1RA
}

// reduce the outcome of the last psi update into d_status (together with the PCG scalars);
// guess_start: first synchronisation of a solve with the projection guess (sums its partial arrays,
// sets S_BB / S_TOL2, resets the iteration counters); rr_part: residual partials to sum into S_RR
// workgroups of the projection guess's dot-product pass (k_multi_dot): one per CU -- every workgroup ends with
// a reduction of 2 K + 2 double-double sums, and the status kernel (ONE workgroup, on the step's critical
// path) adds that many partials per sum; four waves per CU with 13 16-byte loads per lane in flight keep
// the stream busy.
static inline int guess_grid(const tdgl_ctx *ctx) { return std::min(ctx->npart, 256); }
// one process per GPU with at most G_RANK_STRIDE ranks: the ranks' double-double totals are gathered exactly
static inline bool guess_rank_totals(const tdgl_ctx *ctx) {
    static const bool off = getenv("TDGL_GUESS_NO_GATHER") != nullptr;  // (tests: the path of more than 16 ranks)
    return (ctx->world > 1 || ctx->comm != nullptr || ctx->ipc != nullptr) && ctx->world <= G_RANK_STRIDE && !off;
}

static void publish_status(tdgl_ctx *ctx, bool guess_start = false, const double *rr_part = nullptr) {
    const bool psi = ctx->psi_status_pending;
    const bool rank_totals = guess_rank_totals(ctx);
    hipLaunchKernelGGL(k_publish_status, dim3(1), dim3(BLOCK), 0, ctx->stream, ctx->status_dev, ctx->scal.p,
                       psi ? ctx->psi_dmax_part.p : (const double *)nullptr,
                       psi ? ctx->psi_fail_part.p : (const int32_t *)nullptr, ctx->psi_blocks,
                       ctx->d_gdot.n ? ctx->d_gdot.p : (double *)nullptr,
                       guess_start ? (rank_totals ? ctx->part_gdot_rank.p : ctx->part_gdot.p) : (const double *)nullptr, ctx->g_count,
                       rank_totals ? ctx->world : guess_grid(ctx), rank_totals ? (int)G_RANK_STRIDE : (int)NB, (double)ctx->n_global,
                       ctx->popt.rtol * ctx->popt.rtol, rr_part);
    ctx->psi_status_pending = false;
}

// part 0: all edges; 1: edges between owned sites; 2: edges touching a ghost site
static void launch_edge_currents(tdgl_ctx *ctx, const double2 *psi, const double *mu, double *js,
                                 double *jn, int part = 0) {
    const int64_t base = (part == 2) ? ctx->m_int : 0, end = (part == 1) ? ctx->m_int : ctx->m;
    if (end <= base) return;
    const int grid = grid_for(end - base);
    const double *dadt = ctx->has_dadt ? ctx->e_dAdt.p : (const double *)nullptr;
    if (js && jn)
        hipLaunchKernelGGL((k_edge_currents<true, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, end,
                           ctx->e0.p, ctx->e1.p, ctx->e_inv_len.p, ctx->e_U.p, psi, mu, js, jn, dadt, base);
    else if (js)
        hipLaunchKernelGGL((k_edge_currents<true, false>), dim3(grid), dim3(BLOCK), 0, ctx->stream, end,
                           ctx->e0.p, ctx->e1.p, ctx->e_inv_len.p, ctx->e_U.p, psi, mu, js, jn, dadt, base);
    else if (jn)
        hipLaunchKernelGGL((k_edge_currents<false, true>), dim3(grid), dim3(BLOCK), 0, ctx->stream, end,
                           ctx->e0.p, ctx->e1.p, ctx->e_inv_len.p, ctx->e_U.p, psi, mu, js, jn, dadt, base);
}

static void refresh_ceff(tdgl_ctx *ctx) {
    hipLaunchKernelGGL(k_ceff, dim3(grid_for((int64_t)ctx->lap_pat.n_slices * WAVE)), dim3(BLOCK), 0, ctx->stream,
                       ctx->lap_pat.n_slices, ctx->lap_pat.n_rows, ctx->lap_pat.slice_off.p, ctx->lap_slot_edge.p,
                       ctx->lap_slot_w.p, ctx->e_inv_len.p, ctx->has_dadt ? ctx->e_dAdt.p : (const double *)nullptr,
                       ctx->cvec.p, ctx->ceff.p);
}

static int update_link_scale(tdgl_ctx *ctx, double scale, double dt_prev);  // below
static int apply_time_tables(tdgl_ctx *ctx);                                 // below
static int profile_event(tdgl_ctx *ctx, hipEvent_t *ev);                     // below

static inline int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#include "ipc.inc"
#include "comm.inc"
#include "poisson.inc"
#include "dense.inc"
#include "schur.inc"
#include "screening.inc"
#include "run.inc"

// ---------------------------------------------------------------------------------------
static int set_links_impl(tdgl_ctx *ctx, const double *A, bool dynamic, double dt_prev);

extern "C" int tdgl_set_link_exponents(tdgl_ctx *ctx, const double *A) {
    return set_links_impl(ctx, A, false, 0.0);
}

extern "C" int tdgl_update_link_exponents(tdgl_ctx *ctx, const double *A_new, double dt_prev) {
    if (ctx && !(dt_prev > 0.0)) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_update_link_exponents: dt_prev must be > 0");
    if (ctx && !ctx->have_links) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "call tdgl_set_link_exponents first");
    return set_links_impl(ctx, A_new, true, dt_prev);
}

// everything that follows new values in ctx->e_A: dA/dt (dynamic) or A_prev <- A (static), the
// rhs term of dA/dt, link variables, covariant-Laplacian values
static int finish_links(tdgl_ctx *ctx, bool dynamic, double dt_prev) {
    const size_t bytes = 2 * ctx->m_pad * sizeof(double);
    const int32_t *only_if = nullptr;
    if (dynamic) {
        const int nblk = grid_for(ctx->m);
        if (ctx->link_block_changed.n == 0) {
            HIP_TRY(ctx, ctx->link_block_changed.alloc(nblk));
            HIP_TRY(ctx, ctx->link_changed.alloc(1));
        }
        hipLaunchKernelGGL(k_dadt, dim3(nblk), dim3(BLOCK), 0, ctx->stream, ctx->m, 1.0 / dt_prev,
                           ctx->e_A.p, ctx->e_Aprev.p, ctx->e_dirx.p, ctx->e_diry.p, ctx->e_inv_len.p, ctx->e_dAdt.p,
                           ctx->link_block_changed.p);
        hipLaunchKernelGGL(k_any_flag, dim3(1), dim3(BLOCK), 0, ctx->stream, nblk, ctx->link_block_changed.p,
                           ctx->link_changed.p);
        // (with screening the links are rebuilt from A_applied + A_induced in every screening
        // iteration anyway, solver.py:670-673)
        only_if = ctx->link_changed.p;
        ctx->has_dadt = true;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->e_Aprev.p, ctx->e_A.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->has_dadt = false;
    }
    refresh_ceff(ctx);
    hipLaunchKernelGGL(k_link_variables, dim3(grid_for(ctx->m)), dim3(BLOCK), 0, ctx->stream, ctx->m,
                       ctx->e_A.p, ctx->scr_enabled ? ctx->scr_Aind.p : (const double *)nullptr, ctx->e_dirx.p,
                       ctx->e_diry.p, ctx->e_U.p, only_if);
    hipLaunchKernelGGL(k_fill_laplacian, dim3(grid_for(ctx->lap_pat.n_slots)), dim3(BLOCK), 0, ctx->stream,
                       ctx->lap_pat.n_slots, ctx->lap_slot_edge.p, ctx->lap_slot_w.p, ctx->e_U.p, ctx->lap_vals.p,
                       only_if);
    HIP_TRY(ctx, hipGetLastError());
    ctx->have_links = true;
    ctx->lap_valid = false;
    ctx->currents_valid = false;
    return TDGL_OK;
}

static int upload_edge_vectors(tdgl_ctx *ctx, const double *A, double *dst) {
    std::vector<double> tmp(2 * ctx->m_pad, 0.0);
    for (int64_t k = 0; k < ctx->m; ++k) {
        const int32_t e = ctx->edge_perm[k];
        tmp[2 * k] = A[2 * e];
        tmp[2 * k + 1] = A[2 * e + 1];
    }
    HIP_TRY(ctx, hipMemcpyAsync(dst, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // tmp goes out of scope
    return TDGL_OK;
}

static int set_links_impl(tdgl_ctx *ctx, const double *A, bool dynamic, double dt_prev) {
    CTX_GUARD(ctx);
    if (!A) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_link_exponents: null A");
    TDGL_TRY(upload_edge_vectors(ctx, A, ctx->e_A.p));
    ctx->ramp_on = false;
    TDGL_TRY(finish_links(ctx, dynamic, dt_prev));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TDGL_OK;
}

// ---- A(t) = scale(t) * A_base without leaving the device ---------------------------------------
extern "C" int tdgl_set_link_exponents_base(tdgl_ctx *ctx, const double *A_base, double scale) {
    CTX_GUARD(ctx);
    if (!A_base) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_link_exponents_base: null A_base");
    if (ctx->e_Abase.n == 0) HIP_TRY(ctx, ctx->e_Abase.alloc(2 * ctx->m_pad));
    TDGL_TRY(upload_edge_vectors(ctx, A_base, ctx->e_Abase.p));
    ctx->have_base = true;
    ctx->ramp_on = false;
    hipLaunchKernelGGL(k_scale_links, dim3(grid_for(2 * ctx->m_pad)), dim3(BLOCK), 0, ctx->stream, 2 * ctx->m_pad,
                       scale, ctx->e_Abase.p, ctx->e_A.p);
    ctx->link_scale = ctx->link_scale_prev = scale;
    TDGL_TRY(finish_links(ctx, false, 0.0));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TDGL_OK;
}

static int update_link_scale(tdgl_ctx *ctx, double scale, double dt_prev) {
    // nothing moves when the factor has been constant over the last two evaluations (A and
    // dA/dt = 0 are already in place)
    if (scale == ctx->link_scale && scale == ctx->link_scale_prev && ctx->has_dadt) return TDGL_OK;
    hipLaunchKernelGGL(k_scale_links, dim3(grid_for(2 * ctx->m_pad)), dim3(BLOCK), 0, ctx->stream, 2 * ctx->m_pad,
                       scale, ctx->e_Abase.p, ctx->e_A.p);
    ctx->link_scale_prev = ctx->link_scale;
    ctx->link_scale = scale;
    return finish_links(ctx, true, dt_prev);
}

extern "C" int tdgl_update_link_scale(tdgl_ctx *ctx, double scale, double dt_prev) {
    CTX_GUARD(ctx);
    if (!ctx->have_base) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "call tdgl_set_link_exponents_base first");
    if (!(dt_prev > 0.0)) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_update_link_scale: dt_prev must be > 0");
    TDGL_TRY(update_link_scale(ctx, scale, dt_prev));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TDGL_OK;
}

extern "C" int tdgl_set_link_ramp(tdgl_ctx *ctx, int32_t on, double tmin, double tmax, double initial, double final_) {
    CTX_GUARD(ctx);
    if (on && !ctx->have_base) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "call tdgl_set_link_exponents_base first");
    if (on && !(tmax > tmin)) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_link_ramp: tmax must be > tmin");
    ctx->ramp_on = on != 0;
    ctx->ramp_tmin = tmin;
    ctx->ramp_tmax = tmax;
    ctx->ramp_initial = initial;
    ctx->ramp_final = final_;
    return TDGL_OK;
}

extern "C" int tdgl_get_link_scale(tdgl_ctx *ctx, double *scale) {
    if (!ctx || !scale) return TDGL_ERR_ARG;
    *scale = ctx->link_scale;
    return TDGL_OK;
}

extern "C" int tdgl_set_epsilon(tdgl_ctx *ctx, const double *epsilon) {
    CTX_GUARD(ctx);
    if (!epsilon) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_epsilon: null epsilon");
    TDGL_TRY(upload_sites(ctx, epsilon, ctx->eps));
    ctx->have_eps = true;
    direct_policy_reset(ctx);
    return TDGL_OK;
}

// mu_boundary -> boundary term of the Poisson right-hand side (everything queued on the stream)
static int apply_mu_boundary(tdgl_ctx *ctx, const double *mu_boundary) {
    HIP_TRY(ctx, hipMemsetAsync(ctx->cvec.p, 0, ctx->n_pad * sizeof(double), ctx->stream));
    if (ctx->nb > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->b_mu.p, mu_boundary, ctx->nb * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_boundary_term, dim3(grid_for(ctx->nb)), dim3(BLOCK), 0, ctx->stream, ctx->nb,
                           ctx->b_s0.p, ctx->b_s1.p, ctx->b_c0.p, ctx->b_c1.p, ctx->b_mu.p, ctx->cvec.p);
    }
    refresh_ceff(ctx);
    HIP_TRY(ctx, hipGetLastError());
    return TDGL_OK;
}

extern "C" int tdgl_set_mu_boundary(tdgl_ctx *ctx, const double *mu_boundary) {
    CTX_GUARD(ctx);
    if (ctx->nb > 0 && !mu_boundary) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_mu_boundary: null array");
    TDGL_TRY(apply_mu_boundary(ctx, mu_boundary));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->tab_mu_host.empty() && ctx->nb > 0) ctx->tab_mu_host.assign(mu_boundary, mu_boundary + ctx->nb);
    ctx->tab_mu_dev_synced = !ctx->tab_mu_host.empty();  // (the device holds exactly this array now)
    direct_policy_reset(ctx);
    return TDGL_OK;
}

// ---- piecewise-linear time tables, evaluated inside tdgl_run ------------------------------------
// value at time t of the table (t_k, v_k): linear between nodes, constant outside
static double table_value(const std::vector<double> &t, const double *v, double time) {
    const size_t n = t.size();
    if (n == 0) return 0.0;
    if (time <= t[0]) return v[0];
    if (time >= t[n - 1]) return v[n - 1];
    const size_t k = std::upper_bound(t.begin(), t.end(), time) - t.begin();  // t[k-1] <= time < t[k]
    const double t0 = t[k - 1], t1 = t[k];
    return v[k - 1] + (v[k] - v[k - 1]) * ((time - t0) / (t1 - t0));
}

static bool table_times_ok(const double *times, int32_t n) {
    for (int32_t k = 0; k < n; ++k)
        if (!std::isfinite(times[k]) || (k > 0 && !(times[k] > times[k - 1]))) return false;
    return true;
}

extern "C" int tdgl_set_mu_boundary_table(tdgl_ctx *ctx, int32_t n_nodes, const double *times, int32_t n_groups,
                                          const int32_t *group_ptr, const int32_t *group_pos, const double *density) {
    CTX_GUARD(ctx);
    ctx->tab_mu_t.clear();
    ctx->tab_mu_host.clear();
    ctx->tab_mu_dev_synced = false;
    if (n_nodes == 0) return TDGL_OK;  // off
    if (n_nodes < 1 || n_groups < 1 || !times || !group_ptr || !group_pos || !density || !table_times_ok(times, n_nodes))
        TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_mu_boundary_table: bad table (times must increase strictly)");
    for (int32_t k = 0; k < group_ptr[n_groups]; ++k)
        if (group_pos[k] < 0 || group_pos[k] >= ctx->nb)
            TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_mu_boundary_table: boundary position %d out of range", group_pos[k]);
    ctx->tab_mu_t.assign(times, times + n_nodes);
    ctx->tab_mu_dens.assign(density, density + (size_t)n_groups * n_nodes);
    ctx->tab_mu_ptr.assign(group_ptr, group_ptr + n_groups + 1);
    ctx->tab_mu_pos.assign(group_pos, group_pos + group_ptr[n_groups]);
    ctx->tab_mu_last.assign(n_groups, 0.0);                        // solver.py:323: densities start at 0
    ctx->tab_mu_host.assign(std::max<int64_t>(ctx->nb, 1), 0.0);  // ... and mu_boundary at 0 (solver.py:289)
    // device copy for the run-ahead loop (k_ra_mu_table); single-GPU contexts only, like that loop
    ctx->tab_mu_on_device = false;
    if (ctx->nb > 0 && ctx->n_own == ctx->n) {
        std::vector<int32_t> group((size_t)ctx->nb, -1);
        for (int32_t g = 0; g < n_groups; ++g)
            for (int32_t k = group_ptr[g]; k < group_ptr[g + 1]; ++k) group[(size_t)group_pos[k]] = g;
        HIP_TRY(ctx, ctx->d_tab_mu_group.upload(group));
        HIP_TRY(ctx, ctx->d_tab_mu_t.upload(ctx->tab_mu_t));
        HIP_TRY(ctx, ctx->d_tab_mu_dens.upload(ctx->tab_mu_dens));
        if (ctx->d_b_sites.n == 0) {
            std::vector<int32_t> s0((size_t)ctx->nb), s1((size_t)ctx->nb);
            HIP_TRY(ctx, hipMemcpy(s0.data(), ctx->b_s0.p, s0.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(s1.data(), ctx->b_s1.p, s1.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            s0.insert(s0.end(), s1.begin(), s1.end());
            std::sort(s0.begin(), s0.end());
            s0.erase(std::unique(s0.begin(), s0.end()), s0.end());
            HIP_TRY(ctx, ctx->d_b_sites.upload(s0));
            ctx->n_b_sites = (int32_t)s0.size();
        }
        ctx->tab_mu_on_device = true;
    }
    return TDGL_OK;
}

extern "C" int tdgl_set_epsilon_table(tdgl_ctx *ctx, const double *epsilon0, int32_t n_nodes, const double *times,
                                      const double *factor) {
    CTX_GUARD(ctx);
    ctx->tab_eps_t.clear();
    if (n_nodes == 0) return TDGL_OK;  // off
    if (n_nodes < 1 || !epsilon0 || !times || !factor || !table_times_ok(times, n_nodes))
        TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_epsilon_table: bad table (times must increase strictly)");
    if (ctx->tab_eps0.n == 0) HIP_TRY(ctx, ctx->tab_eps0.alloc(ctx->n_pad));
    TDGL_TRY(upload_sites(ctx, epsilon0, ctx->tab_eps0));
    ctx->tab_eps_t.assign(times, times + n_nodes);
    ctx->tab_eps_f.assign(factor, factor + n_nodes);
    ctx->tab_eps_last = NAN;
    ctx->tab_eps_on_device = false;
    if (ctx->n_own == ctx->n) {  // device copy for the run-ahead loop
        HIP_TRY(ctx, ctx->d_tab_eps_t.upload(ctx->tab_eps_t));
        HIP_TRY(ctx, ctx->d_tab_eps_f.upload(ctx->tab_eps_f));
        ctx->tab_eps_on_device = true;
    }
    return TDGL_OK;
}

// update_mu_boundary (solver.py:325-345) and update_epsilon (solver.py:364-381) for tabulated inputs,
// at the time of the step about to be taken
static int apply_time_tables(tdgl_ctx *ctx) {
    if (!ctx->tab_mu_t.empty()) {
        const size_t nn = ctx->tab_mu_t.size(), ng = ctx->tab_mu_last.size();
        bool changed = false;
        for (size_t g = 0; g < ng; ++g) {
            const double d = table_value(ctx->tab_mu_t, ctx->tab_mu_dens.data() + g * nn, ctx->time);
            if (d != ctx->tab_mu_last[g]) {  // solver.py:341: only when the density changed
                ctx->tab_mu_last[g] = d;
                for (int32_t k = ctx->tab_mu_ptr[g]; k < ctx->tab_mu_ptr[g + 1]; ++k) ctx->tab_mu_host[ctx->tab_mu_pos[k]] = d;
                changed = true;
            }
        }
        if (changed) {
            // (the copy of tab_mu_host is asynchronous: wait before the host array can change again)
            TDGL_TRY(apply_mu_boundary(ctx, ctx->tab_mu_host.data()));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            ctx->tab_mu_dev_synced = true;
        }
    }
    if (!ctx->tab_eps_t.empty()) {
        const double f = table_value(ctx->tab_eps_t, ctx->tab_eps_f.data(), ctx->time);
        if (!(f == ctx->tab_eps_last)) {
            hipLaunchKernelGGL(k_scale_links, dim3(grid_for(ctx->n_pad)), dim3(BLOCK), 0, ctx->stream, ctx->n_pad, f,
                               (const double *)ctx->tab_eps0.p, ctx->eps.p);
            ctx->tab_eps_last = f;
            ctx->have_eps = true;
        }
    }
    return TDGL_OK;
}

extern "C" int tdgl_set_state(tdgl_ctx *ctx, const double *psi, const double *mu) {
    CTX_GUARD(ctx);
    if (!psi || !mu) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_state: null array");
    TDGL_TRY(upload_sites(ctx, reinterpret_cast<const double2 *>(psi), ctx->psi[ctx->cur]));
    TDGL_TRY(upload_sites(ctx, mu, ctx->mu));
    ctx->have_state = true;
    ctx->lap_valid = false;
    ctx->currents_valid = false;
    ctx->currents_deferred = false;
    ctx->ra_retries = 0;
    ctx->prev_dt = ctx->prev_dt2 = 0.0;  // no mu history: the next solve starts from mu itself
    ctx->g_count = 0;                     // (nor a projection basis)
    ctx->g_row_pending = false;
    direct_policy_reset(ctx);             // (nor a history for the loop's solver choice)
    return TDGL_OK;
}

extern "C" int tdgl_set_controller(tdgl_ctx *ctx, const tdgl_controller *c) {
    if (!ctx || !c) return TDGL_ERR_ARG;
    if (c->dt_init > c->dt_max) TDGL_FAIL(ctx, TDGL_ERR_ARG, "dt_init must be less than or equal to dt_max.");
    if (!(c->adaptive_time_step_multiplier > 0 && c->adaptive_time_step_multiplier < 1))
        TDGL_FAIL(ctx, TDGL_ERR_ARG, "adaptive_time_step_multiplier must be in (0, 1) (got %g).",
                  c->adaptive_time_step_multiplier);
    if (c->adaptive_window < 0) TDGL_FAIL(ctx, TDGL_ERR_ARG, "adaptive_window must be >= 0 (got %d).", c->adaptive_window);
    ctx->ctl = *c;
    ctx->tentative_dt = c->dt_init;                       // solver.py:319
    ctx->dt_cap = c->adaptive ? c->dt_max : c->dt_init;   // solver.py:320
    ctx->d_psi_sq_vals.clear();                           // solver.py:318
    ctx->runner_dt = c->dt_init;                          // runner.py:262
    ctx->time = 0.0;
    ctx->stage_step = 0;
    return TDGL_OK;
}

extern "C" int tdgl_set_probes(tdgl_ctx *ctx, const int32_t *sites, int32_t n_probe) {
    CTX_GUARD(ctx);
    if (n_probe < 0 || (n_probe > 0 && !sites)) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_set_probes: bad arguments");
    ctx->probes.clear();
    ctx->probe_ring_count = 0;
    for (int k = 0; k < n_probe; ++k) {
        if (sites[k] < 0 || sites[k] >= ctx->n) TDGL_FAIL(ctx, TDGL_ERR_ARG, "probe site %d out of range", sites[k]);
        ctx->probes.push_back(ctx->iperm[sites[k]]);
    }
    if (ctx->h_probe_out) {
        (void)hipHostFree(ctx->h_probe_out);
        ctx->h_probe_out = nullptr;
    }
    if (n_probe > 0) {
        HIP_TRY(ctx, ctx->d_probes.upload(ctx->probes));
        HIP_TRY(ctx, ctx->d_probe_out.alloc((size_t)PROBE_RING_STEPS * 2 * n_probe));
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_probe_out),
                                   (size_t)PROBE_RING_STEPS * 2 * n_probe * sizeof(double)));
    }
    return TDGL_OK;
}

// ---------------------------------------------------------------------------------------
static int ensure_currents(tdgl_ctx *ctx) {
    if (ctx->currents_valid) return TDGL_OK;
    if (!ctx->have_links) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "link exponents not set");
    launch_edge_currents(ctx, ctx->psi[ctx->cur].p, ctx->mu.p, ctx->js.p, ctx->jn.p);
    HIP_TRY(ctx, hipGetLastError());
    ctx->currents_valid = true;
    return TDGL_OK;
}

extern "C" int tdgl_get_state(tdgl_ctx *ctx, double *psi, double *mu, double *supercurrent,
                              double *normal_current) {
    CTX_GUARD(ctx);
    if (!ctx->have_state) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "tdgl_get_state: no state set");
    if (psi) TDGL_TRY(download_sites(ctx, ctx->psi[ctx->cur].p, reinterpret_cast<double2 *>(psi)));
    if (mu) TDGL_TRY(download_sites(ctx, ctx->mu.p, mu));
    if (supercurrent || normal_current) {
        TDGL_TRY(ensure_currents(ctx));
        if (supercurrent) TDGL_TRY(download_edges(ctx, ctx->js.p, supercurrent));
        if (normal_current) TDGL_TRY(download_edges(ctx, ctx->jn.p, normal_current));
    }
    return TDGL_OK;
}

// ---------------------------------------------------------------------------------------
// single-operator entry points (parity tests)
struct Scratch {
    DevBuf<double2> c0, c1;
    DevBuf<double> r0, r1;
};

extern "C" int tdgl_apply_psi_laplacian(tdgl_ctx *ctx, const double *psi, double *out) {
    CTX_GUARD(ctx);
    if (!psi || !out) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    if (!ctx->have_links) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "link exponents not set");
    Scratch s;
    HIP_TRY(ctx, s.c0.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.c1.alloc(ctx->n_pad));
    TDGL_TRY(upload_sites(ctx, reinterpret_cast<const double2 *>(psi), s.c0));
    launch_psi_laplacian(ctx, false, s.c0.p, s.c1.p);
    HIP_TRY(ctx, hipGetLastError());
    return download_sites(ctx, s.c1.p, reinterpret_cast<double2 *>(out));
}

extern "C" int tdgl_supercurrent(tdgl_ctx *ctx, const double *psi, double *out) {
    CTX_GUARD(ctx);
    if (!psi || !out) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    if (!ctx->have_links) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "link exponents not set");
    Scratch s;
    HIP_TRY(ctx, s.c0.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.r0.alloc(ctx->m_pad));
    TDGL_TRY(upload_sites(ctx, reinterpret_cast<const double2 *>(psi), s.c0));
    launch_edge_currents(ctx, s.c0.p, nullptr, s.r0.p, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return download_edges(ctx, s.r0.p, out);
}

// psi_gradient @ psi (operators.py:340-344): [n_edges] complex
extern "C" int tdgl_apply_psi_gradient(tdgl_ctx *ctx, const double *psi, double *out) {
    CTX_GUARD(ctx);
    if (!psi || !out) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    if (!ctx->have_links) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "link exponents not set");
    Scratch s;
    HIP_TRY(ctx, s.c0.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.c1.alloc(ctx->m_pad));
    TDGL_TRY(upload_sites(ctx, reinterpret_cast<const double2 *>(psi), s.c0));
    hipLaunchKernelGGL(k_edge_gradient, dim3(grid_for(ctx->m)), dim3(BLOCK), 0, ctx->stream, ctx->m, ctx->e0.p,
                       ctx->e1.p, ctx->e_inv_len.p, ctx->e_U.p, s.c0.p, s.c1.p);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<double2> tmp(ctx->m);
    HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), s.c1.p, ctx->m * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double2 *o = reinterpret_cast<double2 *>(out);
    for (int64_t k = 0; k < ctx->m; ++k) o[ctx->edge_perm[k]] = tmp[k];
    return TDGL_OK;
}

// divergence @ edge_field (operators.py:59-84): [n_edges] real -> [n_sites]
extern "C" int tdgl_apply_divergence(tdgl_ctx *ctx, const double *edge_field, double *out) {
    CTX_GUARD(ctx);
    if (!edge_field || !out) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    if (ctx->n_own != ctx->n) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_apply_divergence: single-GPU contexts only");
    Scratch s;
    HIP_TRY(ctx, s.r0.alloc(ctx->m_pad));
    HIP_TRY(ctx, s.r1.alloc(ctx->n_pad));
    std::vector<double> tmp(ctx->m_pad, 0.0);
    for (int64_t k = 0; k < ctx->m; ++k) tmp[k] = edge_field[ctx->edge_perm[k]];
    HIP_TRY(ctx, hipMemcpyAsync(s.r0.p, tmp.data(), ctx->m_pad * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ceff, dim3(grid_for((int64_t)ctx->lap_pat.n_slices * WAVE)), dim3(BLOCK), 0, ctx->stream,
                       ctx->lap_pat.n_slices, ctx->lap_pat.n_rows, ctx->lap_pat.slice_off.p, ctx->lap_slot_edge.p,
                       ctx->lap_slot_w.p, ctx->e_inv_len.p, (const double *)s.r0.p, (const double *)nullptr, s.r1.p);
    HIP_TRY(ctx, hipGetLastError());
    return download_sites(ctx, s.r1.p, out);
}

// mu_laplacian @ mu (operators.py:285)
extern "C" int tdgl_apply_mu_laplacian(tdgl_ctx *ctx, const double *mu, double *out) {
    CTX_GUARD(ctx);
    if (!mu || !out) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    if (ctx->n_own != ctx->n) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_apply_mu_laplacian: single-GPU contexts only");
    Scratch s;
    HIP_TRY(ctx, s.r0.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.r1.alloc(ctx->n_pad));
    TDGL_TRY(upload_sites(ctx, mu, s.r0));
    hipLaunchKernelGGL(k_mu_laplacian, dim3(grid_for((int64_t)ctx->lap_pat.n_slices * WAVE)), dim3(BLOCK), 0,
                       ctx->stream, ctx->lap_pat.n_slices, ctx->lap_pat.n_rows, ctx->lap_pat.slice_off.p,
                       ctx->lap_pat.cols.p, ctx->lap_slot_w.p, ctx->lap_diag.p, (const double *)s.r0.p, s.r1.p);
    HIP_TRY(ctx, hipGetLastError());
    return download_sites(ctx, s.r1.p, out);
}

// mu_boundary_laplacian @ mu_boundary (operators.py:188-230, unmasked as built at :286)
extern "C" int tdgl_apply_mu_boundary_laplacian(tdgl_ctx *ctx, const double *mu_boundary, double *out) {
    CTX_GUARD(ctx);
    if ((ctx->nb > 0 && !mu_boundary) || !out) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    Scratch s;
    DevBuf<double> mb;
    HIP_TRY(ctx, s.r0.alloc(ctx->n_pad));
    HIP_TRY(ctx, mb.alloc(std::max<int64_t>(ctx->nb, 1)));
    if (ctx->nb > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(mb.p, mu_boundary, ctx->nb * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_boundary_term, dim3(grid_for(ctx->nb)), dim3(BLOCK), 0, ctx->stream, ctx->nb,
                           ctx->b_s0.p, ctx->b_s1.p, ctx->b_c0.p, ctx->b_c1.p, (const double *)mb.p, s.r0.p);
    }
    HIP_TRY(ctx, hipGetLastError());
    return download_sites(ctx, s.r0.p, out);
}

extern "C" int tdgl_normal_current(tdgl_ctx *ctx, const double *mu, double *out) {
    CTX_GUARD(ctx);
    if (!mu || !out) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    Scratch s;
    HIP_TRY(ctx, s.r0.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.r1.alloc(ctx->m_pad));
    TDGL_TRY(upload_sites(ctx, mu, s.r0));
    launch_edge_currents(ctx, nullptr, s.r0.p, nullptr, s.r1.p);
    HIP_TRY(ctx, hipGetLastError());
    return download_edges(ctx, s.r1.p, out);
}

extern "C" int tdgl_psi_update(tdgl_ctx *ctx, const double *psi, const double *mu, double dt,
                               double *psi_out, double *abs_sq_out, int32_t *ok) {
    CTX_GUARD(ctx);
    if (!psi || !mu || !psi_out || !ok) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    if (!ctx->have_links || !ctx->have_eps) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "link exponents / epsilon not set");
    Scratch s;
    DevBuf<double2> lap, pnew;
    HIP_TRY(ctx, s.c0.alloc(ctx->n_pad));
    HIP_TRY(ctx, lap.alloc(ctx->n_pad));
    HIP_TRY(ctx, pnew.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.r0.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.r1.alloc(ctx->n_pad));
    TDGL_TRY(upload_sites(ctx, reinterpret_cast<const double2 *>(psi), s.c0));
    TDGL_TRY(upload_sites(ctx, mu, s.r0));
    launch_psi_laplacian(ctx, false, s.c0.p, lap.p);
    launch_psi_update(ctx, s.c0.p, s.r0.p, lap.p, dt, pnew.p, s.r1.p);
    publish_status(ctx);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status.p, sizeof(StepStatus), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *ok = ctx->h_status->fail_flag ? 0 : 1;
    TDGL_TRY(download_sites(ctx, pnew.p, reinterpret_cast<double2 *>(psi_out)));
    if (abs_sq_out) TDGL_TRY(download_sites(ctx, s.r1.p, abs_sq_out));
    return TDGL_OK;
}

extern "C" int tdgl_poisson_rhs(tdgl_ctx *ctx, const double *psi, double *rhs) {
    CTX_GUARD(ctx);
    if (!psi || !rhs) TDGL_FAIL(ctx, TDGL_ERR_ARG, "null argument");
    if (!ctx->have_links) TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "link exponents not set");
    Scratch s;
    HIP_TRY(ctx, s.c0.alloc(ctx->n_pad));
    HIP_TRY(ctx, s.c1.alloc(ctx->n_pad));
    TDGL_TRY(upload_sites(ctx, reinterpret_cast<const double2 *>(psi), s.c0));
    launch_psi_laplacian(ctx, true, s.c0.p, s.c1.p);  // writes ctx->bvec = -a * rhs
    HIP_TRY(ctx, hipGetLastError());
    std::vector<double> b(ctx->n);
    HIP_TRY(ctx, hipMemcpyAsync(b.data(), ctx->bvec.p, ctx->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> area(ctx->n);
    HIP_TRY(ctx, hipMemcpyAsync(area.data(), ctx->area.p, ctx->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < ctx->n; ++i) rhs[ctx->perm[i]] = -b[i] / area[i];
    return TDGL_OK;
}

// ---------------------------------------------------------------------------------------
// an event for the in-run timers: from the pool when it has one (no creation in a timed region)
static int profile_event(tdgl_ctx *ctx, hipEvent_t *ev) {
    if (!ctx->prof_pool.empty()) {
        *ev = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return TDGL_OK;
    }
    HIP_TRY(ctx, hipEventCreate(ev));
    return TDGL_OK;
}

extern "C" int tdgl_profile_enable(tdgl_ctx *ctx, int32_t on) {
    if (!ctx) return TDGL_ERR_ARG;
    if (on) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        while (ctx->prof_pool.size() < 1024) {  // 448 timed steps + the 64 sampled A p launches per read-out
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->prof_pool.push_back(e);
        }
    }
    ctx->profile = on != 0;
    ctx->prof_launches = 0;
    ctx->prof_ms = 0.0;
    ctx->prof2_launches = 0;
    ctx->prof2_ms = 0.0;
    ctx->prof2_budget = on ? 64 : 0;
    ctx->prof2_seen = 0;
    ctx->prof3_launches = 0;
    ctx->prof3_ms = 0.0;
    return TDGL_OK;
}

static int profile_drain(tdgl_ctx *ctx) {
    for (auto &pr : ctx->prof_pending) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(pr.second));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
        ctx->prof_ms += ms;
        ctx->prof_launches += 1;
        ctx->prof_pool.push_back(pr.first);
        ctx->prof_pool.push_back(pr.second);
    }
    ctx->prof_pending.clear();
    for (auto &pr : ctx->prof2_pending) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(pr.second));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
        ctx->prof2_ms += ms;
        ctx->prof2_launches += 1;
        ctx->prof_pool.push_back(pr.first);
        ctx->prof_pool.push_back(pr.second);
    }
    ctx->prof2_pending.clear();
    for (auto &pr : ctx->prof3_pending) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(pr.second));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
        ctx->prof3_ms += ms;
        ctx->prof3_launches += 1;
        ctx->prof_pool.push_back(pr.first);
        ctx->prof_pool.push_back(pr.second);
    }
    ctx->prof3_pending.clear();
    return TDGL_OK;
}

extern "C" int tdgl_profile_read_pcg(tdgl_ctx *ctx, int64_t *launches, double *total_ms) {
    CTX_GUARD(ctx);
    TDGL_TRY(profile_drain(ctx));
    if (launches) *launches = ctx->prof2_launches;
    if (total_ms) *total_ms = ctx->prof2_ms;
    return TDGL_OK;
}

extern "C" int tdgl_profile_read_direct(tdgl_ctx *ctx, int64_t *solves, double *total_ms) {
    CTX_GUARD(ctx);
    TDGL_TRY(profile_drain(ctx));
    if (solves) *solves = ctx->prof3_launches;
    if (total_ms) *total_ms = ctx->prof3_ms;
    return TDGL_OK;
}

extern "C" int tdgl_profile_read(tdgl_ctx *ctx, int64_t *launches, double *total_ms) {
    CTX_GUARD(ctx);
    TDGL_TRY(profile_drain(ctx));
    if (launches) *launches = ctx->prof_launches;
    if (total_ms) *total_ms = ctx->prof_ms;
    return TDGL_OK;
}

// What an event pair reads with nothing between the two records: the marker packets' own
// processing.  bench.py reports it next to the in-run kernel durations so that they can be
// reconciled with rocprofv3's dispatch durations (which do not contain it).
extern "C" int tdgl_profile_event_overhead(tdgl_ctx *ctx, int32_t reps, double *avg_ms) {
    CTX_GUARD(ctx);
    if (reps < 1 || !avg_ms) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_profile_event_overhead: reps >= 1 and avg_ms required");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    for (int i = 0; i < reps; ++i) {
        hipEvent_t e0, e1;
        HIP_TRY(ctx, hipEventCreate(&e0));
        HIP_TRY(ctx, hipEventCreate(&e1));
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        total += ms;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    *avg_ms = total / reps;
    return TDGL_OK;
}

extern "C" int tdgl_time_kernel(tdgl_ctx *ctx, int32_t kernel, int32_t reps, double *avg_ms) {
    CTX_GUARD(ctx);
    if (!avg_ms || reps <= 0) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_time_kernel: bad arguments");
    if (!ctx->have_links || !ctx->have_state || !ctx->have_eps)
        TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "tdgl_time_kernel: set link exponents, epsilon and state first");
    if ((kernel == 4 || kernel == 5) && ctx->levels.empty())
        TDGL_FAIL(ctx, TDGL_ERR_NOT_READY, "tdgl_time_kernel: no AMG hierarchy");
    DevBuf<double2> tmp_c;
    DevBuf<double> tmp_r, tmp_r2;
    DevBuf<unsigned> bar_cnt;
    DevBuf<int> bar_err;
    DevBuf<double2> big;
    if (kernel >= 16 && kernel <= 21) HIP_TRY(ctx, big.alloc((size_t)1 << 26));  // 1 GiB
    HIP_TRY(ctx, bar_cnt.alloc(1));
    HIP_TRY(ctx, bar_err.alloc(32));
    HIP_TRY(ctx, tmp_c.alloc(ctx->n_pad));
    HIP_TRY(ctx, tmp_r.alloc(std::max<int64_t>(std::max(ctx->n_pad, ctx->m_pad), 256 * BLOCK)));
    HIP_TRY(ctx, tmp_r2.alloc(std::max(ctx->n_pad, ctx->m_pad)));
    const double2 *psi = ctx->psi[ctx->cur].p;
    auto once = [&]() -> int {
        switch (kernel) {
            case 0: launch_psi_laplacian(ctx, false, psi, tmp_c.p); break;
            case 1: launch_psi_laplacian(ctx, true, psi, tmp_c.p); break;
            case 2: launch_psi_update(ctx, psi, ctx->mu.p, ctx->lap[ctx->cur].p, 1e-4, tmp_c.p, nullptr); break;
            case 3: launch_edge_currents(ctx, psi, ctx->mu.p, tmp_r.p, tmp_r2.p); break;
            case 4: poisson_spmv_level0(ctx, ctx->mu.p, tmp_r.p); break;
            case 5: vcycle(ctx, ctx->bvec.p, /*result*/ nullptr); break;
            case 6:
                hipLaunchKernelGGL(k_copy_d2, dim3(grid_for(ctx->n_pad)), dim3(BLOCK), 0, ctx->stream,
                                   ctx->n_pad, psi, tmp_c.p);
                break;
            case 7:
                if (!ctx->scr_enabled) return TDGL_ERR_ARG;
                launch_induced(ctx);
                break;
            case 8:  // two trivial kernels, the second on the communication stream and back
                hipLaunchKernelGGL(k_copy_d2, dim3(1), dim3(BLOCK), 0, ctx->stream, 64, psi, tmp_c.p);
                (void)hipEventRecord(ctx->ev_pack, ctx->stream);
                (void)hipStreamWaitEvent(ctx->comm_stream, ctx->ev_pack, 0);
                hipLaunchKernelGGL(k_copy_d2, dim3(1), dim3(BLOCK), 0, ctx->comm_stream, 64, psi, tmp_c.p + 64);
                (void)hipEventRecord(ctx->ev_halo, ctx->comm_stream);
                (void)hipStreamWaitEvent(ctx->stream, ctx->ev_halo, 0);
                break;
            case 9:  // the same two kernels in one stream
                hipLaunchKernelGGL(k_copy_d2, dim3(1), dim3(BLOCK), 0, ctx->stream, 64, psi, tmp_c.p);
                hipLaunchKernelGGL(k_copy_d2, dim3(1), dim3(BLOCK), 0, ctx->stream, 64, psi, tmp_c.p + 64);
                break;
            case 10: case 11: case 12: case 13: case 14: case 15: {
                // 100 device-wide barriers inside one launch (k_barrier_bench): 10 / 11 grid.sync() with 32 /
                // 256 workgroups; 12 / 13 counter + agent-scope fences with 32 / 256; 14 the 32 workgroups
                // of one XCD (every 8th of 256) with agent-scope fences; 15 the same with L2-local fences
                const int mode = kernel <= 11 ? 0 : (kernel <= 14 ? 1 : 2);
                int stride = kernel >= 14 ? 8 : 1;
                const int grid = (kernel == 10 || kernel == 12) ? 32 : 256;
                int iters = 100, workers = grid / stride;
                (void)hipMemsetAsync(bar_cnt.p, 0, sizeof(unsigned), ctx->stream);
                unsigned *cnt = bar_cnt.p;
                double *data = tmp_r.p;
                int *err = bar_err.p;
                if (mode == 0) {
                    void *args[] = {&iters, &stride, &workers, &cnt, &data, &err};
                    if (hipLaunchCooperativeKernel(reinterpret_cast<void *>(k_barrier_bench<0>), dim3(grid), dim3(BLOCK), args, 0,
                                                   ctx->stream) != hipSuccess)
                        return TDGL_ERR_HIP;
                } else if (mode == 1) {
                    hipLaunchKernelGGL(k_barrier_bench<1>, dim3(grid), dim3(BLOCK), 0, ctx->stream, iters, stride, workers, cnt, data, err);
                } else {
                    hipLaunchKernelGGL(k_barrier_bench<2>, dim3(grid), dim3(BLOCK), 0, ctx->stream, iters, stride, workers, cnt, data, err);
                }
                break;
            }
            case 16:  // read-only stream over 1 GiB (HBM-cold: four times the Infinity Cache)
                hipLaunchKernelGGL(k_stream_read<4>, dim3(256 * 8), dim3(BLOCK), 0, ctx->stream, (int64_t)big.n, (const double2 *)big.p,
                                   tmp_r.p);
                break;
            case 18:  // ... 8 loads per lane in flight
                hipLaunchKernelGGL(k_stream_read<8>, dim3(256 * 8), dim3(BLOCK), 0, ctx->stream, (int64_t)big.n, (const double2 *)big.p,
                                   tmp_r.p);
                break;
            case 19:  // ... 16 workgroups per CU
                hipLaunchKernelGGL(k_stream_read<4>, dim3(256 * 16), dim3(BLOCK), 0, ctx->stream, (int64_t)big.n, (const double2 *)big.p,
                                   tmp_r.p);
                break;
            case 20:  // ... one pass, no loop: a workgroup per 256 x 2 entries
                hipLaunchKernelGGL(k_stream_read<2>, dim3((unsigned)(big.n / (BLOCK * 2))), dim3(BLOCK), 0, ctx->stream, (int64_t)big.n,
                                   (const double2 *)big.p, tmp_r.p);
                break;
            case 21:  // one workgroup reading 2 MiB from L2, 50 times
                hipLaunchKernelGGL(k_one_cu_read, dim3(1), dim3(1024), 0, ctx->stream, 50, (int64_t)(1 << 17), (const double2 *)big.p,
                                   tmp_r.p);
                break;
            case 17:  // copy 512 MiB -> 512 MiB
                hipLaunchKernelGGL(k_stream_copy<4>, dim3(256 * 8), dim3(BLOCK), 0, ctx->stream, (int64_t)big.n / 2,
                                   (const double2 *)big.p, big.p + big.n / 2);
                break;
            default: return TDGL_ERR_ARG;
        }
        return TDGL_OK;
    };
    for (int i = 0; i < 3; ++i)
        if (once() != TDGL_OK) TDGL_FAIL(ctx, TDGL_ERR_ARG, "tdgl_time_kernel: unknown kernel %d", kernel);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < reps; ++i) (void)once();
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipGetLastError());
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms = (double)ms / reps;
    if (kernel >= 10 && kernel <= 15) {  // stale reads, block -> XCD mapping: reported through the error string
        int h[17] = {0};
        HIP_TRY(ctx, hipMemcpy(h, bar_err.p, sizeof(h), hipMemcpyDeviceToHost));
        char buf[256];
        int off = snprintf(buf, sizeof(buf), "barrier bench %d: stale reads %d; XCC_ID of blocks 0-15:", kernel, h[0]);
        for (int b = 0; b < 16 && off < (int)sizeof(buf) - 4; ++b) off += snprintf(buf + off, sizeof(buf) - off, " %d", h[1 + b]);
        ctx->err = buf;
    }
    ctx->psi_status_pending = false;  // (the psi-update kernel may have run on scratch data)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TDGL_OK;
}
