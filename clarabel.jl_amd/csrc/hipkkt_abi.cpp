// C ABI of libclarabel_hipkkt.so (see include/hipkkt.h for the contract and the reference interfaces each entry point
// replaces): creation / destruction, getters, value updates, residuals (N4), scaling (N1), timing and diagnostics.
// The factorisation and solve entry points live in hipkkt_factor.cpp / hipkkt_solve.cpp (file map: hipkkt_internal.h).
#include "hipkkt_internal.h"

using namespace hipkkt;
using namespace hipkkt_host;

namespace hipkkt_host {
thread_local std::string g_create_error;
DebugOpts &debug_opts() { static DebugOpts o; return o; }
bool verbose() { static const bool v = getenv("HIPKKT_VERBOSE") != nullptr; return v; }   // (the one place that reads it)
}

extern "C" {

void hipkkt_default_opts(hipkkt_opts *o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->index_base = 0;
    o->supernode_max_width = kMaxSnWidth;
    o->relax_supernodes = 1;
    o->update_policy = 2;
    o->update_batch = 0;   // automatic
    o->front_min_panels = 0;
    o->dynamic_reg_eps = 1e-13;
    o->dynamic_reg_delta = 2e-7;
    o->amd_dense_scale = 1.5;
    o->user_perm = nullptr;
}

// Gives the process-wide cache of device slabs (runtime_pool.h: up to 4 GiB per device, kept across handles so that a batch of small
// problems does not pay hipMalloc / hipFree per handle) back to the driver -- for embedding hosts (Julia, torch) under memory pressure.
int32_t hipkkt_trim_cache(int32_t device_id) {
    if (hipSetDevice(device_id) != hipSuccess) return HIPKKT_ERR_DEVICE;
    RuntimePool::get().trim(device_id);
    return HIPKKT_OK;
}

int32_t hipkkt_is_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? n : 0;
}

int32_t hipkkt_create(int32_t device_id, int64_t N, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, const int64_t *dsigns, const hipkkt_opts *opts, hipkkt_handle *out) {
    if (!out || N < 0 || !colptr || (!rowval && N) || !opts) { g_create_error = "null argument"; return HIPKKT_ERR_ARGUMENT; }
    *out = nullptr;
    hipkkt_solver *S = nullptr;
    try {
        S = new hipkkt_solver();
        S->device = device_id;
        S->opts = *opts;
        S->l1 = false;
        const int64_t base = opts->index_base;
        KKTImage &K = S->img;
        K.N = N;
        K.colptr.resize(N + 1);
        for (int64_t j = 0; j <= N; j++) K.colptr[j] = colptr[j] - base;
        const int64_t nnz = K.colptr[N];
        K.rowval.resize(nnz);
        K.nzval.resize(nnz);
        for (int64_t q = 0; q < nnz; q++) { K.rowval[q] = rowval[q] - base; K.nzval[q] = nzval ? nzval[q] : 0.0; }
        K.dsigns.resize(N);
        for (int64_t j = 0; j < N; j++) K.dsigns[j] = dsigns ? dsigns[j] : 1;
        K.diag_full.resize(N);
        for (int64_t j = 0; j < N; j++) {
            if (K.colptr[j + 1] <= K.colptr[j] || K.rowval[K.colptr[j + 1] - 1] != j) {
                g_create_error = "KKT must be :triu with the diagonal stored last in every column";
                delete S;
                return HIPKKT_ERR_ARGUMENT;
            }
            K.diag_full[j] = K.colptr[j + 1] - 1;
        }
    } catch (const std::bad_alloc &) {
        delete S;
        g_create_error = "out of memory";
        return HIPKKT_ERR_ALLOC;
    }
    return finish_create(S, opts, out);
}

int32_t hipkkt_create_from_parts(int32_t device_id, int64_t n, int64_t m, const int64_t *Pcolptr,
                                 const int64_t *Prowval, const double *Pnzval, const int64_t *Acolptr,
                                 const int64_t *Arowval, const double *Anzval, int64_t ncones,
                                 const int64_t *cone_numel, const int32_t *cone_hs_dense,
                                 const int32_t *cone_sparse_kind, const int64_t *cone_dim1, const hipkkt_opts *opts,
                                 hipkkt_handle *out) {
    if (!out || !opts || n < 0 || m < 0 || !Pcolptr || !Acolptr) { g_create_error = "null argument"; return HIPKKT_ERR_ARGUMENT; }
    *out = nullptr;
    hipkkt_solver *S = nullptr;
    try {
        S = new hipkkt_solver();
        S->device = device_id;
        S->opts = *opts;
        S->l1 = true;
        const int64_t base = opts->index_base;
        std::vector<int64_t> Pp(n + 1), Ap(n + 1);
        for (int64_t j = 0; j <= n; j++) { Pp[j] = Pcolptr[j] - base; Ap[j] = Acolptr[j] - base; }
        std::vector<int64_t> Pi(Pp[n]), Ai(Ap[n]);
        for (int64_t q = 0; q < Pp[n]; q++) Pi[q] = Prowval[q] - base;
        for (int64_t q = 0; q < Ap[n]; q++) Ai[q] = Arowval[q] - base;
        std::vector<int64_t> dim1(ncones, 0);
        if (cone_dim1) for (int64_t c = 0; c < ncones; c++) dim1[c] = cone_dim1[c];
        if (ncones > 0 && cone_numel && cone_hs_dense && cone_sparse_kind) {
            S->cone_numel.assign(cone_numel, cone_numel + ncones);
            S->cone_hs_dense.assign(cone_hs_dense, cone_hs_dense + ncones);
            S->cone_sparse_kind.assign(cone_sparse_kind, cone_sparse_kind + ncones);
        }
        // the image is assembled by count -> scan -> fill kernels on the device (assemble_dev.hip); the debug switch HOST_ASSEMBLY
        // (testing build only) selects the host twin (assemble.cpp), which the GPU tests compare the device image with
        std::string err;
        if (debug_opts().host_assembly) {
            err = assemble_kkt(n, m, Pp.data(), Pi.data(), Pnzval, Ap.data(), Ai.data(), Anzval, ncones, cone_numel, cone_hs_dense,
                               cone_sparse_kind, dim1.data(), S->img);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            try {
                init_runtime(S);
                S->runtime_ready = true;
            } catch (const DeviceError &e) {
                g_create_error = e.msg; delete S; return HIPKKT_ERR_DEVICE;
            }
            S->t_init_runtime = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            err = assemble_kkt_device((void *)S->stream, n, m, Pp.data(), Pi.data(), Pnzval, Ap.data(), Ai.data(), Anzval, ncones, cone_numel,
                                      cone_hs_dense, cone_sparse_kind, dim1.data(), S->img);
        }
        if (!err.empty()) { g_create_error = err; delete S; return HIPKKT_ERR_ARGUMENT; }
    } catch (const std::bad_alloc &) {
        delete S;
        g_create_error = "out of memory";
        return HIPKKT_ERR_ALLOC;
    }
    S->t_created = std::chrono::steady_clock::now();
    return finish_create(S, opts, out);
}

void hipkkt_destroy(hipkkt_handle h) {
    if (h && verbose()) {
        const auto t0 = std::chrono::steady_clock::now();
        delete h;
        fprintf(stderr, "hipkkt: destroy %.2f ms\n", 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        return;
    }
    delete h;
}

int32_t hipkkt_get_dims(hipkkt_handle h, int64_t *o) {
    if (!h || !o) return HIPKKT_ERR_ARGUMENT;
    const KKTImage &K = h->img;
    const HostPlan &P = h->plan;
    o[0] = K.N; o[1] = K.n; o[2] = K.m; o[3] = K.p; o[4] = h->nnzK; o[5] = K.nHs; o[6] = (int64_t)K.smaps.size();
    o[7] = K.nnzP; o[8] = K.nnzA; o[9] = P.nnzL; o[10] = P.nsuper; o[11] = P.nlevels; o[12] = P.panel_doubles;
    o[13] = (int64_t)P.upd_tasks.size(); o[14] = P.etree_height; o[15] = P.ordering_used;
    return HIPKKT_OK;
}

int32_t hipkkt_info(hipkkt_handle h, int64_t *nnzA, int64_t *nnzL) {
    if (!h) return HIPKKT_ERR_ARGUMENT;
    if (nnzA) *nnzA = h->nnzK;
    if (nnzL) *nnzL = h->plan.nnzL;
    return HIPKKT_OK;
}

int32_t hipkkt_get_cost_model(hipkkt_handle h, double *o) {
    if (!h || !o) return HIPKKT_ERR_ARGUMENT;
    const HostPlan &P = h->plan;
    const double N = P.N, nnzK = (double)P.nnzK, nnzL = (double)P.nnzL;
    o[0] = P.flops_colcount;
    o[1] = P.flops_exec;
    o[2] = 4.0 * nnzL + N;
    o[3] = 8.0 * (nnzK + nnzL + N);
    o[4] = 2.0 * (8.0 + 4.0) * nnzL + 8.0 * 5.0 * N;
    o[5] = (8.0 + 4.0) * nnzK + 8.0 * 3.0 * N;
    o[6] = P.flops_update;
    o[7] = P.flops_update_dense;
    return HIPKKT_OK;
}

int32_t hipkkt_get_kkt(hipkkt_handle h, int64_t *colptr, int64_t *rowval, double *nzval) {
    HK_ENTER(h)
    const KKTImage &K = S->img;
    const int64_t base = S->opts.index_base;
    if (colptr) for (int64_t j = 0; j <= K.N; j++) colptr[j] = K.colptr[j] + base;
    if (rowval) for (int64_t q = 0; q < S->nnzK; q++) rowval[q] = K.rowval[q] + base;
    if (nzval) copy_sync(S->stream, nzval, S->dp.kval, (size_t)S->nnzK * sizeof(double), hipMemcpyDeviceToHost);
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_get_perm(hipkkt_handle h, int64_t *perm) {
    if (!h || !perm) return HIPKKT_ERR_ARGUMENT;
    const hipkkt_solver *T = (h->using_fallback && h->fallback) ? h->fallback : h;   // the order of the factorisation in use
    for (int k = 0; k < h->N; k++) perm[k] = T->plan.perm[k] + h->opts.index_base;
    return HIPKKT_OK;
}

int32_t hipkkt_get_dsigns(hipkkt_handle h, int64_t *dsigns) {
    if (!h || !dsigns) return HIPKKT_ERR_ARGUMENT;
    for (int k = 0; k < h->N; k++) dsigns[k] = h->img.dsigns[k];
    return HIPKKT_OK;
}

int32_t hipkkt_get_map(hipkkt_handle h, int32_t which, int64_t *out) {
    if (!h || !out) return HIPKKT_ERR_ARGUMENT;
    const KKTImage &K = h->img;
    const std::vector<int64_t> *v = nullptr;
    switch (which) {
        case 0: v = &K.mapP; break;
        case 1: v = &K.mapA; break;
        case 2: v = &K.mapHs; break;
        case 3: v = &K.diagP; break;
        case 4: v = &K.diag_full; break;
        default: return HIPKKT_ERR_ARGUMENT;
    }
    for (size_t i = 0; i < v->size(); i++) out[i] = (*v)[i] + h->opts.index_base;
    return HIPKKT_OK;
}

int32_t hipkkt_get_sparse_map(hipkkt_handle h, int64_t i, int32_t which, int64_t *out, int64_t *len) {
    if (!h || i < 0 || i >= (int64_t)h->img.smaps.size() || which < 0 || which > 3) return HIPKKT_ERR_ARGUMENT;
    const SparseMap &sm = h->img.smaps[i];
    if (which == 3) {
        if (len) *len = sm.pdim;
        if (out) for (int t = 0; t < sm.pdim; t++) out[t] = sm.D[t] + h->opts.index_base;
    } else {
        if (len) *len = (int64_t)sm.vec[which].size();
        if (out) for (size_t q = 0; q < sm.vec[which].size(); q++) out[q] = sm.vec[which][q] + h->opts.index_base;
    }
    return HIPKKT_OK;
}

// ---- value updates ----------------------------------------------------------------------------

int32_t hipkkt_update_values(hipkkt_handle h, const int64_t *index, const double *values, int64_t k) {
    HK_ENTER(h)
    if (k < 0 || (k && (!index || !values))) return HIPKKT_ERR_ARGUMENT;
    if (k == 0) return HIPKKT_OK;
    S->ensure_stage(k);
    std::vector<int64_t> idx(k);
    for (int64_t i = 0; i < k; i++) {
        idx[i] = index[i] - S->opts.index_base;
        if (idx[i] < 0 || idx[i] >= S->nnzK) { S->err = "index out of range"; return HIPKKT_ERR_ARGUMENT; }
    }
    HK_CHECK(hipMemcpyAsync(S->d_stage_idx, idx.data(), k * sizeof(int64_t), hipMemcpyHostToDevice, S->stream));
    HK_CHECK(hipMemcpyAsync(S->d_stage, values, k * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_scatter_values(S->stream, S->dp.kval, S->d_stage_idx, S->d_stage, k, 1.0);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_scale_values(hipkkt_handle h, const int64_t *index, int64_t k, double scale) {
    HK_ENTER(h)
    if (k < 0 || (k && !index)) return HIPKKT_ERR_ARGUMENT;
    if (k == 0) return HIPKKT_OK;
    S->ensure_stage(k);
    std::vector<int64_t> idx(k);
    for (int64_t i = 0; i < k; i++) {
        idx[i] = index[i] - S->opts.index_base;
        if (idx[i] < 0 || idx[i] >= S->nnzK) { S->err = "index out of range"; return HIPKKT_ERR_ARGUMENT; }
    }
    HK_CHECK(hipMemcpyAsync(S->d_stage_idx, idx.data(), k * sizeof(int64_t), hipMemcpyHostToDevice, S->stream));
    launch_scale_values(S->stream, S->dp.kval, S->d_stage_idx, k, scale);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_set_hs_dev(hipkkt_handle h, const double *hs_dev, int64_t nHs) {
    HK_ENTER(h)
    if (!S->l1 || nHs != S->img.nHs || (nHs && !hs_dev)) { S->err = "set_hs: wrong length / not an L1 handle"; return HIPKKT_ERR_ARGUMENT; }
    launch_scatter_values(S->stream, S->dp.kval, S->d_mapHs, hs_dev, nHs, -1.0);
    // no host synchronisation (round 6: a device-resident input needs none; it cost a round trip + a cold start of the next launch
    // per IPM iteration).  The factorisation that follows runs on the same stream; the second solve context's stream is made to
    // wait for this event the next time it is used (hipkkt_solve.cpp solve_begin).  hs_dev must stay valid until the next call that
    // synchronises (hipkkt_refactor, any solve) has returned.
    HK_CHECK(hipEventRecord(S->ev3, S->stream));
    S->kval_event_pending = true;
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_set_hs(hipkkt_handle h, const double *hs, int64_t nHs) {
    HK_ENTER(h)
    if (!S->l1 || nHs != S->img.nHs || (nHs && !hs)) { S->err = "set_hs: wrong length / not an L1 handle"; return HIPKKT_ERR_ARGUMENT; }
    S->ensure_stage(nHs);
    HK_CHECK(hipMemcpyAsync(S->d_stage, hs, nHs * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_scatter_values(S->stream, S->dp.kval, S->d_mapHs, S->d_stage, nHs, -1.0);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_block_products(hipkkt_handle h, const double *x, const double *z, double *Px, double *ATz, double *Ax) {
    HK_ENTER(h)
    const int64_t n = S->img.n, m = S->img.m;
    if (!S->l1 || (n && !x) || (m && !z)) { S->err = "block_products: bad arguments / not an L1 handle"; return HIPKKT_ERR_ARGUMENT; }
    S->ensure_stage(3 * n + 2 * m);     // x | z | Px | ATz | Ax
    double *dx = S->d_stage, *dz = dx + n, *dPx = dz + m, *dATz = dPx + n, *dAx = dATz + n;
    if (n) HK_CHECK(hipMemcpyAsync(dx, x, n * sizeof(double), hipMemcpyHostToDevice, S->stream));
    if (m) HK_CHECK(hipMemcpyAsync(dz, z, m * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_block_products(S->stream, S->dp, dx, dz, dPx, dATz, dAx, (int)n, (int)m);
    if (Px && n) HK_CHECK(hipMemcpyAsync(Px, dPx, n * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    if (ATz && n) HK_CHECK(hipMemcpyAsync(ATz, dATz, n * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    if (Ax && m) HK_CHECK(hipMemcpyAsync(Ax, dAx, m * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

// residuals_update! on the device.  q and b become resident with hipkkt_set_qb.
int32_t hipkkt_set_qb(hipkkt_handle h, const double *q, const double *b) {
    HK_ENTER(h)
    const int64_t n = S->img.n, m = S->img.m;
    if (!S->l1 || (n && !q) || (m && !b)) { S->err = "set_qb: bad arguments / not an L1 handle"; return HIPKKT_ERR_ARGUMENT; }
    if (!S->d_qb) {
        S->d_qb = S->dalloc<double>(n + m);
        S->d_res_in = S->dalloc<double>(n + 2 * m);                  // x | z | s
        S->d_res_out = S->dalloc<double>(3 * n + 2 * m + 8);         // rx | rz | rx_inf | rz_inf | Px | 5 scalars
        S->d_res_part = S->dalloc<double>(4 * (size_t)residual_blocks((int)n, (int)m) + 4);
    }
    if (n) HK_CHECK(hipMemcpyAsync(S->d_qb, q, n * sizeof(double), hipMemcpyHostToDevice, S->stream));
    if (m) HK_CHECK(hipMemcpyAsync(S->d_qb + n, b, m * sizeof(double), hipMemcpyHostToDevice, S->stream));
    HK_CHECK(hipStreamSynchronize(S->stream));
    S->red_have_const = false;   // the resident constant-rhs solution (x2, z2) solved the OLD [-q; b]: never combine it with the new terms
    return HIPKKT_OK;
    HK_LEAVE
}

static int32_t residuals_impl(hipkkt_solver *S, const double *xzs_dev, double tau, double kappa, double *out_dev, double *scal5) {
    const int64_t n = S->img.n, m = S->img.m;
    launch_residuals(S->stream, S->dp, xzs_dev, xzs_dev + n, xzs_dev + n + m, S->d_qb, S->d_qb + n, tau, kappa, out_dev, S->d_res_part,
                     S->d_res_out + 3 * n + 2 * m, (int)n, (int)m);
    copy_sync(S->stream, scal5, S->d_res_out + 3 * n + 2 * m, 5 * sizeof(double), hipMemcpyDeviceToHost);
    return HIPKKT_OK;
}

int32_t hipkkt_residuals(hipkkt_handle h, const double *x, const double *z, const double *s, double tau, double kappa, double *rx,
                         double *rz, double *rx_inf, double *rz_inf, double *Px, double *scal5) {
    HK_ENTER(h)
    const int64_t n = S->img.n, m = S->img.m;
    if (!S->l1 || !S->d_qb || !scal5 || (n && !x) || (m && (!z || !s))) { S->err = "residuals: call hipkkt_set_qb first / bad arguments"; return HIPKKT_ERR_ARGUMENT; }
    if (n) HK_CHECK(hipMemcpyAsync(S->d_res_in, x, n * sizeof(double), hipMemcpyHostToDevice, S->stream));
    if (m) HK_CHECK(hipMemcpyAsync(S->d_res_in + n, z, m * sizeof(double), hipMemcpyHostToDevice, S->stream));
    if (m) HK_CHECK(hipMemcpyAsync(S->d_res_in + n + m, s, m * sizeof(double), hipMemcpyHostToDevice, S->stream));
    const int32_t rc = residuals_impl(S, S->d_res_in, tau, kappa, S->d_res_out, scal5);
    double *o = S->d_res_out;
    struct { double *dst; const double *src; int64_t len; } cp[5] = {{rx, o, n}, {rz, o + n, m}, {rx_inf, o + n + m, n}, {rz_inf, o + 2 * n + m, m}, {Px, o + 2 * n + 2 * m, n}};
    for (auto &c : cp)
        if (c.dst && c.len) HK_CHECK(hipMemcpyAsync(c.dst, c.src, c.len * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    HK_CHECK(hipStreamSynchronize(S->stream));
    return rc;
    HK_LEAVE
}

int32_t hipkkt_residuals_dev(hipkkt_handle h, const double *xzs_dev, double tau, double kappa, double *out_dev, double *scal5) {
    HK_ENTER(h)
    if (!S->l1 || !S->d_qb || !xzs_dev || !out_dev || !scal5) { S->err = "residuals_dev: call hipkkt_set_qb first / bad arguments"; return HIPKKT_ERR_ARGUMENT; }
    return residuals_impl(S, xzs_dev, tau, kappa, out_dev, scal5);
    HK_LEAVE
}

int32_t hipkkt_set_hs_psd(hipkkt_handle h, int64_t npsd, const int64_t *hs_off, const int64_t *dim, const double *w_all) {
    HK_ENTER(h)
    if (!S->l1 || npsd < 0 || (npsd && (!hs_off || !dim || !w_all))) { S->err = "set_hs_psd: bad arguments / not an L1 handle"; return HIPKKT_ERR_ARGUMENT; }
    int64_t total = 0;
    for (int64_t c = 0; c < npsd; c++) {
        const int64_t n = dim[c], numel = n * (n + 1) / 2, nent = numel * (numel + 1) / 2;
        if (n < 1 || n > 30000 || hs_off[c] < 0 || hs_off[c] + nent > S->img.nHs) { S->err = "set_hs_psd: block outside the Hs vector"; return HIPKKT_ERR_ARGUMENT; }
        total += n * n;
    }
    if (npsd == 0) return HIPKKT_OK;
    S->ensure_stage(total);
    HK_CHECK(hipMemcpyAsync(S->d_stage, w_all, (size_t)total * sizeof(double), hipMemcpyHostToDevice, S->stream));
    int64_t woff = 0;
    for (int64_t c = 0; c < npsd; c++) {
        launch_psd_hs(S->stream, S->dp.kval, S->d_mapHs, hs_off[c], S->d_stage + woff, (int)dim[c]);
        woff += dim[c] * dim[c];
    }
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

// ---- N1: update_scaling! + get_Hs! of the symmetric cones on the device (scaling.hip) ---------------------------------
// kinds[c]: 0 ZeroCone, 1 NonnegativeCone, 2 SecondOrderCone, 3 PSDTriangleCone, anything else = a cone whose block the
// caller keeps setting through hipkkt_set_hs / hipkkt_set_genpow (ref: the SupportedCone types of cone_types.jl / cone_api.py)
int32_t hipkkt_set_cone_types(hipkkt_handle h, int64_t ncones, const int32_t *kinds) {
    HK_ENTER(h)
    if (!S->l1 || ncones != (int64_t)S->cone_numel.size() || (ncones && !kinds)) { S->err = "set_cone_types: not an L1 handle / wrong number of cones"; return HIPKKT_ERR_ARGUMENT; }
    const int64_t m = S->img.m;
    S->sc_ready = false;          // until this call has validated and uploaded its tables
    std::vector<signed char> kind((size_t)std::max<int64_t>(m, 1), 2);
    std::vector<int64_t> rowhs((size_t)std::max<int64_t>(m, 1), 0), socdesc;
    S->sc_psd_hs.clear(); S->sc_psd_n.clear(); S->sc_psd_total = 0; S->sc_nsoc = 0;
    int64_t row = 0, hs = 0;
    int sparse_idx = 0;
    for (int64_t c = 0; c < ncones; c++) {
        const int64_t numel = S->cone_numel[c];
        const bool dense = S->cone_hs_dense[c] != 0;
        const int sk = S->cone_sparse_kind[c];
        const int64_t blk = dense ? numel * (numel + 1) / 2 : numel;
        if (kinds[c] == 0 || kinds[c] == 1) {
            if (dense || sk != 0) { S->err = "set_cone_types: a Zero / Nonnegative cone has a diagonal Hs block and no expansion"; return HIPKKT_ERR_ARGUMENT; }
            for (int64_t i = 0; i < numel; i++) { kind[row + i] = (signed char)kinds[c]; rowhs[row + i] = hs + i; }
        } else if (kinds[c] == 2) {
            int64_t uv0 = -1, ord = -1;
            if (sk == 1) {
                ord = S->soc_of_sparse[sparse_idx];
                uv0 = S->soc_off[ord];
                if (dense || S->soc_off[ord + 1] - uv0 != numel) { S->err = "set_cone_types: sparse second-order cone does not match its expansion map"; return HIPKKT_ERR_ARGUMENT; }
            } else if (!dense || numel < 2 || numel > 4 || sk != 0) {
                // cone_types.jl:86-118: dim <= SOC_NO_EXPANSION_MAX_SIZE (4) is the dense form, everything larger the sparse one
                S->err = "set_cone_types: a second-order cone is either sparse-expanded or dense with dim <= 4"; return HIPKKT_ERR_ARGUMENT;
            }
            const int64_t d5[5] = {row, numel, hs, uv0, ord};
            socdesc.insert(socdesc.end(), d5, d5 + 5);
            S->sc_nsoc++;
        } else if (kinds[c] == 3) {
            int64_t n = (int64_t)((std::sqrt(8.0 * (double)numel + 1.0) - 1.0) * 0.5 + 0.5);
            if (!dense || sk != 0 || n * (n + 1) / 2 != numel) { S->err = "set_cone_types: a PSD triangle cone has a dense block of triangular size"; return HIPKKT_ERR_ARGUMENT; }
            S->sc_psd_hs.push_back(hs);
            S->sc_psd_n.push_back(n);
            S->sc_psd_total += n * n;
        }
        if (sk != 0) sparse_idx++;
        row += numel;
        hs += blk;
    }
    if (row != m || hs != S->img.nHs) { S->sc_ready = false; S->err = "set_cone_types: cone sizes do not add up to m / the Hs vector"; return HIPKKT_ERR_ARGUMENT; }
    if (socdesc.empty()) socdesc.assign(5, 0);
    // device buffers: allocated on the first call; a later call (same cone sizes, possibly other kinds) re-uses them when they are
    // large enough -- slab memory is only returned when the handle is destroyed, so repeated calls must not allocate again
    if (!S->d_sc_kind || (int64_t)socdesc.size() > S->sc_cap_socdesc || S->sc_psd_total > S->sc_cap_psd) {
        S->d_sc_kind = S->dalloc<signed char>(kind.size());
        S->d_sc_rowhs = S->dalloc<int64_t>(rowhs.size());
        S->d_sc_socdesc = S->dalloc<int64_t>(socdesc.size());
        S->d_sc_sz = S->dalloc<double>(2 * m);
        S->d_sc_wl = S->dalloc<double>(2 * m);
        fill_async(S->stream, S->d_sc_wl, 0, (size_t)std::max<int64_t>(2 * m, 1) * sizeof(double));   // w, lambda stay zero on the rows of cones the kernels do not write (PSD, others)
        S->d_sc_eta = S->dalloc<double>(socdesc.size() / 5);
        S->d_sc_R = S->dalloc<double>(S->sc_psd_total);
        S->d_sc_W = S->dalloc<double>(S->sc_psd_total);
        S->d_sc_fail = S->dalloc<int>(1);
        S->sc_cap_socdesc = (int64_t)socdesc.size();
        S->sc_cap_psd = S->sc_psd_total;
    }
    copy_sync(S->stream, S->d_sc_kind, kind.data(), kind.size() * sizeof(signed char), hipMemcpyHostToDevice);
    copy_sync(S->stream, S->d_sc_rowhs, rowhs.data(), rowhs.size() * sizeof(int64_t), hipMemcpyHostToDevice);
    copy_sync(S->stream, S->d_sc_socdesc, socdesc.data(), socdesc.size() * sizeof(int64_t), hipMemcpyHostToDevice);
    S->sc_ready = true;
    return HIPKKT_OK;
    HK_LEAVE
}

// s, z (length m), psd_R (concatenated n x n column-major R factors, NULL = leave the PSD blocks to hipkkt_set_hs_psd) and the three
// outputs (w and lambda of length m, eta per second-order cone; any may be NULL) are host pointers, or device pointers when `dev`
static int32_t update_scaling_impl(hipkkt_handle h, const double *s, const double *z, const double *psd_R, double *w_out,
                                   double *lambda_out, double *soc_eta_out, int32_t *scaling_ok, bool dev) {
    HK_ENTER(h)
    if (!S->sc_ready) { S->err = "update_scaling: call hipkkt_set_cone_types first"; return HIPKKT_ERR_ARGUMENT; }
    const int64_t m = S->img.m;
    if (m && (!s || !z)) { S->err = "update_scaling: null s / z"; return HIPKKT_ERR_ARGUMENT; }
    const double *ds = s, *dz = z, *dR = psd_R;
    if (!dev) {
        if (m) {
            HK_CHECK(hipMemcpyAsync(S->d_sc_sz, s, m * sizeof(double), hipMemcpyHostToDevice, S->stream));
            HK_CHECK(hipMemcpyAsync(S->d_sc_sz + m, z, m * sizeof(double), hipMemcpyHostToDevice, S->stream));
        }
        ds = S->d_sc_sz; dz = S->d_sc_sz + m;
        if (psd_R && S->sc_psd_total) {
            HK_CHECK(hipMemcpyAsync(S->d_sc_R, psd_R, (size_t)S->sc_psd_total * sizeof(double), hipMemcpyHostToDevice, S->stream));
            dR = S->d_sc_R;
        }
    }
    double *dw = S->d_sc_wl, *dl = S->d_sc_wl + m;
    launch_zero_words(S->stream, S->d_sc_fail, 1);
    launch_scaling_diag(S->stream, S->d_sc_kind, S->d_sc_rowhs, S->d_mapHs, ds, dz, dw, dl, S->dp.kval, m);
    launch_scaling_soc(S->stream, S->sc_nsoc, S->d_sc_socdesc, S->d_mapHs, ds, dz, dw, dl, S->d_sc_eta, S->d_soc_u, S->d_soc_v,
                       S->d_soc_eta2, S->dp.kval, S->d_sc_fail);
    if (S->nsoc > 0)      // u, v, D entries of the sparse cones (the same kernel hipkkt_set_soc_batch uses)
        launch_soc_batch(S->stream, S->dp.kval, S->d_soc_uidx, S->d_soc_vidx, S->d_soc_cone, S->d_soc_u, S->d_soc_v, S->d_soc_eta2,
                         S->soc_total, S->d_soc_didx, S->nsoc);
    if (dR) {
        int64_t off = 0;
        for (size_t c = 0; c < S->sc_psd_n.size(); c++) {
            const int n = (int)S->sc_psd_n[c];
            launch_psd_rrt(S->stream, dR + off, S->d_sc_W + off, n);
            launch_psd_hs(S->stream, S->dp.kval, S->d_mapHs, S->sc_psd_hs[c], S->d_sc_W + off, n);
            off += (int64_t)n * n;
        }
    }
    const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (w_out && m) HK_CHECK(hipMemcpyAsync(w_out, dw, m * sizeof(double), kind, S->stream));
    if (lambda_out && m) HK_CHECK(hipMemcpyAsync(lambda_out, dl, m * sizeof(double), kind, S->stream));
    if (soc_eta_out && S->sc_nsoc) HK_CHECK(hipMemcpyAsync(soc_eta_out, S->d_sc_eta, S->sc_nsoc * sizeof(double), kind, S->stream));
    int fail = 0;
    copy_sync(S->stream, &fail, S->d_sc_fail, sizeof(int), hipMemcpyDeviceToHost);
    if (scaling_ok) *scaling_ok = fail ? 0 : 1;
    return HIPKKT_OK;
    HK_LEAVE
}
int32_t hipkkt_update_scaling(hipkkt_handle h, const double *s, const double *z, const double *psd_R, double *w_out,
                              double *lambda_out, double *soc_eta_out, int32_t *scaling_ok) {
    return update_scaling_impl(h, s, z, psd_R, w_out, lambda_out, soc_eta_out, scaling_ok, false);
}
int32_t hipkkt_update_scaling_dev(hipkkt_handle h, const double *s_dev, const double *z_dev, const double *psd_R_dev, double *w_out_dev,
                                  double *lambda_out_dev, double *soc_eta_out_dev, int32_t *scaling_ok) {
    return update_scaling_impl(h, s_dev, z_dev, psd_R_dev, w_out_dev, lambda_out_dev, soc_eta_out_dev, scaling_ok, true);
}

int32_t hipkkt_set_soc_batch(hipkkt_handle h, int64_t nsoc, const double *eta2, const double *u_all, const double *v_all,
                             int64_t total) {
    HK_ENTER(h)
    if (!S->l1 || nsoc != S->nsoc || total != S->soc_total) { S->err = "set_soc_batch: size mismatch"; return HIPKKT_ERR_ARGUMENT; }
    if (nsoc == 0) return HIPKKT_OK;
    HK_CHECK(hipMemcpyAsync(S->d_soc_u, u_all, total * sizeof(double), hipMemcpyHostToDevice, S->stream));
    HK_CHECK(hipMemcpyAsync(S->d_soc_v, v_all, total * sizeof(double), hipMemcpyHostToDevice, S->stream));
    HK_CHECK(hipMemcpyAsync(S->d_soc_eta2, eta2, nsoc * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_soc_batch(S->stream, S->dp.kval, S->d_soc_uidx, S->d_soc_vidx, S->d_soc_cone, S->d_soc_u, S->d_soc_v,
                     S->d_soc_eta2, total, S->d_soc_didx, (int)nsoc);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_set_soc(hipkkt_handle h, int64_t sparse_idx, double eta2, const double *u, const double *v, int64_t dim) {
    HK_ENTER(h)
    if (!S->l1 || sparse_idx < 0 || sparse_idx >= (int64_t)S->img.smaps.size()) return HIPKKT_ERR_ARGUMENT;
    const int o = S->soc_of_sparse[sparse_idx];
    if (o < 0 || S->soc_off[o + 1] - S->soc_off[o] != dim) { S->err = "set_soc: not a SOC map / wrong dim"; return HIPKKT_ERR_ARGUMENT; }
    const int64_t off = S->soc_off[o];
    HK_CHECK(hipMemcpyAsync(S->d_soc_u + off, u, dim * sizeof(double), hipMemcpyHostToDevice, S->stream));
    HK_CHECK(hipMemcpyAsync(S->d_soc_v + off, v, dim * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_scatter_values(S->stream, S->dp.kval, S->d_soc_uidx + off, S->d_soc_u + off, dim, -eta2);
    launch_scatter_values(S->stream, S->dp.kval, S->d_soc_vidx + off, S->d_soc_v + off, dim, -eta2);
    const double dv[2] = {-eta2, eta2};
    HK_CHECK(hipMemcpyAsync(S->d_stage, dv, 2 * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_scatter_values(S->stream, S->dp.kval, S->d_soc_didx + 2 * o, S->d_stage, 2, 1.0);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_set_genpow(hipkkt_handle h, int64_t sparse_idx, double sqrtmu, const double *p, const double *q,
                          const double *r) {
    HK_ENTER(h)
    if (!S->l1 || sparse_idx < 0 || sparse_idx >= (int64_t)S->img.smaps.size()) return HIPKKT_ERR_ARGUMENT;
    const SparseMap &sm = S->img.smaps[sparse_idx];
    if (sm.kind != 2) { S->err = "set_genpow: not a GenPow map"; return HIPKKT_ERR_ARGUMENT; }
    const double *src[3] = {q, r, p};
    for (int t = 0; t < 3; t++) {
        const int64_t k = (int64_t)sm.vec[t].size();
        if (!k) continue;
        S->ensure_stage(k);
        HK_CHECK(hipMemcpyAsync(S->d_stage_idx, sm.vec[t].data(), k * sizeof(int64_t), hipMemcpyHostToDevice, S->stream));
        HK_CHECK(hipMemcpyAsync(S->d_stage, src[t], k * sizeof(double), hipMemcpyHostToDevice, S->stream));
        launch_scatter_values(S->stream, S->dp.kval, S->d_stage_idx, S->d_stage, k, -sqrtmu);
        HK_CHECK(hipStreamSynchronize(S->stream));
    }
    const double dv[3] = {-1.0, -1.0, 1.0};
    HK_CHECK(hipMemcpyAsync(S->d_stage_idx, sm.D, 3 * sizeof(int64_t), hipMemcpyHostToDevice, S->stream));
    HK_CHECK(hipMemcpyAsync(S->d_stage, dv, 3 * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_scatter_values(S->stream, S->dp.kval, S->d_stage_idx, S->d_stage, 3, 1.0);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_update_P(hipkkt_handle h, const double *Pnzval, int64_t nnzP) {
    HK_ENTER(h)
    if (!S->l1 || nnzP != S->img.nnzP) { S->err = "update_P: wrong length"; return HIPKKT_ERR_ARGUMENT; }
    if (!nnzP) return HIPKKT_OK;
    S->ensure_stage(nnzP);
    HK_CHECK(hipMemcpyAsync(S->d_stage, Pnzval, nnzP * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_scatter_values(S->stream, S->dp.kval, S->d_mapP, S->d_stage, nnzP, 1.0);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_update_A(hipkkt_handle h, const double *Anzval, int64_t nnzA) {
    HK_ENTER(h)
    if (!S->l1 || nnzA != S->img.nnzA) { S->err = "update_A: wrong length"; return HIPKKT_ERR_ARGUMENT; }
    if (!nnzA) return HIPKKT_OK;
    S->ensure_stage(nnzA);
    HK_CHECK(hipMemcpyAsync(S->d_stage, Anzval, nnzA * sizeof(double), hipMemcpyHostToDevice, S->stream));
    launch_scatter_values(S->stream, S->dp.kval, S->d_mapA, S->d_stage, nnzA, 1.0);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_get_timing(hipkkt_handle h, double *o) {
    if (!h || !o) return HIPKKT_ERR_ARGUMENT;
    o[0] = h->t_last_factor; o[1] = h->t_last_solve; o[2] = h->t_acc_factor; o[3] = h->t_acc_solve;
    o[4] = (double)h->n_factor; o[5] = (double)h->n_solvecalls; o[6] = (double)h->n_ldlsolves; o[7] = h->t_last_update;
    return HIPKKT_OK;
}

int32_t hipkkt_abi_version(void) { return HIPKKT_ABI_VERSION; }

int32_t hipkkt_box_probe(int32_t device_id, double *out, int64_t cap) {
    if (!out || cap < 0) return HIPKKT_ERR_ARGUMENT;
    double o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    try {
        if (hipkkt::box_probe(device_id, o) != 0) return HIPKKT_ERR_DEVICE;
    } catch (...) { return HIPKKT_ERR_DEVICE; }
    for (int64_t i = 0; i < cap && i < 8; i++) out[i] = o[i];
    return HIPKKT_OK;
}

int32_t hipkkt_get_profile(hipkkt_handle h, double *out, int64_t cap) {
    if (!h || !out || cap < 0) return HIPKKT_ERR_ARGUMENT;
    const double o[12] = {h->t_last_update, h->prof_dense4_ms, h->prof_dense4_flops, (double)h->prof_dense4_launches, h->prof_fb_ms,
                          (double)h->prof_fb_launches, (double)h->prof_fb_panels, h->prof_fb_flops, h->prof_extra_tiles, h->prof_extra_flops,
                          (double)h->last_npolish, (double)h->n_accurate_factorisations};
    for (int64_t i = 0; i < cap && i < 12; i++) out[i] = o[i];   // never more than the caller's buffer holds
    return HIPKKT_OK;
}

int32_t hipkkt_get_profile_launches(hipkkt_handle h, double *ms, double *flops, double *tiles, int64_t cap, int64_t *count) {
    if (!h) return HIPKKT_ERR_ARGUMENT;
    const int64_t n = (int64_t)h->prof_launch_ms.size();
    if (count) *count = n;
    for (int64_t i = 0; i < n && i < cap; i++) {
        if (ms) ms[i] = h->prof_launch_ms[i];
        if (flops) flops[i] = h->prof_launch_flops[i];
        if (tiles) tiles[i] = h->prof_launch_tiles[i];
    }
    return HIPKKT_OK;
}

int32_t hipkkt_get_counters(hipkkt_handle h, int64_t *out, int64_t cap) {
    if (!h || !out || cap < 0) return HIPKKT_ERR_ARGUMENT;
    int64_t o[15];
    o[0] = h->n_sweep_timeouts; o[1] = h->use_persist ? 1 : 0; o[2] = h->n_twin_refactors; o[3] = h->fallback ? 1 : 0;
    o[4] = h->using_fallback ? 1 : 0; o[5] = h->plan.ordering_used; o[6] = (int64_t)h->plan.fronts.size(); o[7] = h->nseg;
    o[8] = (int64_t)h->fbatches.size(); o[9] = h->use_front_block ? 1 : 0;
    plan_cache_counts(&o[10], &o[11]);   // process-wide: symbolic plans taken from / not found in the plan cache
    o[12] = h->fb_streamed ? 1 : 0;
    o[13] = h->n_accurate_factorisations;
    o[14] = 0;                                                            // gather entries that run on the side stream (split_gather_stages)
    for (int l = 0; l < h->plan.nlevels && l < (int)h->gath_split.size(); l++)
        if (h->gath_split[l] >= 0) o[14] += h->plan.gath_stage_ptr[l + 1] - h->plan.gath_stage_ptr[l] - h->gath_split[l];
    for (int64_t i = 0; i < cap && i < 15; i++) out[i] = o[i];
    return HIPKKT_OK;
}

// developer diagnostic (not part of the plugin contract): internal vectors of the last LDL solve / plan tables as doubles.
// what: 0 = the permuted right-hand side, 1 = z (forward result / D), 2 = x (permuted), 3 = ubuf, 4 = the unregularised KKT values,
// 5 = D and 6 = 1/D of the last factorisation (permuted order), 7 / 8 = u / v of the sparse second-order cones (concatenated), 10 = sn_first, 11 = sn_level,
// 12 = rows per supernode, 13 = sn_parent, 14 = persistent-sweep membership (1 = item of a segment launch)
int32_t hipkkt_debug_dump(hipkkt_handle h, int32_t what, double *out, int64_t cap, int64_t *len) {
    HK_ENTER(h)
    const HostPlan &P = S->plan;
    auto dev = [&](const double *p, int64_t n) {
        if (len) *len = n;
        if (out && cap >= n) copy_sync(S->stream, out, p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    };
    auto host = [&](int64_t n, auto f) {
        if (len) *len = n;
        if (out && cap >= n) for (int64_t i = 0; i < n; i++) out[i] = (double)f(i);
    };
    switch (what) {
        case 0: dev(S->d_y, S->N); break;
        case 1: dev(S->d_z, S->N); break;
        case 2: dev(S->d_xp, S->N); break;
        case 3: dev(S->dp.ubuf, P.ubuf_len); break;
        case 4: dev(S->dp.kval, S->nnzK); break;
        case 5: dev(S->dp.D, S->N); break;
        case 6: dev(S->dp.Dinv, S->N); break;
        case 9:
            if (!S->d_fb_trace) return HIPKKT_ERR_ARGUMENT;
            dev((const double *)S->d_fb_trace, (int64_t)S->fbatches.size() * 128);   // raw int64 stamps (100 MHz) in double-sized words
            break;
        case 17: {    // refinement state of the last refined solve on context 0: ||e|| before the last step, ||b||, ||e|| after it, steps
            const RefineState *r = S->ctx[0].h_rs;
            const double v[4] = {r->lastnorme, r->normb, r->norme, (double)r->steps};
            host(4, [&](int64_t i) { return v[i]; });
            break;
        }
        case 18: dev(S->ctx[0].d_e, S->N); break;       // residual b - K x of the last SpMV on context 0 (original ordering)
        case 19: host(1, [&](int64_t) { return (double)P.dtri.size(); }); break;   // dense triangles of K outside the symmetric view
        case 15: dev(S->d_fb_stream, std::max<int64_t>(S->fb_stream_doubles, 1)); break;   // stream records of the front batches (raw)
        case 16: dev(S->d_fb_scratch, (int64_t)kFbScratch * (int64_t)std::max<size_t>(S->fbatches.size(), 1)); break;
        case 20: {   // the dense tiles of the plan, 6 values each: stage, tasks, sum of source widths, tasks through a tile map, full-tile
                     // flag, sum over tasks of rows x columns x width (development: what a stage's update launch is made of)
            std::vector<double> v;
            const int nl = (int)P.upd_stage_ndense.size();
            for (int l = 0; l < nl; l++)
                for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l] + P.upd_stage_ndense[l]; g++) {
                    const UpdGroup &G = P.upd_groups[g];
                    double sumK = 0, mapped = 0, vol = 0;
                    bool full = true;
                    for (int t = G.task_begin; t < G.task_end; t++) {
                        const UpdTask &T = P.upd_tasks[t];
                        const int K = P.sn_first[T.src + 1] - P.sn_first[T.src];
                        sumK += K; vol += (double)T.nrows * T.ncols * K;
                        if (T.geom & (1 << 17)) mapped += 1;
                        if (!(T.geom & (1 << 16)) || T.nrows != 64 || T.ncols != 64 || (K & 7)) full = false;
                    }
                    const double rec[6] = {(double)l, (double)(G.task_end - G.task_begin), sumK, mapped, full ? 1.0 : 0.0, vol};
                    v.insert(v.end(), rec, rec + 6);
                }
            host((int64_t)v.size(), [&](int64_t i) { return v[i]; });
            break;
        }
        case 21: {   // per supernode solved in a persistent segment sweep (dump 14): level, width, rows, longest and mean gather list
                     // of its row slots (development: what a hop of k_fwd_seg waits for)
            std::vector<double> v;
            for (int s = 0; s < P.nsuper; s++) {
                if (!(P.sn_front[s] < 0 && P.sn_level[s] >= S->seg_lstar[S->seg_of_level[P.sn_level[s]]])) continue;
                const int64_t a = P.sn_rowptr[s], b = P.sn_rowptr[s + 1];
                int64_t mx = 0;
                for (int64_t q = a; q < b; q++) mx = std::max<int64_t>(mx, P.g_ptr[q + 1] - P.g_ptr[q]);
                const double rec[5] = {(double)P.sn_level[s], (double)(P.sn_first[s + 1] - P.sn_first[s]), (double)(b - a), (double)mx,
                                       (double)(P.g_ptr[b] - P.g_ptr[a]) / (double)std::max<int64_t>(b - a, 1)};
                v.insert(v.end(), rec, rec + 5);
            }
            host((int64_t)v.size(), [&](int64_t i) { return v[i]; });
            break;
        }
        case 7: dev(S->d_soc_u, S->soc_total); break;
        case 8: dev(S->d_soc_v, S->soc_total); break;
        case 10: host(P.nsuper + 1, [&](int64_t i) { return P.sn_first[i]; }); break;
        case 11: host(P.nsuper, [&](int64_t i) { return P.sn_level[i]; }); break;
        case 12: host(P.nsuper, [&](int64_t i) { return P.sn_rowptr[i + 1] - P.sn_rowptr[i]; }); break;
        case 13: host(P.nsuper, [&](int64_t i) { return P.sn_parent[i]; }); break;
        case 14: host(P.nsuper, [&](int64_t i) { return P.sn_front[i] < 0 && P.sn_level[i] >= S->seg_lstar[S->seg_of_level[P.sn_level[i]]] ? 1 : 0; }); break;
        default: return HIPKKT_ERR_ARGUMENT;
    }
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_reset_timing(hipkkt_handle h) {
    if (!h) return HIPKKT_ERR_ARGUMENT;
    h->t_acc_factor = h->t_acc_solve = 0;
    h->n_factor = h->n_solvecalls = h->n_ldlsolves = 0;
    return HIPKKT_OK;
}

int32_t hipkkt_set_profiling(hipkkt_handle h, int32_t enable) {
    if (!h) return HIPKKT_ERR_ARGUMENT;
    h->profiling = enable != 0;
    h->profiling_no_extra = enable == 2;
    return HIPKKT_OK;
}

// D = A(16x4) * B(4x16) through the matrix-core path used by the update kernel; returns the max
// abs deviation from the host product (layout self-test), or a negative status
int32_t hipkkt_selftest_mfma(int32_t device_id, double *max_err) {
    if (hipSetDevice(device_id) != hipSuccess) return HIPKKT_ERR_DEVICE;
    double A[64], B[64], Dh[256], Dd[256];
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 4; k++) A[i * 4 + k] = 1.0 + i * 0.37 - k * 1.13 + (i * k) * 0.05;
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 16; j++) B[k * 16 + j] = -0.5 + j * 0.21 + k * 0.77 - (j * j) * 0.013;  // asymmetric
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j];
            Dh[i * 16 + j] = s;
        }
    double *dA = nullptr, *dB = nullptr, *dD = nullptr;
    if (hipMalloc((void **)&dA, sizeof(A)) != hipSuccess || hipMalloc((void **)&dB, sizeof(B)) != hipSuccess ||
        hipMalloc((void **)&dD, sizeof(Dd)) != hipSuccess)
        return HIPKKT_ERR_ALLOC;
    (void)hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    launch_mfma_probe(nullptr, dA, dB, dD);
    hipError_t e = hipMemcpy(Dd, dD, sizeof(Dd), hipMemcpyDeviceToHost);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    if (e != hipSuccess) return HIPKKT_ERR_DEVICE;
    double me = 0;
    for (int i = 0; i < 256; i++) me = std::max(me, std::fabs(Dd[i] - Dh[i]));
    if (max_err) *max_err = me;
    return me < 1e-12 ? HIPKKT_OK : HIPKKT_NUMERICAL_FAILURE;
}

int32_t hipkkt_debug_is_testing_build(void) {
#ifdef HIPKKT_TESTING
    return 1;
#else
    return 0;
#endif
}

// the switches of struct DebugOpts (hipkkt_internal.h), by name; value NULL = default
int32_t hipkkt_debug_set(const char *key, const char *value) {
    DebugOpts &o = debug_opts();
    const DebugOpts def;
    if (!key) { o = def; return HIPKKT_OK; }
#ifndef HIPKKT_TESTING
    (void)value;
    g_create_error = "hipkkt_debug_set: this is the production library (no -DHIPKKT_TESTING): switches keep their defaults";
    return HIPKKT_ERR_ARGUMENT;
#else
    const std::string k(key);
    auto is0 = [&] { return value && value[0] == '0'; };
    auto is1 = [&] { return value && value[0] == '1'; };
    // a number, checked to its end (atof turned "0.0" and any non-numeric text into threshold 0)
    auto num = [&](double *out) {
        char *end = nullptr;
        const double v = strtod(value, &end);
        while (end && (*end == ' ' || *end == '\t')) end++;
        if (!end || end == value || *end != 0 || !(v == v)) return false;
        *out = v;
        return true;
    };
    double v = 0;
    if (k == "PLAN_CACHE") o.plan_cache = !is0();
    else if (k == "FB_EXTRA") o.fb_extra = !is0();
    else if (k == "FB_STREAM") o.fb_stream = !is0();
    else if (k == "FB_V2") o.fb_v2 = !is0();
    else if (k == "FORCE_TWIN") o.force_twin = is1();
    else if (k == "NO_GRAPH") o.no_graph = is1();
    else if (k == "NO_PERSIST") o.no_persist = is1();
    else if (k == "FULL_TILES") o.full_tiles = !is0();
    else if (k == "FRONT_BLOCK") o.front_block = !is0();
    else if (k == "SPLIT_K") o.split_k = !is0();
    else if (k == "GATHER_OVERLAP") o.gather_overlap = !is0();
    else if (k == "GATHER_SORT") o.gather_sort = !is0();
    else if (k == "GATHER_SIDE_BLOCKS") { if (!value) o.gather_side_blocks = def.gather_side_blocks; else if (num(&v) && v >= 0) o.gather_side_blocks = (int)v; else return HIPKKT_ERR_ARGUMENT; }
    else if (k == "DENSE_TRI") o.dense_tri = !is0();
    else if (k == "ORDERING") o.ordering_amd = value && value[0] == 'a';
    else if (k == "NO_FRONT") o.no_front = is1();
    else if (k == "HOST_ASSEMBLY") o.host_assembly = is1();
    else if (k == "FRONT_BLOCK_MIN_ROWS") { if (!value) o.front_block_min_rows = def.front_block_min_rows; else if (num(&v)) o.front_block_min_rows = (int)v; else return HIPKKT_ERR_ARGUMENT; }
    else if (k == "SUPERHOP") { if (!value) o.superhop = def.superhop; else if (num(&v)) o.superhop = (int)v; else return HIPKKT_ERR_ARGUMENT; }
    else if (k == "DEBUG_FLAGS") { if (!value) o.debug_flags = 0; else if (num(&v)) o.debug_flags = (int)v; else return HIPKKT_ERR_ARGUMENT; }
    else if (k == "SPIN_LIMIT") { if (!value) o.spin_limit = def.spin_limit; else if (num(&v) && v >= 0) o.spin_limit = (long long)v; else return HIPKKT_ERR_ARGUMENT; }
    else if (k == "PERSIST_RETRY") { if (!value) o.persist_retry = def.persist_retry; else if (num(&v) && v >= 0) o.persist_retry = (long long)v; else return HIPKKT_ERR_ARGUMENT; }
    else if (k == "ACCURATE") {
        // threshold on the largest |entry| of a wide block's explicit inverse: any value equal to 0 = never, negative = every block
        if (!value) o.accurate = def.accurate;
        else if (!num(&v)) return HIPKKT_ERR_ARGUMENT;
        else o.accurate = v == 0.0 ? 1e300 : v;
    } else if (k == "FB_EXTRA_PW") {
        o.fb_extra_pw = def.fb_extra_pw; o.fb_pen1 = def.fb_pen1; o.fb_pen2 = def.fb_pen2;
        if (value) {
            int m = o.fb_extra_pw; double a = o.fb_pen1, b = o.fb_pen2;
            if (sscanf(value, "%d,%lf,%lf", &m, &a, &b) < 1) return HIPKKT_ERR_ARGUMENT;
            o.fb_extra_pw = std::max(1, std::min(m, 2)); o.fb_pen1 = a; o.fb_pen2 = b;
        }
    } else {
        g_create_error = "hipkkt_debug_set: unknown switch " + k;
        return HIPKKT_ERR_ARGUMENT;
    }
    return HIPKKT_OK;
#endif
}

const char *hipkkt_last_error(hipkkt_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
