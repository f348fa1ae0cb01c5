// C ABI of libclarabel_hipkkt.so (see include/hipkkt.h for the contract and the reference
// interfaces each entry point replaces).  Host orchestration only: symbolic analysis, device
// residency, stream / hipGraph management and the control flow of iterative refinement.  All
// numeric work is in kernels.hip.
#include "../../include/hipkkt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "assemble.h"
#include "device_plan.h"
#include "kernels.h"
#include "runtime_pool.h"
#include "symbolic.h"

using namespace hipkkt;

namespace {

thread_local std::string g_create_error;

struct DeviceError {
    std::string msg;
};

#define HK_CHECK(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw DeviceError{std::string(#expr) + ": " + hipGetErrorString(e_)};                 \
    } while (0)

// Synchronous copy / fill on a handle's OWN stream.  The legacy (NULL) stream is never used: an operation on it from
// one host thread fails while another thread captures a hipGraph ("would make the legacy stream depend on a capturing
// stream"), and handles are meant to be driven concurrently from several threads.
inline void copy_sync(hipStream_t st, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (!bytes) return;
    HK_CHECK(hipMemcpyAsync(dst, src, bytes, kind, st));
    HK_CHECK(hipStreamSynchronize(st));
}
inline void fill_async(hipStream_t st, void *dst, int value, size_t bytes) {
    if (bytes) HK_CHECK(hipMemsetAsync(dst, value, bytes, st));
}

struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    bool valid = false;
    // parameters baked into the captured kernel arguments
    int static_enable = -1;
    double eps_const = 0, eps_prop = 0;
};

}  // namespace

// One solve context = everything a refined KKT solve mutates: its stream, work vectors, the hand-off / counter words
// of the persistent sweeps (a private copy of those DevPlan pointers), the device-side refinement state and its
// captured graphs.  Two contexts let two right-hand sides be solved CONCURRENTLY on one factorisation (SURVEY
// section 8(f) row N2: the constant-rhs solve and the affine solve of an IPM iteration): the sweeps are bound by
// dependency latency, not by throughput, so two of them overlap almost perfectly.
constexpr int kNumCtx = 2;
struct SolveCtx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevPlan dp{};                 // S->dp with this context's ubuf / pbuf / slots / sync words / scal / flags
    double *d_b = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_e = nullptr, *d_corr = nullptr;
    double *d_y = nullptr, *d_z = nullptr, *d_xp = nullptr;
    RefineState *d_rs = nullptr, *h_rs = nullptr;   // device state, pinned host copy
    int *h_flags = nullptr;                         // pinned copy of dp.flags
    GraphSlot g_ldl, g_first, g_step;               // plain LDL solve (d_b -> d_x0) / refined solve incl. its first step / one more step
    double g_reltol = -1, g_abstol = -1, g_stop = -1;   // parameters baked into g_first / g_step
    int64_t g_maxit = -1;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    double last_ms = 0;
    int64_t last_steps = 0;
    bool ir_used = false;
    const double *result() const { return (ir_used && h_rs->cur) ? d_x1 : d_x0; }
};

struct hipkkt_solver {
    int device = 0;
    SolveCtx ctx[kNumCtx];
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;          // far Schur updates run here, overlapped with the critical path
    std::vector<hipEvent_t> fork_events;
    bool use_side = true;
    int far_wgs = 256;   // grid bound of the look-ahead (far) update launches; 0 = one workgroup per 4 tiles
    hipkkt_opts opts{};
    bool l1 = false;
    KKTImage img;      // L1: assembled image; L0: colptr/rowval/nzval/dsigns copied in
    HostPlan plan;
    DevPlan dp{};
    std::vector<std::pair<void *, size_t>> allocs;   // device slabs (RuntimePool blocks: pointer, capacity)
    std::string err;

    int N = 0;
    int64_t nnzK = 0;
    // solve item lists (256-row blocks) per level
    std::vector<FacItem> slv_items, bwd_items;
    std::vector<int> slv_lvl_ptr, bwd_lvl_ptr;
    std::vector<int> reg_lvl_sn, reg_lvl_ptr;   // supernodes of every level that are NOT front panels
    std::vector<int> lvl_wnarrow, all_lvl_wnarrow;   // [nlevels] widest narrow supernode of the level
    std::vector<int> lvl_nnarrow;                // [nlevels] the first lvl_nnarrow[l] supernodes of a level's list are narrow (one thread each)
    // the same level lists over ALL supernodes (front panels included), appended to the same item arrays: the path without
    // any persistent kernel, taken after a sweep time-out
    std::vector<int> all_slv_lvl_ptr, all_bwd_lvl_ptr, all_reg_lvl_ptr;
    std::vector<int> all_lvl_nnarrow;
    // persistent sweeps over the regular supernodes: segments = level ranges between front kernels
    bool use_persist = true;
    double t_init_runtime = 0;           // seconds (HIPKKT_VERBOSE)
    std::chrono::steady_clock::time_point t_created{};   // when the assembly returned
    // front batches factored by one launch each (front_block.hip); a time-out inside one downgrades the handle to one launch per panel
    bool use_front_block = true;
    std::vector<FrontBatch> fbatches;
    std::vector<int> lvl_fb;             // [nlevels] index of the batch that starts at this level, -2 inside a batch, -1 otherwise
    std::vector<int> fb_last_level;      // per batch
    int *d_fb_sync = nullptr;
    double *d_fb_scratch = nullptr;
    long long *d_fb_trace = nullptr;     // HIPKKT_FB_TRACE=1: wall-clock stamps of the first 8 workgroups of every batch (debug_dump 9)
    bool persist_allowed = true;         // false: HIPKKT_NO_PERSIST (never tried)
    int64_t persist_retry_at = -1;       // after a sweep time-out: the LDL-solve count at which the persistent kernels are tried again
    int64_t persist_backoff = 0;         // doubles with every time-out (64, 128, ...); HIPKKT_PERSIST_RETRY=0 disables the retry
    int64_t n_sweep_timeouts = 0;
    int64_t n_twin_refactors = 0;        // factorisations repeated on the robust-order twin
    int nseg = 0;
    std::vector<int> seg_of_level;               // [nlevels]
    std::vector<int> fseg_ptr, bseg_ptr;         // [nseg+1] into slv_items / pbwd_items
    std::vector<int> seg_lo, seg_hi, seg_lstar;  // per segment: level range [lo, hi] and the first level handled by the
                                                 // persistent kernels (wide bottom levels keep one launch per level)
    std::vector<FacItem> pbwd_items;
    int wmax_all = 1;
    int inv_nsmall = 0, inv_wsmall = 1, inv_nwide = 0;   // split of the diagonal-block inversions (kernels.hip)
    std::vector<int64_t> p_off;
    std::vector<int64_t> gath_heavy_ptr;   // [nlevels+1] into the list of heavy gather entries (kernels.hip k_update_gather_heavy)

    // device index arrays for value updates
    int64_t *d_mapHs = nullptr, *d_mapP = nullptr, *d_mapA = nullptr, *d_diag_full = nullptr;
    // all sparse SOC cones concatenated
    int64_t soc_total = 0;
    int nsoc = 0;
    std::vector<int64_t> soc_off;  // per sparse map (SOC only) offset into the concatenated arrays
    std::vector<int> soc_of_sparse; // sparse-map index -> soc ordinal or -1
    int64_t *d_soc_uidx = nullptr, *d_soc_vidx = nullptr, *d_soc_didx = nullptr;
    int *d_soc_cone = nullptr;
    double *d_soc_u = nullptr, *d_soc_v = nullptr, *d_soc_eta2 = nullptr;
    // N1: update_scaling! / get_Hs! on the device (hipkkt_set_cone_types + hipkkt_update_scaling, scaling.hip)
    std::vector<int64_t> cone_numel;                       // as given to hipkkt_create_from_parts
    std::vector<int32_t> cone_hs_dense, cone_sparse_kind;
    bool sc_ready = false;
    int sc_nsoc = 0;                                       // ALL second-order cones (sparse and dense), cone order
    std::vector<int64_t> sc_psd_hs, sc_psd_n;              // per PSD cone: first Hs entry, matrix dimension
    int64_t sc_psd_total = 0;                              // sum of n * n
    signed char *d_sc_kind = nullptr;
    int64_t *d_sc_rowhs = nullptr, *d_sc_socdesc = nullptr;
    double *d_sc_sz = nullptr, *d_sc_wl = nullptr, *d_sc_eta = nullptr, *d_sc_R = nullptr, *d_sc_W = nullptr;
    int *d_sc_fail = nullptr;

    // vectors
    double *d_b = nullptr, *d_x = nullptr, *d_dx = nullptr, *d_e = nullptr;
    double *d_sin = nullptr, *d_sout = nullptr, *d_y = nullptr, *d_z = nullptr, *d_xp = nullptr;
    double *d_qb = nullptr, *d_res_in = nullptr, *d_res_out = nullptr, *d_res_part = nullptr;   // residuals_update! on the device (N4)
    double *d_stage = nullptr;   // staging for host-supplied values
    int64_t *d_stage_idx = nullptr;
    int64_t stage_cap = 0;
    double *h_scal = nullptr;    // pinned read-back area
    int *h_flags = nullptr;

    GraphSlot g_factor;
    bool use_graph = true;
    bool runtime_ready = false;   // init_runtime done (streams, events, pinned areas)
    bool poison = false;
    PlanOptions plan_opts;       // as used for the current plan
    // robust-order twin (minimum degree on K), created on the first factorisation that fails in the
    // "variables last" order; every later factorisation still tries the fast order first
    hipkkt_solver *fallback = nullptr;
    // the twin's symbolic analysis runs on a host thread from the moment the cheap order is chosen (finish_create)
    std::unique_ptr<hipkkt_solver> twin_pending;
    std::future<std::string> twin_future;
    std::shared_ptr<std::atomic<bool>> twin_cancel;   // set when the twin turns out not to be needed
    bool using_fallback = false;
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    double t_last_factor = 0, t_last_solve = 0, t_acc_factor = 0, t_acc_solve = 0, t_last_update = 0;
    int64_t n_factor = 0, n_solvecalls = 0, n_ldlsolves = 0, n_rhs_solved = 0;
    double last_eps = 0;
    double prof_fb_flops = 0;
    double prof_fb_ms = 0;               // last profiled refactorisation: the k_front_block launches
    int prof_fb_launches = 0, prof_fb_panels = 0;
    double prof_dense4_ms = 0, prof_dense4_flops = 0;   // last profiled refactorisation: k_update_dense<4,4> alone
    int prof_dense4_launches = 0;
    std::vector<double> prof_launch_ms, prof_launch_flops, prof_launch_tiles;   // per k_update_dense<4,4> launch of that refactorisation
    int64_t last_nreg = 0;

    // Device memory comes from a few slabs (bump allocation, 256-byte aligned) instead of one hipMalloc per array:
    // a handle owns ~80 arrays, and on the small problems of a batch the ~160 hipMalloc / hipFree calls were a
    // third of the set-up + tear-down time.
    char *slab_cur = nullptr;
    size_t slab_left = 0;
    template <class T>
    T *dalloc(size_t n) {
        if (n == 0) n = 1;
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (bytes > slab_left) {
            size_t slab = std::max<size_t>(bytes, (size_t)8 << 20);
            void *p = RuntimePool::get().dev_alloc(device, slab, &slab);
            if (!p) throw std::bad_alloc();
            allocs.push_back({p, slab});
            slab_cur = (char *)p;
            slab_left = slab;
        }
        void *p = slab_cur;
        slab_cur += bytes;
        slab_left -= bytes;
        if (poison) (void)hipMemsetAsync(p, 0xFF, n * sizeof(T), stream);   // debugging aid (HIPKKT_POISON=1): NaNs in every fresh buffer
        return (T *)p;
    }
    template <class T>
    T *upload(const std::vector<T> &v) {
        T *p = dalloc<T>(v.size());
        if (!v.empty()) copy_sync(stream, p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return p;
    }
    void ensure_stage(int64_t n) {
        if (n <= stage_cap) return;
        int64_t cap = std::max<int64_t>(n, 2 * stage_cap);
        d_stage = dalloc<double>(cap);
        d_stage_idx = dalloc<int64_t>(cap);
        stage_cap = cap;
    }
    ~hipkkt_solver() {
        if (twin_cancel) twin_cancel->store(true);
        if (twin_future.valid()) twin_future.wait();     // the thread reads twin_pending's image (a cancelled one ends at its next phase)
        twin_pending.reset();
        delete fallback;
        (void)hipSetDevice(device);
        // everything below goes back to the process-wide cache (runtime_pool.h): the streams must be idle first
        RuntimePool &rp = RuntimePool::get();
        if (stream) (void)hipStreamSynchronize(stream);
        if (side) (void)hipStreamSynchronize(side);
        for (SolveCtx &C : ctx)
            if (C.own_stream && C.stream) (void)hipStreamSynchronize(C.stream);
        if (g_factor.exec) (void)hipGraphExecDestroy(g_factor.exec);
        for (SolveCtx &C : ctx) {
            for (GraphSlot *g : {&C.g_ldl, &C.g_first, &C.g_step})
                if (g->exec) (void)hipGraphExecDestroy(g->exec);
            rp.pinned_free(device, C.h_rs);
            rp.pinned_free(device, C.h_flags);
            for (hipEvent_t e : {C.ev_a, C.ev_b}) rp.event_put(device, e);
            if (C.own_stream) rp.stream_put(device, 2, C.stream);
        }
        for (auto &a : allocs) rp.dev_free(device, a.first, a.second);
        rp.pinned_free(device, h_scal);
        rp.pinned_free(device, h_flags);
        for (hipEvent_t e : {ev0, ev1, ev2, ev3}) rp.event_put(device, e);
        for (hipEvent_t e : fork_events) (void)hipEventDestroy(e);
        rp.stream_put(device, 1, side);
        rp.stream_put(device, 0, stream);
    }
};

namespace {

// seg_sync = [forward tickets | backward tickets] padded to whole 128-byte lines, then fdone / bdone / pdone [nsuper each],
// then the error word (kernels.hip seg_sync())
static size_t seg_sync_ints(int nseg, int nsuper) { return (size_t)((2 * nseg + 31) & ~31) + 3 * (size_t)nsuper + 16; }

void init_runtime(hipkkt_solver *S) {
    { const char *pz = getenv("HIPKKT_POISON"); S->poison = pz && pz[0] == '1'; }
    HK_CHECK(hipSetDevice(S->device));
    RuntimePool &rp = RuntimePool::get();
    auto need = [&](void *p) { if (!p) throw DeviceError{"creating a stream / event / pinned buffer failed"}; return p; };
    {
        // the critical path (panel factorisations) gets the higher priority
        S->stream = (hipStream_t)need(rp.stream_get(S->device, 0));
        S->side = (hipStream_t)need(rp.stream_get(S->device, 1));
        // measured on MI355X (cfg 2a): forking the far updates gives no net gain inside a hipGraph -- the far
        // kernel fills every CU and the panel kernels on the critical path slow down by what the overlap
        // saves -- so the fork is opt-in (HIPKKT_SIDE_STREAM=1)
        const char *ns = getenv("HIPKKT_SIDE_STREAM");
        S->use_side = ns && ns[0] == '1';
        const char *fw = getenv("HIPKKT_FAR_WGS");
        if (fw) S->far_wgs = atoi(fw);
    }
    static_assert(SC_COUNT * sizeof(double) <= RuntimePool::kPinned && sizeof(RefineState) <= RuntimePool::kPinned, "pinned chunk too small");
    for (hipEvent_t *e : {&S->ev0, &S->ev1, &S->ev2, &S->ev3}) *e = (hipEvent_t)need(rp.event_get(S->device));
    S->h_scal = (double *)need(rp.pinned_alloc(S->device));
    S->h_flags = (int *)need(rp.pinned_alloc(S->device));
    memset(S->h_scal, 0, SC_COUNT * sizeof(double));
    memset(S->h_flags, 0, FL_COUNT * sizeof(int));
    const char *ng = getenv("HIPKKT_NO_GRAPH");
    if (ng && ng[0] == '1') S->use_graph = false;
    for (int c = 0; c < kNumCtx; c++) {
        SolveCtx &C = S->ctx[c];
        if (c == 0) C.stream = S->stream;
        else { C.stream = (hipStream_t)need(rp.stream_get(S->device, 2)); C.own_stream = true; }
        C.ev_a = (hipEvent_t)need(rp.event_get(S->device));
        C.ev_b = (hipEvent_t)need(rp.event_get(S->device));
        C.h_rs = (RefineState *)need(rp.pinned_alloc(S->device));
        C.h_flags = (int *)need(rp.pinned_alloc(S->device));
        memset(C.h_rs, 0, sizeof(RefineState));
        memset(C.h_flags, 0, FL_COUNT * sizeof(int));
    }
}

// (re)builds every device-resident structure from S->plan and S->img (values included)
static void build_front_batches(hipkkt_solver *S);
void setup_device(hipkkt_solver *S) {
    HK_CHECK(hipSetDevice(S->device));
    for (GraphSlot *g : {&S->g_factor, &S->ctx[0].g_ldl, &S->ctx[0].g_first, &S->ctx[0].g_step, &S->ctx[1].g_ldl, &S->ctx[1].g_first,
                         &S->ctx[1].g_step}) {
        if (g->exec) (void)hipGraphExecDestroy(g->exec);
        *g = GraphSlot();
    }
    if (!S->allocs.empty() && S->stream) (void)hipStreamSynchronize(S->stream);   // hipFree used to wait implicitly
    for (auto &a : S->allocs) RuntimePool::get().dev_free(S->device, a.first, a.second);
    S->allocs.clear();
    S->slab_cur = nullptr;
    S->slab_left = 0;
    S->slv_items.clear(); S->bwd_items.clear(); S->reg_lvl_sn.clear(); S->pbwd_items.clear();
    {
        const char *np_ = getenv("HIPKKT_NO_PERSIST");
        S->use_persist = !(np_ && np_[0] == '1');
        S->persist_allowed = S->use_persist;
        S->persist_retry_at = -1;
    }
    S->soc_off.clear(); S->soc_of_sparse.clear();
    S->nsoc = 0; S->soc_total = 0; S->wmax_all = 1;
    S->stage_cap = 0; S->d_stage = nullptr; S->d_stage_idx = nullptr;
    S->d_qb = S->d_res_in = S->d_res_out = S->d_res_part = nullptr;

    HostPlan &P = S->plan;
    const int N = P.N;
    S->N = N;
    S->nnzK = P.nnzK;
    // solve items: 64-row blocks (kSlvRows in kernels.hip)
    S->slv_lvl_ptr.assign(P.nlevels + 1, 0);
    S->bwd_lvl_ptr.assign(P.nlevels + 1, 0);
    S->p_off.assign(P.nsuper + 1, 0);
    for (int s = 0; s < P.nsuper; s++) {
        int w = P.sn_first[s + 1] - P.sn_first[s];
        int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
        int64_t nb = std::max<int64_t>(1, (r - w + 63) / 64);
        S->p_off[s + 1] = S->p_off[s] + nb * w;
    }
    S->reg_lvl_ptr.assign(P.nlevels + 1, 0);
    const char *nn = getenv("HIPKKT_NO_NARROW");
    const bool allow_narrow = !(nn && nn[0] == '1');
    // ---- segments of the persistent sweeps = level ranges between two front kernels; inside a segment the wide bottom
    //      levels (thousands of leaf supernodes) are cheaper as one launch per level, the persistent kernels take over
    //      from the first level with fewer than kPersistMaxItems items (seg_lstar)
    std::vector<int> dep_ptr(P.nsuper + 1, 0), dep_idx, sn_nitems(P.nsuper, 0), sn_bparent(P.nsuper, -1);
    std::vector<int> rows_seg;
    {
        S->seg_of_level.assign(P.nlevels, 0);
        std::vector<char> boundary(P.nlevels + 1, 0);
        for (const FrontDesc &F : P.fronts) boundary[F.level_last] = 1;   // the front runs after regular level level_last
        int sg = 0;
        for (int l = 0; l < P.nlevels; l++) { S->seg_of_level[l] = sg; if (boundary[l]) sg++; }
        S->nseg = sg + 1;
        std::vector<int> lvl_items(P.nlevels, 0);      // forward items of the regular (non-front) supernodes of a level
        for (int s = 0; s < P.nsuper; s++) {
            if (P.sn_front[s] >= 0) continue;
            const int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
            const int w = P.sn_first[s + 1] - P.sn_first[s];
            lvl_items[P.sn_level[s]] += (int)std::max<int64_t>(1, (r - w + 63) / 64);
        }
        const int kPersistMaxItems = 1024;
        S->seg_lo.assign(S->nseg, P.nlevels); S->seg_hi.assign(S->nseg, -1); S->seg_lstar.assign(S->nseg, 0);
        for (int l = 0; l < P.nlevels; l++) {
            const int g = S->seg_of_level[l];
            S->seg_lo[g] = std::min(S->seg_lo[g], l);
            S->seg_hi[g] = std::max(S->seg_hi[g], l);
        }
        for (int g = 0; g < S->nseg; g++) {
            int ls = S->seg_lo[g];
            while (ls <= S->seg_hi[g] && lvl_items[ls] >= kPersistMaxItems) ls++;
            S->seg_lstar[g] = ls;
        }
    }
    // Level lists of the per-level solve kernels, appended to slv_items / bwd_items / reg_lvl_sn.  Variant 0 leaves out the
    // panels of the fronts (the persistent front kernels solve those); variant 1 holds EVERY supernode and is what a
    // handle falls back to after a persistent sweep timed out (front kernels included: no persistent kernel at all).
    // On a level that gets its own launches the NARROW supernodes (<= kNarrowW columns, <= kNarrowR rows below the
    // block: the leaves) are listed first and solved one THREAD each (k_fwd_narrow / k_bwd_narrow); they have no items.
    std::vector<char> has_child(P.nsuper, 0);
    for (int c = 0; c < P.nsuper; c++)
        if (P.sn_parent[c] >= 0) has_child[P.sn_parent[c]] = 1;
    auto build_level_lists = [&](bool with_fronts, std::vector<int> &slv_ptr, std::vector<int> &bwd_ptr, std::vector<int> &reg_ptr,
                                 std::vector<int> &nnarrow, std::vector<int> &wnarrow) {
        slv_ptr.assign(P.nlevels + 1, (int)S->slv_items.size());
        bwd_ptr.assign(P.nlevels + 1, (int)S->bwd_items.size());
        reg_ptr.assign(P.nlevels + 1, (int)S->reg_lvl_sn.size());
        nnarrow.assign(P.nlevels, 0);
        wnarrow.assign(P.nlevels, 1);
        std::vector<int> nar, reg;
        for (int l = 0; l < P.nlevels; l++) {
            const bool own_launches = with_fronts || l < S->seg_lstar[S->seg_of_level[l]];
            nar.clear(); reg.clear();
            bool all_tiny = true;    // every supernode of the level has <= 4 columns and <= 16 rows below the block
            for (int q = P.lvl_ptr[l]; q < P.lvl_ptr[l + 1]; q++) {
                int s = P.lvl_sn[q];
                int w = P.sn_first[s + 1] - P.sn_first[s];
                S->wmax_all = std::max(S->wmax_all, w);
                if (!with_fronts && P.sn_front[s] >= 0) continue;      // solved by the persistent front kernels
                int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
                all_tiny = all_tiny && w <= 4 && r - w <= 16;
            }
            for (int q = P.lvl_ptr[l]; q < P.lvl_ptr[l + 1]; q++) {
                int s = P.lvl_sn[q];
                int w = P.sn_first[s + 1] - P.sn_first[s];
                if (!with_fronts && P.sn_front[s] >= 0) continue;
                int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
                // one thread per supernode pays for every gather of a child's update vector with a serial memory round trip:
                // a level goes to the thread kernels as a whole when all of it is tiny; otherwise only its childless
                // supernodes (leaves: nothing to gather) do, up to kNarrowW x kNarrowR
                // (a thread walks its w x (r - w) panel entries one strided load after the other: bounded work per thread, and
                // only worth it where the workgroup-per-item kernel would need several rounds of the chip: >= 2048 leaves)
                const bool thr = all_tiny || (!has_child[s] && w <= kNarrowW && r - w <= kNarrowR && (int64_t)w * (r - w) <= 128);
                (allow_narrow && own_launches && thr ? nar : reg).push_back(s);
            }
            if (nar.size() < (all_tiny ? 256u : 2048u)) { reg.insert(reg.end(), nar.begin(), nar.end()); std::sort(reg.begin(), reg.end()); nar.clear(); }
            nnarrow[l] = (int)nar.size();
            for (int s : nar) wnarrow[l] = std::max(wnarrow[l], P.sn_first[s + 1] - P.sn_first[s]);
            S->reg_lvl_sn.insert(S->reg_lvl_sn.end(), nar.begin(), nar.end());
            for (int s : reg) {
                int w = P.sn_first[s + 1] - P.sn_first[s];
                int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
                int nb = (int)std::max<int64_t>(1, (r - w + 63) / 64);
                for (int b = 0; b < nb; b++) S->slv_items.push_back({s, b});
                if (nb > 1)
                    for (int b = 0; b < nb; b++) S->bwd_items.push_back({s, b});
                S->reg_lvl_sn.push_back(s);
            }
            slv_ptr[l + 1] = (int)S->slv_items.size();
            bwd_ptr[l + 1] = (int)S->bwd_items.size();
            reg_ptr[l + 1] = (int)S->reg_lvl_sn.size();
        }
    };
    build_level_lists(false, S->slv_lvl_ptr, S->bwd_lvl_ptr, S->reg_lvl_ptr, S->lvl_nnarrow, S->lvl_wnarrow);
    if (!P.fronts.empty() || S->nseg > 0) build_level_lists(true, S->all_slv_lvl_ptr, S->all_bwd_lvl_ptr, S->all_reg_lvl_ptr, S->all_lvl_nnarrow, S->all_lvl_wnarrow);
    // ---- persistent sweeps: dependency lists, backward item order
    {
        auto seg_of = [&](int s) { return S->seg_of_level[P.sn_level[s]]; };
        auto persistent = [&](int s) { return P.sn_level[s] >= S->seg_lstar[seg_of(s)]; };
        std::vector<std::vector<int>> kids(P.nsuper);
        for (int c = 0; c < P.nsuper; c++) {
            const int p = P.sn_parent[c];
            if (P.sn_front[c] >= 0) continue;
            const int64_t r = P.sn_rowptr[c + 1] - P.sn_rowptr[c];
            const int w = P.sn_first[c + 1] - P.sn_first[c];
            sn_nitems[c] = (int)std::max<int64_t>(1, (r - w + 63) / 64);
            if (p >= 0 && P.sn_front[p] < 0 && seg_of(p) == seg_of(c) && persistent(c) && persistent(p)) {
                kids[p].push_back(c);
                sn_bparent[c] = p;
            }
        }
        for (int s = 0; s < P.nsuper; s++) {
            dep_ptr[s + 1] = dep_ptr[s] + (int)kids[s].size();
            dep_idx.insert(dep_idx.end(), kids[s].begin(), kids[s].end());
        }
        // tagged hand-off of the persistent backward sweep: mark the rows whose x is produced inside the same launch
        {
            auto in_seg_kernel = [&](int s) { return P.sn_front[s] < 0 && persistent(s); };
            std::vector<int> col_sn(N, -1);
            for (int c = 0; c < P.nsuper; c++)
                for (int k = P.sn_first[c]; k < P.sn_first[c + 1]; k++) col_sn[k] = c;
            rows_seg.assign(P.sn_rows.begin(), P.sn_rows.end());
            for (int s = 0; s < P.nsuper; s++) {
                if (!in_seg_kernel(s)) continue;
                for (int64_t slot = P.sn_rowptr[s]; slot < P.sn_rowptr[s + 1]; slot++) {
                    const int a = col_sn[P.sn_rows[slot]];
                    if (a != s && a >= 0 && in_seg_kernel(a) && seg_of(a) == seg_of(s)) rows_seg[(size_t)slot] = P.sn_rows[slot] | 0x40000000;
                }
            }
        }
        // forward segments: slv_items is level-ordered, a segment is a level range
        S->fseg_ptr.assign(2 * S->nseg, 0);   // [2g] first persistent item, [2g+1] end, of segment g
        for (int g = 0; g < S->nseg; g++) {
            const int ls = std::min(S->seg_lstar[g], P.nlevels);
            S->fseg_ptr[2 * g] = S->seg_hi[g] >= 0 ? S->slv_lvl_ptr[std::min(ls, S->seg_hi[g] + 1)] : 0;
            S->fseg_ptr[2 * g + 1] = S->seg_hi[g] >= 0 ? S->slv_lvl_ptr[S->seg_hi[g] + 1] : 0;
        }
        // backward items: segments in DESCENDING order of level; inside a segment levels descending, partial
        // blocks of a level before its finalisers
        S->bseg_ptr.assign(S->nseg + 1, 0);
        for (int g = S->nseg - 1; g >= 0; g--) {
            for (int l = P.nlevels - 1; l >= 0; l--) {
                if (S->seg_of_level[l] != g || l < S->seg_lstar[g]) continue;
                for (int q = S->reg_lvl_ptr[l]; q < S->reg_lvl_ptr[l + 1]; q++) {
                    const int s = S->reg_lvl_sn[q];
                    if (sn_nitems[s] > 1)
                        for (int b = 0; b < sn_nitems[s]; b++) S->pbwd_items.push_back({s, b});
                }
                for (int q = S->reg_lvl_ptr[l]; q < S->reg_lvl_ptr[l + 1]; q++) S->pbwd_items.push_back({S->reg_lvl_sn[q], -1});
            }
            S->bseg_ptr[S->nseg - g] = (int)S->pbwd_items.size();   // bseg_ptr is indexed by launch order
        }
    }
    std::vector<signed char> sgn_perm(N), kdiag(S->nnzK, 0);
    for (int k = 0; k < N; k++) sgn_perm[k] = (signed char)(S->img.dsigns[P.perm[k]] >= 0 ? 1 : -1);
    for (int j = 0; j < N; j++)
        for (int64_t q = S->img.colptr[j]; q < S->img.colptr[j + 1]; q++)
            if (S->img.rowval[q] == j) kdiag[q] = (signed char)(S->img.dsigns[j] >= 0 ? 1 : -1);

    DevPlan &D = S->dp;
    D.sn_first = S->upload(P.sn_first);
    D.sn_rowptr = S->upload(P.sn_rowptr);
    D.sn_rows = S->upload(P.sn_rows);
    D.sn_panel = S->upload(P.sn_panel);
    D.sn_diag = S->upload(P.sn_diag);
    D.u_off = S->upload(P.u_off);
    D.p_off = S->upload(S->p_off);
    D.lt_off = S->upload(P.lt_off);
    D.bwd_items = S->upload(S->bwd_items);
    D.lvl_sn = S->upload(S->reg_lvl_sn);
    D.perm = S->upload(P.perm);
    D.sgn_perm = S->upload(sgn_perm);
    {
        // diagonal blocks wider than 16 columns (and at most 64, the blocked kernel's size) are inverted by
        // k_invert_diag_wide, the rest by the one-wave kernel with LDS sized for the widest of them
        std::vector<int> small, wide;
        S->inv_wsmall = 1;
        for (int s = 0; s < P.nsuper; s++) {
            const int w = P.sn_first[s + 1] - P.sn_first[s];
            if (w > 16 && w <= 64) wide.push_back(s);
            else { small.push_back(s); S->inv_wsmall = std::max(S->inv_wsmall, w); }
        }
        S->inv_nsmall = (int)small.size();
        S->inv_nwide = (int)wide.size();
        small.insert(small.end(), wide.begin(), wide.end());
        D.inv_list = S->upload(small);
    }
    D.fac_items = S->upload(P.fac_items);
    {
        std::vector<FacRec> recs(P.fac_items.size());
        for (size_t q = 0; q < recs.size(); q++) {
            const int s = P.fac_items[q].sn;
            recs[q] = {P.sn_panel[s], P.sn_diag[s], P.lt_off[s], P.sn_first[s], P.sn_first[s + 1] - P.sn_first[s],
                       (int32_t)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]), P.fac_items[q].blk};
        }
        D.fac_recs = S->upload(recs);
    }
    D.fac_jit = S->upload(P.fac_jit);
    D.slv_items = S->upload(S->slv_items);
    D.rel = S->upload(P.rel);
    D.upd_tasks = S->upload(P.upd_tasks);
    D.upd_groups = S->upload(P.upd_groups);
    {
        std::vector<DenseGroup> dg(P.upd_groups.size());
        for (size_t q = 0; q < dg.size(); q++) {
            const UpdGroup &G = P.upd_groups[q];
            const int t = G.tgt;
            const int rt = (int)(P.sn_rowptr[t + 1] - P.sn_rowptr[t]);
            dg[q] = {P.sn_panel[t] + G.row_base, rt, std::min(kUpdRows, rt - G.row_base), P.sn_first[t + 1] - P.sn_first[t],
                     G.task_begin, G.task_end, 0};
        }
        D.dgroups = S->upload(dg);
    }
    D.upd_tmap = S->upload(P.upd_tmap);
    {
        std::vector<DenseTask> dt(P.upd_tasks.size());
        for (size_t q = 0; q < dt.size(); q++) {
            const UpdTask &T = P.upd_tasks[q];
            const int s = T.src;
            dt[q] = {P.sn_panel[s], (int32_t)((P.sn_rowptr[s + 1] - P.sn_rowptr[s]) * 8), P.sn_first[s + 1] - P.sn_first[s],
                     P.sn_first[s], T.row_lo, T.nrows, T.col_lo, T.ncols, T.geom, T.vt_begin, 0};
        }
        D.dtasks = S->upload(dt);
    }
    D.gath_tgt = S->upload(P.gath_tgt);
    D.gath_pptr = S->upload(P.gath_pptr);
    {
        std::vector<GathPair> gp(P.gath_src.size());
        for (size_t q = 0; q < gp.size(); q++) {
            const int s = P.gath_sn[q];
            gp[q] = {P.gath_src[q], P.gath_dj[q], (int32_t)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]), P.sn_first[s + 1] - P.sn_first[s], P.sn_first[s]};
        }
        D.gath_pairs = S->upload(gp);
        std::vector<int64_t> heavy;
        S->gath_heavy_ptr.assign(P.nlevels + 1, 0);
        for (int l = 0; l < P.nlevels; l++) {
            for (int64_t e = P.gath_stage_ptr[l]; e < P.gath_stage_ptr[l + 1]; e++)
                if (P.gath_pptr[e + 1] - P.gath_pptr[e] > kGathHeavy) heavy.push_back(e);
            S->gath_heavy_ptr[l + 1] = (int64_t)heavy.size();
        }
        D.gath_heavy = S->upload(heavy);
    }
    D.g_ptr = S->upload(P.g_ptr);
    D.g_idx = S->upload(P.g_idx);
    D.kmap = S->upload(P.kmap);
    D.kdiag_sign = S->upload(kdiag);
    D.sym_rowptr = S->upload(P.sym_rowptr);
    D.sym_col = S->upload(P.sym_col);
    D.sym_q = S->upload(P.sym_q);
    {
        std::vector<int> lr;
        const int thr = long_row_threshold();
        for (int i = 0; i < N; i++)
            if (P.sym_rowptr[i + 1] - P.sym_rowptr[i] > thr) lr.push_back(i);
        D.long_rows = S->upload(lr);
        D.n_long_rows = (int)lr.size();
    }
    D.front_panels = S->upload(P.front_panels);
    D.front_gptr = S->upload(P.front_gptr);
    D.front_gidx = S->upload(P.front_gidx);
    D.pbwd_items = S->upload(S->pbwd_items);
    {
        std::vector<int> dep_total(P.nsuper, 0);
        for (int s = 0; s < P.nsuper; s++)
            for (int q = dep_ptr[s]; q < dep_ptr[s + 1]; q++) dep_total[s] += sn_nitems[dep_idx[q]];
        D.dep_total = S->upload(dep_total);
    }
    D.sn_nitems = S->upload(sn_nitems);
    D.sn_bparent = S->upload(sn_bparent);
    D.nseg = S->nseg;
    {
        const char *tk = getenv("HIPKKT_SEG_TICKET");      // 0: item = blockIdx (A/B timing of the ticket's cost)
        D.seg_ticket = tk ? atoi(tk) : 3;                  // bit 0: forward sweep, bit 1: backward sweep
        const char *sl = getenv("HIPKKT_SPIN_LIMIT");      // tests force a sweep time-out with a tiny bound
        D.spin_limit = sl ? (unsigned)strtoul(sl, nullptr, 10) : (1u << 20);
    }
    {
        const size_t nsync = seg_sync_ints(S->nseg, P.nsuper);
        D.seg_sync = S->dalloc<int>(nsync);
        fill_async(S->stream, D.seg_sync, 0, nsync * sizeof(int));
    }
    D.front_sync = S->dalloc<int>(std::max(P.front_sync_ints, 16));
    fill_async(S->stream, D.front_sync, 0, (size_t)std::max(P.front_sync_ints, 16) * sizeof(int));
    build_front_batches(S);
    D.kval = S->upload(S->img.nzval);
    D.Lx = S->dalloc<double>(P.panel_doubles);
    D.Ldiag = S->dalloc<double>(P.diag_doubles);
    D.Linv = S->dalloc<double>(P.diag_doubles);
    D.LinvT = S->dalloc<double>(P.diag_doubles);
    D.LT = S->dalloc<double>(P.lt_off[P.nsuper]);
    D.D = S->dalloc<double>(N);
    D.Dinv = S->dalloc<double>(N);
    D.ubuf = S->dalloc<double>(P.ubuf_len);
    D.pbuf = S->dalloc<double>(S->p_off[P.nsuper]);
    {
        // slots start out all-zero = invalid in every epoch
        const size_t nx = (size_t)std::max(N, 1), np_ = (size_t)std::max<int64_t>(S->p_off[P.nsuper], 1);
        D.xseg = (FrontSlot *)S->dalloc<double>(2 * nx);
        D.pseg = (FrontSlot *)S->dalloc<double>(2 * np_);
        fill_async(S->stream, D.xseg, 0, 16 * nx);
        fill_async(S->stream, D.pseg, 0, 16 * np_);
        D.seg_epoch = S->dalloc<int>(4);
        fill_async(S->stream, D.seg_epoch, 0, 4 * sizeof(int));
        D.rows_seg = S->upload(rows_seg);
    }
    D.scal = S->dalloc<double>(SC_COUNT);
    D.flags = S->dalloc<int>(FL_COUNT);
    fill_async(S->stream, D.scal, 0, SC_COUNT * sizeof(double));
    fill_async(S->stream, D.flags, 0, FL_COUNT * sizeof(int));
    fill_async(S->stream, D.Dinv, 0, (size_t)std::max(N, 1) * sizeof(double));
    fill_async(S->stream, D.Ldiag, 0, (size_t)std::max<int64_t>(P.diag_doubles, 1) * sizeof(double));

    S->d_diag_full = S->upload(S->img.diag_full);
    if (S->l1) {
        S->d_mapHs = S->upload(S->img.mapHs);
        S->d_mapP = S->upload(S->img.mapP);
        S->d_mapA = S->upload(S->img.mapA);
        // concatenated SOC expansion maps
        std::vector<int64_t> uidx, vidx, didx;
        std::vector<int> coneof;
        S->soc_of_sparse.assign(S->img.smaps.size(), -1);
        for (size_t i = 0; i < S->img.smaps.size(); i++) {
            const SparseMap &sm = S->img.smaps[i];
            if (sm.kind != 1) continue;
            S->soc_of_sparse[i] = S->nsoc;
            S->soc_off.push_back((int64_t)uidx.size());
            for (size_t q = 0; q < sm.vec[0].size(); q++) {
                uidx.push_back(sm.vec[0][q]);
                vidx.push_back(sm.vec[1][q]);
                coneof.push_back(S->nsoc);
            }
            didx.push_back(sm.D[0]);
            didx.push_back(sm.D[1]);
            S->nsoc++;
        }
        S->soc_off.push_back((int64_t)uidx.size());
        S->soc_total = (int64_t)uidx.size();
        S->d_soc_uidx = S->upload(uidx);
        S->d_soc_vidx = S->upload(vidx);
        S->d_soc_didx = S->upload(didx);
        S->d_soc_cone = S->upload(coneof);
        S->d_soc_u = S->dalloc<double>(S->soc_total);
        S->d_soc_v = S->dalloc<double>(S->soc_total);
        S->d_soc_eta2 = S->dalloc<double>(S->nsoc);
    }
    for (int c = 0; c < kNumCtx; c++) {
        SolveCtx &C = S->ctx[c];
        for (double **v : {&C.d_b, &C.d_x0, &C.d_x1, &C.d_e, &C.d_corr, &C.d_y, &C.d_z, &C.d_xp}) {
            *v = S->dalloc<double>(N);
            fill_async(S->stream, *v, 0, (size_t)std::max(N, 1) * sizeof(double));
        }
        C.d_rs = (RefineState *)S->dalloc<double>(sizeof(RefineState) / sizeof(double) + 1);
        fill_async(S->stream, C.d_rs, 0, sizeof(RefineState));
        C.dp = D;
        if (c > 0) {   // private copies of everything a solve writes besides its vectors
            const size_t nx = (size_t)std::max(N, 1), np_ = (size_t)std::max<int64_t>(S->p_off[P.nsuper], 1);
            const size_t nsync = seg_sync_ints(S->nseg, P.nsuper), nfs = (size_t)std::max(P.front_sync_ints, 16);
            C.dp.ubuf = S->dalloc<double>(P.ubuf_len);
            C.dp.pbuf = S->dalloc<double>(S->p_off[P.nsuper]);
            C.dp.xseg = (FrontSlot *)S->dalloc<double>(2 * nx);
            C.dp.pseg = (FrontSlot *)S->dalloc<double>(2 * np_);
            fill_async(S->stream, C.dp.xseg, 0, 16 * nx);
            fill_async(S->stream, C.dp.pseg, 0, 16 * np_);
            C.dp.seg_epoch = S->dalloc<int>(4);
            fill_async(S->stream, C.dp.seg_epoch, 0, 4 * sizeof(int));
            C.dp.seg_sync = S->dalloc<int>(nsync);
            fill_async(S->stream, C.dp.seg_sync, 0, nsync * sizeof(int));
            C.dp.front_sync = S->dalloc<int>(nfs);
            fill_async(S->stream, C.dp.front_sync, 0, nfs * sizeof(int));
            C.dp.scal = S->dalloc<double>(SC_COUNT);
            C.dp.flags = S->dalloc<int>(FL_COUNT);
            fill_async(S->stream, C.dp.scal, 0, SC_COUNT * sizeof(double));
            fill_async(S->stream, C.dp.flags, 0, FL_COUNT * sizeof(int));
        }
        C.ir_used = false;
        C.h_rs->cur = 0;
    }
    // legacy names = context 0
    S->d_b = S->ctx[0].d_b; S->d_x = S->ctx[0].d_x0; S->d_dx = S->ctx[0].d_x1; S->d_e = S->ctx[0].d_e;
    S->d_sin = S->ctx[0].d_b; S->d_sout = S->ctx[0].d_corr; S->d_y = S->ctx[0].d_y; S->d_z = S->ctx[0].d_z; S->d_xp = S->ctx[0].d_xp;
    S->ensure_stage(std::max<int64_t>(1024, std::max<int64_t>(S->img.nHs, N)));
    HK_CHECK(hipStreamSynchronize(S->stream));
}

// One launch of k_front_block per qualifying update batch of a front (symbolic.cpp front_batches).  HIPKKT_FRONT_BLOCK=0: never.
static void build_front_batches(hipkkt_solver *S) {
    const HostPlan &P = S->plan;
    S->fbatches.clear(); S->fb_last_level.clear();
    S->lvl_fb.assign(std::max(P.nlevels, 1), -1);
    const char *e = getenv("HIPKKT_FRONT_BLOCK");
    if (e && e[0] == '0') S->use_front_block = false;
    if (S->use_front_block)
        for (const FrontBatchHost &H : front_batches(P, S->plan_opts.update_policy, kFbMax)) {
            const FrontDesc &F = P.fronts[H.front];
            FrontBatch B;
            B.fp_off = F.fp_off + H.p0;
            B.nb = H.nb;
            B.r0 = P.front_panels[F.fp_off + H.p0].r;
            B.nblk = (B.r0 + 63) / 64;
            B.sync_off = 128 * (int)S->fbatches.size();
            B.scratch_off = kFbScratch * (int64_t)S->fbatches.size();
            S->lvl_fb[H.level_first] = (int)S->fbatches.size();
            for (int l = H.level_first + 1; l <= H.level_last; l++) S->lvl_fb[l] = -2;
            S->fbatches.push_back(B);
            S->fb_last_level.push_back(H.level_last);
        }
    if (getenv("HIPKKT_VERBOSE")) fprintf(stderr, "hipkkt: %zu front batch(es) factored by one launch each (fronts %zu, update batch %d)\n", S->fbatches.size(), P.fronts.size(), P.update_batch_used);
    const size_t nb_ = std::max<size_t>(S->fbatches.size(), 1);
    S->d_fb_sync = S->dalloc<int>(128 * nb_);
    S->d_fb_scratch = S->dalloc<double>((size_t)kFbScratch * nb_);
    fill_async(S->stream, S->d_fb_sync, 0, 128 * nb_ * sizeof(int));
    if (getenv("HIPKKT_FB_TRACE")) {
        S->d_fb_trace = (long long *)S->dalloc<double>(nb_ * 128);
        fill_async(S->stream, S->d_fb_trace, 0, nb_ * 128 * sizeof(double));
    }
}

// ---- enqueue helpers (no synchronisation inside; capturable) ---------------------------------

// narrow levels (w <= 8): LDS-resident kernel; wide panels: the register-resident 8-wave kernel
void enqueue_factor_level(hipkkt_solver *S, int l) {
    const HostPlan &P = S->plan;
    const int n = P.fac_lvl_ptr[l + 1] - P.fac_lvl_ptr[l];
    if (P.fac_lvl_maxw[l] <= 8)
        launch_factor_level(S->stream, S->dp, P.fac_lvl_ptr[l], n, P.fac_lvl_maxw[l], S->opts.dynamic_reg_eps,
                            S->opts.dynamic_reg_delta);
    else
        launch_factor_panel(S->stream, S->dp, P.fac_lvl_ptr[l], n, S->opts.dynamic_reg_eps, S->opts.dynamic_reg_delta,
                            P.lvl_fused[l] != 0);
}

// Schur-complement updates applied after level l is factored: dense register tiles (matrix cores),
// per-entry gather lists (tiny scattered contributions), relative-index scatter (whatever is left).
// The three kinds own disjoint target tiles, so their order inside a stage is immaterial.
void enqueue_updates(hipkkt_solver *S, int l, bool split_far = false) {
    const HostPlan &P = S->plan;
    hipStream_t st = S->stream;
    if (l + 1 < P.nlevels && P.lvl_fused[l + 1]) return;   // applied inside the next level's panel kernel
    const int g0 = P.upd_stage_ptr[l], nd = P.upd_stage_ndense[l], ng = P.upd_stage_ngather[l];
    launch_update_dense(st, S->dp, g0, nd - (split_far ? P.upd_stage_nfar[l] : 0), 0, nd > 0 && P.upd_stage_flops_dense[l] >= 1.5e6 * nd);
    launch_update_gather(st, S->dp, P.gath_stage_ptr[l], P.gath_stage_ptr[l + 1] - P.gath_stage_ptr[l], S->gath_heavy_ptr[l],
                         S->gath_heavy_ptr[l + 1] - S->gath_heavy_ptr[l]);
    launch_update_stage(st, S->dp, g0 + nd + ng, P.upd_stage_ptr[l + 1] - g0 - nd - ng);
}

void enqueue_factor(hipkkt_solver *S, int static_enable, double eps_const, double eps_prop) {
    const HostPlan &P = S->plan;
    hipStream_t st = S->stream;
    launch_zero_words(st, S->dp.scal, 2);                 // SC_MAXDIAG  (kernels.hip: why not hipMemsetAsync)
    launch_zero_words(st, S->dp.flags, FL_COUNT);
    launch_maxabs_gather(st, S->dp.kval, S->d_diag_full, S->N, (unsigned long long *)S->dp.scal + SC_MAXDIAG);
    HK_CHECK(hipMemsetAsync(S->dp.Lx, 0, (size_t)P.panel_doubles * sizeof(double), st));
    launch_init_panels(st, S->dp, S->nnzK, static_enable, eps_const, eps_prop);
    // Far updates (targets more than `lookahead` levels ahead) are forked to the side stream right after the
    // level's factorisation and joined before the next batch end touches the same targets (symbolic.h).
    auto new_event = [&]() {
        hipEvent_t e = nullptr;
        HK_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        S->fork_events.push_back(e);
        return e;
    };
    const bool fork = S->use_side && P.lookahead > 0;
    hipEvent_t pending = nullptr;
    int pending_level = -1;
    const bool fb = S->use_front_block && !S->fbatches.empty();
    if (fb) launch_zero_words(st, S->d_fb_sync, 128 * (int)S->fbatches.size());
    for (int l = 0; l < P.nlevels; l++) {
        if (fb && S->lvl_fb[l] != -1) {
            // a front's update batch: one launch for its panels and their just-in-time updates, then the batch's far stage
            if (S->lvl_fb[l] >= 0)
                launch_front_block(st, S->dp, S->fbatches[S->lvl_fb[l]], S->d_fb_sync, S->d_fb_scratch, S->opts.dynamic_reg_eps,
                                   S->opts.dynamic_reg_delta, S->d_fb_trace);
            const bool last = l + 1 >= P.nlevels || S->lvl_fb[l + 1] != -2;
            if (!last) continue;                          // the stages inside the batch are applied by the kernel itself
        } else {
            enqueue_factor_level(S, l);
        }
        const int nfar = fork ? P.upd_stage_nfar[l] : 0;
        if (pending && (nfar > 0 || l >= pending_level + P.lookahead)) {
            HK_CHECK(hipStreamWaitEvent(st, pending, 0));
            pending = nullptr;
        }
        enqueue_updates(S, l, nfar > 0);
        if (nfar > 0) {   // forked AFTER the near updates: the far tiles must not compete with them for the CUs
            hipEvent_t e1 = new_event(), e2 = new_event();
            HK_CHECK(hipEventRecord(e1, st));
            HK_CHECK(hipStreamWaitEvent(S->side, e1, 0));
            launch_update_dense(S->side, S->dp, P.upd_stage_ptr[l] + P.upd_stage_ndense[l] - nfar, nfar, S->far_wgs);
            HK_CHECK(hipEventRecord(e2, S->side));
            pending = e2;
            pending_level = l;
        }
    }
    if (pending) HK_CHECK(hipStreamWaitEvent(st, pending, 0));
    launch_invert_diag(st, S->dp, S->inv_nsmall, S->inv_wsmall, S->inv_nwide);
}

// in -> out (original ordering on both sides) on context C
void enqueue_ldl_solve(hipkkt_solver *S, SolveCtx &C, const double *in, double *out) {
    const HostPlan &P = S->plan;
    hipStream_t st = C.stream;
    const DevPlan &D = C.dp;
    launch_permute_in(st, in, D.perm, C.d_y, S->N, D.seg_epoch, D.seg_sync, 2 * S->nseg);
    // one launch per level (wide bottom levels, and every level on the fallback path); the leaves of such a level
    // take the thread-per-supernode kernels
    const bool all = !S->use_persist;   // no persistent kernel at all: level lists over every supernode
    const std::vector<int> &slvp = all ? S->all_slv_lvl_ptr : S->slv_lvl_ptr, &bwdp = all ? S->all_bwd_lvl_ptr : S->bwd_lvl_ptr,
                           &regp = all ? S->all_reg_lvl_ptr : S->reg_lvl_ptr;
    const std::vector<int> &nnar = all ? S->all_lvl_nnarrow : S->lvl_nnarrow, &wnar = all ? S->all_lvl_wnarrow : S->lvl_wnarrow;
    auto fwd_level = [&](int l) {
        launch_fwd_narrow(st, D, regp[l], nnar[l], wnar[l], C.d_y, C.d_z);
        launch_fwd_level(st, D, slvp[l], slvp[l + 1] - slvp[l], C.d_y, C.d_z);
    };
    auto bwd_level = [&](int l) {
        launch_bwd_partial(st, D, bwdp[l], bwdp[l + 1] - bwdp[l], C.d_xp);
        launch_bwd_final(st, D, regp[l] + nnar[l], regp[l + 1] - regp[l] - nnar[l], C.d_z, C.d_xp, out);
        launch_bwd_narrow(st, D, regp[l], nnar[l], wnar[l], C.d_z, C.d_xp, out);
    };
    if (S->use_persist) {
        // one persistent launch per segment of regular levels, front kernels in between
        bool first = true;
        for (int g = 0; g < S->nseg; g++) {
            for (int l = S->seg_lo[g]; l < std::min(S->seg_lstar[g], S->seg_hi[g] + 1); l++) fwd_level(l);   // wide bottom levels
            const int n = S->fseg_ptr[2 * g + 1] - S->fseg_ptr[2 * g];
            if (n > 0) { launch_fwd_seg(st, D, g, S->fseg_ptr[2 * g], n, P.nsuper, first ? 1 : 0, C.d_y, C.d_z); first = false; }
            for (const FrontDesc &F : P.fronts)
                if (S->seg_of_level[F.level_last] == g) launch_front_fwd(st, D, F, C.d_y, C.d_z);
        }
        first = true;
        for (int g = S->nseg - 1; g >= 0; g--) {
            for (const FrontDesc &F : P.fronts)
                if (S->seg_of_level[F.level_last] == g) launch_front_bwd(st, D, F, C.d_z, C.d_xp, out);
            const int k = S->nseg - 1 - g;     // launch order index
            const int n = S->bseg_ptr[k + 1] - S->bseg_ptr[k];
            if (n > 0) { launch_bwd_seg(st, D, g, S->bseg_ptr[k], n, P.nsuper, first ? 1 : 0, C.d_z, C.d_xp, out); first = false; }
            for (int l = std::min(S->seg_lstar[g], S->seg_hi[g] + 1) - 1; l >= S->seg_lo[g]; l--) bwd_level(l);
        }
        return;
    }
    for (int l = 0; l < P.nlevels; l++) fwd_level(l);
    for (int l = P.nlevels - 1; l >= 0; l--) bwd_level(l);
}

template <class F>
void run_graphed(hipkkt_solver *S, hipStream_t stream, GraphSlot &slot, bool reusable, F &&enqueue) {
    if (!S->use_graph) { enqueue(); return; }
    if (!(slot.valid && reusable)) {
        if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; slot.valid = false; }
        hipGraph_t graph = nullptr;
        HK_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
        try {
            enqueue();
        } catch (...) {
            (void)hipStreamEndCapture(stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        HK_CHECK(hipStreamEndCapture(stream, &graph));
        hipError_t e = hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { slot.exec = nullptr; S->use_graph = false; enqueue(); return; }
        slot.valid = true;
    }
    HK_CHECK(hipGraphLaunch(slot.exec, stream));
}

// the downgrade to per-level kernels after a sweep time-out is temporary
void maybe_retry_persistent(hipkkt_solver *S) {
    if (!S->use_persist && S->persist_allowed && S->persist_retry_at >= 0 && S->n_ldlsolves >= S->persist_retry_at) {
        S->use_persist = true;
        S->persist_retry_at = -1;
        for (SolveCtx &C : S->ctx) C.g_ldl.valid = C.g_first.valid = C.g_step.valid = false;
    }
}

double slot_value(const hipkkt_solver *S, int slot) {
    double v;
    memcpy(&v, (const char *)S->h_scal + slot * sizeof(double), sizeof(double));
    return v;  // the slot holds the raw bit pattern of a non-negative double (or NaN)
}

void read_scalars(hipkkt_solver *S) {
    HK_CHECK(hipMemcpyAsync(S->h_flags, S->dp.flags, FL_COUNT * sizeof(int), hipMemcpyDeviceToHost, S->stream));
    HK_CHECK(hipMemcpyAsync(S->h_scal, S->dp.scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    HK_CHECK(hipStreamSynchronize(S->stream));
}

// one refinement step on the device: correction solve, candidate = iterate + correction, its residual, the decision
void enqueue_refine_step(hipkkt_solver *S, SolveCtx &C, double reltol, double abstol, int64_t max_iter, double stop_ratio) {
    hipStream_t st = C.stream;
    enqueue_ldl_solve(S, C, C.d_e, C.d_corr);
    launch_refine_add(st, C.d_rs, C.d_x0, C.d_x1, C.d_corr, S->N);
    launch_zero_words(st, (char *)C.dp.scal + SC_NORME * sizeof(double), 2);
    launch_spmv_residual_cand(st, C.dp, C.d_b, C.d_rs, C.d_x0, C.d_x1, C.d_e, S->N, (unsigned long long *)C.dp.scal + SC_NORME);
    launch_refine_decide(st, C.d_rs, C.dp.scal, 1, reltol, abstol, (int)std::min<int64_t>(max_iter, 1 << 30), stop_ratio);
}

// bit 0: a front sweep / forward segment sweep gave up, bit 2: the backward segment sweep gave up
static inline bool sweep_failed(const SolveCtx &C) {
    if (C.h_flags[FL_FRONTFAIL] & ~7) {   // never written by this library (seen once: a small memset node of a captured
                                           // graph wrote garbage under rocprofv3): report it, do not act on it
        static bool told = false;
        if (!told)
            fprintf(stderr, "hipkkt: unexpected value in the flag words: %x %x %x %x\n", C.h_flags[0], C.h_flags[1], C.h_flags[2], C.h_flags[3]);
        told = true;
    }
    return (C.h_flags[FL_FRONTFAIL] & 7) != 0;
}

// A persistent sweep kernel gave up (bounded spin expired: the workgroups were not dispatched in the order the
// hardware was shared with something that starved a hand-off).  Re-arm every hand-off word, drop to the per-level
// kernels and tell the caller to repeat the solve.  The downgrade is temporary: after 64 further LDL solves (doubling
// with every time-out; HIPKKT_PERSIST_RETRY=<n> sets the first interval, 0 = never) the persistent kernels are tried
// again.  Returns false when there is nothing left to fall back to.
bool recover_from_sweep_failure(hipkkt_solver *S) {
    if (!S->use_persist) return false;
    const HostPlan &P = S->plan;
    for (SolveCtx &C : S->ctx) HK_CHECK(hipStreamSynchronize(C.stream));
    S->n_sweep_timeouts++;
    {
        const char *rt = getenv("HIPKKT_PERSIST_RETRY");
        const int64_t first = rt ? atoll(rt) : 64;
        S->persist_backoff = S->persist_backoff > 0 ? 2 * S->persist_backoff : first;
        S->persist_retry_at = first > 0 ? S->n_ldlsolves + S->persist_backoff : -1;
    }
    fprintf(stderr, "hipkkt: a persistent sweep kernel timed out (flags 0x%x 0x%x); per-level solve kernels for the next %lld LDL solves\n",
            S->ctx[0].h_flags[FL_FRONTFAIL], S->ctx[1].h_flags[FL_FRONTFAIL], (long long)(S->persist_retry_at >= 0 ? S->persist_backoff : -1));
    const size_t nsync = seg_sync_ints(S->nseg, P.nsuper);
    for (SolveCtx &C : S->ctx) {
        fill_async(S->stream, C.dp.seg_sync, 0, nsync * sizeof(int));
        fill_async(S->stream, C.dp.front_sync, 0, (size_t)std::max(P.front_sync_ints, 16) * sizeof(int));
        fill_async(S->stream, C.dp.flags + FL_FRONTFAIL, 0, sizeof(int));
        C.h_flags[FL_FRONTFAIL] = 0;
        C.g_ldl.valid = C.g_first.valid = C.g_step.valid = false;
    }
    HK_CHECK(hipStreamSynchronize(S->stream));
    S->use_persist = false;
    return true;
}

// ref: kktsolver_solve! + _iterative_refinement (kktsolver_directldl.jl:346-449); C.d_b holds b.  Enqueues the first
// solve, its residual, the device-side decision and ONE refinement step (decided on the device whether it counts), then
// the read-back of the state -- no synchronisation: several contexts can be started before any is finished.
void solve_begin(hipkkt_solver *S, SolveCtx &C, int ir_enable, double reltol, double abstol, int64_t max_iter, double stop_ratio) {
    hipStream_t st = C.stream;
    HK_CHECK(hipEventRecord(C.ev_a, st));
    C.ir_used = ir_enable != 0;
    if (ir_enable) {
        const bool same = C.g_reltol == reltol && C.g_abstol == abstol && C.g_maxit == max_iter && C.g_stop == stop_ratio;
        run_graphed(S, st, C.g_first, same, [&] {
            enqueue_ldl_solve(S, C, C.d_b, C.d_x0);
            launch_zero_words(st, (char *)C.dp.scal + SC_NORMB * sizeof(double), 4);
            launch_norm_inf(st, C.d_b, S->N, (unsigned long long *)C.dp.scal + SC_NORMB);
            launch_spmv_residual(st, C.dp, C.d_b, C.d_x0, C.d_e, S->N, (unsigned long long *)C.dp.scal + SC_NORME);
            launch_refine_decide(st, C.d_rs, C.dp.scal, 0, reltol, abstol, (int)std::min<int64_t>(max_iter, 1 << 30), stop_ratio);
            if (max_iter > 0) enqueue_refine_step(S, C, reltol, abstol, max_iter, stop_ratio);
        });
        if (!same) { C.g_step.valid = false; C.g_reltol = reltol; C.g_abstol = abstol; C.g_maxit = max_iter; C.g_stop = stop_ratio; }
        S->n_ldlsolves += max_iter > 0 ? 2 : 1;
    } else {
        run_graphed(S, st, C.g_ldl, true, [&] {
            enqueue_ldl_solve(S, C, C.d_b, C.d_x0);
            launch_zero_words(st, C.dp.flags, 1);
            launch_check_finite(st, C.d_x0, S->N, C.dp.flags);
        });
        S->n_ldlsolves += 1;
    }
}

// copy the accepted iterate of a started solve to a device buffer (first nm entries), still without synchronising
void solve_copy_out_dev(hipkkt_solver *S, SolveCtx &C, double *out_dev, int nm) {
    if (!out_dev) return;
    if (C.ir_used) launch_refine_copy_out(C.stream, C.d_rs, C.d_x0, C.d_x1, out_dev, nm);
    else HK_CHECK(hipMemcpyAsync(out_dev, C.d_x0, (size_t)nm * sizeof(double), hipMemcpyDeviceToDevice, C.stream));
}

// waits for a started solve, runs further refinement steps while the device says so, reports like the reference
int32_t solve_finish(hipkkt_solver *S, SolveCtx &C, int64_t *ir_steps, double *out_dev, int nm) {
    hipStream_t st = C.stream;
    auto readback = [&] {
        if (C.ir_used) HK_CHECK(hipMemcpyAsync(C.h_rs, C.d_rs, sizeof(RefineState), hipMemcpyDeviceToHost, st));
        HK_CHECK(hipMemcpyAsync(C.h_flags, C.dp.flags, FL_COUNT * sizeof(int), hipMemcpyDeviceToHost, st));
        HK_CHECK(hipEventRecord(C.ev_b, st));
        HK_CHECK(hipStreamSynchronize(st));
    };
    readback();
    bool more = false;
    while (C.ir_used && C.h_rs->active && !sweep_failed(C)) {   // rare: more than one step needed
        run_graphed(S, st, C.g_step, true, [&] { enqueue_refine_step(S, C, C.g_reltol, C.g_abstol, C.g_maxit, C.g_stop); });
        S->n_ldlsolves += 1;
        more = true;
        readback();
    }
    if (more && out_dev) {   // the accepted iterate changed after the copy that solve_copy_out_dev enqueued
        solve_copy_out_dev(S, C, out_dev, nm);
        HK_CHECK(hipEventRecord(C.ev_b, st));
        HK_CHECK(hipStreamSynchronize(st));
    }
    float ms = 0;
    HK_CHECK(hipEventElapsedTime(&ms, C.ev_a, C.ev_b));
    C.last_ms = ms;
    C.last_steps = C.ir_used ? C.h_rs->steps : 0;
    if (ir_steps) *ir_steps = C.last_steps;
    if (sweep_failed(C)) { S->err = "persistent solve kernel timed out"; return HIPKKT_ERR_DEVICE; }
    const bool ok = C.ir_used ? C.h_rs->fail == 0 : C.h_flags[FL_NONFINITE] == 0;
    return ok ? HIPKKT_OK : HIPKKT_NUMERICAL_FAILURE;
}

// nrhs (<= kNumCtx) right-hand sides already in ctx[c].d_b: solved concurrently, results optionally copied to out_dev[c]
int32_t solve_many(hipkkt_solver *S, int nrhs, int ir_enable, double reltol, double abstol, int64_t max_iter, double stop_ratio,
                   int64_t *ir_steps, double *const *out_dev, int nm) {
    for (int attempt = 0; attempt < 2; attempt++) {
        maybe_retry_persistent(S);
        for (int c = 0; c < nrhs; c++) {
            solve_begin(S, S->ctx[c], ir_enable, reltol, abstol, max_iter, stop_ratio);
            solve_copy_out_dev(S, S->ctx[c], out_dev ? out_dev[c] : nullptr, nm);
        }
        int32_t rc = HIPKKT_OK;
        double ms = 0;
        for (int c = 0; c < nrhs; c++) {
            const int32_t r = solve_finish(S, S->ctx[c], ir_steps ? ir_steps + c : nullptr, out_dev ? out_dev[c] : nullptr, nm);
            if (r < 0 || (r > 0 && rc == HIPKKT_OK)) rc = r < 0 ? r : (rc < 0 ? rc : r);
            ms = std::max(ms, S->ctx[c].last_ms);
        }
        bool timed_out = false;
        for (int c = 0; c < nrhs; c++) timed_out = timed_out || sweep_failed(S->ctx[c]);
        if (timed_out && recover_from_sweep_failure(S)) continue;   // repeat everything on the per-level kernels
        S->t_last_solve = ms;
        S->t_acc_solve += ms;
        S->n_solvecalls++;
        S->n_rhs_solved += nrhs;
        S->d_x = const_cast<double *>(S->ctx[0].result());
        return rc;
    }
    return HIPKKT_ERR_DEVICE;
}

// the solver that holds the current factorisation (the robust-order twin after a fallback)
hipkkt_solver *solve_target(hipkkt_solver *S) { return (S->using_fallback && S->fallback) ? S->fallback : S; }
void account_fallback_solve(hipkkt_solver *S, hipkkt_solver *T) {
    if (T == S) return;
    S->t_last_solve = T->t_last_solve;
    S->t_acc_solve += T->t_last_solve;
    S->n_solvecalls++;
    S->n_rhs_solved += 1;
}

int32_t finish_create(hipkkt_solver *S, const hipkkt_opts *opts, hipkkt_handle *out) {
    PlanOptions po;
    po.max_width = opts->supernode_max_width > 0 ? opts->supernode_max_width : kMaxSnWidth;
    po.relax = opts->relax_supernodes != 0;
    po.update_policy = opts->update_policy;
    if (opts->update_batch > 0) po.update_batch = opts->update_batch;
    po.amd_dense_scale = opts->amd_dense_scale > 0 ? opts->amd_dense_scale : 1.5;
    po.front_min_panels = opts->front_min_panels == 0 ? 4 : std::max(0, opts->front_min_panels);
    po.n_hold = S->l1 ? (int)S->img.n : 0;
    {
        const char *ns = getenv("HIPKKT_SIDE_STREAM");
        po.split_far = ns && ns[0] == '1';
        const char *mr = getenv("HIPKKT_FRONT_BLOCK_MIN_ROWS");   // tests: 0, so that small fronts take the front-batch kernel too
        if (mr) po.front_block_min_width = atoi(mr);
        const char *dc = getenv("HIPKKT_DENSE_COVER");   // A/B: threshold between the tile path and the gather lists
        if (dc && atof(dc) > 0) po.dense_min_cover = atof(dc);
        const char *nf = getenv("HIPKKT_FUSE_JIT");    // experiment: just-in-time updates inside the panel kernel
        if (nf && nf[0] == '1') po.fuse_jit = true;
        const char *nx = getenv("HIPKKT_XCD_ORDER");   // measured: no effect on cfg 2a (L2 locality is not the limiter)
        po.xcd_order = nx && nx[0] == '1';
    }
    {
        const char *nh = getenv("HIPKKT_ORDERING");   // "amd": minimum degree on K only
        if (nh && nh[0] == 'a') po.n_hold = 0;
    }
    {
        const char *nf = getenv("HIPKKT_NO_FRONT");
        if (nf && nf[0] == '1') po.front_min_panels = 0;
    }
    std::vector<int64_t> up;
    const int64_t *uperm = nullptr;
    if (opts->user_perm) {
        up.resize(S->img.N);
        for (int64_t k = 0; k < S->img.N; k++) up[k] = opts->user_perm[k] - opts->index_base;
        uperm = up.data();
    }
    if (S->img.N >= ((int64_t)1 << 31)) { g_create_error = "N exceeds int32"; delete S; return HIPKKT_ERR_ARGUMENT; }
    {
        // The "cone rows first" order can break down on an ill-conditioned iterate; the factorisation is
        // then repeated on a twin handle in the minimum-degree order.  Its symbolic analysis is seconds of host work on the
        // problems that take this path (dense PSD blocks), so it starts on a host thread as soon as the minimum-degree order is
        // known and the cheap order is about to be evaluated against it -- speculatively: if the cheap order is not chosen the
        // thread is cancelled at its next phase boundary (HIPKKT_TWIN_AHEAD=0: analysed only when it is needed).
        const char *ta = getenv("HIPKKT_TWIN_AHEAD");
        if (!(ta && ta[0] == '0') && S->l1)
            po.on_alternative_order = [S, &po](const std::vector<int> &perm_md) {
                std::unique_ptr<hipkkt_solver> T(new hipkkt_solver());
                T->device = S->device;
                T->opts = S->opts;
                T->l1 = S->l1;
                T->img = S->img;
                PlanOptions po2 = po;
                po2.n_hold = 0;
                po2.on_alternative_order = nullptr;
                S->twin_cancel = std::make_shared<std::atomic<bool>>(false);
                po2.cancel = S->twin_cancel.get();
                T->plan_opts = po2;
                hipkkt_solver *Tp = T.get();
                std::vector<int64_t> pv(perm_md.begin(), perm_md.end());
                S->twin_pending = std::move(T);
                S->twin_future = std::async(std::launch::async, [Tp, pv, po2]() {
                    return build_plan((int)Tp->img.N, Tp->img.colptr.data(), Tp->img.rowval.data(), pv.data(), po2, Tp->plan);
                });
            };
    }
    const auto t_a = std::chrono::steady_clock::now();
    std::string err = build_plan((int)S->img.N, S->img.colptr.data(), S->img.rowval.data(), uperm, po, S->plan);
    po.on_alternative_order = nullptr;
    if (!err.empty()) { g_create_error = err; delete S; return HIPKKT_ERR_ARGUMENT; }
    if (S->plan.ordering_used != 1 && S->twin_cancel) S->twin_cancel->store(true);   // speculative twin not needed: the thread stops at its next phase
    S->plan_opts = po;
    const auto t_b = std::chrono::steady_clock::now();
    try {
        if (!S->runtime_ready) init_runtime(S);
        S->runtime_ready = true;
        setup_device(S);
        if (getenv("HIPKKT_VERBOSE"))
            fprintf(stderr, "hipkkt: N %d nnzL %lld levels %d ordering %d: symbolic %.2f ms (%s), device set-up %.2f ms, runtime objects %.2f ms\n", S->plan.N, (long long)S->plan.nnzL,
                    S->plan.nlevels, S->plan.ordering_used, 1e3 * std::chrono::duration<double>(t_b - t_a).count(), S->plan.timing_note.c_str(),
                    1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_b).count(), 1e3 * S->t_init_runtime);
    } catch (const DeviceError &e) {
        g_create_error = e.msg; delete S; return HIPKKT_ERR_DEVICE;
    } catch (const std::bad_alloc &) {
        g_create_error = "out of (device) memory"; delete S; return HIPKKT_ERR_ALLOC;
    }
    *out = S;
    return HIPKKT_OK;
}

}  // namespace

#define HK_ENTER(h)                                                      \
    if (!(h)) return HIPKKT_ERR_ARGUMENT;                                \
    hipkkt_solver *S = (h);                                              \
    try {                                                                \
        if (hipSetDevice(S->device) != hipSuccess) { S->err = "hipSetDevice failed"; return HIPKKT_ERR_DEVICE; }

#define HK_LEAVE                                                         \
    }                                                                    \
    catch (const DeviceError &e) { S->err = e.msg; return HIPKKT_ERR_DEVICE; } \
    catch (const std::bad_alloc &) { S->err = "out of memory"; return HIPKKT_ERR_ALLOC; } \
    catch (...) { S->err = "internal error"; return HIPKKT_ERR_INTERNAL; }

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
        // the image is assembled by count -> scan -> fill kernels on the device (assemble_dev.hip); HIPKKT_HOST_ASSEMBLY=1
        // selects the host twin (assemble.cpp), which the GPU tests compare the device image with
        const char *ha = getenv("HIPKKT_HOST_ASSEMBLY");
        std::string err;
        if (ha && ha[0] == '1') {
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
    if (h && getenv("HIPKKT_VERBOSE")) {
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
    for (int k = 0; k < h->N; k++) perm[k] = h->plan.perm[k] + h->opts.index_base;
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
    HK_CHECK(hipStreamSynchronize(S->stream));
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
    if (row != m || hs != S->img.nHs) { S->err = "set_cone_types: cone sizes do not add up to m / the Hs vector"; return HIPKKT_ERR_ARGUMENT; }
    S->d_sc_kind = S->upload(kind);
    S->d_sc_rowhs = S->upload(rowhs);
    if (socdesc.empty()) socdesc.assign(5, 0);
    S->d_sc_socdesc = S->upload(socdesc);
    S->d_sc_sz = S->dalloc<double>(2 * m);
    S->d_sc_wl = S->dalloc<double>(2 * m);
    S->d_sc_eta = S->dalloc<double>(S->sc_nsoc);
    S->d_sc_R = S->dalloc<double>(S->sc_psd_total);
    S->d_sc_W = S->dalloc<double>(S->sc_psd_total);
    S->d_sc_fail = S->dalloc<int>(1);
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

// ---- factor ------------------------------------------------------------------------------------

static int32_t refactor_once(hipkkt_handle h, int32_t static_reg_enable, double eps_const, double eps_prop,
                             double *eps_used, int64_t *n_dynamic_reg);

// The "variables last" order (plan.ordering_used == 1) can be far cheaper than minimum degree on K but
// eliminates the ill-conditioned cone blocks first; if a factorisation in that order ends with a
// non-finite pivot, the handle is rebuilt ONCE with the minimum-degree order on K (the reference's
// choice) and the factorisation is repeated, so robustness is never worse than with that order.
int32_t hipkkt_refactor(hipkkt_handle h, int32_t static_reg_enable, double eps_const, double eps_prop,
                        double *eps_used, int64_t *n_dynamic_reg) {
    if (h) h->using_fallback = false;
    int32_t rc = refactor_once(h, static_reg_enable, eps_const, eps_prop, eps_used, n_dynamic_reg);
    if (rc != HIPKKT_NUMERICAL_FAILURE || !h || h->plan.ordering_used != 1) return rc;
    hipkkt_solver *S = h;
    try {
        if (hipSetDevice(S->device) != hipSuccess) return rc;
        if (!S->fallback) {
            // built in a local owner: a set-up that throws half-way (e.g. device OOM) must not leave a twin with null
            // streams / device pointers behind -- the next failing factorisation then simply tries again
            std::unique_ptr<hipkkt_solver> T;
            const auto t_a = std::chrono::steady_clock::now();
            if (S->twin_future.valid()) {               // analysed ahead on a host thread (finish_create): wait for it
                const std::string err = S->twin_future.get();
                T = std::move(S->twin_pending);
                if (!err.empty()) T.reset();
                else T->plan_opts.cancel = nullptr;
            }
            if (!T) {
                T.reset(new hipkkt_solver());
                T->device = S->device;
                T->opts = S->opts;
                T->l1 = S->l1;
                T->img = S->img;
                PlanOptions po = S->plan_opts;
                po.n_hold = 0;
                std::string err = build_plan((int)T->img.N, T->img.colptr.data(), T->img.rowval.data(), nullptr, po, T->plan);
                if (!err.empty()) return rc;
                T->plan_opts = po;
            }
            const auto t_b = std::chrono::steady_clock::now();
            init_runtime(T.get());
            setup_device(T.get());
            if (getenv("HIPKKT_VERBOSE"))
                fprintf(stderr, "hipkkt: robust-order twin (minimum degree on K): N %d nnzL %lld levels %d: waited %.2f ms for its symbolic analysis (%s), device set-up %.2f ms\n",
                        T->plan.N, (long long)T->plan.nnzL, T->plan.nlevels, 1e3 * std::chrono::duration<double>(t_b - t_a).count(),
                        T->plan.timing_note.c_str(), 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_b).count());
            S->fallback = T.release();
        }
        copy_sync(S->stream, S->fallback->dp.kval, S->dp.kval, (size_t)S->nnzK * sizeof(double), hipMemcpyDeviceToDevice);
    } catch (...) {
        S->err = "building the fallback (minimum-degree) factorisation failed";
        return HIPKKT_ERR_DEVICE;
    }
    rc = refactor_once(S->fallback, static_reg_enable, eps_const, eps_prop, eps_used, n_dynamic_reg);
    S->using_fallback = true;
    S->n_twin_refactors++;
    S->last_eps = S->fallback->last_eps;
    S->last_nreg = S->fallback->last_nreg;
    S->t_last_factor += S->fallback->t_last_factor;      // the failed attempt + the repeated one
    S->t_acc_factor += S->fallback->t_last_factor;
    return rc;
}

static int32_t refactor_once(hipkkt_handle h, int32_t static_reg_enable, double eps_const, double eps_prop,
                             double *eps_used, int64_t *n_dynamic_reg) {
    HK_ENTER(h)
    HK_CHECK(hipEventRecord(S->ev0, S->stream));
    if (S->profiling) {
        // eager, with the dense-update launches timed separately (adds event overhead)
        const HostPlan &P = S->plan;
        hipStream_t st = S->stream;
        launch_zero_words(st, S->dp.scal, 2);
        launch_zero_words(st, S->dp.flags, FL_COUNT);
        launch_maxabs_gather(st, S->dp.kval, S->d_diag_full, S->N, (unsigned long long *)S->dp.scal + SC_MAXDIAG);
        HK_CHECK(hipMemsetAsync(S->dp.Lx, 0, (size_t)P.panel_doubles * sizeof(double), st));
        launch_init_panels(st, S->dp, S->nnzK, static_reg_enable, eps_const, eps_prop);
        std::vector<hipEvent_t> evs, evd;   // evs: all update kernels of a stage; evd: its k_update_dense<4,4> launch alone
        std::vector<int> evd_level;
        const bool fb = S->use_front_block && !S->fbatches.empty();
        if (fb) launch_zero_words(st, S->d_fb_sync, 128 * (int)S->fbatches.size());
        std::vector<hipEvent_t> evf;        // around every k_front_block launch
        int fb_panels = 0;
        double fb_flops = 0;                // update flops of the stages inside the batches (executed by k_front_block)
        for (int l = 0; l < P.nlevels; l++) {
            if (fb && S->lvl_fb[l] != -1) {
                if (S->lvl_fb[l] >= 0) {
                    hipEvent_t a, b;
                    HK_CHECK(hipEventCreate(&a));
                    HK_CHECK(hipEventCreate(&b));
                    HK_CHECK(hipEventRecord(a, st));
                    launch_front_block(st, S->dp, S->fbatches[S->lvl_fb[l]], S->d_fb_sync, S->d_fb_scratch, S->opts.dynamic_reg_eps,
                                       S->opts.dynamic_reg_delta, S->d_fb_trace);
                    HK_CHECK(hipEventRecord(b, st));
                    evf.push_back(a);
                    evf.push_back(b);
                    fb_panels += S->fbatches[S->lvl_fb[l]].nb;
                }
                if (l + 1 < P.nlevels && S->lvl_fb[l + 1] == -2) { fb_flops += P.upd_stage_flops_dense[l]; continue; }   // applied by the kernel
            } else {
                enqueue_factor_level(S, l);
            }
            const bool fused_next = l + 1 < P.nlevels && P.lvl_fused[l + 1];   // applied by the next panel kernel
            if (P.upd_stage_ptr[l + 1] > P.upd_stage_ptr[l] && !fused_next) {
                hipEvent_t a, b, c2;
                HK_CHECK(hipEventCreate(&a));
                HK_CHECK(hipEventCreate(&b));
                HK_CHECK(hipEventRecord(a, st));
                const int g0 = P.upd_stage_ptr[l], nd = P.upd_stage_ndense[l], ng = P.upd_stage_ngather[l];
                launch_update_dense(st, S->dp, g0, nd, 0, nd > 0 && P.upd_stage_flops_dense[l] >= 1.5e6 * nd);
                if (nd > 384) {   // the one-wavefront-per-tile variant (see launch_update_dense)
                    HK_CHECK(hipEventCreate(&c2));
                    HK_CHECK(hipEventRecord(c2, st));
                    evd.push_back(a);
                    evd.push_back(c2);
                    evd_level.push_back(l);
                }
                launch_update_gather(st, S->dp, P.gath_stage_ptr[l], P.gath_stage_ptr[l + 1] - P.gath_stage_ptr[l], S->gath_heavy_ptr[l],
                         S->gath_heavy_ptr[l + 1] - S->gath_heavy_ptr[l]);
                launch_update_stage(st, S->dp, g0 + nd + ng, P.upd_stage_ptr[l + 1] - g0 - nd - ng);
                HK_CHECK(hipEventRecord(b, st));
                evs.push_back(a);
                evs.push_back(b);
            }
        }
        launch_invert_diag(st, S->dp, S->inv_nsmall, S->inv_wsmall, S->inv_nwide);
        HK_CHECK(hipStreamSynchronize(st));
        double tot = 0;
        for (size_t i = 0; i + 1 < evs.size(); i += 2) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, evs[i], evs[i + 1]);
            tot += ms;
        }
        S->prof_dense4_ms = 0; S->prof_dense4_flops = 0; S->prof_dense4_launches = 0;
        S->prof_launch_ms.clear(); S->prof_launch_flops.clear(); S->prof_launch_tiles.clear();
        for (size_t i = 0; i + 1 < evd.size(); i += 2) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, evd[i], evd[i + 1]);
            S->prof_dense4_ms += ms;
            S->prof_dense4_flops += P.upd_stage_flops_dense[evd_level[i / 2]];
            S->prof_dense4_launches++;
            S->prof_launch_ms.push_back(ms);
            S->prof_launch_flops.push_back(P.upd_stage_flops_dense[evd_level[i / 2]]);
            S->prof_launch_tiles.push_back(P.upd_stage_ndense[evd_level[i / 2]]);
        }
        for (size_t i = 1; i < evd.size(); i += 2) (void)hipEventDestroy(evd[i]);
        for (hipEvent_t e : evs) (void)hipEventDestroy(e);
        S->prof_fb_ms = 0; S->prof_fb_launches = (int)(evf.size() / 2); S->prof_fb_panels = fb_panels; S->prof_fb_flops = fb_flops;
        for (size_t i = 0; i + 1 < evf.size(); i += 2) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, evf[i], evf[i + 1]);
            S->prof_fb_ms += ms;
        }
        for (hipEvent_t e : evf) (void)hipEventDestroy(e);
        S->t_last_update = tot;
    } else {
        GraphSlot &g = S->g_factor;
        const bool same = g.static_enable == static_reg_enable && g.eps_const == eps_const && g.eps_prop == eps_prop;
        run_graphed(S, S->stream, g, same, [&] { enqueue_factor(S, static_reg_enable, eps_const, eps_prop); });
        g.static_enable = static_reg_enable; g.eps_const = eps_const; g.eps_prop = eps_prop;
    }
    HK_CHECK(hipEventRecord(S->ev1, S->stream));
    HK_CHECK(hipMemcpyAsync(S->h_flags, S->dp.flags, FL_COUNT * sizeof(int), hipMemcpyDeviceToHost, S->stream));
    read_scalars(S);
    float ms = 0;
    HK_CHECK(hipEventElapsedTime(&ms, S->ev0, S->ev1));
    S->t_last_factor = ms;
    S->t_acc_factor += ms;
    S->n_factor++;
    if (S->h_flags[FL_FACFAIL] && S->use_front_block) {
        // a spin of k_front_block ran out (a stalled workgroup): repeat this factorisation with one launch per panel, and keep that
        S->use_front_block = false;
        S->g_factor.valid = false;
        fprintf(stderr, "hipkkt: a hand-off of the front-batch factorisation timed out; repeating it with one launch per panel (kept from now on)\n");
        S->n_sweep_timeouts++;
        return refactor_once(h, static_reg_enable, eps_const, eps_prop, eps_used, n_dynamic_reg);
    }
    const double maxdiag = slot_value(S, SC_MAXDIAG);
    S->last_eps = static_reg_enable ? eps_const + eps_prop * maxdiag : 0.0;
    S->last_nreg = S->h_flags[FL_NREG];
    if (eps_used) *eps_used = S->last_eps;
    if (n_dynamic_reg) *n_dynamic_reg = S->last_nreg;
    return S->h_flags[FL_NONFINITE] ? HIPKKT_NUMERICAL_FAILURE : HIPKKT_OK;
    HK_LEAVE
}

// ---- solve -------------------------------------------------------------------------------------

int32_t hipkkt_setrhs(hipkkt_handle h, const double *rhsx, const double *rhsz) {
    HK_ENTER(h)
    const int64_t n = S->img.n, m = S->img.m;
    if (!S->l1 || (n && !rhsx) || (m && !rhsz)) return HIPKKT_ERR_ARGUMENT;
    if (n) HK_CHECK(hipMemcpyAsync(S->d_b, rhsx, n * sizeof(double), hipMemcpyHostToDevice, S->stream));
    if (m) HK_CHECK(hipMemcpyAsync(S->d_b + n, rhsz, m * sizeof(double), hipMemcpyHostToDevice, S->stream));
    if (S->img.p) HK_CHECK(hipMemsetAsync(S->d_b + n + m, 0, S->img.p * sizeof(double), S->stream));
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_setrhs_dev(hipkkt_handle h, const double *rhs_dev) {
    HK_ENTER(h)
    if (!S->l1 || !rhs_dev) return HIPKKT_ERR_ARGUMENT;
    launch_set_rhs(S->stream, S->d_b, rhs_dev, (int)(S->img.n + S->img.m), S->N);
    HK_CHECK(hipStreamSynchronize(S->stream));
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_solve(hipkkt_handle h, double *lhsx, double *lhsz, int32_t ir_enable, double reltol, double abstol,
                     int64_t max_iter, double stop_ratio, int64_t *ir_steps) {
    HK_ENTER(h)
    if (!S->l1) return HIPKKT_ERR_ARGUMENT;
    hipkkt_solver *T = solve_target(S);
    if (T != S) copy_sync(S->stream, T->ctx[0].d_b, S->ctx[0].d_b, (size_t)S->N * sizeof(double), hipMemcpyDeviceToDevice);
    int32_t rc = solve_many(T, 1, ir_enable, reltol, abstol, max_iter, stop_ratio, ir_steps, nullptr, 0);
    account_fallback_solve(S, T);
    if (rc == HIPKKT_OK) {  // ref: kktsolver_getlhs! only on success
        const int64_t n = S->img.n, m = S->img.m;
        const double *x = T->ctx[0].result();
        if (lhsx && n) copy_sync(S->stream, lhsx, x, n * sizeof(double), hipMemcpyDeviceToHost);
        if (lhsz && m) copy_sync(S->stream, lhsz, x + n, m * sizeof(double), hipMemcpyDeviceToHost);
    }
    return rc;
    HK_LEAVE
}

int32_t hipkkt_solve_dev(hipkkt_handle h, double *lhs_dev, int32_t ir_enable, double reltol, double abstol,
                         int64_t max_iter, double stop_ratio, int64_t *ir_steps) {
    HK_ENTER(h)
    if (!S->l1) return HIPKKT_ERR_ARGUMENT;
    hipkkt_solver *T = solve_target(S);
    if (T != S) copy_sync(S->stream, T->ctx[0].d_b, S->ctx[0].d_b, (size_t)S->N * sizeof(double), hipMemcpyDeviceToDevice);
    double *outs[1] = {lhs_dev};
    int32_t rc = solve_many(T, 1, ir_enable, reltol, abstol, max_iter, stop_ratio, ir_steps, outs, (int)(S->img.n + S->img.m));
    account_fallback_solve(S, T);
    return rc;
    HK_LEAVE
}

// several right-hand sides on one factorisation, two at a time on concurrent solve contexts
static int32_t solve_multi_impl(hipkkt_solver *S, int64_t nrhs, const double *rhsx, const double *rhsz, const double *rhs_dev,
                                double *lhsx, double *lhsz, double *lhs_dev, int32_t ir_enable, double reltol, double abstol,
                                int64_t max_iter, double stop_ratio, int64_t *ir_steps) {
    const int64_t n = S->img.n, m = S->img.m, p = S->img.p;
    hipkkt_solver *T = solve_target(S);
    int32_t rc_all = HIPKKT_OK;
    for (int64_t r0 = 0; r0 < nrhs; r0 += kNumCtx) {
        const int k = (int)std::min<int64_t>(kNumCtx, nrhs - r0);
        double *outs[kNumCtx] = {nullptr, nullptr};
        for (int c = 0; c < k; c++) {
            SolveCtx &C = T->ctx[c];
            const int64_t r = r0 + c;
            if (rhs_dev) {
                launch_set_rhs(C.stream, C.d_b, rhs_dev + r * (n + m), (int)(n + m), T->N);
            } else {
                if (n) HK_CHECK(hipMemcpyAsync(C.d_b, rhsx + r * n, n * sizeof(double), hipMemcpyHostToDevice, C.stream));
                if (m) HK_CHECK(hipMemcpyAsync(C.d_b + n, rhsz + r * m, m * sizeof(double), hipMemcpyHostToDevice, C.stream));
                if (p) HK_CHECK(hipMemsetAsync(C.d_b + n + m, 0, p * sizeof(double), C.stream));
            }
            if (lhs_dev) outs[c] = lhs_dev + r * (n + m);
        }
        const int32_t rc = solve_many(T, k, ir_enable, reltol, abstol, max_iter, stop_ratio, ir_steps ? ir_steps + r0 : nullptr,
                                      lhs_dev ? outs : nullptr, (int)(n + m));
        if (T != S) { S->t_last_solve = T->t_last_solve; S->t_acc_solve += T->t_last_solve; S->n_solvecalls++; S->n_rhs_solved += k; }
        if (rc < 0) return rc;
        if (rc > 0) rc_all = rc;
        if (rc == HIPKKT_OK && (lhsx || lhsz))
            for (int c = 0; c < k; c++) {
                const double *x = T->ctx[c].result();
                const int64_t r = r0 + c;
                if (lhsx && n) copy_sync(S->stream, lhsx + r * n, x, n * sizeof(double), hipMemcpyDeviceToHost);
                if (lhsz && m) copy_sync(S->stream, lhsz + r * m, x + n, m * sizeof(double), hipMemcpyDeviceToHost);
            }
    }
    return rc_all;
}

int32_t hipkkt_solve_multi(hipkkt_handle h, int64_t nrhs, const double *rhsx, const double *rhsz, double *lhsx, double *lhsz,
                           int32_t ir_enable, double reltol, double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps) {
    HK_ENTER(h)
    if (!S->l1 || nrhs < 0 || (nrhs && ((S->img.n && !rhsx) || (S->img.m && !rhsz)))) return HIPKKT_ERR_ARGUMENT;
    return solve_multi_impl(S, nrhs, rhsx, rhsz, nullptr, lhsx, lhsz, nullptr, ir_enable, reltol, abstol, max_iter, stop_ratio, ir_steps);
    HK_LEAVE
}

int32_t hipkkt_solve_multi_dev(hipkkt_handle h, int64_t nrhs, const double *rhs_dev, double *lhs_dev, int32_t ir_enable, double reltol,
                               double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps) {
    HK_ENTER(h)
    if (!S->l1 || nrhs < 0 || (nrhs && !rhs_dev)) return HIPKKT_ERR_ARGUMENT;
    return solve_multi_impl(S, nrhs, nullptr, nullptr, rhs_dev, nullptr, nullptr, lhs_dev, ir_enable, reltol, abstol, max_iter, stop_ratio, ir_steps);
    HK_LEAVE
}

int32_t hipkkt_ldl_solve(hipkkt_handle h, double *x, const double *b) {
    if (h && h->using_fallback && h->fallback) return hipkkt_ldl_solve(h->fallback, x, b);
    HK_ENTER(h)
    if (!x || !b) return HIPKKT_ERR_ARGUMENT;
    SolveCtx &C = S->ctx[0];
    HK_CHECK(hipMemcpyAsync(C.d_b, b, (size_t)S->N * sizeof(double), hipMemcpyHostToDevice, C.stream));
    int32_t rc = solve_many(S, 1, 0, 0.0, 0.0, 0, 0.0, nullptr, nullptr, 0);
    if (rc < 0) return rc;
    // ref: solve!(ldlsolver,K,x,b) returns whatever the triangular solves produce; a non-finite result is the caller's to detect
    copy_sync(S->stream, x, C.d_x0, (size_t)S->N * sizeof(double), hipMemcpyDeviceToHost);
    return HIPKKT_OK;
    HK_LEAVE
}

int32_t hipkkt_get_timing(hipkkt_handle h, double *o) {
    if (!h || !o) return HIPKKT_ERR_ARGUMENT;
    o[0] = h->t_last_factor; o[1] = h->t_last_solve; o[2] = h->t_acc_factor; o[3] = h->t_acc_solve;
    o[4] = (double)h->n_factor; o[5] = (double)h->n_solvecalls; o[6] = (double)h->n_ldlsolves; o[7] = h->t_last_update;
    return HIPKKT_OK;
}

int32_t hipkkt_get_profile(hipkkt_handle h, double *o) {
    if (!h || !o) return HIPKKT_ERR_ARGUMENT;
    o[0] = h->t_last_update; o[1] = h->prof_dense4_ms; o[2] = h->prof_dense4_flops; o[3] = (double)h->prof_dense4_launches;
    o[4] = h->prof_fb_ms; o[5] = (double)h->prof_fb_launches; o[6] = (double)h->prof_fb_panels; o[7] = h->prof_fb_flops;
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

int32_t hipkkt_get_counters(hipkkt_handle h, int64_t *o) {
    if (!h || !o) return HIPKKT_ERR_ARGUMENT;
    o[0] = h->n_sweep_timeouts; o[1] = h->use_persist ? 1 : 0; o[2] = h->n_twin_refactors; o[3] = h->fallback ? 1 : 0;
    o[4] = h->using_fallback ? 1 : 0; o[5] = h->plan.ordering_used; o[6] = (int64_t)h->plan.fronts.size(); o[7] = h->nseg;
    o[8] = (int64_t)h->fbatches.size(); o[9] = h->use_front_block ? 1 : 0; o[10] = o[11] = 0;
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

const char *hipkkt_last_error(hipkkt_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
