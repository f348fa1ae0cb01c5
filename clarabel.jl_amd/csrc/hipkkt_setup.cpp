// Runtime objects, device residency of a symbolic plan and handle creation (see hipkkt_internal.h for the file map).
#include "hipkkt_internal.h"

using namespace hipkkt;
using namespace hipkkt_host;

namespace hipkkt_host {

// Process-wide cache of symbolic plans keyed by the KKT pattern + the plan options: a batch of problems with
// IDENTICAL structure (the same model re-solved with new data, MPC, parameter sweeps) pays the ordering + symbolic analysis once.
// Patterns are compared exactly (hash first); a hit deep-copies the plan into the handle.  (Debug switch PLAN_CACHE=0 disables it.)
namespace {
struct PlanCache {
    struct Entry {
        uint64_t key;
        std::string optkey;
        std::vector<int64_t> colptr, rowval;
        std::shared_ptr<const HostPlan> plan;
    };
    std::mutex mu;
    std::deque<Entry> entries;
    int64_t hits = 0, misses = 0;
    static constexpr size_t kMaxEntries = 8;
    static constexpr int64_t kMaxNnzL = 40000000;      // bigger plans (hundreds of MB of work lists) are not kept
    static PlanCache &get() { static PlanCache c; return c; }
    static bool enabled() { return debug_opts().plan_cache; }   // (read ONCE per create call below)
    static uint64_t fnv(uint64_t h, const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
        return h;
    }
    std::shared_ptr<const HostPlan> find(uint64_t key, const std::string &optkey, const KKTImage &K) {
        std::lock_guard<std::mutex> lk(mu);
        for (const Entry &e : entries)
            if (e.key == key && e.optkey == optkey && e.colptr == K.colptr && e.rowval == K.rowval) { hits++; return e.plan; }
        misses++;
        return nullptr;
    }
    void put(uint64_t key, const std::string &optkey, const KKTImage &K, const HostPlan &P) {
        if (P.nnzL > kMaxNnzL || P.ordering_used == 1) return;   // ("cone rows first" plans come with a speculative twin analysis that a cache hit would skip)
        Entry e{key, optkey, K.colptr, K.rowval, std::make_shared<const HostPlan>(P)};
        std::lock_guard<std::mutex> lk(mu);
        entries.push_front(std::move(e));
        while (entries.size() > kMaxEntries) entries.pop_back();
    }
};
}  // namespace
void plan_cache_counts(int64_t *hits, int64_t *misses) {
    PlanCache &c = PlanCache::get();
    std::lock_guard<std::mutex> lk(c.mu);
    *hits = c.hits; *misses = c.misses;
}

void init_runtime(hipkkt_solver *S) {
    HK_CHECK(hipSetDevice(S->device));
    RuntimePool &rp = RuntimePool::get();
    auto need = [&](void *p) { if (!p) throw DeviceError{"creating a stream / event / pinned buffer failed"}; return p; };
    S->stream = (hipStream_t)need(rp.stream_get(S->device, 0));
    // A third stream per handle that never carries work.  Measured (cfg 4, 6 worker processes on one GPU, round 4): with only the main
    // stream (high priority) and the second solve context's stream a process keeps 2 hardware queues and the batch runs at 560 IPM
    // iterations/s; with this idle low-priority stream next to them 1070-1130; with two idle ones 930; a normal-priority main stream
    // 820-880.  One problem alone (cfg 2a) does not notice.  How the runtime spreads streams over hardware queues is not documented:
    // the stream count is kept where the measurement put it (it was the "side" stream of rounds 1-3, dropped once and missed).
    S->idle_stream = (hipStream_t)need(rp.stream_get(S->device, 1));
    {
        // test / experiment switches (hipkkt_debug_set; the production library keeps the defaults), latched per handle
        const DebugOpts &o = debug_opts();
        S->fb_extra = o.fb_extra;            // false: every far stage applies all of its tiles in its own launch (A/B timing, bit-identity test)
        S->fb_streamed = o.fb_stream;        // false: the pivot chain of k_front_block hands over L11^-T D^-1 after all 64 pivots (round-3 form)
        S->fb_v2 = o.fb_v2 && S->fb_streamed;   // false: the first form of the front-batch kernel (front_block.hip; the round-3 chain exists only there)
#ifndef HIPKKT_TESTING
        S->fb_streamed = S->fb_v2 = true;    // (the production library does not contain the first form)
#endif
        S->accurate_threshold = o.accurate;  // largest |entry| of a wide block's explicit inverse above which its solves take a refinement step
        S->force_twin = o.force_twin;        // tests: every successful factorisation in the cheap order counts as broken down (hipkkt_refactor)
    }
    static_assert(SC_COUNT * sizeof(double) <= RuntimePool::kPinned && sizeof(RefineState) <= RuntimePool::kPinned, "pinned chunk too small");
    for (hipEvent_t *e : {&S->ev0, &S->ev1, &S->ev2, &S->ev3, &S->ev_fork, &S->ev_join}) *e = (hipEvent_t)need(rp.event_get(S->device));
    S->h_scal = (double *)need(rp.pinned_alloc(S->device));
    S->h_flags = (int *)need(rp.pinned_alloc(S->device));
    S->h_scal_red = S->h_scal + 16;     // second half of the same pinned chunk (10 doubles; RuntimePool::kPinned = 256 bytes)
    static_assert((16 + 10) * sizeof(double) <= RuntimePool::kPinned, "pinned chunk too small for the reduced-solve scalars");
    memset(S->h_scal, 0, SC_COUNT * sizeof(double));
    memset(S->h_flags, 0, FL_COUNT * sizeof(int));
    if (debug_opts().no_graph) S->use_graph = false;
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
static void order_far_stages(hipkkt_solver *S);
static void split_gather_stages(hipkkt_solver *S);
static int64_t plan_split_k(hipkkt_solver *S, std::vector<DenseGroup> &dg);
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
        S->use_persist = !debug_opts().no_persist;
        S->persist_allowed = S->use_persist;
        S->persist_retry_at = -1;
    }
    S->soc_off.clear(); S->soc_of_sparse.clear();
    S->nsoc = 0; S->soc_total = 0; S->wmax_all = 1;
    S->stage_cap = 0; S->d_stage = nullptr; S->d_stage_idx = nullptr;
    S->d_qb = S->d_res_in = S->d_res_out = S->d_res_part = nullptr;
    S->d_red = S->d_red_part = nullptr;
    S->d_sc_kind = nullptr; S->sc_cap_socdesc = S->sc_cap_psd = 0; S->sc_ready = false;   // (slab memory of an earlier set-up is gone)
    S->red_have_const = false;

    order_far_stages(S);
    split_gather_stages(S);
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
    const bool allow_narrow = true;
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
    D.slv_items = S->upload(S->slv_items);
    D.rel = S->upload(P.rel);
    D.upd_tasks = S->upload(P.upd_tasks);
    D.upd_groups = S->upload(P.upd_groups);
    std::vector<DenseGroup> dg(P.upd_groups.size());       // uploaded after the front batches are known (plan_split_k appends to it)
    {
        const bool no_full_tiles = !debug_opts().full_tiles;   // bit-identity test of the full-tile core
        for (size_t q = 0; q < dg.size(); q++) {
            const UpdGroup &G = P.upd_groups[q];
            const int t = G.tgt;
            const int rt = (int)(P.sn_rowptr[t + 1] - P.sn_rowptr[t]);
            // pad bit 0: a FULL tile (dense_tile.h dense_tile_core_full): 64 x 64, every task lands contiguously on all of it
            bool full = G.dense == 1 && rt - G.row_base >= kUpdRows && P.sn_first[t + 1] - P.sn_first[t] == 64 && !no_full_tiles;
            for (int u = G.task_begin; u < G.task_end && full; u++) {
                const UpdTask &T = P.upd_tasks[u];
                const int K = P.sn_first[T.src + 1] - P.sn_first[T.src];
                full = !(T.geom & (1 << 17)) && (T.geom & 0xFFFF) == 0 && T.nrows >= 64 && T.ncols >= 64 && K >= 8 && (K & 7) == 0;
            }
            dg[q] = {P.sn_panel[t] + G.row_base, rt, std::min(kUpdRows, rt - G.row_base), P.sn_first[t + 1] - P.sn_first[t],
                     G.task_begin, G.task_end, full ? 1 : 0};
        }
    }
    // (never empty: the dense tile core requests "the next task's map" unconditionally and reads map 0 for tasks without one)
    D.upd_tmap = P.upd_tmap.size() >= 128 ? S->upload(P.upd_tmap) : S->upload(std::vector<int16_t>(128, (int16_t)-1));
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
        S->gath_heavy_split.assign(std::max(P.nlevels, 1), 0);
        for (int l = 0; l < P.nlevels; l++) {
            const int64_t e_split = S->gath_split[l] >= 0 ? P.gath_stage_ptr[l] + S->gath_split[l] : P.gath_stage_ptr[l + 1];
            for (int64_t e = P.gath_stage_ptr[l]; e < P.gath_stage_ptr[l + 1]; e++)
                if (P.gath_pptr[e + 1] - P.gath_pptr[e] > kGathHeavy) {
                    heavy.push_back(e);
                    if (e < e_split) S->gath_heavy_split[l]++;
                }
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
    {
        // dense triangles of K (symbolic.h HostPlan::dtri): one workgroup of k_spmv_dense_tri per 64 rows of a triangle
        std::vector<DenseTriStrip> strips;
        for (const DenseTri &T : P.dtri)
            for (int i0 = 0; i0 < T.d; i0 += 64) strips.push_back(DenseTriStrip{T.c0, T.d, i0, 0, T.col0});
        D.dtri_strips = S->upload(strips);
        D.n_dtri_strips = (int)strips.size();
        D.dtri_col = S->upload(P.dtri_col);
        D.dense_acc = nullptr;                 // per solve context (below)
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
        D.seg_ticket = 3;                                  // bit 0: forward sweep, bit 1: backward sweep take their items by atomic ticket
        D.spin_limit = debug_opts().spin_limit >= 0 ? (unsigned)debug_opts().spin_limit : (1u << 20);   // tests force a sweep time-out with a tiny bound
#ifdef HIPKKT_TESTING
        D.dbg = debug_opts().debug_flags;                  // timing experiments only: results are WRONG when set (device_plan.h DevPlan::dbg)
        if (D.dbg) fprintf(stderr, "hipkkt: DEBUG_FLAGS = %d: timing experiment, the results of this handle are WRONG\n", D.dbg);
#else
        D.dbg = 0;
#endif
        D.sn_polish = S->dalloc<int>((size_t)std::max(P.nsuper, 1));
        fill_async(S->stream, D.sn_polish, 0, (size_t)std::max(P.nsuper, 1) * sizeof(int));
        D.polish_tau = S->accurate_threshold;
    }
    {
        const size_t nsync = seg_sync_ints(S->nseg, P.nsuper);
        D.seg_sync = S->dalloc<int>(nsync);
        fill_async(S->stream, D.seg_sync, 0, nsync * sizeof(int));
    }
    D.front_sync = S->dalloc<int>(std::max(P.front_sync_ints, 16));
    fill_async(S->stream, D.front_sync, 0, (size_t)std::max(P.front_sync_ints, 16) * sizeof(int));
    build_front_batches(S);
    const int64_t split_scratch = plan_split_k(S, dg);
    S->split_scratch_doubles = split_scratch;
    D.dgroups = S->upload(dg);
    D.kval = S->upload(S->img.nzval);
    D.Lx = S->dalloc<double>(P.panel_doubles + split_scratch);
    if (split_scratch) fill_async(S->stream, D.Lx + P.panel_doubles, 0, (size_t)split_scratch * sizeof(double));
    D.Ldiag = S->dalloc<double>(P.diag_doubles);
    D.Linv = S->dalloc<double>(P.diag_doubles);
    D.LinvT = S->dalloc<double>(P.diag_doubles);
    D.LT = S->dalloc<double>(P.lt_off[P.nsuper]);
    D.SbInv = S->dalloc<double>(P.sbinv_doubles);
    fill_async(S->stream, D.SbInv, 0, (size_t)std::max<int64_t>(P.sbinv_doubles, 1) * sizeof(double));
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
        if (D.n_dtri_strips > 0) {
            C.dp.dense_acc = S->dalloc<double>(N);     // rows outside the triangles stay 0 for ever, the others are rewritten by every SpMV
            fill_async(S->stream, C.dp.dense_acc, 0, (size_t)std::max(N, 1) * sizeof(double));
        }
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

// Far stage of every front batch whose successor (the next batch of the same front) follows at once: its dense tiles are reordered
// [columns of the next batch | rest] (stable).  The next k_front_block launch needs only the first part before it starts; tiles from
// the END of the rest may ride in that launch as extra workgroups (hipkkt_factor.cpp fb_extra_tiles_of_stage).  Runs BEFORE the
// work lists go to the device: it reorders the dense groups of those stages inside S->plan.
static void order_far_stages(hipkkt_solver *S) {
    HostPlan &Pm = S->plan;
    const HostPlan &P = S->plan;
    const auto hb = front_batches(P, S->plan_opts.update_policy, kFbMax);
    const size_t nbh = hb.size();
    S->next_batch.assign(nbh, hipkkt_solver::NextBatch());
    if (!S->fb_extra || !debug_opts().front_block) return;
    std::vector<std::vector<int>> panel_batch(P.fronts.size());
    for (size_t fi = 0; fi < P.fronts.size(); fi++) panel_batch[fi].assign((size_t)P.fronts[fi].np, -1);
    for (size_t q = 0; q < nbh; q++)
        for (int t = 0; t < hb[q].nb; t++) panel_batch[(size_t)hb[q].front][(size_t)(hb[q].p0 + t)] = (int)q;
    for (size_t b = 0; b < nbh; b++) {
        const bool next = b + 1 < nbh && hb[b + 1].front == hb[b].front && hb[b + 1].p0 == hb[b].p0 + hb[b].nb;
        hipkkt_solver::NextBatch &A = S->next_batch[b];
        const int l = hb[b].level_last, g0 = P.upd_stage_ptr[l], nd = P.upd_stage_ndense[l];
        const FrontDesc &F = P.fronts[(size_t)hb[b].front];
        auto near = [&](const UpdGroup &G) {
            if (!next || P.sn_front[G.tgt] != hb[b].front) return false;
            return panel_batch[(size_t)hb[b].front][(size_t)(G.tgt - F.s0)] == (int)b + 1;
        };
        auto gb = Pm.upd_groups.begin() + g0, ge = gb + nd;
        A.ncrit = (int)(std::stable_partition(gb, ge, near) - gb);
        A.has_next = next;
        A.next_blk = next ? (P.front_panels[P.fronts[(size_t)hb[b + 1].front].fp_off + hb[b + 1].p0].r + 63) / 64 : 0;
    }
}

// The per-entry gather of the bottom update batch is one long launch (cfg 2a: 4.0e6 target entries, 355 us) in front of a string of
// small, dependency-bound launches (the levels of the next batch: 13 launches, ~200 us, a few hundred workgroups each) -- and 97 % of
// its entries land in panels that nothing reads or writes before that next batch's FAR stage.  A batch-end stage with at least
// kGatherSplitMin (10^6) such entries is reordered [targets up to the next batch's last level | targets beyond]; the second part runs on the
// side stream next to the next batch's levels and is joined before its far stage (hipkkt_factor.cpp).  Why that is safe: with the
// batched schedule (symbolic.cpp, stage = min(level(t) - 1, batch end of the source)) every stage s strictly inside the next batch
// holds only targets of level s + 1, and a level's panel kernels touch their own panels only.  Entries keep their pair lists and
// their order inside each part: same arithmetic, bit for bit.
constexpr int64_t kGatherSplitMin = 1000000;   // (200 000 at first: a batch of small problems, six processes per GPU, lost 20 % of its rate
                                               // to the parallel branch in the factorisation graphs of the problems that qualified -- cfg 4,
                                               // 1070 -> 860 IPM iterations/s; a launch of a few hundred microseconds is what is worth hiding)
// ORDER OF THE ENTRIES (round 6).  The plan lists a stage's entries in target order (tile, column, row): consecutive threads own
// consecutive rows of a target column -- and take their operands from whatever small source panel put something there: every lane of
// a wavefront reads another cache line (cfg 2a: 4.0e6 entries x 28 operand loads, 355 us, bound by the rate at which the L1 looks up
// divergent lines).  The kernel does not care in which order it meets the entries (each owns its target), so a stage of at least
// kGatherSortMin entries is re-sorted by the SOURCE of each entry's first pair (stable counting sort: within a source the target
// order remains, i.e. column by column with the source's rows ascending): the lanes of a wavefront then read consecutive rows of one
// source column, the column operand and the pivot are one address for all of them, and only the single read-modify-write of the
// target scatters.  Same pair lists, same arithmetic per entry: bit-identical results.
constexpr int64_t kGatherSortMin = 50000;
static void split_gather_stages(hipkkt_solver *S) {
    HostPlan &P = S->plan;
    S->gath_split.assign(std::max(P.nlevels, 1), -1);
    const int B = P.update_batch_used;
    const bool may_split = S->plan_opts.update_policy == 2 && B >= 2 && debug_opts().gather_overlap;
    for (int l = 0; l < P.nlevels; l++) {
        const int64_t e0 = P.gath_stage_ptr[l], e1 = P.gath_stage_ptr[l + 1], ne = e1 - e0;
        if (ne < kGatherSortMin || !debug_opts().gather_sort) continue;
        // part of every entry: 1 = deferred to the side stream
        std::vector<char> far((size_t)ne, 0);
        int64_t nfar = 0;
        if (may_split && ne >= kGatherSplitMin && l % B == B - 1 && l + B < P.nlevels) {
            for (int64_t e = e0; e < e1; e++) {
                const int t = (int)(std::upper_bound(P.sn_panel.begin(), P.sn_panel.begin() + P.nsuper, P.gath_tgt[e]) - P.sn_panel.begin()) - 1;
                far[(size_t)(e - e0)] = P.sn_level[t] > l + B;
                nfar += far[(size_t)(e - e0)];
            }
            if (nfar < kGatherSplitMin) { std::fill(far.begin(), far.end(), 0); nfar = 0; }
        }
        // stable counting sort by (part, source of the first pair).  [Measured and rejected, round 6: keys that also keep the
        // entries of a target slab (2^12 .. 2^24 doubles) or of a 64-byte target line together: 1 - 2 % slower on cfg 2a.]
        const size_t nkeys = (size_t)2 * (size_t)P.nsuper;
        std::vector<int64_t> cnt(nkeys + 1, 0);
        auto key = [&](int64_t e) { return (size_t)far[(size_t)(e - e0)] * (size_t)P.nsuper + (size_t)P.gath_sn[P.gath_pptr[e]]; };
        for (int64_t e = e0; e < e1; e++) cnt[key(e) + 1]++;
        for (size_t k = 0; k < nkeys; k++) cnt[k + 1] += cnt[k];
        std::vector<int64_t> ord((size_t)ne);
        for (int64_t e = e0; e < e1; e++) ord[(size_t)cnt[key(e)]++] = e;
        const int64_t p0 = P.gath_pptr[e0], p1 = P.gath_pptr[e1];
        std::vector<int64_t> tgt((size_t)ne), pptr((size_t)ne), src((size_t)(p1 - p0));
        std::vector<int32_t> dj((size_t)(p1 - p0)), sn((size_t)(p1 - p0));
        int64_t w = 0;
        for (int64_t q = 0; q < ne; q++) {
            const int64_t e = ord[(size_t)q];
            tgt[(size_t)q] = P.gath_tgt[e];
            for (int64_t pq = P.gath_pptr[e]; pq < P.gath_pptr[e + 1]; pq++, w++) { src[(size_t)w] = P.gath_src[pq]; dj[(size_t)w] = P.gath_dj[pq]; sn[(size_t)w] = P.gath_sn[pq]; }
            pptr[(size_t)q] = p0 + w;
        }
        std::copy(tgt.begin(), tgt.end(), P.gath_tgt.begin() + e0);
        std::copy(pptr.begin(), pptr.end(), P.gath_pptr.begin() + e0 + 1);
        std::copy(src.begin(), src.end(), P.gath_src.begin() + p0);
        std::copy(dj.begin(), dj.end(), P.gath_dj.begin() + p0);
        std::copy(sn.begin(), sn.end(), P.gath_sn.begin() + p0);
        if (nfar > 0) S->gath_split[l] = ne - nfar;
        if (verbose()) fprintf(stderr, "hipkkt: gather stage %d: %lld entries sorted by source, %lld of them deferred to the side stream (targets beyond level %d)\n", l, (long long)ne, (long long)nfar, l + B);
    }
}

// Split-K (kernels.hip k_split_reduce).  A dense launch lasts as long as its longest tile, and a tile's time is its number of
// contributions: in cfg 5 the 136 tiles of the variables' block receive 20 cones x 4 panels = 80+ contributions per update batch next
// to thousands of tiles with 4-10 (measured: a 1788-tile launch took 809 us, of which the other tiles need ~100).  Tiles with at
// least twice the stage's typical number of contributions are cut into chunks of about that size, one wavefront per chunk; the
// original group record is switched off (wt = 0: dense_tile returns at once), the chunk records are appended to `dg`, partial
// tiles live behind the panels in Lx.  Stages inside the front batches are left alone (their tile order is significant).  Returns the
// scratch doubles needed.
static int64_t plan_split_k(hipkkt_solver *S, std::vector<DenseGroup> &dg) {
    const HostPlan &P = S->plan;
    S->split_group_begin.assign(std::max(P.nlevels, 1), 0);
    S->split_group_count.assign(std::max(P.nlevels, 1), 0);
    S->split_rec_ptr.assign(P.nlevels + 1, 0);
    std::vector<SplitRec> recs;
    int64_t scratch_max = 0;
    if (!debug_opts().split_k) { S->d_split_recs = S->upload(recs); return 0; }   // (A/B timing, comparison test)
    for (int l = 0; l < P.nlevels; l++) {
        S->split_rec_ptr[l + 1] = S->split_rec_ptr[l];
        const int g0 = P.upd_stage_ptr[l], nd = P.upd_stage_ndense[l];
        if (nd == 0 || S->lvl_fb[l] != -1) continue;
        int64_t ntasks = 0;
        for (int g = g0; g < g0 + nd; g++) ntasks += dg[g].task_end - dg[g].task_begin;
        // chunk size = the stage's typical tile (average contributions per tile, at least 4).  Only launches of the one-wavefront-
        // per-tile kernel qualify (<= 384 tiles go to the strip kernel, 8 wavefronts per tile already), and only when the longest
        // tile outlasts two typical ones by >= 6 contributions (~40 us at 64 columns each): the chunk launch and the reduction
        // cost ~30 us of their own (measured on cfg 1 / 2a / 2b / 3, where a looser rule split a few tiles and lost 0.03-0.05 ms)
        const int chunk = std::max(4, (int)((ntasks + nd - 1) / nd));
        int max_nt = 0;
        for (int g = g0; g < g0 + nd; g++) max_nt = std::max(max_nt, dg[g].task_end - dg[g].task_begin);
        if (nd <= 384 || max_nt < 2 * chunk + 6) continue;
        int64_t scratch = 0;
        S->split_group_begin[l] = (int)dg.size();
        for (int g = g0; g < g0 + nd; g++) {
            const int nt = dg[g].task_end - dg[g].task_begin;
            const int parts = std::min(16, nt / chunk);
            if (nt < 2 * chunk || parts < 2) continue;
            recs.push_back({dg[g].tile_off, P.panel_doubles + scratch, dg[g].rt, dg[g].nrt, dg[g].wt, parts});
            for (int q = 0; q < parts; q++) {
                const int tb = dg[g].task_begin + (int)((int64_t)nt * q / parts), te = dg[g].task_begin + (int)((int64_t)nt * (q + 1) / parts);
                dg.push_back({P.panel_doubles + scratch, 64, dg[g].nrt, dg[g].wt, tb, te, dg[g].pad});
                scratch += 4096;
            }
            dg[g].wt = 0;                                   // switched off
        }
        S->split_group_count[l] = (int)dg.size() - S->split_group_begin[l];
        S->split_rec_ptr[l + 1] = (int)recs.size();
        scratch_max = std::max(scratch_max, scratch);
    }
    S->d_split_recs = S->upload(recs);
    if (verbose() && !recs.empty())
        fprintf(stderr, "hipkkt: split-K: %zu target tiles cut into %zu chunks, %.1f MB of partial tiles\n", recs.size(), dg.size() - P.upd_groups.size(), scratch_max * 8e-6);
    return scratch_max;
}

// One launch of k_front_block per qualifying update batch of a front (symbolic.cpp front_batches).  HIPKKT_FRONT_BLOCK=0: never.
static void build_front_batches(hipkkt_solver *S) {
    const HostPlan &P = S->plan;
    S->fbatches.clear(); S->fb_last_level.clear();
    S->lvl_fb.assign(std::max(P.nlevels, 1), -1);
    if (!debug_opts().front_block) S->use_front_block = false;
    if (S->use_front_block)
        for (const FrontBatchHost &H : front_batches(P, S->plan_opts.update_policy, kFbMax)) {
            const FrontDesc &F = P.fronts[H.front];
            FrontBatch B;
            B.fp_off = F.fp_off + H.p0;
            B.nb = H.nb;
            B.r0 = P.front_panels[F.fp_off + H.p0].r;
            B.nblk = (B.r0 + 63) / 64;
            B.i_base = 0; B.i_end = B.nblk; B.tick = 0; B.pad = 0; B.x_begin = 0; B.x_count = 0;
            B.sync_off = 128 * (int)S->fbatches.size();
            B.scratch_off = kFbScratch * (int64_t)S->fbatches.size();
            B.stream_off = kFbStream * (int64_t)S->fbatches.size();
            S->lvl_fb[H.level_first] = (int)S->fbatches.size();
            for (int l = H.level_first + 1; l <= H.level_last; l++) S->lvl_fb[l] = -2;
            S->fbatches.push_back(B);
            S->fb_last_level.push_back(H.level_last);
        }
    if (verbose()) fprintf(stderr, "hipkkt: %zu front batch(es) factored by one launch each (fronts %zu, update batch %d)\n", S->fbatches.size(), P.fronts.size(), P.update_batch_used);
    const size_t nb_ = std::max<size_t>(S->fbatches.size(), 1);
    S->d_fb_sync = S->dalloc<int>(128 * nb_);
    S->d_fb_scratch = S->dalloc<double>((size_t)kFbScratch * nb_);
    fill_async(S->stream, S->d_fb_sync, 0, 128 * nb_ * sizeof(int));
    S->fb_stream_doubles = (S->fb_streamed || S->fb_v2) ? (int64_t)kFbStream * (int64_t)nb_ : 0;
    S->d_fb_stream = S->dalloc<double>((size_t)std::max<int64_t>(S->fb_stream_doubles, 1));
    if (getenv("HIPKKT_FB_TRACE")) {
        S->d_fb_trace = (long long *)S->dalloc<double>(nb_ * 128);
        fill_async(S->stream, S->d_fb_trace, 0, nb_ * 128 * sizeof(double));
    }
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
        // debug switch DENSE_TRI=0: dense Hs triangles stay in the symmetric view (A/B timing, parity test of k_spmv_dense_tri)
        po.dense_tri_first_col = (S->l1 && debug_opts().dense_tri) ? (int)S->img.n : -1;
    }
    {
        if (debug_opts().front_block_min_rows >= 0) po.front_block_min_width = debug_opts().front_block_min_rows;   // tests: 0, so that small fronts take the front-batch kernel too
        if (debug_opts().superhop >= 0) po.superhop = debug_opts().superhop;   // 0: one hop per panel in the front sweeps; N: fronts of >= N panels go super-block by super-block
    }
    {
        if (debug_opts().ordering_amd) po.n_hold = 0;   // minimum degree on K only
    }
    {
        if (debug_opts().no_front) po.front_min_panels = 0;
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
        // thread is cancelled at its next phase boundary.
        if (S->l1)
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
                S->twin_go = std::make_shared<std::atomic<int>>(0);
                std::shared_ptr<std::atomic<int>> go = S->twin_go;
                S->twin_future = std::async(std::launch::async, [Tp, pv, po2, go]() {
                    std::string err = build_plan((int)Tp->img.N, Tp->img.colptr.data(), Tp->img.rowval.data(), pv.data(), po2, Tp->plan);
                    if (!err.empty()) return err;
                    // Round 6: the twin's device residency too (its work lists are hundreds of MB on the problems that take this path:
                    // 0.32 - 0.40 s of set-up on cfg 5, paid until now by the FIRST factorisation that breaks down -- half of that
                    // problem's whole IPM time).  Here it overlaps the owner's own analysis and set-up; a cancelled speculation
                    // (the cheap order was not chosen after all) never gets here.
                    // ... but only once the owner has settled on the cheap order (small problems finish this analysis before the
                    // owner has compared the two orders: without the wait every batch problem of cfg 4 set up a twin it never used)
                    // ... and only for a twin whose set-up is worth hiding (>= 2e7 entries of L: tenths of a second).  A small twin is set
                    // up in milliseconds when it is needed, and until then costs its process three more HIP streams -- which is what
                    // a batch of small problems is most sensitive to (cfg 4 fell from 1070 to 890 IPM iterations/s
                    // when every batch problem in the cheap order kept a resident twin).
                    if (Tp->plan.nnzL < 20000000) return err;
                    while (go->load(std::memory_order_acquire) == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
                    if (go->load(std::memory_order_acquire) != 1 || (po2.cancel && po2.cancel->load(std::memory_order_relaxed))) return std::string("cancelled");
                    try {
                        init_runtime(Tp);
                        setup_device(Tp);
                        Tp->device_ready = true;
                    } catch (const DeviceError &e) {
                        return std::string("twin device set-up: ") + e.msg;       // (the failing factorisation will try again, synchronously)
                    } catch (const std::bad_alloc &) {
                        return std::string("twin device set-up: out of memory");
                    }
                    return err;
                });
            };
    }
    const auto t_a = std::chrono::steady_clock::now();
    // plan cache: same pattern, same options => the same plan
    uint64_t ckey = 0;
    std::string optkey;
    std::shared_ptr<const HostPlan> cached;
    const bool plan_cache_on = PlanCache::enabled();
    if (plan_cache_on) {
        char buf[256];
        snprintf(buf, sizeof buf, "%d|%d|%d|%d|%.17g|%.17g|%d|%d|%d|%d|%d|%d|%d|%d|%d", po.max_width, (int)po.relax, po.update_policy, po.update_batch,
                 po.amd_dense_scale, po.dense_min_cover, po.n_hold, po.front_block_min_width, po.front_min_panels, po.superhop, po.nd_mode,
                 po.nd_leaf, uperm ? 1 : 0, po.dense_tri_first_col, po.dense_tri_min_dim);
        optkey = buf;
        ckey = PlanCache::fnv(1469598103934665603ull, S->img.colptr.data(), S->img.colptr.size() * sizeof(int64_t));
        ckey = PlanCache::fnv(ckey, S->img.rowval.data(), S->img.rowval.size() * sizeof(int64_t));
        if (uperm) ckey = PlanCache::fnv(ckey, uperm, (size_t)S->img.N * sizeof(int64_t));
        cached = PlanCache::get().find(ckey, optkey, S->img);
    }
    std::string err;
    if (cached) {
        S->plan = *cached;
        S->plan.timing_note = "plan cache hit";
    } else {
        err = build_plan((int)S->img.N, S->img.colptr.data(), S->img.rowval.data(), uperm, po, S->plan);
        if (err.empty() && plan_cache_on) PlanCache::get().put(ckey, optkey, S->img, S->plan);
    }
    po.on_alternative_order = nullptr;
    if (!err.empty()) { g_create_error = err; delete S; return HIPKKT_ERR_ARGUMENT; }
    if (S->plan.ordering_used != 1 && S->twin_cancel) S->twin_cancel->store(true);   // speculative twin not needed: the thread stops at its next phase
    if (S->twin_go) S->twin_go->store((S->plan.ordering_used == 1 && err.empty()) ? 1 : 2, std::memory_order_release);   // ... or goes on to its device set-up
    S->plan_opts = po;
    const auto t_b = std::chrono::steady_clock::now();
    try {
        if (!S->runtime_ready) init_runtime(S);
        S->runtime_ready = true;
        setup_device(S);
        if (verbose())
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

}  // namespace hipkkt_host
