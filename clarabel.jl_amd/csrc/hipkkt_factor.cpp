// The factorisation: launch sequence (levels, front batches, Schur updates), its hipGraph, hipkkt_refactor and the
// robust-order twin (see hipkkt_internal.h for the file map).
#include "hipkkt_internal.h"

using namespace hipkkt;
using namespace hipkkt_host;

namespace hipkkt_host {

// ---- enqueue helpers (no synchronisation inside; capturable) ---------------------------------

// narrow levels (w <= 8): LDS-resident kernel; wide panels: the register-resident 8-wave kernel
void enqueue_factor_level(hipkkt_solver *S, int l) {
    const HostPlan &P = S->plan;
    const int n = P.fac_lvl_ptr[l + 1] - P.fac_lvl_ptr[l];
    if (P.fac_lvl_maxw[l] <= 8)
        launch_factor_level(S->stream, S->dp, P.fac_lvl_ptr[l], n, P.fac_lvl_maxw[l], S->opts.dynamic_reg_eps,
                            S->opts.dynamic_reg_delta);
    else
        launch_factor_panel(S->stream, S->dp, P.fac_lvl_ptr[l], n, S->opts.dynamic_reg_eps, S->opts.dynamic_reg_delta);
}

// one update batch of a front (front_block2.hip; the first form of the kernel, front_block.hip, exists in the testing build only)
static void enqueue_front_batch(hipkkt_solver *S, const FrontBatch &B) {
#ifdef HIPKKT_TESTING
    if (!S->fb_v2) {
        launch_front_block(S->stream, S->dp, B, S->d_fb_sync, S->d_fb_scratch, S->d_fb_stream, S->opts.dynamic_reg_eps,
                           S->opts.dynamic_reg_delta, S->fb_streamed, S->d_fb_trace);
        return;
    }
#endif
    launch_front_block2(S->stream, S->dp, B, S->d_fb_sync, S->d_fb_scratch, S->d_fb_stream, S->opts.dynamic_reg_eps,
                        S->opts.dynamic_reg_delta, S->d_fb_trace);
}

// split-K part of a stage's dense updates (hipkkt_setup.cpp plan_split_k): the chunk tiles, then their fixed-order reduction
void enqueue_split_k(hipkkt_solver *S, int l) {
    if (S->split_group_count[l] <= 0) return;
    launch_update_dense(S->stream, S->dp, S->split_group_begin[l], S->split_group_count[l], false);
    launch_split_reduce(S->stream, S->dp, S->d_split_recs + S->split_rec_ptr[l], S->split_rec_ptr[l + 1] - S->split_rec_ptr[l]);
}

// The stage's per-entry gather lists; a split stage (hipkkt_setup.cpp split_gather_stages) runs its deferred part on the side stream
void enqueue_gather(hipkkt_solver *S, int l) {
    const HostPlan &P = S->plan;
    const int64_t e0 = P.gath_stage_ptr[l], n = P.gath_stage_ptr[l + 1] - e0, h0 = S->gath_heavy_ptr[l], nh = S->gath_heavy_ptr[l + 1] - h0;
    const int64_t head = S->gath_split[l];
    if (head < 0 || S->side_join_level >= 0) { launch_update_gather(S->stream, S->dp, e0, n, h0, nh); return; }
    const int64_t hh = S->gath_heavy_split[l];
    launch_update_gather(S->stream, S->dp, e0, head, h0, hh);
    HK_CHECK(hipEventRecord(S->ev_fork, S->stream));
    HK_CHECK(hipStreamWaitEvent(S->idle_stream, S->ev_fork, 0));
    launch_update_gather(S->idle_stream, S->dp, e0 + head, n - head, h0 + hh, nh - hh, debug_opts().gather_side_blocks);
    HK_CHECK(hipEventRecord(S->ev_join, S->idle_stream));
    S->side_join_level = l + P.update_batch_used;
}
// ... which everything from the far stage of the next batch on must wait for
void join_side(hipkkt_solver *S, int l) {
    if (S->side_join_level < 0 || l < S->side_join_level) return;
    HK_CHECK(hipStreamWaitEvent(S->stream, S->ev_join, 0));
    S->side_join_level = -1;
}

// Schur-complement updates applied after level l is factored: dense register tiles (matrix cores),
// per-entry gather lists (tiny scattered contributions), relative-index scatter (whatever is left).
// The three kinds own disjoint target tiles, so their order inside a stage is immaterial.
// dense_skip_tail: the last tiles of the stage's dense list ride in the next k_front_block launch instead.
void enqueue_updates(hipkkt_solver *S, int l, int dense_skip_tail = 0) {
    const HostPlan &P = S->plan;
    hipStream_t st = S->stream;
    const int g0 = P.upd_stage_ptr[l], nd = P.upd_stage_ndense[l], ng = P.upd_stage_ngather[l];
    join_side(S, l);
    launch_update_dense(st, S->dp, g0, nd - dense_skip_tail, nd > 0 && P.upd_stage_flops_dense[l] >= 1.5e6 * nd);
    enqueue_split_k(S, l);
    enqueue_gather(S, l);
    launch_update_stage(st, S->dp, g0 + nd + ng, P.upd_stage_ptr[l + 1] - g0 - nd - ng);
}

// How many of a front batch's far tiles ride in the next k_front_block launch (enqueue_factor; HIPKKT_FB_EXTRA=0: none).  That launch
// has `next_blk` workgroups of its own, one per compute unit (100 KB of LDS), and lasts ~112 us: every other compute unit can take one
// extra workgroup = four wavefronts, each alone on its SIMD, with one tile (launch +10 us) or two tiles one after the other (+28 us)
// per wavefront.  The stage's own launch then holds M = nd - r tiles; its cost is a step function of M (rounds of one tile per SIMD;
// measured on MI355X at K = 320: 1024 tiles 77 us, 2048 130, 3072 194; a remainder of <= 256 tiles as strips +13, <= 512 as halves
// +20; <= 384 tiles in all: 20 us + 0.065 per tile; <= 768: 66 us).  Choose tiles per wavefront and M to minimise the sum; M sits
// on a step (what would not move it below the next one is given back).  Never tiles of the next batch's columns (the first `ncrit`
// groups of the stage).
static double dense_stage_cost_us(int m) {
    if (m <= 0) return 0.0;
    if (m <= 384) return 20.0 + 0.065 * m;
    if (m <= 768) return 66.0;
    const int k = m / 1024, rem = m % 1024;
    const double base = k == 0 ? 0.0 : k == 1 ? 77.0 : 65.0 * k;
    if (rem == 0) return base;
    if (rem <= 256) return base + 13.0;
    if (rem <= 512) return base + 20.0;
    return k == 0 ? 77.0 : 65.0 * (k + 1);
}
int fb_extra_tiles_of_stage(int nd, int ncrit, int next_blk, int *per_wave_out) {
    constexpr int cus = 254, rmin = 64;
    // what the panel launch gains in duration (us) with one / two tiles per wavefront (round 5: one tile per wavefront since the
    // panel chain got shorter than two tile times); experiments: debug switch FB_EXTRA_PW=<max tiles per wavefront>,<penalty 1>,<penalty 2>
    const DebugOpts &dbo = debug_opts();
    const int pw_max = dbo.fb_extra_pw;
    const double pen[3] = {0.0, dbo.fb_pen1, dbo.fb_pen2};
    double best = dense_stage_cost_us(nd);
    int best_r = 0, best_pw = 1;
    for (int pw = 1; pw <= pw_max; pw++) {
        const int rmax = std::min(nd - ncrit, std::max(0, 4 * pw * (cus - next_blk)));
        if (rmax < rmin) continue;
        const int mmin = nd - rmax;
        int m;
        if (mmin <= 384) m = mmin;
        else if (mmin <= 768) m = 768;
        else {
            const int k = mmin / 1024, rem = mmin % 1024;
            m = 1024 * k + (rem == 0 ? 0 : rem <= 256 ? 256 : rem <= 512 ? 512 : 1024);
        }
        m = std::max(std::min(m, nd), ncrit);
        const int r = nd - m;
        if (r < rmin) continue;
        const double c = dense_stage_cost_us(m) + pen[pw];
        if (c < best - 1.0) { best = c; best_r = r; best_pw = pw; }
    }
    *per_wave_out = best_pw;
    return best_r;
}

void enqueue_factor(hipkkt_solver *S, int static_enable, double eps_const, double eps_prop) {
    const HostPlan &P = S->plan;
    hipStream_t st = S->stream;
    S->side_join_level = -1;
    launch_zero_words(st, S->dp.scal, 2);                 // SC_MAXDIAG  (kernels.hip: why not hipMemsetAsync)
    launch_zero_words(st, S->dp.flags, FL_COUNT);
    launch_maxabs_gather(st, S->dp.kval, S->d_diag_full, S->N, (unsigned long long *)S->dp.scal + SC_MAXDIAG);
    HK_CHECK(hipMemsetAsync(S->dp.Lx, 0, (size_t)(P.panel_doubles + S->split_scratch_doubles) * sizeof(double), st));   // (incl. the split-K partial tiles: a factorisation that aborted between a chunk launch and its reduce must not leave sums behind)
    launch_init_panels(st, S->dp, S->nnzK, static_enable, eps_const, eps_prop);
    const bool fb = S->use_front_block && !S->fbatches.empty();
    if (fb) launch_fb_reset(st, S->d_fb_sync, 128 * (int)S->fbatches.size(), S->d_fb_stream, S->fb_stream_doubles);
    int cur_bi = -1;
    int extra_begin = 0, extra_count = 0, extra_pw = 1;   // dense tiles handed to the next k_front_block launch, tiles per wavefront there
    for (int l = 0; l < P.nlevels; l++) {
        if (fb && S->lvl_fb[l] != -1) {
            // a front's update batch: one launch for its panels and their just-in-time updates, then the batch's far stage
            if (S->lvl_fb[l] >= 0) {
                cur_bi = S->lvl_fb[l];
                join_side(S, S->fb_last_level[(size_t)cur_bi]);       // (a batch that reaches beyond the join level touches deferred targets)
                FrontBatch B = S->fbatches[(size_t)cur_bi];
                B.x_begin = extra_begin; B.x_count = extra_count; B.pad = extra_pw;     // far tiles of the stage before (fb_extra_tiles_of_stage)
                extra_begin = extra_count = 0;
                enqueue_front_batch(S, B);
            }
            const bool last = l + 1 >= P.nlevels || S->lvl_fb[l + 1] != -2;
            if (!last) continue;                          // the stages inside the batch are applied by the kernel itself
        } else {
            enqueue_factor_level(S, l);
        }
        int skip = 0;
        if (fb && S->fb_extra && cur_bi >= 0 && S->lvl_fb[l] != -1 && S->next_batch[(size_t)cur_bi].has_next) {
            // The far stage of a front batch runs in rounds of 1024 tiles (one per SIMD; 2048 with two co-resident wavefronts); what is
            // left after the whole rounds keeps a fraction of the device busy for a full tile time.  Those tiles -- taken from the END
            // of the [columns of the next batch | rest] order, so the next panel kernel does not need them -- ride in the next
            // k_front_block launch as extra workgroups on the compute units it leaves idle.
            const hipkkt_solver::NextBatch &A = S->next_batch[(size_t)cur_bi];
            const int r = fb_extra_tiles_of_stage(P.upd_stage_ndense[l], A.ncrit, A.next_blk, &extra_pw);
            if (r > 0 && P.upd_stage_flops_dense[l] >= 1.5e6 * P.upd_stage_ndense[l]) {
                skip = r;
                extra_begin = P.upd_stage_ptr[l] + P.upd_stage_ndense[l] - r;
                extra_count = r;
            }
        }
        enqueue_updates(S, l, skip);
    }
    join_side(S, P.nlevels);
    launch_invert_diag(st, S->dp, S->inv_nsmall, S->inv_wsmall, S->inv_nwide);
    for (const FrontDesc &F : P.fronts) launch_invert_super(st, S->dp, F);   // super-block inverses for the front sweeps
}

}  // namespace hipkkt_host

extern "C" {

int32_t hipkkt_debug_extra_tiles(int32_t nd, int32_t ncrit, int32_t next_blk, int32_t *per_wave) {
    int pw = 1;
    const int r = fb_extra_tiles_of_stage(nd, ncrit, next_blk, &pw);
    if (per_wave) *per_wave = pw;
    return r;
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
    // tests: HIPKKT_FORCE_TWIN=1 treats every successful factorisation in the cheap order as broken down, so that the robust-order
    // twin can be checked in ITS permutation on problems a scalar CPU factorisation finishes in seconds
    // (read once per handle in init_runtime: no getenv on the refactor path)
    if (h && h->force_twin && rc == HIPKKT_OK && h && h->plan.ordering_used == 1) rc = HIPKKT_NUMERICAL_FAILURE;
    if (rc != HIPKKT_NUMERICAL_FAILURE || !h || h->plan.ordering_used != 1) return rc;
    hipkkt_solver *S = h;
    try {
        if (hipSetDevice(S->device) != hipSuccess) return rc;
        if (!S->fallback) {
            // built in a local owner: a set-up that throws half-way (e.g. device OOM) must not leave a twin with null
            // streams / device pointers behind -- the next failing factorisation then simply tries again
            std::unique_ptr<hipkkt_solver> T;
            const auto t_a = std::chrono::steady_clock::now();
            if (S->twin_future.valid()) {               // analysed (and made device-resident) ahead on a host thread (finish_create): wait for it
                const std::string err = S->twin_future.get();
                T = std::move(S->twin_pending);
                if (!err.empty() && err.rfind("twin device set-up", 0) == 0) {
                    // the analysis is fine, only its residency failed on the thread: release what it got and repeat that part here
                    std::unique_ptr<hipkkt_solver> T2(new hipkkt_solver());
                    T2->device = T->device; T2->opts = T->opts; T2->l1 = T->l1; T2->img = std::move(T->img); T2->plan = std::move(T->plan);
                    T2->plan_opts = T->plan_opts;
                    T = std::move(T2);
                    T->plan_opts.cancel = nullptr;
                } else if (!err.empty()) T.reset();
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
            if (!T->device_ready) {
                init_runtime(T.get());
                setup_device(T.get());
            }
            if (verbose())
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
    S->red_have_const = false;   // the resident constant-rhs solution (N2) belongs to the previous factorisation
    HK_CHECK(hipEventRecord(S->ev0, S->stream));
    if (S->profiling) {
        // eager, with the dense-update launches timed separately (adds event overhead)
        const HostPlan &P = S->plan;
        hipStream_t st = S->stream;
        S->side_join_level = -1;
        launch_zero_words(st, S->dp.scal, 2);
        launch_zero_words(st, S->dp.flags, FL_COUNT);
        launch_maxabs_gather(st, S->dp.kval, S->d_diag_full, S->N, (unsigned long long *)S->dp.scal + SC_MAXDIAG);
        HK_CHECK(hipMemsetAsync(S->dp.Lx, 0, (size_t)(P.panel_doubles + S->split_scratch_doubles) * sizeof(double), st));   // (incl. the split-K partial tiles: a factorisation that aborted between a chunk launch and its reduce must not leave sums behind)
        launch_init_panels(st, S->dp, S->nnzK, static_reg_enable, eps_const, eps_prop);
        std::vector<hipEvent_t> evs, evd;   // evs: all update kernels of a stage; evd: its k_update_dense<4,4> launch alone
        std::vector<int> evd_level;
        const bool fb = S->use_front_block && !S->fbatches.empty();
        if (fb) launch_fb_reset(st, S->d_fb_sync, 128 * (int)S->fbatches.size(), S->d_fb_stream, S->fb_stream_doubles);
        std::vector<hipEvent_t> evf;        // around every k_front_block launch
        int fb_panels = 0;
        double fb_flops = 0;                // update flops of the stages inside the batches (executed by k_front_block)
        int cur_bi = -1, extra_begin = 0, extra_count = 0, extra_pw = 1;
        std::vector<double> evd_flops;      // flops / tiles of the timed dense launches (a launch may have handed tiles to the next k_front_block)
        std::vector<int> evd_tiles;
        S->prof_extra_tiles = 0; S->prof_extra_flops = 0;
        auto dense_flops = [&](int gb, int n) {
            double f = 0;
            for (int g = gb; g < gb + n; g++)
                for (int q = P.upd_groups[g].task_begin; q < P.upd_groups[g].task_end; q++) {
                    const UpdTask &T = P.upd_tasks[q];
                    f += 2.0 * T.nrows * T.ncols * (P.sn_first[T.src + 1] - P.sn_first[T.src]);
                }
            return f;
        };
        for (int l = 0; l < P.nlevels; l++) {
            if (fb && S->lvl_fb[l] != -1) {
                if (S->lvl_fb[l] >= 0) {
                    hipEvent_t a, b;
                    HK_CHECK(hipEventCreate(&a));
                    HK_CHECK(hipEventCreate(&b));
                    HK_CHECK(hipEventRecord(a, st));
                    cur_bi = S->lvl_fb[l];
                    join_side(S, S->fb_last_level[(size_t)cur_bi]);
                    FrontBatch B = S->fbatches[(size_t)cur_bi];
                    B.x_begin = extra_begin; B.x_count = extra_count; B.pad = extra_pw;
                    extra_begin = extra_count = 0;
                    enqueue_front_batch(S, B);
                    HK_CHECK(hipEventRecord(b, st));
                    evf.push_back(a);
                    evf.push_back(b);
                    fb_panels += S->fbatches[S->lvl_fb[l]].nb;
                }
                if (l + 1 < P.nlevels && S->lvl_fb[l + 1] == -2) { fb_flops += P.upd_stage_flops_dense[l]; continue; }   // applied by the kernel
            } else {
                enqueue_factor_level(S, l);
            }
            if (P.upd_stage_ptr[l + 1] > P.upd_stage_ptr[l]) {
                hipEvent_t a, b, c2;
                HK_CHECK(hipEventCreate(&a));
                HK_CHECK(hipEventCreate(&b));
                HK_CHECK(hipEventRecord(a, st));
                const int g0 = P.upd_stage_ptr[l], nd = P.upd_stage_ndense[l], ng = P.upd_stage_ngather[l];
                join_side(S, l);
                int skip = 0;                                              // (the same rule as enqueue_factor)
                if (fb && S->fb_extra && !S->profiling_no_extra && cur_bi >= 0 && S->lvl_fb[l] != -1 && S->next_batch[(size_t)cur_bi].has_next &&
                    P.upd_stage_flops_dense[l] >= 1.5e6 * nd) {
                    skip = fb_extra_tiles_of_stage(nd, S->next_batch[(size_t)cur_bi].ncrit, S->next_batch[(size_t)cur_bi].next_blk, &extra_pw);
                    if (skip > 0) { extra_begin = g0 + nd - skip; extra_count = skip; }
                }
                const double xf = skip > 0 ? dense_flops(g0 + nd - skip, skip) : 0.0;
                S->prof_extra_tiles += skip; S->prof_extra_flops += xf;
                launch_update_dense(st, S->dp, g0, nd - skip, nd > 0 && P.upd_stage_flops_dense[l] >= 1.5e6 * nd);
                enqueue_split_k(S, l);
                if (nd - skip > 384) {   // the one-wavefront-per-tile variant (see launch_update_dense)
                    HK_CHECK(hipEventCreate(&c2));
                    HK_CHECK(hipEventRecord(c2, st));
                    evd.push_back(a);
                    evd.push_back(c2);
                    evd_level.push_back(l);
                    evd_flops.push_back(P.upd_stage_flops_dense[l] - xf);
                    evd_tiles.push_back(nd - skip);
                }
                enqueue_gather(S, l);
                launch_update_stage(st, S->dp, g0 + nd + ng, P.upd_stage_ptr[l + 1] - g0 - nd - ng);
                HK_CHECK(hipEventRecord(b, st));
                evs.push_back(a);
                evs.push_back(b);
            }
        }
        join_side(S, P.nlevels);
        launch_invert_diag(st, S->dp, S->inv_nsmall, S->inv_wsmall, S->inv_nwide);
        for (const FrontDesc &F : P.fronts) launch_invert_super(st, S->dp, F);   // super-block inverses for the front sweeps
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
            S->prof_dense4_flops += evd_flops[i / 2];
            S->prof_dense4_launches++;
            S->prof_launch_ms.push_back(ms);
            S->prof_launch_flops.push_back(evd_flops[i / 2]);
            S->prof_launch_tiles.push_back(evd_tiles[i / 2]);
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
    S->kval_event_pending = false;       // (the main stream has been synchronised: every value update before it is complete)
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
    S->last_npolish = S->h_flags[FL_NPOLISH];            // wide diagonal blocks whose solves take a refinement step (kernels.hip k_invert_diag_wide)
    if (S->last_npolish > 0) S->n_accurate_factorisations++;
    const double maxdiag = slot_value(S, SC_MAXDIAG);
    S->last_eps = static_reg_enable ? eps_const + eps_prop * maxdiag : 0.0;
    S->last_nreg = S->h_flags[FL_NREG];
    if (eps_used) *eps_used = S->last_eps;
    if (n_dynamic_reg) *n_dynamic_reg = S->last_nreg;
    return S->h_flags[FL_NONFINITE] ? HIPKKT_NUMERICAL_FAILURE : HIPKKT_OK;
    HK_LEAVE
}

}  // extern "C"
