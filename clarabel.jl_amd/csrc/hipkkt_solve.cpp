// LDL solves on a solve context, iterative refinement decided on the device, recovery from a sweep time-out, and the solve entry
// points of the C ABI (see hipkkt_internal.h for the file map).
#include "hipkkt_internal.h"

using namespace hipkkt;
using namespace hipkkt_host;

namespace hipkkt_host {

// in -> out (original ordering on both sides) on context C
void enqueue_ldl_solve(hipkkt_solver *S, SolveCtx &C, const double *in, double *out, int *zero, int nzero) {
    const HostPlan &P = S->plan;
    hipStream_t st = C.stream;
    const DevPlan &D = C.dp;
    launch_permute_in(st, in, D.perm, C.d_y, S->N, D.seg_epoch, D.seg_sync, 2 * S->nseg, zero, nzero);
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

// the downgrade to per-level kernels after a sweep time-out is temporary
void maybe_retry_persistent(hipkkt_solver *S) {
    if (!S->use_persist && S->persist_allowed && S->persist_retry_at >= 0 && S->n_ldlsolves >= S->persist_retry_at) {
        S->use_persist = true;
        S->persist_retry_at = -1;
        for (SolveCtx &C : S->ctx) C.g_ldl.valid = C.g_first.valid = C.g_step.valid = false;
    }
}

// one refinement step on the device: correction solve, candidate = iterate + correction, its residual, the decision
void enqueue_refine_step(hipkkt_solver *S, SolveCtx &C, double reltol, double abstol, int64_t max_iter, double stop_ratio) {
    hipStream_t st = C.stream;
    enqueue_ldl_solve(S, C, C.d_e, C.d_corr, (int *)(C.dp.scal + SC_NORME), 2);   // (||e|| of the candidate starts from zero)
    launch_refine_add(st, C.d_rs, C.d_x0, C.d_x1, C.d_corr, S->N);
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
// with every time-out; debug switch PERSIST_RETRY=<n>: the first interval, 0 = never) the persistent kernels are tried
// again.  Returns false when there is nothing left to fall back to.
bool recover_from_sweep_failure(hipkkt_solver *S) {
    if (!S->use_persist) return false;
    const HostPlan &P = S->plan;
    for (SolveCtx &C : S->ctx) HK_CHECK(hipStreamSynchronize(C.stream));
    S->n_sweep_timeouts++;
    {
        const int64_t first = debug_opts().persist_retry >= 0 ? debug_opts().persist_retry : 64;
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
    if (S->kval_event_pending && st != S->stream) HK_CHECK(hipStreamWaitEvent(st, S->ev3, 0));   // an asynchronous hipkkt_set_hs_dev on the main stream
    HK_CHECK(hipEventRecord(C.ev_a, st));
    C.ir_used = ir_enable != 0;
    if (ir_enable) {
        const bool same = C.g_reltol == reltol && C.g_abstol == abstol && C.g_maxit == max_iter && C.g_stop == stop_ratio;
        run_graphed(S, st, C.g_first, same, [&] {
            enqueue_ldl_solve(S, C, C.d_b, C.d_x0, (int *)(C.dp.scal + SC_NORMB), 4);       // (||b||, ||e|| start from zero)
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
    C.extra_steps = false;
    while (C.ir_used && C.h_rs->active && !sweep_failed(C)) {   // rare: more than one step needed
        run_graphed(S, st, C.g_step, true, [&] { enqueue_refine_step(S, C, C.g_reltol, C.g_abstol, C.g_maxit, C.g_stop); });
        S->n_ldlsolves += 1;
        more = true;
        readback();
    }
    C.extra_steps = more;
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

}  // namespace hipkkt_host

extern "C" {

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
    // (no host synchronisation: the solve that reads d_b runs on the same stream; rhs_dev must stay valid until that solve has returned)
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

// SURVEY section 8(f) row N2, second half: kkt_solve! (kktsystem.jl:135-215) between the caller's cone algebra and mul_Hs!.
// Both solves are started before anything is waited for; the reduction (dots, dtau, axpys) and the copies of the step are
// enqueued behind them speculatively; the host synchronises once.  Only if a solve needed more refinement steps than the one that
// is part of its graph (rare) the reduction is repeated after those steps.
static int32_t kkt_solve_reduced_impl(hipkkt_solver *S, const double *rhs_x, const double *workz, const double *var_x,
                                      const double *in_dev, const double *scal_in, int32_t const_pending, double *lhs_x,
                                      double *lhs_z, double *lhs_dev, double *scal_out, int32_t ir_enable, double reltol,
                                      double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps) {
    const int64_t n = S->img.n, m = S->img.m, p = S->img.p;
    const int nm = (int)(n + m);
    if (!S->d_red) {
        S->d_red = S->dalloc<double>(3 * (size_t)S->N + n + 16);                          // x1z1 | x2z2 | lhs | var_x | scalars
        S->d_red_part = S->dalloc<double>(8 * (size_t)residual_blocks((int)n, (int)m) + 8);
        S->red_have_const = false;
    }
    double *d_s1 = S->d_red, *d_s2 = d_s1 + S->N, *d_lhs = d_s2 + S->N, *d_xv = d_lhs + S->N, *d_sc = d_xv + n;
    if (!const_pending && !S->red_have_const) { S->err = "kkt_solve_reduced: no constant-rhs solution resident (pass const_pending = 1 after a refactorisation)"; return HIPKKT_ERR_ARGUMENT; }
    hipkkt_solver *T = solve_target(S);
    const double tau = scal_in[0], kappa = scal_in[1], rhs_tau = scal_in[2], rhs_kappa = scal_in[3];
    for (int attempt = 0; attempt < 2; attempt++) {
        maybe_retry_persistent(T);
        SolveCtx &A = T->ctx[0], &Bc = T->ctx[1];
        // right-hand sides
        if (in_dev) {
            launch_set_rhs(A.stream, A.d_b, in_dev, nm, T->N);
            HK_CHECK(hipMemcpyAsync(d_xv, in_dev + nm, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, A.stream));
        } else {
            if (n) HK_CHECK(hipMemcpyAsync(A.d_b, rhs_x, n * sizeof(double), hipMemcpyHostToDevice, A.stream));
            if (m) HK_CHECK(hipMemcpyAsync(A.d_b + n, workz, m * sizeof(double), hipMemcpyHostToDevice, A.stream));
            if (p) HK_CHECK(hipMemsetAsync(A.d_b + n + m, 0, p * sizeof(double), A.stream));
            if (n) HK_CHECK(hipMemcpyAsync(d_xv, var_x, n * sizeof(double), hipMemcpyHostToDevice, A.stream));
        }
        solve_begin(T, A, ir_enable, reltol, abstol, max_iter, stop_ratio);
        solve_copy_out_dev(T, A, d_s1, nm);
        if (const_pending) {
            launch_const_rhs(Bc.stream, Bc.d_b, S->d_qb, (int)n, nm, T->N);
            solve_begin(T, Bc, ir_enable, reltol, abstol, max_iter, stop_ratio);
            solve_copy_out_dev(T, Bc, d_s2, nm);
            HK_CHECK(hipEventRecord(S->ev2, Bc.stream));
            HK_CHECK(hipStreamWaitEvent(A.stream, S->ev2, 0));
        }
        auto reduce = [&]() {
            launch_reduced(A.stream, S->dp, d_s1, d_s2, d_xv, S->d_qb, S->d_qb + n, tau, kappa, rhs_tau, rhs_kappa, S->d_red_part, d_sc,
                           d_lhs, (int)n, (int)m);
            if (lhs_dev) HK_CHECK(hipMemcpyAsync(lhs_dev, d_lhs, (size_t)nm * sizeof(double), hipMemcpyDeviceToDevice, A.stream));
            if (lhs_x && n) HK_CHECK(hipMemcpyAsync(lhs_x, d_lhs, n * sizeof(double), hipMemcpyDeviceToHost, A.stream));
            if (lhs_z && m) HK_CHECK(hipMemcpyAsync(lhs_z, d_lhs + n, m * sizeof(double), hipMemcpyDeviceToHost, A.stream));
            HK_CHECK(hipMemcpyAsync(S->h_scal_red, d_sc, 10 * sizeof(double), hipMemcpyDeviceToHost, A.stream));
        };
        reduce();
        int64_t st1 = 0, st2 = 0;
        int32_t rc = solve_finish(T, A, &st1, d_s1, nm);
        bool redo = A.extra_steps;
        double ms = A.last_ms;
        if (const_pending) {
            const int32_t r2 = solve_finish(T, Bc, &st2, d_s2, nm);
            redo = redo || Bc.extra_steps;
            ms = std::max(ms, Bc.last_ms);
            if (r2 < 0 || (r2 > 0 && rc == HIPKKT_OK)) rc = r2 < 0 ? r2 : (rc < 0 ? rc : r2);
        }
        bool timed_out = sweep_failed(A) || (const_pending && sweep_failed(Bc));
        if (timed_out && recover_from_sweep_failure(T)) continue;   // repeat everything on the per-level kernels
        if (timed_out) return HIPKKT_ERR_DEVICE;
        if (redo && rc == HIPKKT_OK) {                              // the accepted iterates changed after the speculative reduction
            if (const_pending) { HK_CHECK(hipEventRecord(S->ev2, Bc.stream)); HK_CHECK(hipStreamWaitEvent(A.stream, S->ev2, 0)); }
            reduce();
        }
        HK_CHECK(hipStreamSynchronize(A.stream));
        T->t_last_solve = ms; T->t_acc_solve += ms; T->n_solvecalls++; T->n_rhs_solved += const_pending ? 2 : 1;
        if (T != S) { S->t_last_solve = ms; S->t_acc_solve += ms; S->n_solvecalls++; S->n_rhs_solved += const_pending ? 2 : 1; }
        if (rc == HIPKKT_OK && const_pending) S->red_have_const = true;
        if (ir_steps) { ir_steps[0] = st1; ir_steps[1] = st2; }
        if (scal_out) memcpy(scal_out, S->h_scal_red, 10 * sizeof(double));
        return rc;
    }
    return HIPKKT_ERR_DEVICE;
}

int32_t hipkkt_kkt_solve_reduced(hipkkt_handle h, const double *rhs_x, const double *workz, const double *var_x, const double *scal_in4,
                                 int32_t const_pending, double *lhs_x, double *lhs_z, double *scal_out10, int32_t ir_enable,
                                 double reltol, double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps2) {
    HK_ENTER(h)
    const int64_t n = S->img.n, m = S->img.m;
    if (!S->l1 || !S->d_qb || !scal_in4 || !scal_out10 || (n && (!rhs_x || !var_x)) || (m && !workz)) {
        S->err = "kkt_solve_reduced: call hipkkt_set_qb first / bad arguments";
        return HIPKKT_ERR_ARGUMENT;
    }
    return kkt_solve_reduced_impl(S, rhs_x, workz, var_x, nullptr, scal_in4, const_pending, lhs_x, lhs_z, nullptr, scal_out10, ir_enable,
                                  reltol, abstol, max_iter, stop_ratio, ir_steps2);
    HK_LEAVE
}

int32_t hipkkt_kkt_solve_reduced_dev(hipkkt_handle h, const double *in_dev, const double *scal_in4, int32_t const_pending,
                                     double *lhs_dev, double *scal_out10, int32_t ir_enable, double reltol, double abstol,
                                     int64_t max_iter, double stop_ratio, int64_t *ir_steps2) {
    HK_ENTER(h)
    if (!S->l1 || !S->d_qb || !scal_in4 || !scal_out10 || !in_dev) {
        S->err = "kkt_solve_reduced_dev: call hipkkt_set_qb first / bad arguments";
        return HIPKKT_ERR_ARGUMENT;
    }
    return kkt_solve_reduced_impl(S, nullptr, nullptr, nullptr, in_dev, scal_in4, const_pending, nullptr, nullptr, lhs_dev, scal_out10,
                                  ir_enable, reltol, abstol, max_iter, stop_ratio, ir_steps2);
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

}  // extern "C"
