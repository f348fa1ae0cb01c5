// Developer tool (not product): prints per-level statistics of the symbolic plan for a KKT pattern
// read from a raw file written by tools/plan_stats.py.  Build:
//   g++ -O2 -std=c++17 -I clarabel.jl_amd/csrc tools/plan_stats.cpp clarabel.jl_amd/csrc/symbolic.cpp clarabel.jl_amd/csrc/ordering.cpp -o /tmp/plan_stats
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <chrono>
#include "symbolic.h"
using namespace hipkkt;
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    int64_t N, nnz;
    fread(&N, 8, 1, f); fread(&nnz, 8, 1, f);
    std::vector<int64_t> Ap(N + 1), Ai(nnz);
    fread(Ap.data(), 8, N + 1, f); fread(Ai.data(), 8, nnz, f);
    fclose(f);
    PlanOptions opt;
    if (argc > 2) opt.max_width = atoi(argv[2]);
    if (argc > 3) opt.update_policy = atoi(argv[3]);
    if (argc > 4) opt.n_hold = atoi(argv[4]);
    if (argc > 5) opt.nd_mode = atoi(argv[5]);
    if (argc > 6) opt.nd_leaf = atoi(argv[6]);
    HostPlan P;
    auto t0 = std::chrono::steady_clock::now();
    std::string err = build_plan((int)N, Ap.data(), Ai.data(), nullptr, opt, P);
    auto t1 = std::chrono::steady_clock::now();
    if (!err.empty()) { printf("err %s\n", err.c_str()); return 1; }
    printf("N %d nnzK %ld nnzL %ld nsuper %d nlevels %d panel_doubles %ld flops_colcount %.3e flops_update %.3e flops_exec %.3e plan_s %.3f\n",
           P.N, (long)P.nnzK, (long)P.nnzL, P.nsuper, P.nlevels, (long)P.panel_doubles, P.flops_colcount, P.flops_update, P.flops_exec,
           std::chrono::duration<double>(t1 - t0).count());
    printf("phases [%s]\n", P.timing_note.c_str());
    { auto fbs = front_batches(P, opt.update_policy, 5); printf("front batches %zu:", fbs.size()); for (auto &h : fbs) printf(" (p%d x%d)", h.p0, h.nb); printf("\n");
      for (auto &F : P.fronts) printf("front: np %d cw %d W %d rF %d levels %d..%d\n", F.np, F.cw, F.W, F.rF, F.level_first, F.level_last); }
    printf("ntasks %zu ngroups %zu ordering_used %d fronts %zu | model: md %.3f ms (%d levels)  nd %.3f ms (%d levels)\n", P.upd_tasks.size(), P.upd_groups.size(), P.ordering_used, P.fronts.size(), 1e3 * P.cost_md_seconds, P.cost_md_levels, 1e3 * P.cost_nd_seconds, P.cost_nd_levels);
    for (int l = 0; l < P.nlevels; l++) {   // per stage: per-entry gather lists (k_update_gather)
        const int64_t e0 = P.gath_stage_ptr[l], e1 = P.gath_stage_ptr[l + 1];
        if (e1 == e0) continue;
        int64_t maxp = 0, pairs = 0; double fma = 0, maxf = 0;
        for (int64_t e = e0; e < e1; e++) {
            const int64_t np = P.gath_pptr[e + 1] - P.gath_pptr[e];
            double f = 0;
            for (int64_t q = P.gath_pptr[e]; q < P.gath_pptr[e + 1]; q++) f += P.sn_first[P.gath_sn[q] + 1] - P.sn_first[P.gath_sn[q]];
            maxp = std::max(maxp, np); pairs += np; fma += f; maxf = std::max(maxf, f);
        }
        printf("  gather stage %d: %ld entries, %ld pairs (max %ld per entry), %.3e fma (max %.0f per entry)\n", l, (long)(e1 - e0), (long)pairs, (long)maxp, fma, maxf);
    }
    for (int l = 0; l < P.nlevels; l++) {   // per stage: dense tiles by average coverage per task, and sub-blocks touched per task
        double cov = 0, nt = 0, sb = 0, kk = 0; int ng = 0;
        int hist[6] = {0, 0, 0, 0, 0, 0};
        for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l] + P.upd_stage_ndense[l]; g++) {
            const UpdGroup &G = P.upd_groups[g];
            const int ft = P.sn_first[G.tgt];
            double c = 0;
            for (int q = G.task_begin; q < G.task_end; q++) {
                const UpdTask &T = P.upd_tasks[q];
                c += (double)T.nrows * T.ncols;
                unsigned rb = 0, cb = 0;
                const int *srows = &P.sn_rows[P.sn_rowptr[T.src]];
                const int *rel = &P.rel[T.rel_off];
                for (int i = 0; i < T.nrows; i++) rb |= 1u << ((rel[T.row_lo + i - T.col_lo] - G.row_base) >> 4);
                for (int j = 0; j < T.ncols; j++) cb |= 1u << ((srows[T.col_lo + j] - ft) >> 4);
                sb += __builtin_popcount(rb) * __builtin_popcount(cb);
                kk += P.sn_first[T.src + 1] - P.sn_first[T.src];
            }
            const double per = c / (G.task_end - G.task_begin);
            hist[per < 100 ? 0 : per < 200 ? 1 : per < 400 ? 2 : per < 800 ? 3 : per < 1600 ? 4 : 5]++;
            cov += c; nt += G.task_end - G.task_begin; ng++;
        }
        if (ng > 50) printf("  stage %d: %d dense tiles, %.0f tasks/tile, %.0f entries/task, %.1f sub-blocks/task, K %.1f; tiles by entries/task <100:%d <200:%d <400:%d <800:%d <1600:%d more:%d\n",
                            l, ng, nt / ng, cov / nt, sb / nt, kk / nt, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
    }
    printf("%5s %7s %7s %7s %9s %12s %8s %8s\n", "lvl", "nsn", "maxw", "maxr", "sum_rw", "upd_flops", "groups", "facitems");
    std::vector<double> lf(P.nlevels, 0.0), lfd(P.nlevels, 0.0), fills(P.nlevels, 0.0);
    for (int l = 0; l < P.nlevels; l++)
        for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l + 1]; g++)
            for (int q = P.upd_groups[g].task_begin; q < P.upd_groups[g].task_end; q++) {
                const UpdTask &t = P.upd_tasks[q];
                int w = P.sn_first[t.src + 1] - P.sn_first[t.src];
                lf[l] += 2.0 * t.nrows * t.ncols * w;
                if (P.upd_groups[g].dense) lfd[l] += 2.0 * t.nrows * t.ncols * w;
                fills[l] += (double)t.nrows * t.ncols / 4096.0;
            }
    {   // pair statistics of the non-dense groups: sum nrows*ncols, by stage
        for (int l = 0; l < P.nlevels; l++) {
            double pairs = 0, pk = 0; int ng = 0;
            for (int g = P.upd_stage_ptr[l] + P.upd_stage_ndense[l]; g < P.upd_stage_ptr[l + 1]; g++) {
                ng++;
                for (int q = P.upd_groups[g].task_begin; q < P.upd_groups[g].task_end; q++) {
                    const UpdTask &t = P.upd_tasks[q];
                    pairs += (double)t.nrows * t.ncols;
                    pk += (double)t.nrows * t.ncols * (P.sn_first[t.src + 1] - P.sn_first[t.src]);
                }
            }
            if (ng) printf("sparse stage %d: groups %d pairs %.3e fma %.3e\n", l, ng, pairs, pk);
        }
    }
    for (int l = 0; l < P.nlevels; l++) {
        int maxw = 0; int64_t maxr = 0, srw = 0;
        for (int q = P.lvl_ptr[l]; q < P.lvl_ptr[l + 1]; q++) {
            int s = P.lvl_sn[q];
            int w = P.sn_first[s + 1] - P.sn_first[s];
            int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
            maxw = std::max(maxw, w); maxr = std::max(maxr, r); srw += r * w;
        }
        int ntk = 0;
        for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l + 1]; g++) ntk += P.upd_groups[g].task_end - P.upd_groups[g].task_begin;
        printf("%5d %7d %7d %7ld %9ld %12.3e %8d %8d   dense: %6d groups (far %5d) %10.3e flops; tasks %7d avgfill %.2f\n", l, P.lvl_ptr[l + 1] - P.lvl_ptr[l], maxw, (long)maxr, (long)srw, lf[l],
               P.upd_stage_ptr[l + 1] - P.upd_stage_ptr[l], P.fac_lvl_ptr[l + 1] - P.fac_lvl_ptr[l], P.upd_stage_ndense[l], 0, lfd[l], ntk, ntk ? fills[l] / ntk : 0.0);
    }
    if (argc > 6) {   // debug: who owns ubuf position argv[5], as seen from supernode argv[6]
        const int64_t upos = atoll(argv[5]);
        const int sn = atoi(argv[6]);
        for (int c = 0; c < P.nsuper; c++) {
            const int64_t r = P.sn_rowptr[c + 1] - P.sn_rowptr[c];
            const int w = P.sn_first[c + 1] - P.sn_first[c];
            if (upos >= P.u_off[c] && upos < P.u_off[c] + (r - w))
                printf("ubuf %ld: owner sn %d (level %d, w %d, r %ld, front %d, parent %d) row %ld ; consumer sn %d level %d front %d parent %d\n", (long)upos, c,
                       P.sn_level[c], w, (long)r, P.sn_front[c], P.sn_parent[c], (long)(upos - P.u_off[c]), sn, P.sn_level[sn], P.sn_front[sn], P.sn_parent[sn]);
        }
        for (int l = 0; l < P.nlevels; l++) {
            int n = 0;
            for (int q = P.lvl_ptr[l]; q < P.lvl_ptr[l + 1]; q++) {
                const int s2 = P.lvl_sn[q];
                if (P.sn_front[s2] >= 0) continue;
                n += (int)std::max<int64_t>(1, (P.sn_rowptr[s2 + 1] - P.sn_rowptr[s2] - (P.sn_first[s2 + 1] - P.sn_first[s2]) + 63) / 64);
            }
            if (l < 12) printf("level %d: %d regular items\n", l, n);
        }
    }
    // K (source width) histogram of the dense tasks of the stage with the most dense tasks
    {
        int best = 0; int64_t bestn = -1;
        for (int l = 0; l < P.nlevels; l++) {
            int64_t n = 0;
            for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l] + P.upd_stage_ndense[l]; g++) n += P.upd_groups[g].task_end - P.upd_groups[g].task_begin;
            if (n > bestn) { bestn = n; best = l; }
        }
        std::vector<int64_t> hk(65, 0), hr(65, 0);
        for (int g = P.upd_stage_ptr[best]; g < P.upd_stage_ptr[best] + P.upd_stage_ndense[best]; g++)
            for (int q = P.upd_groups[g].task_begin; q < P.upd_groups[g].task_end; q++) {
                const UpdTask &t = P.upd_tasks[q];
                hk[std::min(64, P.sn_first[t.src + 1] - P.sn_first[t.src])]++;
                hr[std::min(64, (int)t.nrows)]++;
            }
        printf("stage %d dense-task K histogram:", best);
        for (int k = 1; k <= 64; k++) if (hk[k]) printf(" %d:%ld", k, (long)hk[k]);
        printf("\nstage %d dense-task nrows histogram:", best);
        for (int k = 1; k <= 64; k++) if (hr[k]) printf(" %d:%ld", k, (long)hr[k]);
        printf("\n");
    }
    return 0;
}
