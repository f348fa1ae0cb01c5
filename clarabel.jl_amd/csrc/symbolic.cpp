// Symbolic analysis: ordering -> elimination tree -> postorder -> column counts -> (relaxed)
// supernodes -> width-capped panels -> row structures -> levels -> static work lists.
// See symbolic.h / DESIGN.md §4.  Host only; no numeric work.
#include "symbolic.h"

#include <algorithm>
#include <future>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>

namespace hipkkt {

namespace {

// permuted upper-triangular pattern: column max(pi,pj) holds row min(pi,pj)
void permuted_upper(int N, const int64_t *Ap, const int64_t *Ai, const std::vector<int> &iperm,
                    std::vector<int64_t> &up, std::vector<int> &ui) {
    up.assign(N + 1, 0);
    for (int j = 0; j < N; j++)
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int pi = iperm[Ai[p]], pj = iperm[j];
            up[std::max(pi, pj) + 1]++;
        }
    for (int j = 0; j < N; j++) up[j + 1] += up[j];
    ui.resize(up[N]);
    std::vector<int64_t> nxt(up.begin(), up.end() - 1);
    for (int j = 0; j < N; j++)
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int pi = iperm[Ai[p]], pj = iperm[j];
            ui[nxt[std::max(pi, pj)]++] = std::min(pi, pj);
        }
}

// elimination tree + strictly-lower column counts of L by row-subtree traversal (O(nnz(L)))
void etree_counts(int N, const std::vector<int64_t> &up, const std::vector<int> &ui,
                  std::vector<int> &parent, std::vector<int> &cnt) {
    parent.assign(N, -1);
    cnt.assign(N, 0);
    std::vector<int> work(N, -1);
    for (int j = 0; j < N; j++) {
        work[j] = j;
        for (int64_t p = up[j]; p < up[j + 1]; p++) {
            int i = ui[p];
            while (work[i] != j) {
                if (parent[i] < 0) parent[i] = j;
                cnt[i]++;
                work[i] = j;
                i = parent[i];
            }
        }
    }
}

void postorder(int N, const std::vector<int> &parent, std::vector<int> &post) {
    std::vector<int> head(N, -1), next(N, -1), stack;
    for (int j = N - 1; j >= 0; j--)
        if (parent[j] >= 0) { next[j] = head[parent[j]]; head[parent[j]] = j; }
    post.clear();
    post.reserve(N);
    for (int r = 0; r < N; r++) {
        if (parent[r] >= 0) continue;
        stack.push_back(r);
        while (!stack.empty()) {
            int p = stack.back();
            int c = head[p];
            if (c < 0) { post.push_back(p); stack.pop_back(); }
            else { head[p] = next[c]; stack.push_back(c); }
        }
    }
}

// Cost of one KKT iteration unit (1 factorisation + ~6 triangular solve pairs) predicted for an elimination order, in
// seconds: dependency latency per level of the (fundamental, width-capped) supernodal elimination tree + factor flops
// at the rate the dense kernels sustain + the solves' traffic.  Constants measured on MI355X (DESIGN.md section 7:
// cfg 2b 4.3 ms / 150 levels per factorisation, 6 us per level and solve; cfg 2a 11.7 TFLOP/s over the whole
// factorisation).  Only used to choose between candidate orders.
struct OrderCost {
    double flops = 0, nnzL = 0, seconds = 0;
    int levels = 0;
};
OrderCost order_cost(int N, const int64_t *Ap, const int64_t *Ai, const std::vector<int> &perm, int maxw) {
    std::vector<int> ip(N);
    for (int k = 0; k < N; k++) ip[perm[k]] = k;
    std::vector<int64_t> up;
    std::vector<int> ui, parent, cnt, post;
    permuted_upper(N, Ap, Ai, ip, up, ui);
    etree_counts(N, up, ui, parent, cnt);
    postorder(N, parent, post);
    std::vector<int> newlab(N), par2(N), cnt2(N);
    for (int k = 0; k < N; k++) newlab[post[k]] = k;
    for (int k = 0; k < N; k++) {
        const int o = post[k];
        par2[k] = parent[o] < 0 ? -1 : newlab[parent[o]];
        cnt2[k] = cnt[o];
    }
    OrderCost c;
    // supernodes: fundamental, cut at maxw columns; level = longest chain of supernodes below
    std::vector<int> sn_of(N), sn_last;
    int ns = 0, width = 0;
    for (int j = 0; j < N; j++) {
        const bool cont = j > 0 && par2[j - 1] == j && cnt2[j - 1] == cnt2[j] + 1 && width < maxw;
        if (!cont) { ns++; width = 0; sn_last.push_back(j); }
        else sn_last.back() = j;
        width++;
        sn_of[j] = ns - 1;
        c.flops += (double)cnt2[j] * cnt2[j] + 3.0 * cnt2[j];
        c.nnzL += cnt2[j];
    }
    std::vector<int> level(ns, 0);
    for (int s = 0; s < ns; s++) {
        const int pj = par2[sn_last[s]];
        if (pj >= 0) level[sn_of[pj]] = std::max(level[sn_of[pj]], level[s] + 1);
        c.levels = std::max(c.levels, level[s] + 1);
    }
    c.seconds = c.levels * 64e-6 + c.flops / 12e12 + 6.0 * 24.0 * c.nnzL / 1.5e12;
    return c;
}

struct TaskKey {
    int32_t stage, tgt, rb, src;
    int32_t task;
};

}  // namespace

std::string build_plan(int N, const int64_t *Ap, const int64_t *Ai, const int64_t *user_perm,
                       const PlanOptions &opt, HostPlan &P) {
    P = HostPlan();
    P.N = N;
    P.nnzK = Ap[N];
    const int maxw = std::max(1, std::min(opt.max_width, kMaxSnWidth));

    auto t_last = std::chrono::steady_clock::now();
    bool cancelled = false;
    auto lap = [&](const char *name) {
        if (opt.cancel && opt.cancel->load(std::memory_order_relaxed)) cancelled = true;
        const auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof buf, "%s %.2f ", name, 1e3 * std::chrono::duration<double>(now - t_last).count());
        P.timing_note += buf;
        t_last = now;
    };
    // ---- 1. ordering
    std::vector<int> perm0;
    if (user_perm) {
        perm0.resize(N);
        std::vector<char> seen(N, 0);
        for (int k = 0; k < N; k++) {
            int64_t o = user_perm[k];
            if (o < 0 || o >= N || seen[o]) return "user_perm is not a permutation";
            seen[o] = 1;
            perm0[k] = (int)o;
        }
        P.ordering_used = 2;
    } else {
        amd_order(N, Ap, Ai, opt.amd_dense_scale, perm0);
        if ((int)perm0.size() != N) return "internal: ordering size mismatch";
        // the alternatives below are only worth their set-up time when minimum degree leaves something to gain: an
        // expensive factorisation (-> cone rows first) or a long dependent chain (-> nested dissection)
        const OrderCost c_md = order_cost(N, Ap, Ai, perm0, maxw);
        P.cost_md_seconds = c_md.seconds;
        P.cost_md_levels = c_md.levels;
        if (opt.n_hold > 0 && opt.n_hold < N && c_md.flops > 1e9) {
            // second candidate: eliminate the cone rows first, the variables last; keep the cheaper one
            if (opt.on_alternative_order) opt.on_alternative_order(perm0);     // perm0 = minimum degree on K (speculative head start)
            std::vector<int> perm1;
            std::vector<char> hold(N, 0);
            for (int i = 0; i < opt.n_hold; i++) hold[i] = 1;
            amd_order(N, Ap, Ai, opt.amd_dense_scale, perm1, hold.data());
            if ((int)perm1.size() == N) {
                const OrderCost c_b = order_cost(N, Ap, Ai, perm1, maxw);
                if (c_b.flops < 0.7 * c_md.flops) {
                    perm0.swap(perm1); P.ordering_used = 1; P.cost_md_seconds = c_b.seconds; P.cost_md_levels = c_b.levels;
                }
            }
        }
    }
    if (!user_perm && opt.nd_mode > 0 && N >= 64 &&
        (opt.nd_mode >= 2 || (Ap[N] <= (int64_t)64 * N && P.cost_md_levels >= 36))) {   // (dense blocks: no separators; short chains: nothing to
                                                                                   // gain -- on the 256 batch problems nested dissection never won below 41 levels, and evaluating it cost more than the minimum-degree order itself)
        // third candidate: nested dissection (ordering.cpp) -- far fewer dependent levels on banded / grid-like systems
        std::vector<int> permN;
        nd_order(N, Ap, Ai, opt.amd_dense_scale, opt.nd_leaf, permN);
        if ((int)permN.size() == N) {
            const OrderCost c1 = order_cost(N, Ap, Ai, permN, maxw);
            P.cost_nd_seconds = c1.seconds;
            P.cost_nd_levels = c1.levels;
            if (opt.nd_mode >= 2 || c1.seconds < 0.8 * P.cost_md_seconds) { perm0.swap(permN); P.ordering_used = 3; }
        }
    }
    std::vector<int> iperm0(N);
    for (int k = 0; k < N; k++) iperm0[perm0[k]] = k;

    lap("ordering");

    if (cancelled) return "cancelled";
    // ---- 2-4. etree, postorder, final permutation
    std::vector<int64_t> up;
    std::vector<int> ui, parent, cnt, post;
    permuted_upper(N, Ap, Ai, iperm0, up, ui);
    etree_counts(N, up, ui, parent, cnt);
    postorder(N, parent, post);
    P.perm.resize(N);
    P.iperm.resize(N);
    {
        std::vector<int> newlab(N);
        for (int k = 0; k < N; k++) { newlab[post[k]] = k; P.perm[k] = perm0[post[k]]; }
        for (int k = 0; k < N; k++) P.iperm[P.perm[k]] = k;
        std::vector<int> par2(N), cnt2(N);
        for (int k = 0; k < N; k++) {
            int o = post[k];
            par2[k] = parent[o] < 0 ? -1 : newlab[parent[o]];
            cnt2[k] = cnt[o];
        }
        parent.swap(par2);
        cnt.swap(cnt2);
    }
    permuted_upper(N, Ap, Ai, P.iperm, up, ui);
    // lower pattern by columns (transpose of up/ui): column j -> rows i > j
    std::vector<int64_t> lp(N + 1, 0);
    std::vector<int> li;
    {
        for (int j = 0; j < N; j++)
            for (int64_t p = up[j]; p < up[j + 1]; p++)
                if (ui[p] != j) lp[ui[p] + 1]++;
        for (int j = 0; j < N; j++) lp[j + 1] += lp[j];
        li.resize(lp[N]);
        std::vector<int64_t> nxt(lp.begin(), lp.end() - 1);
        for (int j = 0; j < N; j++)
            for (int64_t p = up[j]; p < up[j + 1]; p++)
                if (ui[p] != j) li[nxt[ui[p]]++] = j;
    }
    P.nnzL = 0;
    P.flops_colcount = 0;
    for (int j = 0; j < N; j++) {
        P.nnzL += cnt[j];
        P.flops_colcount += (double)cnt[j] * cnt[j] + 3.0 * cnt[j];
    }
    {
        std::vector<int> h(N, 1);
        int H = 0;
        for (int j = 0; j < N; j++) {
            if (parent[j] >= 0) h[parent[j]] = std::max(h[parent[j]], h[j] + 1);
            H = std::max(H, h[j]);
        }
        P.etree_height = H;
    }

    // ---- 5. supernodes (maximal: same structure as the next column)
    std::vector<char> is_start(N + 1, 0);
    is_start[N] = 1;
    for (int j = 0; j < N; j++)
        is_start[j] = (j == 0) || !(parent[j - 1] == j && cnt[j - 1] == cnt[j] + 1);
    // ---- 6. relaxed amalgamation of a child chain into its parent (contiguous columns only)
    if (opt.relax && N > 0) {
        std::vector<int64_t> zz(N, 0);  // explicit zeros carried by the supernode starting at column f
        int f = 0;
        while (f < N) {
            int l = f;
            while (!is_start[l + 1]) l++;
            if (l + 1 >= N) break;
            int pf = l + 1;
            int pl = pf;
            while (!is_start[pl + 1]) pl++;
            bool merged = false;
            if (parent[l] == pf) {
                int64_t wc = l - f + 1, wp = pl - pf + 1, hc = cnt[l], hp = cnt[pl];
                int64_t w = wc + wp;
                int64_t znew = wc * (wp + hp - hc);
                int64_t z = zz[f] + zz[pf] + znew;
                double total = 0.5 * (double)w * (double)(w + 1) + (double)w * (double)hp;
                double frac = total > 0 ? (double)z / total : 0.0;
                bool ok = (w <= 4) || (w <= 16 && frac < 0.8) || (w <= 48 && frac < 0.1) || (frac < 0.05);
                if (ok && w <= maxw) {
                    is_start[pf] = 0;
                    zz[f] = z;
                    merged = true;
                }
            }
            if (!merged) f = pf;
        }
    }
    // ---- 7. width cap: split wide supernodes into a chain of balanced chunks
    std::vector<std::array<int, 3>> splits;   // (first column, #chunks, chunk width)
    {
        int f = 0;
        while (f < N) {
            int l = f;
            while (!is_start[l + 1]) l++;
            int w = l - f + 1;
            if (w > maxw) {
                int nch = (w + maxw - 1) / maxw;
                int cw = (w + nch - 1) / nch;
                // chains long enough to become a front keep full-width panels (the last one takes the remainder): k_front_block
                // (front_block.hip) factors whole update batches of 64-column panels in one launch
                if (opt.front_min_panels > 0 && nch >= opt.front_min_panels && maxw == kMaxSnWidth && w >= opt.front_block_min_width) cw = maxw;
                for (int c = f + cw; c <= l; c += cw) is_start[c] = 1;
                splits.push_back({f, (w + cw - 1) / cw, cw});
            }
            f = l + 1;
        }
    }
    // ---- 8. supernode arrays
    P.sn_of_col.resize(N);
    for (int j = 0; j < N; j++) {
        if (is_start[j]) P.sn_first.push_back(j);
        P.sn_of_col[j] = (int)P.sn_first.size() - 1;
    }
    P.nsuper = (int)P.sn_first.size();
    P.sn_first.push_back(N);
    const int S = P.nsuper;

    // ---- 9. row structures (children before parents: postorder guarantees index order)
    P.sn_rowptr.assign(S + 1, 0);
    P.sn_parent.assign(S, -1);
    P.sn_level.assign(S, 0);
    {
        std::vector<std::vector<int>> children(S);
        std::vector<int> mark(N, -1), tmp;
        for (int s = 0; s < S; s++) {
            int f = P.sn_first[s], l = P.sn_first[s + 1] - 1;
            tmp.clear();
            for (int j = f; j <= l; j++) { mark[j] = s; }
            for (int j = f; j <= l; j++)
                for (int64_t p = lp[j]; p < lp[j + 1]; p++) {
                    int i = li[p];
                    if (i > l && mark[i] != s) { mark[i] = s; tmp.push_back(i); }
                }
            for (int c : children[s]) {
                const int *cr = &P.sn_rows[P.sn_rowptr[c]];
                int64_t nr = P.sn_rowptr[c + 1] - P.sn_rowptr[c];
                for (int64_t q = 0; q < nr; q++) {
                    int i = cr[q];
                    if (i > l && mark[i] != s) { mark[i] = s; tmp.push_back(i); }
                }
            }
            std::sort(tmp.begin(), tmp.end());
            for (int j = f; j <= l; j++) P.sn_rows.push_back(j);
            P.sn_rows.insert(P.sn_rows.end(), tmp.begin(), tmp.end());
            P.sn_rowptr[s + 1] = (int64_t)P.sn_rows.size();
            if (!tmp.empty()) {
                int ps = P.sn_of_col[tmp[0]];
                P.sn_parent[s] = ps;
                children[ps].push_back(s);
                P.sn_level[ps] = std::max(P.sn_level[ps], P.sn_level[s] + 1);
            }
            std::vector<int>().swap(children[s]);
        }
    }
    // ---- 10. storage offsets
    P.sn_panel.assign(S + 1, 0);
    P.sn_diag.assign(S + 1, 0);
    P.u_off.assign(S + 1, 0);
    P.lt_off.assign(S + 1, 0);
    for (int s = 0; s < S; s++) {
        int64_t w = P.sn_first[s + 1] - P.sn_first[s];
        int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
        int64_t sz = (r * w + 7) & ~(int64_t)7;
        P.sn_panel[s + 1] = P.sn_panel[s] + sz;
        P.sn_diag[s + 1] = P.sn_diag[s] + ((w * w + 7) & ~(int64_t)7);
        P.u_off[s + 1] = P.u_off[s] + (r - w);
        P.lt_off[s + 1] = P.lt_off[s] + (((r - w) * w + 7) & ~(int64_t)7);
    }
    P.panel_doubles = P.sn_panel[S];
    P.diag_doubles = P.sn_diag[S];
    P.ubuf_len = P.u_off[S];
    if (P.ubuf_len >= (int64_t)1 << 31) return "problem too large: off-diagonal row count exceeds int32";

    // ---- 11. levels
    P.nlevels = 0;
    for (int s = 0; s < S; s++) P.nlevels = std::max(P.nlevels, P.sn_level[s] + 1);
    P.lvl_ptr.assign(P.nlevels + 1, 0);
    for (int s = 0; s < S; s++) P.lvl_ptr[P.sn_level[s] + 1]++;
    for (int l = 0; l < P.nlevels; l++) P.lvl_ptr[l + 1] += P.lvl_ptr[l];
    P.lvl_sn.resize(S);
    {
        std::vector<int> nxt(P.lvl_ptr.begin(), P.lvl_ptr.end() - 1);
        for (int s = 0; s < S; s++) P.lvl_sn[nxt[P.sn_level[s]]++] = s;
    }

    lap("structure");

    if (cancelled) return "cancelled";
    // ---- 12. scatter map of the original nonzeros into the panels.  Independent of the work lists built below: on big images
    //          (the dense PSD blocks of an SDP: 1.6e7 binary searches) it runs on a second host thread meanwhile.
    P.kmap.resize(P.nnzK);
    P.diag_dst.assign(N, -1);
    auto kmap_job = [&P, N, Ap, Ai, &opt]() -> const char * {
        for (int j = 0; j < N; j++) {
            if ((j & 1023) == 0 && opt.cancel && opt.cancel->load(std::memory_order_relaxed)) return "cancelled";
            for (int64_t q = Ap[j]; q < Ap[j + 1]; q++) {
                int pi = P.iperm[Ai[q]], pj = P.iperm[j];
                int c = std::min(pi, pj), r = std::max(pi, pj);
                int s = P.sn_of_col[c];
                const int *rows = &P.sn_rows[P.sn_rowptr[s]];
                int64_t nr = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
                const int *it = std::lower_bound(rows, rows + nr, r);
                if (it == rows + nr || *it != r) return "internal: entry outside the symbolic structure";
                int64_t dest = P.sn_panel[s] + (it - rows) + (int64_t)(c - P.sn_first[s]) * nr;
                P.kmap[q] = dest;
                if (pi == pj) P.diag_dst[c] = dest;
            }
        }
        for (int k = 0; k < N; k++)
            if (P.diag_dst[k] < 0) return "KKT matrix has a column without a diagonal entry";
        return nullptr;
    };
    std::future<const char *> kmap_future;      // (a std::async future joins in its destructor: early returns below are safe)
    if (P.nnzK > ((int64_t)1 << 21)) kmap_future = std::async(std::launch::async, kmap_job);
    else if (const char *e = kmap_job()) return e;

    // ---- 13. factor items
    P.fac_lvl_ptr.assign(P.nlevels + 1, 0);
    P.fac_lvl_maxw.assign(P.nlevels, 1);
    for (int l = 0; l < P.nlevels; l++) {
        for (int q = P.lvl_ptr[l]; q < P.lvl_ptr[l + 1]; q++) {
            int s = P.lvl_sn[q];
            int w = P.sn_first[s + 1] - P.sn_first[s];
            int64_t r = P.sn_rowptr[s + 1] - P.sn_rowptr[s];
            int nb = (int)std::max<int64_t>(1, (r - w + kFacRows - 1) / kFacRows);
            for (int b = 0; b < nb; b++) P.fac_items.push_back({s, b});
            P.fac_lvl_maxw[l] = std::max(P.fac_lvl_maxw[l], w);
            P.flops_exec += (double)w * w * w / 3.0 + (double)(r - w) * w * w;
        }
        P.fac_lvl_ptr[l + 1] = (int)P.fac_items.size();
    }

    // batch length of the batched right-looking schedule: 4 levels; 5 when a front of >= 64 panels dominates (measured,
    // cfg 2a 5.33 -> 5.18 ms per factorisation; shorter fronts and front-less problems are faster with 4: cfg 1 0.50
    // vs 0.75 ms, cfg 3 3.75 vs 4.09 ms)
    int update_batch = opt.update_batch;
    if (update_batch <= 0) {
        int max_chunks = 0;
        for (const auto &sp : splits) max_chunks = std::max(max_chunks, sp[1]);
        update_batch = max_chunks >= 64 ? 5 : 4;
    }
    P.update_batch_used = update_batch;

    lap("kmap+items");

    if (cancelled) return "cancelled";
    // ---- 14. update tasks, owned by target row-blocks
    {
        std::vector<TaskKey> keys;
        {
            int64_t max_r = 0;
            for (int s = 0; s < S; s++) max_r = std::max<int64_t>(max_r, P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
            P.rel.resize((size_t)max_r);
            for (int64_t q = 0; q < max_r; q++) P.rel[q] = (int)q;     // the shared identity segment (see below)
        }
        for (int s = 0; s < S; s++) {
            // the longest phase of a big analysis (1.2e7 tasks for the robust-order twin of an SDP): a cancelled speculation must not
            // hold its owner's destructor for seconds -- poll the flag here too, not only at the phase boundaries
            if ((s & 255) == 0 && opt.cancel && opt.cancel->load(std::memory_order_relaxed)) return "cancelled";
            int w = P.sn_first[s + 1] - P.sn_first[s];
            const int *rows = &P.sn_rows[P.sn_rowptr[s]];
            int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
            int a = w;
            while (a < r) {
                int t = P.sn_of_col[rows[a]];
                int b = a;
                while (b < r && P.sn_of_col[rows[b]] == t) b++;
                const int *trows = &P.sn_rows[P.sn_rowptr[t]];
                int tr = (int)(P.sn_rowptr[t + 1] - P.sn_rowptr[t]);
                if (P.rel.size() + (size_t)(r - a) >= ((size_t)1 << 31)) return "problem too large: relative index table exceeds int32";
                int rel_off = (int)P.rel.size();
                // positions of rows[a..r) inside the target's row list (both sorted)
                if (tr == r - a) {
                    // rows[a..r) is a subset of the target's rows (elimination-tree structure) with the same cardinality: the two
                    // lists are equal and the positions are 0, 1, 2, ... -- every such pair (the panels of a front above all: a
                    // 400-panel front would need 1.4e9 entries of its own) shares the identity segment at the start of rel[]
                    rel_off = 0;
                } else if ((int64_t)(r - a) * 16 < tr) {
                    const int *lo = trows;
                    for (int i = a; i < r; i++) {
                        lo = std::lower_bound(lo, trows + tr, rows[i]);
                        if (lo == trows + tr || *lo != rows[i]) return "internal: source row missing in target structure";
                        P.rel.push_back((int)(lo - trows));
                    }
                } else {
                    int q = 0;
                    for (int i = a; i < r; i++) {
                        while (q < tr && trows[q] < rows[i]) q++;
                        if (q == tr || trows[q] != rows[i]) return "internal: source row missing in target structure";
                        P.rel.push_back(q);
                    }
                }
                // when is (s -> t) applied?  any stage in [level(s), level(t)-1] is valid.
                //   0 right-looking: as soon as s is factored          (max parallelism, C tile re-read per source)
                //   1 left-looking : just before t is factored
                //   2 batched      : sources are batched over `update_batch` consecutive levels so that a
                //                    target tile is loaded once per batch (K = batch*w), near targets just in time
                int stage;
                if (opt.update_policy == 1) stage = P.sn_level[t] - 1;
                else if (opt.update_policy == 2) {
                    int B = update_batch;
                    int ls = P.sn_level[s];
                    int batch_end = ls - (ls % B) + (B - 1);
                    stage = std::min(P.sn_level[t] - 1, batch_end);
                } else stage = P.sn_level[s];
                // split the source rows by the target row-block they land in
                int i = a;
                while (i < r) {
                    int rb = P.rel[rel_off + (i - a)] / kUpdRows;
                    int e = i;
                    while (e < r && P.rel[rel_off + (e - a)] / kUpdRows == rb) e++;
                    UpdTask tk;
                    tk.src = s; tk.row_lo = i; tk.nrows = e - i; tk.col_lo = a; tk.ncols = b - a;
                    tk.rel_off = rel_off; tk.vt_begin = 0; tk.geom = 0;
                    keys.push_back({stage, t, rb, s, (int)P.upd_tasks.size()});
                    P.upd_tasks.push_back(tk);
                    P.flops_update += 2.0 * (double)(e - i) * (double)(b - a) * (double)w;
                    i = e;
                }
                a = b;
            }
        }
        lap("ut:gen");
        if (cancelled) return "cancelled";
        // order: (stage, target, row block, source, task).  The keys are generated source by source, task by task, i.e. already in
        // (source, task) order: three stable counting sorts -- by row block, by target, by stage -- finish the job in O(n)
        // (a comparison sort of the 1.2e7 keys of an SDP twin took half a second).
        {
            std::vector<TaskKey> tmp(keys.size());
            std::vector<int64_t> cnt;
            auto pass = [&](auto field, int64_t nbuckets) {
                cnt.assign((size_t)nbuckets + 1, 0);
                for (const TaskKey &k : keys) cnt[(size_t)field(k) + 1]++;
                for (int64_t b = 0; b < nbuckets; b++) cnt[b + 1] += cnt[b];
                for (const TaskKey &k : keys) tmp[(size_t)cnt[(size_t)field(k)]++] = k;
                keys.swap(tmp);
            };
            int max_rb = 0;
            for (const TaskKey &k : keys) max_rb = std::max(max_rb, k.rb);
            pass([](const TaskKey &k) { return k.rb; }, (int64_t)max_rb + 1);
            pass([](const TaskKey &k) { return k.tgt; }, S);
            pass([](const TaskKey &k) { return k.stage; }, P.nlevels);
        }
        lap("ut:sort");
        if (cancelled) return "cancelled";
        std::vector<UpdTask> sorted(keys.size());
        P.upd_stage_ptr.assign(P.nlevels + 1, 0);
        for (size_t q = 0; q < keys.size(); q++) {
            sorted[q] = P.upd_tasks[keys[q].task];
            bool newgrp = (q == 0) || keys[q].stage != keys[q - 1].stage || keys[q].tgt != keys[q - 1].tgt ||
                          keys[q].rb != keys[q - 1].rb;
            if (newgrp) {
                P.upd_groups.push_back({keys[q].tgt, keys[q].rb * kUpdRows, (int)q, (int)q + 1, 0, 0});
                P.upd_stage_ptr[keys[q].stage + 1]++;
            } else {
                P.upd_groups.back().task_end = (int)q + 1;
            }
            sorted[q].vt_begin = P.upd_groups.back().nvt;
            P.upd_groups.back().nvt += (sorted[q].ncols + 15) / 16;
        }
        P.upd_tasks.swap(sorted);
        for (int l = 0; l < P.nlevels; l++) P.upd_stage_ptr[l + 1] += P.upd_stage_ptr[l];
        P.flops_exec += P.flops_update;
        // ---- 14b. classify target tiles: a tile is "dense" when every contribution lands on a
        //           contiguous row range x contiguous column range of it (true for the panels of one
        //           wide front, e.g. the dense root) and the tile is reasonably filled.  Dense tiles are
        //           accumulated in registers over ALL their sources by k_update_dense; the others go
        //           through the relative-index scatter of k_update_stage.  Dense groups are moved to the
        //           front of their stage.
        lap("ut:groups");
        if (cancelled) return "cancelled";
        P.upd_stage_ndense.assign(P.nlevels, 0);
        P.upd_stage_flops_dense.assign(P.nlevels, 0.0);
        {
            // stage of a group = position in upd_stage_ptr (groups are stage-ordered at this point)
        }
        int64_t gather_pairs_bound = 0;     // pairs of the per-entry gather lists (before the upper-triangle entries are dropped)
        for (auto &G : P.upd_groups) {
            const int t = G.tgt, ft = P.sn_first[t];
            bool all_contig = true;
            double covered = 0;
            for (int q = G.task_begin; q < G.task_end; q++) {
                UpdTask &T = P.upd_tasks[q];
                const int *srows = &P.sn_rows[P.sn_rowptr[T.src]];
                const int *rel = &P.rel[T.rel_off];
                const int r0 = rel[T.row_lo - T.col_lo], r1 = rel[T.row_lo + T.nrows - 1 - T.col_lo];
                const int c0 = srows[T.col_lo] - ft, c1 = srows[T.col_lo + T.ncols - 1] - ft;
                const bool contig = (r1 - r0 == T.nrows - 1) && (c1 - c0 == T.ncols - 1);
                T.geom = ((r0 - G.row_base) & 255) | ((c0 & 255) << 8) | (contig ? 1 << 16 : 0);
                all_contig = all_contig && contig;
                covered += (double)T.nrows * T.ncols;
            }
            const double ntasks = G.task_end - G.task_begin;
            const double fill = covered / (ntasks * kUpdRows * kMaxSnWidth);
            // dense enough per contribution -> matrix-core path; otherwise the contributions are single
            // entries of tiny leaf supernodes -> per-entry gather
            G.dense = ((all_contig && fill >= 0.4) || fill >= 0.3 || covered >= opt.dense_min_cover * ntasks) ? 1 : 2;
            if (G.dense != 1) gather_pairs_bound += (int64_t)covered;
        }
        // A stage with a dense launch of its own (>= 128 tiles) absorbs its thin tiles of few contributions (<= 8) as a few more tiles of
        // that launch; as per-entry gather lists they are a launch of their own AFTER it whose duration is one long dependent dot
        // product (cfg 5: 158 such tiles behind 6288 dense ones, 70 us per stage, 0.49 ms of a 5.1 ms factorisation -> 4.63 ms).
        // Not in stages without such a launch: with the rule applied everywhere cfg 2a / cfg 1 / cfg 3 lose 1.5 / 5 / 1 % (tiles with
        // tile maps as small launches of the sparse tree cost more than their gather lists); with it they are unchanged (measured).
        for (int l = 0; l < P.nlevels; l++) {
            int nd0 = 0;
            for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l + 1]; g++) nd0 += P.upd_groups[g].dense == 1;
            if (nd0 < 128) continue;
            for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l + 1]; g++) {
                UpdGroup &G = P.upd_groups[g];
                if (G.dense == 2 && G.task_end - G.task_begin <= 8) G.dense = 1;
            }
        }
        for (auto &G : P.upd_groups) {
            if (G.dense != 1) continue;
            const int ft = P.sn_first[G.tgt];
            for (int q = G.task_begin; q < G.task_end; q++) {
                const UpdTask &T = P.upd_tasks[q];
                P.flops_update_dense += 2.0 * T.nrows * T.ncols * (P.sn_first[T.src + 1] - P.sn_first[T.src]);
            }
            // contributions that do not land contiguously get explicit tile maps (tile row / column ->
            // source row offset or -1): k_update_dense then gathers its operands through them
            for (int q = G.task_begin; q < G.task_end; q++) {
                UpdTask &T = P.upd_tasks[q];
                if (T.geom & (1 << 16)) continue;
                const int *srows = &P.sn_rows[P.sn_rowptr[T.src]];
                const int *rel = &P.rel[T.rel_off];
                T.vt_begin = (int)(P.upd_tmap.size() / 128);
                T.geom |= 1 << 17;
                const size_t base = P.upd_tmap.size();
                P.upd_tmap.resize(base + 128, (int16_t)-1);
                for (int i = 0; i < T.nrows; i++) P.upd_tmap[base + (rel[T.row_lo + i - T.col_lo] - G.row_base)] = (int16_t)i;
                for (int j = 0; j < T.ncols; j++) P.upd_tmap[base + 64 + (srows[T.col_lo + j] - ft)] = (int16_t)j;
            }
        }
        lap("ut:classify");
        if (cancelled) return "cancelled";
        P.upd_stage_ngather.assign(P.nlevels, 0);
        P.gath_stage_ptr.assign(P.nlevels + 1, 0);
        P.gath_pptr.push_back(0);
        P.gath_src.reserve(gather_pairs_bound); P.gath_dj.reserve(gather_pairs_bound); P.gath_sn.reserve(gather_pairs_bound);
        P.gath_tgt.reserve(gather_pairs_bound); P.gath_pptr.reserve(gather_pairs_bound + 1);
        std::vector<int> gcount;
        for (int l = 0; l < P.nlevels; l++) {
            auto b = P.upd_groups.begin() + P.upd_stage_ptr[l], e = P.upd_groups.begin() + P.upd_stage_ptr[l + 1];
            auto mid = std::stable_partition(b, e, [](const UpdGroup &g) { return g.dense == 1; });
            auto mid2 = std::stable_partition(mid, e, [](const UpdGroup &g) { return g.dense == 2; });
            // the dense tiles of a stage in descending order of their number of contributions (in classes of 4, target order inside a
            // class): a launch ends with its longest tile, and the launch of the sparse tree into a front (cfg 2a: 3828 tiles of 1 - 29
            // contributions on 2048 wavefront slots) otherwise starts long tiles in its last round
            // (only launches of more than one round of wavefront slots: the small just-in-time launches lose more locality than they gain)
            if (mid - b >= 2048)
                std::stable_sort(b, mid, [](const UpdGroup &x, const UpdGroup &y) { return (x.task_end - x.task_begin) / 4 > (y.task_end - y.task_begin) / 4; });
            P.upd_stage_ndense[l] = (int)(mid - b);
            P.upd_stage_ngather[l] = (int)(mid2 - mid);
            for (auto it = b; it != mid; ++it)
                for (int q = it->task_begin; q < it->task_end; q++) {
                    const UpdTask &T = P.upd_tasks[q];
                    P.upd_stage_flops_dense[l] += 2.0 * T.nrows * T.ncols * (P.sn_first[T.src + 1] - P.sn_first[T.src]);
                }
            // gather lists of this stage: every (target entry, source) pair, grouped by target entry and, within an entry, in
            // the task order (fixed summation order).  A group's entries live in one 64-row block of one panel, so the
            // grouping is a counting sort per group over <= 64 * 64 keys (a global stable sort of the pairs was 2/3 of the
            // symbolic analysis on cfg 3 and most of the set-up time of the mid-size batch problems).
            for (auto it = mid; it != mid2; ++it) {
                const UpdGroup &G = *it;
                const int t = G.tgt, ft = P.sn_first[t], wt = P.sn_first[t + 1] - ft;
                const int64_t rt = P.sn_rowptr[t + 1] - P.sn_rowptr[t];
                // gcount is all zero between groups; only the key range a group touches is scanned and cleared again
                if (gcount.size() < (size_t)kUpdRows * kMaxSnWidth + 2) gcount.assign((size_t)kUpdRows * kMaxSnWidth + 2, 0);
                int kmin = kUpdRows * kMaxSnWidth, kmax = -1;
                auto for_pairs = [&](auto &&f) {
                    for (int q = G.task_begin; q < G.task_end; q++) {
                        const UpdTask &T = P.upd_tasks[q];
                        const int *srows = &P.sn_rows[P.sn_rowptr[T.src]];
                        const int *rel = &P.rel[T.rel_off];
                        for (int j = 0; j < T.ncols; j++) {
                            const int cpos = srows[T.col_lo + j] - ft;
                            for (int i = 0; i < T.nrows; i++) {
                                const int rpos = rel[T.row_lo + i - T.col_lo];
                                if (rpos < wt && rpos < cpos) continue;     // strictly upper part of the diagonal block
                                f((rpos - G.row_base) + kUpdRows * cpos, T, i, j);
                            }
                        }
                    }
                };
                for_pairs([&](int key, const UpdTask &, int, int) { gcount[key + 1]++; kmin = std::min(kmin, key); kmax = std::max(kmax, key); });
                const int64_t pbase = (int64_t)P.gath_src.size();
                int64_t total = 0;
                for (int k = kmin; k <= kmax; k++) {    // entries with pairs, in key order; gcount becomes the cursor
                    const int c = gcount[k + 1];
                    gcount[k + 1] = (int)total;                    // (shifted by one: gcount[k + 1] = start of key k)
                    if (c) {
                        const int rpos = G.row_base + (int)(k % kUpdRows), cpos = (int)(k / kUpdRows);
                        P.gath_tgt.push_back(P.sn_panel[t] + rpos + (int64_t)cpos * rt);
                        P.gath_pptr.push_back(pbase + total + c);
                    }
                    total += c;
                }
                P.gath_src.resize(pbase + total);
                P.gath_dj.resize(pbase + total);
                P.gath_sn.resize(pbase + total);
                for_pairs([&](int key, const UpdTask &T, int i, int j) {
                    const int64_t d = pbase + gcount[key + 1]++;
                    P.gath_src[d] = P.sn_panel[T.src] + T.row_lo + i;
                    P.gath_dj[d] = (int32_t)(T.col_lo + j - (T.row_lo + i));
                    P.gath_sn[d] = (int32_t)T.src;
                });
                for (int k = kmin; k <= kmax + 1 && k >= 0; k++) gcount[k] = 0;
                if (kmax >= 0) gcount[kmax + 1] = 0;
            }
            P.gath_stage_ptr[l + 1] = (int64_t)P.gath_tgt.size();
        }
    }

    lap("ut:lists");

    if (cancelled) return "cancelled";
    lap("ut:jit");

    if (cancelled) return "cancelled";
    if (kmap_future.valid())
        if (const char *e = kmap_future.get()) return e;
    lap("kmap-join");
    if (cancelled) return "cancelled";
    // ---- 15. gather lists for the forward solve (multifrontal style): every panel row slot
    //          (s, li) collects the update-vector entries of the CHILDREN of s that land on it.
    //          Fan-in per slot <= #children; each ubuf entry is consumed exactly once, by the parent.
    {
        const int64_t nslots = (int64_t)P.sn_rows.size();
        P.g_ptr.assign(nslots + 1, 0);
        std::vector<int> relp;  // position in the parent's row list of every off-diagonal child row
        relp.resize(P.ubuf_len);
        for (int c = 0; c < S; c++) {
            int p = P.sn_parent[c];
            if (p < 0) continue;
            int w = P.sn_first[c + 1] - P.sn_first[c];
            const int *rows = &P.sn_rows[P.sn_rowptr[c]];
            int r = (int)(P.sn_rowptr[c + 1] - P.sn_rowptr[c]);
            const int *prow = &P.sn_rows[P.sn_rowptr[p]];
            int pr = (int)(P.sn_rowptr[p + 1] - P.sn_rowptr[p]);
            int q = 0;
            for (int i = w; i < r; i++) {
                while (q < pr && prow[q] < rows[i]) q++;
                if (q == pr || prow[q] != rows[i]) return "internal: child row missing in parent structure";
                relp[P.u_off[c] + (i - w)] = q;
                P.g_ptr[P.sn_rowptr[p] + q + 1]++;
            }
        }
        for (int64_t j = 0; j < nslots; j++) P.g_ptr[j + 1] += P.g_ptr[j];
        P.g_idx.resize(P.g_ptr[nslots]);
        std::vector<int64_t> nxt(P.g_ptr.begin(), P.g_ptr.end() - 1);
        for (int c = 0; c < S; c++) {
            int p = P.sn_parent[c];
            if (p < 0) continue;
            int w = P.sn_first[c + 1] - P.sn_first[c];
            int r = (int)(P.sn_rowptr[c + 1] - P.sn_rowptr[c]);
            for (int i = w; i < r; i++) {
                int64_t up = P.u_off[c] + (i - w);
                P.g_idx[nxt[P.sn_rowptr[p] + relp[up]]++] = (int)up;
            }
        }
    }

    // ---- 15b. fronts: chains of panels cut from one wide supernode, for the persistent solve kernels
    P.sn_front.assign(S, -1);
    if (opt.front_min_panels > 0) {
        int sync_ints = 0;
        for (const auto &sp : splits) {
            const int f = sp[0], np = sp[1], cw = sp[2];
            if (np < opt.front_min_panels) continue;
            const int s0 = P.sn_of_col[f];
            const int rF = (int)(P.sn_rowptr[s0 + 1] - P.sn_rowptr[s0]);
            bool ok = true;
            int W = 0;
            for (int p = 0; p < np && ok; p++) {
                const int s = s0 + p;
                const int w = P.sn_first[s + 1] - P.sn_first[s];
                const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
                ok = P.sn_first[s] == f + cw * p && r == rF - cw * p && (p == np - 1 ? w <= cw : w == cw) &&
                     (p == 0 || P.sn_parent[s - 1] == s) && P.sn_level[s] == P.sn_level[s0] + p;
                W += w;
            }
            if (!ok) continue;   // not a nested chain (cannot happen for fundamental supernodes): stay generic
            FrontDesc F{};
            F.s0 = s0; F.np = np; F.cw = cw; F.W = W; F.rF = rF;
            F.nb = np + (rF - W + 63) / 64;
            F.level_first = P.sn_level[s0]; F.level_last = P.sn_level[s0] + np - 1;
            F.gptr_off = (int64_t)P.front_gptr.size();
            F.fp_off = (int64_t)P.front_panels.size();
            F.ubelow_off = P.u_off[s0 + np - 1];
            F.rows_off = P.sn_rowptr[s0];
            F.sync_off = sync_ints;
            // two blocks (forward sweep, backward sweep): {ticket, error, flags[np]} padded to 32 ints (128 B), then np x 64
            // hand-off slots of 16 bytes (value, self-validating tag: kernels.hip front_slot_*)
            // (super-block sweeps use a second set of np x 64 slots per block: the partial right-hand sides r / s)
            F.sync_blk = ((2 + np + 31) & ~31) + 2 * np * 64 * 4;   // multiple of 32 ints: the two blocks share no cache line
            sync_ints += 2 * F.sync_blk;
            F.sb_g = 0; F.nsb = 0; F.sbinv_off = 0;
            if (opt.superhop > 0 && np >= std::max(opt.superhop, kSbMinPanels) && np <= kSbMaxPanels) {
                F.sb_g = kSbG;
                F.nsb = (np + kSbG - 1) / kSbG;
                F.sbinv_off = P.sbinv_doubles;
                P.sbinv_doubles += (int64_t)F.nsb * (kSbG * (kSbG - 1) / 2) * 8192;
            }
            for (int p = 0; p < np; p++) {
                const int s = s0 + p;
                P.front_panels.push_back({P.sn_panel[s], P.lt_off[s], P.sn_diag[s], (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]),
                                          P.sn_first[s + 1] - P.sn_first[s], P.sn_first[s], s});
                P.sn_front[s] = (int)P.fronts.size();
            }
            // external gathers per front row, in panel order: slot (p, i - cw*p); the chain child's own
            // update vector (u_off[s-1] .. u_off[s]) is replaced by the in-kernel accumulation
            std::vector<std::vector<int>> ext(rF);
            for (int p = 0; p < np; p++) {
                const int s = s0 + p;
                const int r = rF - cw * p;
                const int64_t lo = p > 0 ? P.u_off[s - 1] : -1, hi = p > 0 ? P.u_off[s] : -1;
                for (int j = 0; j < r; j++)
                    for (int64_t g = P.g_ptr[P.sn_rowptr[s] + j]; g < P.g_ptr[P.sn_rowptr[s] + j + 1]; g++) {
                        const int u = P.g_idx[g];
                        if (u >= lo && u < hi) continue;
                        ext[cw * p + j].push_back(u);
                    }
            }
            int64_t acc = 0;   // offsets relative to this front's first entry in front_gidx are made absolute
            const int64_t base = (int64_t)P.front_gidx.size();
            for (int i = 0; i < rF; i++) {
                P.front_gptr.push_back(base + acc);
                acc += (int64_t)ext[i].size();
            }
            P.front_gptr.push_back(base + acc);
            for (int i = 0; i < rF; i++) P.front_gidx.insert(P.front_gidx.end(), ext[i].begin(), ext[i].end());
            P.fronts.push_back(F);
        }
        P.front_sync_ints = sync_ints;
    }

    lap("solve-lists");

    if (cancelled) return "cancelled";
    // ---- 16. symmetric CSR view of K in the ORIGINAL ordering (iterative-refinement SpMV)
    //      minus the dense triangles (symbolic.h HostPlan::dtri): greedy from the left, a triangle starts at c0 and grows while column
    //      c0 + j holds >= j + 1 entries, ends on its diagonal and has row c0 exactly j entries before it (rows sorted and distinct: the
    //      j + 1 last entries are then c0 .. c0 + j)
    P.dtri.clear();
    P.dtri_col.clear();
    std::vector<int> tri_c0(N, -1);       // per column: first row of its triangle, -1 outside
    if (opt.dense_tri_first_col >= 0) {
        auto grows = [&](int c0, int j) {
            const int c = c0 + j;
            return c < N && Ap[c + 1] - Ap[c] >= j + 1 && Ai[Ap[c + 1] - 1] == c && Ai[Ap[c + 1] - 1 - j] == c0;
        };
        for (int c = std::max(0, opt.dense_tri_first_col); c < N;) {
            int d = 0;
            while (grows(c, d)) d++;
            if (d >= std::max(2, opt.dense_tri_min_dim)) {
                P.dtri.push_back(DenseTri{c, d, (int64_t)P.dtri_col.size()});
                for (int j = 0; j < d; j++) {
                    tri_c0[c + j] = c;
                    P.dtri_col.push_back(Ap[c + j + 1] - 1 - j);
                }
                c += d;
            } else {
                c++;
            }
        }
    }
    auto in_tri = [&](int i, int j) { return tri_c0[j] >= 0 && i >= tri_c0[j]; };   // (i <= j: column j's rows from c0 on are the triangle's)
    P.sym_rowptr.assign(N + 1, 0);
    for (int j = 0; j < N; j++)
        for (int64_t q = Ap[j]; q < Ap[j + 1]; q++) {
            int i = (int)Ai[q];
            if (in_tri(i, j)) continue;
            P.sym_rowptr[i + 1]++;
            if (i != j) P.sym_rowptr[j + 1]++;
        }
    for (int j = 0; j < N; j++) P.sym_rowptr[j + 1] += P.sym_rowptr[j];
    P.sym_col.resize(P.sym_rowptr[N]);
    P.sym_q.resize(P.sym_rowptr[N]);
    {
        std::vector<int64_t> nxt(P.sym_rowptr.begin(), P.sym_rowptr.end() - 1);
        for (int j = 0; j < N; j++)
            for (int64_t q = Ap[j]; q < Ap[j + 1]; q++) {
                int i = (int)Ai[q];
                if (in_tri(i, j)) continue;
                P.sym_col[nxt[i]] = j; P.sym_q[nxt[i]++] = q;
                if (i != j) { P.sym_col[nxt[j]] = i; P.sym_q[nxt[j]++] = q; }
            }
    }
    lap("symcsr");
    if (cancelled) return "cancelled";
    return "";
}

std::vector<FrontBatchHost> front_batches(const HostPlan &P, int update_policy, int max_nb) {
    std::vector<FrontBatchHost> out;
    const int Bu = P.update_batch_used;
    if (update_policy != 2 || Bu < 2 || Bu > max_nb) return out;
    for (size_t fi = 0; fi < P.fronts.size(); fi++) {
        const FrontDesc &F = P.fronts[fi];
        int p = 0;
        while (p < F.np) {
            const int win = (F.level_first + p) / Bu;
            int q = p;
            while (q < F.np && (F.level_first + q) / Bu == win) q++;
            bool ok = q - p >= 2 && F.cw == 64;
            for (int t = p; t < q && ok; t++) {
                const FrontPanel &fp = P.front_panels[F.fp_off + t];
                const int l = F.level_first + t;
                ok = fp.w == 64 && P.sn_level[fp.sn] == l && P.lvl_ptr[l + 1] - P.lvl_ptr[l] == 1 && P.lvl_sn[P.lvl_ptr[l]] == fp.sn &&
                     fp.r == P.front_panels[F.fp_off + p].r - 64 * (t - p);
                // stage l (between panel t and t + 1): only contributions of this batch's panels to panel t + 1, all dense tiles
                if (ok && t + 1 < q) {
                    const int tgt = P.front_panels[F.fp_off + t + 1].sn, smin = P.front_panels[F.fp_off + p].sn;
                    ok = P.upd_stage_ndense[l] == P.upd_stage_ptr[l + 1] - P.upd_stage_ptr[l];
                    for (int g = P.upd_stage_ptr[l]; g < P.upd_stage_ptr[l + 1] && ok; g++) {
                        ok = P.upd_groups[g].tgt == tgt;
                        for (int u = P.upd_groups[g].task_begin; u < P.upd_groups[g].task_end && ok; u++)
                            ok = P.upd_tasks[u].src >= smin && P.upd_tasks[u].src < tgt && P.sn_front[P.upd_tasks[u].src] == P.sn_front[tgt];
                    }
                }
            }
            if (ok) out.push_back({(int)fi, p, q - p, F.level_first + p, F.level_first + q - 1});
            p = q;
        }
    }
    return out;
}

}  // namespace hipkkt
