// Fill-reducing ordering for the KKT matrix: approximate minimum degree on the quotient graph
// (Amestoy/Davis/Duff-style: element absorption, approximate external degrees, mass elimination,
// hash-based supervariable detection, dense-row deferral).  Host code, runs once per problem
// (SURVEY.md §2 kernel K10; the reference gets its ordering from SuiteSparse AMD through QDLDL.jl
// with amd_dense_scale = 1.5, src/kktsolvers/direct-ldl/directldl_qdldl.jl:18-25).
// Own implementation written for this project; any AMD-class order only changes rounding.
#include "symbolic.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace hipkkt {

namespace {

struct DegreeLists {
    std::vector<int> head, next, prev;
    int mindeg;
    explicit DegreeLists(int n) : head(n + 1, -1), next(n, -1), prev(n, -1), mindeg(n) {}
    void insert(int i, int d) {
        next[i] = head[d];
        prev[i] = -1;
        if (head[d] >= 0) prev[head[d]] = i;
        head[d] = i;
        if (d < mindeg) mindeg = d;
    }
    void remove(int i, int d) {
        if (prev[i] >= 0) next[prev[i]] = next[i];
        else head[d] = next[i];
        if (next[i] >= 0) prev[next[i]] = prev[i];
        next[i] = prev[i] = -1;
    }
};

enum : unsigned char { ST_VAR = 0, ST_ELEM = 1, ST_DEAD = 2, ST_DENSE = 3 };

}  // namespace

// Ap/Ai: upper-triangular (or any) CSC pattern of a symmetric matrix, 0-based; diagonal ignored.
// hold (optional, size n): nodes with hold[i] != 0 are not eligible as pivots until every other node
// has been eliminated ("variables last": the x-block of the KKT matrix is ordered after the cone
// rows, i.e. the Schur-complement / normal-equations order, still with minimum-degree inside each part)
void amd_order(int n, const int64_t *Ap, const int64_t *Ai, double dense_scale, std::vector<int> &perm, const char *hold) {
    perm.clear();
    perm.reserve(n);
    if (n == 0) return;
    // symmetric adjacency without the diagonal
    std::vector<std::vector<int>> vadj(n), eadj(n), evars(n), members(n);
    {
        std::vector<int> cnt(n, 0);
        for (int j = 0; j < n; j++)
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                int i = (int)Ai[p];
                if (i != j) { cnt[i]++; cnt[j]++; }
            }
        for (int i = 0; i < n; i++) vadj[i].reserve(cnt[i]);
        for (int j = 0; j < n; j++)
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                int i = (int)Ai[p];
                if (i != j) { vadj[i].push_back(j); vadj[j].push_back(i); }
            }
        for (int i = 0; i < n; i++) {  // duplicates (if both triangles were given)
            auto &v = vadj[i];
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        }
    }
    std::vector<unsigned char> state(n, ST_VAR);
    std::vector<int> nv(n, 1), deg(n, 0), esize(n, 0), w(n, 0);
    std::vector<int> mark(n, 0), wstamp(n, 0), cmp(n, 0);
    std::vector<unsigned> hashv(n, 0);
    int tag = 0, cmptag = 0;

    // dense rows are deferred to the end (AMD's "dense" control, default 10*sqrt(n), scaled)
    if (dense_scale <= 0) dense_scale = 1.0;
    double thr = std::max(16.0, dense_scale * 10.0 * std::sqrt((double)n));
    std::vector<int> dense_nodes;
    for (int i = 0; i < n; i++)
        if ((double)vadj[i].size() > thr) { state[i] = ST_DENSE; dense_nodes.push_back(i); }
    int nlive = n - (int)dense_nodes.size();
    DegreeLists dl(n);
    std::vector<char> held(n, 0), inlist(n, 0);
    bool holding = false;
    for (int i = 0; i < n; i++) {
        if (state[i] != ST_VAR) continue;
        int d = 0;
        for (int j : vadj[i]) d += (state[j] == ST_VAR);
        deg[i] = d;
        if (hold && hold[i]) { held[i] = 1; holding = true; continue; }
        dl.insert(i, d);
        inlist[i] = 1;
    }

    std::vector<int> Lp, newE, newA, bucket;
    int nel = 0;
    while (nel < nlive) {
        while (dl.mindeg < n && dl.head[dl.mindeg] < 0) dl.mindeg++;
        if (dl.mindeg >= n) {
            if (!holding) break;       // cannot happen: nel < nlive implies a live variable exists
            holding = false;           // phase 2: release the held variables with their current degrees
            dl.mindeg = n;
            for (int i = 0; i < n; i++)
                if (held[i] && state[i] == ST_VAR && nv[i] > 0) { dl.insert(i, deg[i]); inlist[i] = 1; }
            std::fill(held.begin(), held.end(), 0);
            continue;
        }
        int p = dl.head[dl.mindeg];
        dl.remove(p, deg[p]);
        inlist[p] = 0;
        // ---- form the new element Lp
        tag++;
        mark[p] = tag;
        Lp.clear();
        for (int i : vadj[p])
            if (state[i] == ST_VAR && nv[i] > 0 && mark[i] != tag) { mark[i] = tag; Lp.push_back(i); }
        for (int e : eadj[p]) {
            if (state[e] != ST_ELEM) continue;
            for (int i : evars[e])
                if (state[i] == ST_VAR && nv[i] > 0 && mark[i] != tag) { mark[i] = tag; Lp.push_back(i); }
            state[e] = ST_DEAD;  // absorbed into p
            std::vector<int>().swap(evars[e]);
        }
        std::vector<int>().swap(vadj[p]);
        std::vector<int>().swap(eadj[p]);
        state[p] = ST_ELEM;
        nel += nv[p];
        perm.push_back(p);
        int degme = 0;
        for (int i : Lp) degme += nv[i];

        // ---- w[e] = |Le \ Lp| for every element touching Lp
        for (int i : Lp)
            for (int e : eadj[i]) {
                if (state[e] != ST_ELEM) continue;
                if (wstamp[e] != tag) { wstamp[e] = tag; w[e] = esize[e]; }
                w[e] -= nv[i];
            }
        // ---- update the variables of Lp
        for (int i : Lp) {
            if (inlist[i]) { dl.remove(i, deg[i]); inlist[i] = 0; }
            newE.clear();
            newA.clear();
            int dsum = 0;
            unsigned h = 0;
            for (int e : eadj[i]) {
                if (state[e] != ST_ELEM) continue;
                if (w[e] <= 0) {  // Le is a subset of Lp: aggressive absorption
                    state[e] = ST_DEAD;
                    std::vector<int>().swap(evars[e]);
                } else {
                    newE.push_back(e);
                    dsum += w[e];
                    h += (unsigned)e;
                }
            }
            newE.push_back(p);
            h += (unsigned)p;
            for (int j : vadj[i])
                if (state[j] == ST_VAR && nv[j] > 0 && mark[j] != tag) {
                    newA.push_back(j);
                    dsum += nv[j];
                    h += (unsigned)j;
                }
            eadj[i].assign(newE.begin(), newE.end());
            vadj[i].assign(newA.begin(), newA.end());
            if (newE.size() == 1 && newA.empty()) {
                // mass elimination: i is indistinguishable from p
                members[p].push_back(i);
                for (int q : members[i]) members[p].push_back(q);
                std::vector<int>().swap(members[i]);
                degme -= nv[i];
                nel += nv[i];
                nv[i] = 0;
                state[i] = ST_DEAD;
                std::vector<int>().swap(eadj[i]);
            } else {
                deg[i] = std::min(deg[i], dsum);  // partial; the |Lp \ i| term is added below
                hashv[i] = h;
            }
        }
        // ---- supervariable detection inside Lp
        bucket.clear();
        for (int i : Lp)
            if (nv[i] > 0) bucket.push_back(i);
        std::sort(bucket.begin(), bucket.end(), [&](int a, int b) {
            return hashv[a] != hashv[b] ? hashv[a] < hashv[b] : a < b;
        });
        for (size_t a = 0; a < bucket.size(); a++) {
            int i = bucket[a];
            if (nv[i] == 0) continue;
            bool stamped = false;
            for (size_t b = a + 1; b < bucket.size() && hashv[bucket[b]] == hashv[i]; b++) {
                int j = bucket[b];
                if (nv[j] == 0) continue;
                if (eadj[j].size() != eadj[i].size() || vadj[j].size() != vadj[i].size()) continue;
                if (!stamped) {
                    cmptag++;
                    for (int e : eadj[i]) cmp[e] = cmptag;
                    for (int v : vadj[i]) cmp[v] = cmptag;
                    stamped = true;
                }
                bool same = true;
                for (int e : eadj[j])
                    if (cmp[e] != cmptag) { same = false; break; }
                if (same)
                    for (int v : vadj[j])
                        if (cmp[v] != cmptag) { same = false; break; }
                if (!same) continue;
                // merge j into i
                members[i].push_back(j);
                for (int q : members[j]) members[i].push_back(q);
                std::vector<int>().swap(members[j]);
                nv[i] += nv[j];
                nv[j] = 0;
                state[j] = ST_DEAD;
                std::vector<int>().swap(eadj[j]);
                std::vector<int>().swap(vadj[j]);
            }
        }
        // ---- finalise element p and restore the degree lists
        auto &ev = evars[p];
        ev.clear();
        int nleft = nlive - nel;
        for (int i : Lp) {
            if (nv[i] == 0) continue;
            ev.push_back(i);
            int d = deg[i] + degme - nv[i];
            d = std::min(d, nleft - nv[i]);
            if (d < 0) d = 0;
            deg[i] = d;
            if (!(holding && held[i])) { dl.insert(i, d); inlist[i] = 1; }
        }
        esize[p] = degme;
        if (ev.empty()) state[p] = ST_DEAD;
    }
    // expand supervariables: principal pivot followed by everything merged into it
    std::vector<int> out;
    out.reserve(n);
    for (int p : perm) {
        out.push_back(p);
        for (int q : members[p]) out.push_back(q);
    }
    // dense rows last, sparsest first
    std::sort(dense_nodes.begin(), dense_nodes.end(),
              [&](int a, int b) { return vadj[a].size() != vadj[b].size() ? vadj[a].size() < vadj[b].size() : a < b; });
    for (int d : dense_nodes) out.push_back(d);
    perm.swap(out);
}


// ------------------------------------------------------------------------------------------------------------
// Nested dissection by breadth-first level structures (George's automatic nested dissection), minimum degree on the
// leaves.  Why it exists: on the GPU a factorisation / triangular solve costs (levels of the supernodal elimination
// tree) x (a few microseconds of dependency latency) + flops / throughput.  Minimum degree minimises the second
// term only; on banded / grid-like KKT systems (cfg 2b, the Maros-Meszaros-like batch) it produces a CHAIN of ~150
// dependent supernodes where dissection gives a balanced tree of depth ~log2(N / leaf) with separators of the size
// of the bandwidth.  build_plan() evaluates both orders with a latency + throughput model and keeps the cheaper one,
// so problems without small separators (cfg 2a: uniform random pattern) keep the minimum-degree order.
// Host code, once per problem; any order only changes rounding.
// ------------------------------------------------------------------------------------------------------------
namespace {

struct NDGraph {
    int n;
    std::vector<int64_t> xadj;
    std::vector<int> adj;
    std::vector<int> region;     // current region id of every node (-1 = already ordered / separator)
    std::vector<int> lvl;        // scratch: BFS level
    std::vector<int> queue;
    double dense_scale;
    int leaf_size;
    int next_region = 1;
};

// BFS over the nodes of region `rid` from `root`; fills G.queue (visit order) and G.lvl; returns the number of levels
int nd_bfs(NDGraph &G, int rid, int root, std::vector<int> &lvl_ptr) {
    G.queue.clear();
    lvl_ptr.clear();
    G.queue.push_back(root);
    G.lvl[root] = 0;
    lvl_ptr.push_back(0);
    size_t head = 0;
    int cur = 0;
    while (head < G.queue.size()) {
        const int v = G.queue[head];
        if (G.lvl[v] != cur) { cur = G.lvl[v]; lvl_ptr.push_back((int)head); }
        head++;
        for (int64_t p = G.xadj[v]; p < G.xadj[v + 1]; p++) {
            const int u = G.adj[p];
            if (G.region[u] == rid && G.lvl[u] < 0) { G.lvl[u] = cur + 1; G.queue.push_back(u); }
        }
    }
    lvl_ptr.push_back((int)G.queue.size());
    return (int)lvl_ptr.size() - 1;
}

void nd_leaf(NDGraph &G, const std::vector<int> &nodes, std::vector<int> &out) {
    // minimum degree on the induced subgraph (upper-triangular CSC in local numbering)
    const int nl = (int)nodes.size();
    if (nl <= 2) { out.insert(out.end(), nodes.begin(), nodes.end()); return; }
    std::vector<int> &loc = G.lvl;       // scratch reuse: local index of a node, restored to -1 below
    for (int i = 0; i < nl; i++) loc[nodes[i]] = i;
    std::vector<int64_t> cp(nl + 1, 0);
    std::vector<int64_t> ci;
    for (int j = 0; j < nl; j++) {
        const int v = nodes[j];
        for (int64_t p = G.xadj[v]; p < G.xadj[v + 1]; p++) {
            const int u = G.adj[p];
            if (loc[u] >= 0 && loc[u] < j && nodes[loc[u]] == u) ci.push_back(loc[u]);
        }
        ci.push_back(j);
        cp[j + 1] = (int64_t)ci.size();
    }
    for (int i = 0; i < nl; i++) loc[nodes[i]] = -1;
    std::vector<int> lp;
    amd_order(nl, cp.data(), ci.data(), G.dense_scale, lp, nullptr);
    for (int k : lp) out.push_back(nodes[k]);
}

void nd_dissect(NDGraph &G, std::vector<int> nodes, std::vector<int> &out, int depth) {
    const int nn = (int)nodes.size();
    if (nn <= G.leaf_size || depth > 40) { nd_leaf(G, nodes, out); return; }
    const int rid = G.next_region++;
    for (int v : nodes) { G.region[v] = rid; G.lvl[v] = -1; }
    // connected components first: independent subtrees
    std::vector<int> lvl_ptr;
    {
        int nlev = nd_bfs(G, rid, nodes[0], lvl_ptr);
        (void)nlev;
        if ((int)G.queue.size() < nn) {
            std::vector<std::vector<int>> comps;
            comps.emplace_back(G.queue);
            for (int v : nodes)
                if (G.lvl[v] < 0) { nd_bfs(G, rid, v, lvl_ptr); comps.emplace_back(G.queue); }
            for (int v : nodes) G.lvl[v] = -1;
            for (auto &c : comps) nd_dissect(G, std::move(c), out, depth + 1);
            return;
        }
    }
    // pseudo-peripheral root: repeat BFS from a minimum-degree node of the last level while the depth grows
    int root = nodes[0], nlev = (int)lvl_ptr.size() - 1;
    for (int it = 0; it < 4; it++) {
        int best = -1;
        int64_t bestdeg = INT64_MAX;
        for (int q = lvl_ptr[nlev - 1]; q < lvl_ptr[nlev]; q++) {
            const int v = G.queue[q];
            const int64_t d = G.xadj[v + 1] - G.xadj[v];
            if (d < bestdeg) { bestdeg = d; best = v; }
        }
        for (int v : nodes) G.lvl[v] = -1;
        std::vector<int> lp2;
        const int nl2 = nd_bfs(G, rid, best, lp2);
        const bool better = nl2 > nlev;
        root = best; nlev = nl2; lvl_ptr.swap(lp2);
        if (!better) break;
    }
    (void)root;
    if (nlev < 5) { for (int v : nodes) G.lvl[v] = -1; nd_leaf(G, nodes, out); return; }   // no usable level structure
    // separator level: the smallest level among those that leave 30..70 % of the nodes on the near side
    int m = -1;
    int64_t msize = INT64_MAX;
    for (int l = 1; l + 1 < nlev; l++) {
        const int before = lvl_ptr[l], size = lvl_ptr[l + 1] - lvl_ptr[l];
        if (before < 0.3 * nn || before + size > 0.7 * nn) continue;
        if (size < msize) { msize = size; m = l; }
    }
    if (m < 0) {   // one huge level around the middle: take the level that contains the median node
        for (int l = 1; l + 1 < nlev; l++)
            if (lvl_ptr[l + 1] > nn / 2) { m = l; break; }
        if (m < 0) m = nlev / 2;
        msize = lvl_ptr[m + 1] - lvl_ptr[m];
    }
    if (msize > 0.2 * nn) { for (int v : nodes) G.lvl[v] = -1; nd_leaf(G, nodes, out); return; }   // separators too fat: minimum degree does better
    std::vector<int> A, B, S;
    A.reserve(lvl_ptr[m + 1]); B.reserve(nn - lvl_ptr[m + 1]);
    for (int q = 0; q < lvl_ptr[m]; q++) A.push_back(G.queue[q]);
    for (int q = lvl_ptr[m]; q < lvl_ptr[m + 1]; q++) {      // thin the level: only nodes that touch the far side separate
        const int v = G.queue[q];
        bool touches = false;
        for (int64_t p = G.xadj[v]; p < G.xadj[v + 1] && !touches; p++) {
            const int u = G.adj[p];
            touches = G.region[u] == rid && G.lvl[u] == m + 1;
        }
        (touches ? S : A).push_back(v);
    }
    for (int q = lvl_ptr[m + 1]; q < nn; q++) B.push_back(G.queue[q]);
    for (int v : nodes) G.lvl[v] = -1;
    for (int v : S) G.region[v] = -1;
    nd_dissect(G, std::move(A), out, depth + 1);
    nd_dissect(G, std::move(B), out, depth + 1);
    out.insert(out.end(), S.begin(), S.end());
}

}  // namespace

void nd_order(int n, const int64_t *Ap, const int64_t *Ai, double dense_scale, int leaf_size, std::vector<int> &perm) {
    perm.clear();
    perm.reserve(n);
    if (n == 0) return;
    NDGraph G;
    G.n = n;
    G.dense_scale = dense_scale;
    G.leaf_size = std::max(8, leaf_size);
    G.xadj.assign(n + 1, 0);
    for (int j = 0; j < n; j++)
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            const int i = (int)Ai[p];
            if (i != j) { G.xadj[i + 1]++; G.xadj[j + 1]++; }
        }
    for (int i = 0; i < n; i++) G.xadj[i + 1] += G.xadj[i];
    G.adj.resize(G.xadj[n]);
    {
        std::vector<int64_t> nxt(G.xadj.begin(), G.xadj.end() - 1);
        for (int j = 0; j < n; j++)
            for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
                const int i = (int)Ai[p];
                if (i != j) { G.adj[nxt[i]++] = j; G.adj[nxt[j]++] = i; }
            }
    }
    G.region.assign(n, 0);
    G.lvl.assign(n, -1);
    // very dense rows would glue every level structure together: ordered last, like amd_order does
    std::vector<int> nodes, dense;
    const double thresh = std::max(16.0, dense_scale * 10.0 * std::sqrt((double)n));
    for (int i = 0; i < n; i++) {
        if ((double)(G.xadj[i + 1] - G.xadj[i]) > thresh) { dense.push_back(i); G.region[i] = -1; }
        else nodes.push_back(i);
    }
    nd_dissect(G, std::move(nodes), perm, 0);
    std::sort(dense.begin(), dense.end(), [&](int a, int b) {
        const int64_t da = G.xadj[a + 1] - G.xadj[a], db = G.xadj[b + 1] - G.xadj[b];
        return da != db ? da < db : a < b;
    });
    perm.insert(perm.end(), dense.begin(), dense.end());
}

}  // namespace hipkkt
