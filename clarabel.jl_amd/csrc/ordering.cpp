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

}  // namespace hipkkt
