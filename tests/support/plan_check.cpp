// TEST SUPPORT (not product): serial host interpreter of the symbolic plan produced by
#include <cstdio>
#include <algorithm>
// clarabel.jl_amd/csrc/symbolic.cpp.  It executes exactly the work lists the HIP kernels interpret
// (scatter map, per-level factor items, target-owned update tasks, gather lists) so that the index
// structures can be validated against dense linear algebra on a machine without a GPU.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../clarabel.jl_amd/csrc/symbolic.h"

using namespace hipkkt;

// ---- host twins of the super-block front sweeps (clarabel.jl_amd/csrc/front_sweep.hip): the same tile addresses, layouts and
//      summation structure as k_invert_super / k_front_fwd_sb / k_front_bwd_sb, executed serially
namespace {
struct SbEmu {
    const HostPlan &P;
    const std::vector<double> &Lx;
    std::vector<double> LT, Linv, LinvT, SbInv;
    static int64_t tile(const FrontDesc &F, int B, int bl, int cl) {
        return F.sbinv_off + ((int64_t)B * (kSbG * (kSbG - 1) / 2) + bl * (bl - 1) / 2 + cl) * 8192;
    }
    SbEmu(const HostPlan &P_, const std::vector<double> &Lx_, const std::vector<double> &Ld) : P(P_), Lx(Lx_) {
        LT.assign((size_t)P.lt_off[P.nsuper], 0.0);
        Linv.assign((size_t)P.diag_doubles, 0.0);
        LinvT.assign((size_t)P.diag_doubles, 0.0);
        SbInv.assign((size_t)std::max<int64_t>(P.sbinv_doubles, 1), 0.0);
        for (int s = 0; s < P.nsuper; s++) {
            const int w = P.sn_first[s + 1] - P.sn_first[s], r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
            const double *pan = &Lx[P.sn_panel[s]];
            double *lt = &LT[P.lt_off[s]];
            for (int j = w; j < r; j++)
                for (int k = 0; k < w; k++) lt[(size_t)(j - w) * w + k] = pan[j + (size_t)k * r];
            // explicit inverse of the unit-lower diagonal block (column by column), both layouts of k_invert_diag
            const double *ld = &Ld[P.sn_diag[s]];
            std::vector<double> X((size_t)w * w, 0.0);
            for (int j = 0; j < w; j++) {
                X[j + (size_t)j * w] = 1.0;
                for (int k = j; k < w; k++)
                    for (int i = k + 1; i < w; i++) X[i + (size_t)j * w] -= ld[i + (size_t)k * w] * X[k + (size_t)j * w];
            }
            for (int j = 0; j < w; j++)
                for (int i = 0; i < w; i++) {
                    Linv[P.sn_diag[s] + i + (size_t)j * w] = X[i + (size_t)j * w];
                    LinvT[P.sn_diag[s] + j + (size_t)i * w] = X[i + (size_t)j * w];
                }
        }
    }
    void invert_super(const FrontDesc &F) {
        const FrontPanel *fps = &P.front_panels[F.fp_off];
        for (int B = 0; B < F.nsb; B++) {
            const int nbB = std::min(kSbG, F.np - kSbG * B);
            for (int cl = 0; cl + 1 < nbB; cl++) {
                const FrontPanel pc = fps[kSbG * B + cl];
                for (int bl = cl + 1; bl < nbB; bl++) {
                    const FrontPanel pb = fps[kSbG * B + bl];
                    std::vector<double> S(4096, 0.0), O(4096, 0.0);      // [i * 64 + k]
                    for (int kl = cl; kl < bl; kl++) {
                        const FrontPanel pk = fps[kSbG * B + kl];
                        const double *src = &Lx[pk.panel_off + F.cw * (bl - kl)];
                        for (int i = 0; i < 64; i++)
                            for (int k = 0; k < 64; k++) {
                                double a = 0;
                                for (int m = 0; m < 64; m++) {
                                    const double l = (i < pb.w && m < pk.w) ? src[i + (size_t)m * pk.r] : 0.0;
                                    double bv;
                                    if (kl == cl) bv = (m < pc.w && k <= m) ? Linv[pc.diag_off + m + (size_t)k * pc.w] : 0.0;
                                    else bv = SbInv[tile(F, B, kl, cl) + m + 64 * k];
                                    a += l * bv;
                                }
                                S[i * 64 + k] += a;
                            }
                    }
                    for (int i = 0; i < 64; i++)
                        for (int k = 0; k < 64; k++) {
                            double a = 0;
                            for (int m = 0; m < 64; m++) a += ((i < pb.w && m <= i) ? Linv[pb.diag_off + i + (size_t)m * pb.w] : 0.0) * S[m * 64 + k];
                            O[i * 64 + k] = -a;
                        }
                    double *t = &SbInv[tile(F, B, bl, cl)];
                    for (int i = 0; i < 64; i++)
                        for (int k = 0; k < 64; k++) { t[i + 64 * k] = O[i * 64 + k]; t[4096 + i * 64 + k] = O[i * 64 + k]; }
                }
            }
        }
    }
    // y: permuted right-hand side (own rows are overwritten with the solved y_J, like the serial sweep); ub: update vectors
    void fwd(const FrontDesc &F, std::vector<double> &y, std::vector<double> &ub) {
        const FrontPanel *fps = &P.front_panels[F.fp_off];
        const int g = kSbG;
        std::vector<double> ysol((size_t)F.np * 64, 0.0), rvec((size_t)F.np * 64, 0.0);
        for (int b = 0; b < F.nb; b++) {
            const bool own = b < F.np;
            const FrontPanel me = fps[own ? b : 0];
            const int B = own ? b / g : F.nsb, bl = own ? b - g * B : 0;
            const int i0 = own ? F.cw * b : F.W + 64 * (b - F.np);
            const int nrows = own ? me.w : std::min(64, F.rF - i0);
            const int nQ = own ? B : F.nsb;
            for (int lane = 0; lane < nrows; lane++) {
                const int i = i0 + lane;
                double G = 0;
                for (int64_t gq = P.front_gptr[F.gptr_off + i]; gq < P.front_gptr[F.gptr_off + i + 1]; gq++) G += ub[P.front_gidx[gq]];
                const double base = own ? y[me.f + lane] - G : G;
                double tot = 0;
                for (int Q = 0; Q < nQ; Q++)
                    for (int p = 0; p < g; p++) {
                        const int q = g * Q + p;
                        if (q >= F.np) continue;
                        const FrontPanel fq = fps[q];
                        for (int k = 0; k < fq.w; k++) tot += Lx[fq.panel_off + (i - F.cw * q) + (int64_t)k * fq.r] * ysol[(size_t)q * 64 + k];
                    }
                if (!own) ub[F.ubelow_off + (i - F.W)] = base + tot;
                else rvec[(size_t)b * 64 + lane] = base - tot;
            }
            if (!own) continue;
            for (int lane = 0; lane < nrows; lane++) {
                double v = 0;
                for (int p = 0; p <= bl; p++)
                    for (int k = 0; k < 64; k++) {
                        double iv;
                        if (p < bl) iv = SbInv[tile(F, B, bl, p) + lane + 64 * k];
                        else iv = (k <= lane && k < me.w) ? Linv[me.diag_off + lane + (size_t)k * me.w] : 0.0;
                        v += iv * rvec[(size_t)(g * B + p) * 64 + k];
                    }
                ysol[(size_t)b * 64 + lane] = v;
            }
            for (int lane = 0; lane < nrows; lane++) y[me.f + lane] = ysol[(size_t)b * 64 + lane];
        }
    }
    // z: D^-1 y (permuted); x: the permuted solution, final on every row outside the front's own columns
    void bwd(const FrontDesc &F, const std::vector<double> &z, std::vector<double> &x) {
        const FrontPanel *fps = &P.front_panels[F.fp_off];
        const int g = kSbG;
        const int *rows = &P.sn_rows[F.rows_off];
        std::vector<double> xsol((size_t)F.np * 64, 0.0), svec((size_t)F.np * 64, 0.0);
        for (int p = F.np - 1; p >= 0; p--) {
            const FrontPanel me = fps[p];
            const int w = me.w, B = p / g, pl = p - g * B, nbB = std::min(g, F.np - g * B);
            const double *lt = &LT[me.lt_off];
            for (int lane = 0; lane < w; lane++) {
                double a = 0;
                for (int i = F.W; i < F.rF; i++) a += lt[(size_t)(i - F.cw * p - w) * w + lane] * x[rows[i]];
                for (int Q = F.nsb - 1; Q > B; Q--)
                    for (int pp = 0; pp < g; pp++) {
                        const int q = g * Q + pp;
                        if (q >= F.np) continue;
                        for (int jr = 0; jr < fps[q].w; jr++) a += lt[(size_t)(F.cw * (q - p) + jr - w) * w + lane] * xsol[(size_t)q * 64 + jr];
                    }
                svec[(size_t)p * 64 + lane] = z[me.f + lane] - a;
            }
            for (int lane = 0; lane < w; lane++) {
                double v = 0;
                for (int cl = pl; cl < nbB; cl++)
                    for (int i2 = 0; i2 < 64; i2++) {
                        double iv;
                        if (cl > pl) iv = SbInv[tile(F, B, cl, pl) + 4096 + i2 * 64 + lane];
                        else iv = (i2 >= lane && i2 < w) ? LinvT[me.diag_off + lane + (size_t)i2 * w] : 0.0;
                        v += iv * svec[(size_t)(g * B + cl) * 64 + i2];
                    }
                xsol[(size_t)p * 64 + lane] = v;
            }
            for (int lane = 0; lane < w; lane++) x[me.f + lane] = xsol[(size_t)p * 64 + lane];
        }
    }
};
}  // namespace

extern "C" {

// y = K x of the symmetric K the way the refinement's SpMV kernels take it: the symmetric view (without the dense triangles) row by row
// plus, per dense triangle, the row part and the column part of k_spmv_dense_tri (kernels.hip) from the packed columns alone.
// stats: [0] triangles found [1] entries left in the view [2] sum of the triangles' dimensions
int plan_check_symmetric_product(int64_t N, const int64_t *Ap, const int64_t *Ai, const double *Ax, int first_col, int min_dim,
                                 const double *x, double *y, double *stats) {
    HostPlan P;
    PlanOptions opt;
    opt.dense_tri_first_col = first_col;
    opt.dense_tri_min_dim = min_dim;
    std::string err = build_plan((int)N, Ap, Ai, nullptr, opt, P);
    if (!err.empty()) { fprintf(stderr, "build_plan: %s\n", err.c_str()); return -1; }
    for (int i = 0; i < N; i++) {
        double a = 0;
        for (int64_t p = P.sym_rowptr[i]; p < P.sym_rowptr[i + 1]; p++) a += Ax[P.sym_q[p]] * x[P.sym_col[p]];
        y[i] = a;
    }
    int64_t dims = 0;
    for (const DenseTri &T : P.dtri) {
        const int64_t *co = P.dtri_col.data() + T.col0;
        dims += T.d;
        for (int i = 0; i < T.d; i++) {
            double a = 0;
            for (int j = i; j < T.d; j++) a += Ax[co[j] + i] * x[T.c0 + j];     // row part
            for (int r = 0; r < i; r++) a += Ax[co[i] + r] * x[T.c0 + r];       // column part
            y[T.c0 + i] += a;
        }
    }
    stats[0] = (double)P.dtri.size(); stats[1] = (double)P.sym_rowptr[N]; stats[2] = (double)dims;
    return 0;
}

// stats: [0] nsuper [1] nlevels [2] nnzL [3] panel_doubles [4] ntasks [5] ngroups [6] etree_height
//        [7] flops_colcount [8] flops_update [9] flops_exec [10] nreg [11] max group tasks
int plan_check_run(int64_t N, const int64_t *Ap, const int64_t *Ai, const double *Ax, const int64_t *dsigns,
                   const int64_t *user_perm, int max_width, int relax, int policy, double reg_eps,
                   double reg_delta, const double *b, double *x, int64_t *perm_out, double *stats,
                   int symbolic_only) {
    HostPlan P;
    PlanOptions opt;
    opt.max_width = max_width; opt.relax = relax != 0; opt.update_policy = policy & 15; if (policy >> 4) opt.update_batch = policy >> 4;
    { const char *sh = getenv("PLANCHECK_SUPERHOP"); if (sh) opt.superhop = atoi(sh); }
    { const char *fm = getenv("PLANCHECK_FRONT_MIN"); if (fm) opt.front_min_panels = atoi(fm); }
    std::string err = build_plan((int)N, Ap, Ai, user_perm, opt, P);
    if (!err.empty()) { fprintf(stderr, "build_plan: %s\n", err.c_str()); return -1; }
    if (perm_out) for (int k = 0; k < N; k++) perm_out[k] = P.perm[k];
    size_t maxg = 0;
    for (auto &g : P.upd_groups) maxg = std::max(maxg, (size_t)(g.task_end - g.task_begin));
    stats[0] = P.nsuper; stats[1] = P.nlevels; stats[2] = (double)P.nnzL; stats[3] = (double)P.panel_doubles;
    stats[4] = (double)P.upd_tasks.size(); stats[5] = (double)P.upd_groups.size(); stats[6] = P.etree_height;
    stats[7] = P.flops_colcount; stats[8] = P.flops_update; stats[9] = P.flops_exec; stats[10] = 0; stats[11] = (double)maxg;
    {
        size_t nmapped = 0, ndense = 0;
        for (auto &t : P.upd_tasks) nmapped += (t.geom >> 17) & 1;
        for (auto &g : P.upd_groups) ndense += g.dense == 1;
        int nsbf = 0;
        for (auto &F : P.fronts) nsbf += F.sb_g > 0;
        if (getenv("PLANCHECK_VERBOSE")) fprintf(stderr, "plan_check: %zu fronts, %d with super-block sweeps\n", P.fronts.size(), nsbf);
        stats[12] = (double)P.fronts.size(); stats[13] = (double)P.gath_tgt.size(); stats[14] = (double)ndense; stats[15] = (double)nmapped;
    }
    if (symbolic_only) return 0;

    std::vector<double> Lx(P.panel_doubles, 0.0), Ld(P.diag_doubles, 0.0), D(N), Dinv(N);
    for (int64_t q = 0; q < P.nnzK; q++) Lx[P.kmap[q]] = Ax[q];
    int64_t nreg = 0;
    auto W = [&](int s) { return P.sn_first[s + 1] - P.sn_first[s]; };
    auto R = [&](int s) { return (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]); };
    for (int lvl = 0; lvl < P.nlevels; lvl++) {
        // factor items (blk 0 does the diagonal block; every blk its TRSM rows)
        for (int q = P.fac_lvl_ptr[lvl]; q < P.fac_lvl_ptr[lvl + 1]; q++) {
            FacItem it = P.fac_items[q];
            int s = it.sn, w = W(s), r = R(s);
            double *pan = &Lx[P.sn_panel[s]];
            double *ld = &Ld[P.sn_diag[s]];
            int f = P.sn_first[s];
            if (it.blk == 0) {
                std::vector<double> Aw((size_t)w * w);
                for (int j = 0; j < w; j++) for (int i = 0; i < w; i++) Aw[i + (size_t)j * w] = pan[i + (size_t)j * r];
                for (int k = 0; k < w; k++) {
                    double d = Aw[k + (size_t)k * w];
                    double sg = (double)dsigns[P.perm[f + k]];
                    if (d * sg < reg_eps) { d = reg_delta * sg; nreg++; }
                    D[f + k] = d; Dinv[f + k] = 1.0 / d;
                    for (int j = k + 1; j < w; j++)
                        for (int i = j; i < w; i++) Aw[i + (size_t)j * w] -= Aw[i + (size_t)k * w] * Aw[j + (size_t)k * w] / d;
                }
                for (int k = 0; k < w; k++) for (int i = 0; i < w; i++)
                    ld[i + (size_t)k * w] = i > k ? Aw[i + (size_t)k * w] / D[f + k] : (i == k ? 1.0 : 0.0);
            }
        }
        for (int q = P.fac_lvl_ptr[lvl]; q < P.fac_lvl_ptr[lvl + 1]; q++) {
            FacItem it = P.fac_items[q];
            int s = it.sn, w = W(s), r = R(s);
            double *pan = &Lx[P.sn_panel[s]];
            const double *ld = &Ld[P.sn_diag[s]];
            int f = P.sn_first[s];
            int lo = w + it.blk * kFacRows, hi = std::min(r, lo + kFacRows);
            for (int i = lo; i < hi; i++) {
                std::vector<double> y(w);
                for (int k = 0; k < w; k++) {
                    double a = pan[i + (size_t)k * r];
                    for (int j = 0; j < k; j++) a -= y[j] * ld[k + (size_t)j * w];
                    y[k] = a;
                }
                for (int k = 0; k < w; k++) pan[i + (size_t)k * r] = y[k] * Dinv[f + k];
            }
        }
        // update groups of this stage, interpreted the way the device kernels do:
        //   kind 1 (k_update_dense): tile coordinates; a source that does not land contiguously is gathered
        //          through its tile map;  kind 2 (k_update_gather): the per-target-entry pair lists;
        //   kind 0 (k_update_stage): relative-index scatter
        for (int g = P.upd_stage_ptr[lvl]; g < P.upd_stage_ptr[lvl + 1]; g++) {
            UpdGroup G = P.upd_groups[g];
            int t = G.tgt, rt = R(t), ft = P.sn_first[t], wt = W(t);
            double *tp = &Lx[P.sn_panel[t]];
            const int pos = g - P.upd_stage_ptr[lvl];
            const int nd = P.upd_stage_ndense[lvl], ng = P.upd_stage_ngather[lvl];
            if ((G.dense == 1) != (pos < nd) || (G.dense == 2) != (pos >= nd && pos < nd + ng)) return -7;
            if (G.dense == 2) continue;   // applied below from the gather lists
            for (int q = G.task_begin; q < G.task_end; q++) {
                UpdTask T = P.upd_tasks[q];
                int s = T.src, w = W(s), r = R(s), f = P.sn_first[s];
                if (P.sn_level[s] > lvl) return -2;  // source not factored yet
                if (P.sn_level[t] <= lvl) return -3; // target already factored
                const double *sp = &Lx[P.sn_panel[s]];
                const int *srows = &P.sn_rows[P.sn_rowptr[s]];
                if (G.dense == 1) {
                    const int nrt = std::min(kUpdRows, rt - G.row_base);
                    const bool mapped = (T.geom >> 17) & 1;
                    const int16_t *tm = mapped ? &P.upd_tmap[(size_t)T.vt_begin * 128] : nullptr;
                    const int c_r = T.geom & 255, c_c = (T.geom >> 8) & 255;
                    int touched = 0;
                    for (int ii = 0; ii < nrt; ii++) {
                        const int mi = mapped ? tm[ii] : (ii - c_r >= 0 && ii - c_r < T.nrows ? ii - c_r : -1);
                        if (mi < 0) continue;
                        for (int jj = 0; jj < wt; jj++) {
                            const int mj = mapped ? tm[64 + jj] : (jj - c_c >= 0 && jj - c_c < T.ncols ? jj - c_c : -1);
                            if (mj < 0) continue;
                            const int i = T.row_lo + mi, j = T.col_lo + mj;
                            if (P.rel[T.rel_off + (i - T.col_lo)] != G.row_base + ii || srows[j] - ft != jj) return -8;
                            double acc = 0;
                            for (int k = 0; k < w; k++) acc += sp[i + (size_t)k * r] * D[f + k] * sp[j + (size_t)k * r];
                            tp[(G.row_base + ii) + (size_t)jj * rt] -= acc;
                            touched++;
                        }
                    }
                    if (touched != T.nrows * T.ncols) return -9;   // the tile coordinates cover the whole task
                    continue;
                }
                for (int i = T.row_lo; i < T.row_lo + T.nrows; i++) {
                    int rp = P.rel[T.rel_off + (i - T.col_lo)];
                    if (rp < G.row_base || rp >= G.row_base + kUpdRows || rp >= rt) return -4;
                    for (int j = T.col_lo; j < T.col_lo + T.ncols; j++) {
                        int cp = srows[j] - ft;
                        double acc = 0;
                        for (int k = 0; k < w; k++) acc += sp[i + (size_t)k * r] * D[f + k] * sp[j + (size_t)k * r];
                        tp[rp + (size_t)cp * rt] -= acc;
                    }
                }
            }
        }
        for (int64_t e = P.gath_stage_ptr[lvl]; e < P.gath_stage_ptr[lvl + 1]; e++) {
            double acc = 0;
            for (int64_t pq = P.gath_pptr[e]; pq < P.gath_pptr[e + 1]; pq++) {
                const int s = P.gath_sn[pq], w = W(s), r = R(s), f = P.sn_first[s];
                if (P.sn_level[s] > lvl) return -5;
                const double *li = &Lx[P.gath_src[pq]], *lj = li + P.gath_dj[pq];
                for (int k = 0; k < w; k++) acc += li[(size_t)k * r] * D[f + k] * lj[(size_t)k * r];
            }
            Lx[P.gath_tgt[e]] -= acc;
        }
    }
    stats[10] = (double)nreg;
    // solves: y = perm(b); forward by levels with gather lists; D; backward
    std::vector<double> y(N), ub(P.ubuf_len, 0.0);
    for (int k = 0; k < N; k++) y[k] = b[P.perm[k]];
    SbEmu sbe(P, Lx, Ld);
    // PLANCHECK_EXPLICIT_INV: 1 = diagonal-block solves as products with the explicit inverses, like the kernels; 2 = the same with one
    // step of refinement against the factored block itself, y += Linv (b - L y), on the blocks whose inverse has an entry above
    // PLANCHECK_INV_TAU in magnitude (default 0: every block) -- the kernels' polished form (round 5)
    const int xmode = getenv("PLANCHECK_EXPLICIT_INV") ? atoi(getenv("PLANCHECK_EXPLICIT_INV")) : 0;
    const bool xinv = xmode >= 1;
    const double xtau = getenv("PLANCHECK_INV_TAU") ? atof(getenv("PLANCHECK_INV_TAU")) : 0.0;
    int64_t nflag = 0;
    const int xwmax = getenv("PLANCHECK_INV_WMAX") ? atoi(getenv("PLANCHECK_INV_WMAX")) : 1 << 30;   // refine only blocks up to this width
    const int xwmin = getenv("PLANCHECK_INV_WMIN") ? atoi(getenv("PLANCHECK_INV_WMIN")) : 0;         // ... and from this width
    auto flagged = [&](int s_, int w_) {
        if (xmode < 2) return false;
        if (w_ > xwmax || w_ < xwmin) return false;
        double mx = 0;
        const double *li = &sbe.Linv[P.sn_diag[s_]];
        for (int q = 0; q < w_ * w_; q++) mx = std::max(mx, std::fabs(li[q]));
        return mx > xtau;
    };
    // v <- L^-1 v (trans = false) or L^-T v (trans = true) of a w x w unit-lower block through its explicit inverse
    auto inv_apply = [&](int s_, int w_, double *v, bool trans) {
        const double *li = &sbe.Linv[P.sn_diag[s_]];
        const double *ld_ = &Ld[P.sn_diag[s_]];
        std::vector<double> b0(v, v + w_), t(w_), rr(w_);
        auto mul_inv = [&](const std::vector<double> &in, std::vector<double> &out) {
            for (int i = 0; i < w_; i++) {
                double a = 0;
                if (!trans) for (int k = 0; k <= i; k++) a += li[i + (size_t)k * w_] * in[k];
                else for (int k = i; k < w_; k++) a += li[k + (size_t)i * w_] * in[k];
                out[i] = a;
            }
        };
        mul_inv(b0, t);
        if (flagged(s_, w_)) {
            nflag++;
            for (int i = 0; i < w_; i++) {          // r = b - L t  (or L^T t)
                double a = b0[i] - t[i];
                if (!trans) for (int k = 0; k < i; k++) a -= ld_[i + (size_t)k * w_] * t[k];
                else for (int k = i + 1; k < w_; k++) a -= ld_[k + (size_t)i * w_] * t[k];
                rr[i] = a;
            }
            std::vector<double> c(w_);
            mul_inv(rr, c);
            for (int i = 0; i < w_; i++) t[i] += c[i];
        }
        for (int i = 0; i < w_; i++) v[i] = t[i];
    };
    for (int lvl = 0; lvl < P.nlevels; lvl++) {
        for (int q = P.lvl_ptr[lvl]; q < P.lvl_ptr[lvl + 1]; q++) {
            int s = P.lvl_sn[q], w = W(s), r = R(s), f = P.sn_first[s];
            if (P.sn_front[s] >= 0) continue;   // handled by the front sweep below (k_front_fwd)
            const double *pan = &Lx[P.sn_panel[s]];
            const double *ld = &Ld[P.sn_diag[s]];
            const int64_t slot0 = P.sn_rowptr[s];
            for (int k = 0; k < w; k++) {
                double v = 0;
                for (int64_t g = P.g_ptr[slot0 + k]; g < P.g_ptr[slot0 + k + 1]; g++) v += ub[P.g_idx[g]];
                y[f + k] -= v;
            }
            if (xinv) inv_apply(s, w, &y[f], false);        // the kernels' form (k_fwd_level): y_J = Linv * rhs, a lower-triangular GEMV
            else
            for (int k = 0; k < w; k++)
                for (int i = k + 1; i < w; i++) y[f + i] -= ld[i + (size_t)k * w] * y[f + k];
            for (int i = w; i < r; i++) {
                double a = 0;
                for (int64_t g = P.g_ptr[slot0 + i]; g < P.g_ptr[slot0 + i + 1]; g++) a += ub[P.g_idx[g]];
                for (int k = 0; k < w; k++) a += pan[i + (size_t)k * r] * y[f + k];
                ub[P.u_off[s] + (i - w)] = a;
            }
        }
        // fronts that end at this level: the persistent sweep (external gathers + panel-ordered accumulation)
        for (const FrontDesc &F : P.fronts) {
            if (F.level_last != lvl) continue;
            if (F.sb_g > 0) { sbe.invert_super(F); sbe.fwd(F, y, ub); continue; }   // front_sweep.hip
            const FrontPanel *fp = &P.front_panels[F.fp_off];
            std::vector<double> acc(F.rF, 0.0);
            for (int i = 0; i < F.rF; i++)
                for (int64_t g = P.front_gptr[F.gptr_off + i]; g < P.front_gptr[F.gptr_off + i + 1]; g++) acc[i] += ub[P.front_gidx[g]];
            for (int p = 0; p < F.np; p++) {
                const FrontPanel &me = fp[p];
                if (me.sn != F.s0 + p || me.f != P.sn_first[me.sn] || me.w != W(me.sn) || me.r != R(me.sn)) return -10;
                const double *ld = &Ld[P.sn_diag[me.sn]];
                const double *pan = &Lx[me.panel_off];
                for (int k = 0; k < me.w; k++) y[me.f + k] -= acc[F.cw * p + k];
                if (xinv) inv_apply(me.sn, me.w, &y[me.f], false);
                else
                for (int k = 0; k < me.w; k++)
                    for (int i = k + 1; i < me.w; i++) y[me.f + i] -= ld[i + (size_t)k * me.w] * y[me.f + k];
                for (int j = me.w; j < me.r; j++) {      // local row j = front row cw*p + j
                    double a = 0;
                    for (int k = 0; k < me.w; k++) a += pan[j + (size_t)k * me.r] * y[me.f + k];
                    acc[F.cw * p + j] += a;
                }
            }
            for (int i = F.W; i < F.rF; i++) ub[F.ubelow_off + (i - F.W)] = acc[i];
        }
    }
    for (int k = 0; k < N; k++) y[k] *= Dinv[k];
    for (int lvl = P.nlevels - 1; lvl >= 0; lvl--) {
        for (const FrontDesc &F : P.fronts)
            if (F.sb_g > 0 && F.level_last == lvl) { const std::vector<double> zc(y); sbe.bwd(F, zc, y); }   // k_front_bwd_sb
        for (int q = P.lvl_ptr[lvl]; q < P.lvl_ptr[lvl + 1]; q++) {
            int s = P.lvl_sn[q], w = W(s), r = R(s), f = P.sn_first[s];
            if (P.sn_front[s] >= 0 && P.fronts[P.sn_front[s]].sb_g > 0) continue;
            const double *pan = &Lx[P.sn_panel[s]];
            const double *ld = &Ld[P.sn_diag[s]];
            const int *rows = &P.sn_rows[P.sn_rowptr[s]];
            for (int k = 0; k < w; k++) {
                double a = 0;
                for (int i = w; i < r; i++) a += pan[i + (size_t)k * r] * y[rows[i]];
                y[f + k] -= a;
            }
            if (xinv) inv_apply(s, w, &y[f], true);        // k_bwd_final: x_J = Linv^T t
            else
            for (int k = w - 1; k >= 0; k--)
                for (int i = k + 1; i < w; i++) y[f + k] -= ld[i + (size_t)k * w] * y[f + i];
        }
    }
    for (int k = 0; k < N; k++) x[P.perm[k]] = y[k];
    if (xmode >= 2) stats[11] = (double)nflag;   // (diagnostic: block solves that took the refinement step)
    return 0;
}
}
