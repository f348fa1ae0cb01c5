/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see kkt_oracle.h).
 *
 * CPU restatement of Clarabel.jl's DirectLDLKKTSolver (`:qdldl`, :triu shape).
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference/src).  0-based indices throughout.
 */
#include "kkt_oracle.h"
#include "qdldl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int kind;            /* 1 SOC, 2 GenPow */
    int64_t len[3];      /* SOC: u,v ; GenPow: q,r,p */
    int64_t *vec[3];     /* SOC: [0]=u [1]=v ; GenPow: [0]=q [1]=r [2]=p */
    int64_t D[3];
    int pdim;
} sparse_map;

struct oracle_kkt {
    int64_t n, m, p, N, nnzK, nHs, nsparse, nnzP, nnzA;
    int64_t *colptr, *rowval;
    double *nzval;
    /* LDLDataMap (kktsolvers/direct-ldl/directldl_datamaps.jl:170-214) */
    int64_t *mapP, *mapA, *mapHs, *diagP, *diag_full;
    sparse_map *smaps;
    int64_t *dsigns;
    /* kktsolver_directldl.jl:12-18 */
    double *x, *b, *work1, *work2, *hsbuf;
    double diagonal_regularizer;
    qdldl_oracle *ldl;
    /* TEST KNOB, not part of the restatement (0 = the reference's order, the default): 1 sweeps the columns of the refinement residual
     * e = b - K x in DESCENDING order -- the same products, the sums associated the other way round.  tools/seed324_cpu_vs_cpu.py uses
     * it to show what the reference's own refinement branches do under a different association order of that sum (any parallel
     * SpMV has one).  last_norms: ||e|| before / after every refinement step of the last solve (diagnostic). */
    int resid_order;
    double last_norms[16];
    int n_last_norms;
};

/* ---- utils/csc_assembly.jl primitives (counts held in colptr, then scanned) ---- */

/* csc_assembly.jl:245-272 (:triu case) / :44-53: a column lacks a diagonal entry */
static int missing_diag(const int64_t *Pp, const int64_t *Pi, int64_t i) {
    return (Pp[i] == Pp[i + 1]) || (Pi[Pp[i + 1] - 1] != i);
}

oracle_kkt *oracle_kkt_assemble(int64_t n, int64_t m,
                                const int64_t *Pp, const int64_t *Pi, const double *Px,
                                const int64_t *Ap, const int64_t *Ai, const double *Ax,
                                int64_t ncones, const int64_t *cone_numel,
                                const int32_t *cone_hs_dense, const int32_t *cone_sparse_kind,
                                const int64_t *cone_dim1) {
    oracle_kkt *k = (oracle_kkt *)calloc(1, sizeof(*k));
    k->n = n; k->m = m;
    k->nnzP = Pp[n]; k->nnzA = Ap[n];

    /* LDLDataMap ctor, directldl_datamaps.jl:182-212 */
    int64_t nHs = 0, nsparse = 0, p = 0, nnz_vec = 0, mcheck = 0;
    for (int64_t c = 0; c < ncones; c++) {
        int64_t d = cone_numel[c];
        mcheck += d;
        nHs += cone_hs_dense[c] ? d * (d + 1) / 2 : d; /* compositecone_type.jl:126-141 */
        if (cone_sparse_kind[c] == 1) { nsparse++; p += 2; nnz_vec += 2 * d; }
        else if (cone_sparse_kind[c] == 2) { nsparse++; p += 3; nnz_vec += 2 * d; } /* p:d, q:dim1, r:dim2 */
    }
    if (mcheck != m) { free(k); return NULL; }
    k->p = p; k->nHs = nHs; k->nsparse = nsparse;
    int64_t N = n + m + p;
    k->N = N;

    /* directldl_kkt_assembly.jl:27-41 */
    int64_t nnz_diagP = 0;
    for (int64_t i = 0; i < n; i++) nnz_diagP += !missing_diag(Pp, Pi, i);
    int64_t nnzK = k->nnzP + n - nnz_diagP + k->nnzA + nHs + nnz_vec + p;
    k->nnzK = nnzK;

    k->colptr = (int64_t *)calloc(N + 2, sizeof(int64_t));
    k->rowval = (int64_t *)malloc(sizeof(int64_t) * (nnzK ? nnzK : 1));
    k->nzval = (double *)calloc(nnzK ? nnzK : 1, sizeof(double));
    k->mapP = (int64_t *)malloc(sizeof(int64_t) * (k->nnzP ? k->nnzP : 1));
    k->mapA = (int64_t *)malloc(sizeof(int64_t) * (k->nnzA ? k->nnzA : 1));
    k->mapHs = (int64_t *)malloc(sizeof(int64_t) * (nHs ? nHs : 1));
    k->diagP = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    k->diag_full = (int64_t *)malloc(sizeof(int64_t) * (N ? N : 1));
    k->smaps = (sparse_map *)calloc(nsparse ? nsparse : 1, sizeof(sparse_map));
    {
        int64_t s = 0;
        for (int64_t c = 0; c < ncones; c++) {
            if (cone_sparse_kind[c] == 1) { /* SOCExpansionMap, datamaps.jl:8-22 */
                sparse_map *sm = &k->smaps[s++];
                sm->kind = 1; sm->pdim = 2;
                sm->len[0] = sm->len[1] = cone_numel[c];
                sm->vec[0] = (int64_t *)malloc(sizeof(int64_t) * cone_numel[c]);
                sm->vec[1] = (int64_t *)malloc(sizeof(int64_t) * cone_numel[c]);
            } else if (cone_sparse_kind[c] == 2) { /* GenPowExpansionMap, datamaps.jl:81-99 */
                sparse_map *sm = &k->smaps[s++];
                sm->kind = 2; sm->pdim = 3;
                sm->len[0] = cone_dim1[c];
                sm->len[1] = cone_numel[c] - cone_dim1[c];
                sm->len[2] = cone_numel[c];
                for (int v = 0; v < 3; v++)
                    sm->vec[v] = (int64_t *)malloc(sizeof(int64_t) * (sm->len[v] ? sm->len[v] : 1));
            }
        }
    }

    int64_t *cp = k->colptr;
    /* ---- pass 1: _kkt_assemble_colcounts (:triu), directldl_kkt_assembly.jl:52-101 ---- */
    for (int64_t i = 0; i < n; i++) cp[i] += Pp[i + 1] - Pp[i];             /* csc_assembly.jl:76-92 (:N) */
    for (int64_t i = 0; i < n; i++) if (missing_diag(Pp, Pi, i)) cp[i] += 1; /* csc_assembly.jl:41-53 */
    for (int64_t q = 0; q < k->nnzA; q++) cp[Ai[q] + n] += 1;               /* csc_assembly.jl:80-84 (:T) */
    {
        int64_t pcol = n + m, row = n;
        for (int64_t c = 0; c < ncones; c++) {
            int64_t d = cone_numel[c];
            if (!cone_hs_dense[c]) for (int64_t i = 0; i < d; i++) cp[row + i] += 1;   /* :34-37 */
            else for (int64_t i = 0; i < d; i++) cp[row + i] += i + 1;                 /* :19-29 triu */
            if (cone_sparse_kind[c] == 1) {           /* datamaps.jl:27-43 */
                cp[pcol] += d; cp[pcol + 1] += d;
                cp[pcol] += 1; cp[pcol + 1] += 1;
                pcol += 2;
            } else if (cone_sparse_kind[c] == 2) {    /* datamaps.jl:101-121 */
                int64_t d1 = cone_dim1[c], d2 = d - d1;
                cp[pcol] += d1; cp[pcol + 1] += d2; cp[pcol + 2] += d;
                cp[pcol] += 1; cp[pcol + 1] += 1; cp[pcol + 2] += 1;
                pcol += 3;
            }
            row += d;
        }
    }
    /* _csc_colcount_to_colptr, csc_assembly.jl:222-232 */
    {
        int64_t cur = 0;
        for (int64_t i = 0; i <= N; i++) { int64_t cnt = cp[i]; cp[i] = cur; cur += cnt; }
    }
    /* ---- pass 2: _kkt_assemble_fill (:triu), directldl_kkt_assembly.jl:104-175 ---- */
#define PUT(col_, row_, val_, dst_) do { int64_t d_ = cp[col_]++; k->rowval[d_] = (row_); k->nzval[d_] = (val_); dst_ = d_; } while (0)
    int64_t sink;
    for (int64_t i = 0; i < n; i++)                              /* csc_assembly.jl:137-156 (:N) */
        for (int64_t j = Pp[i]; j < Pp[i + 1]; j++) PUT(i, Pi[j], Px[j], k->mapP[j]);
    for (int64_t i = 0; i < n; i++)                              /* csc_assembly.jl:207-220 */
        if (missing_diag(Pp, Pi, i)) PUT(i, i, 0.0, sink);
    for (int64_t i = 0; i < n; i++)                              /* csc_assembly.jl:137-156 (:T) */
        for (int64_t j = Ap[i]; j < Ap[i + 1]; j++) PUT(Ai[j] + n, i, Ax[j], k->mapA[j]);
    {
        int64_t pcol = n + m, row = n, hoff = 0, s = 0;
        for (int64_t c = 0; c < ncones; c++) {
            int64_t d = cone_numel[c];
            if (!cone_hs_dense[c]) {                             /* csc_assembly.jl:193-204 */
                for (int64_t i = 0; i < d; i++) PUT(row + i, row + i, 0.0, k->mapHs[hoff + i]);
                hoff += d;
            } else {                                             /* csc_assembly.jl:174-187 */
                int64_t kidx = 0;
                for (int64_t col = row; col < row + d; col++)
                    for (int64_t r = row; r <= col; r++) { PUT(col, r, 0.0, k->mapHs[hoff + kidx]); kidx++; }
                hoff += d * (d + 1) / 2;
            }
            if (cone_sparse_kind[c] == 1) {                      /* datamaps.jl:45-59: v first, then u */
                sparse_map *sm = &k->smaps[s++];
                for (int64_t i = 0; i < d; i++) PUT(pcol, row + i, 0.0, sm->vec[1][i]);
                for (int64_t i = 0; i < d; i++) PUT(pcol + 1, row + i, 0.0, sm->vec[0][i]);
                PUT(pcol, pcol, 0.0, sm->D[0]);
                PUT(pcol + 1, pcol + 1, 0.0, sm->D[1]);
                pcol += 2;
            } else if (cone_sparse_kind[c] == 2) {               /* datamaps.jl:123-143: q, r, p */
                sparse_map *sm = &k->smaps[s++];
                int64_t d1 = cone_dim1[c], d2 = d - d1;
                for (int64_t i = 0; i < d1; i++) PUT(pcol, row + i, 0.0, sm->vec[0][i]);
                for (int64_t i = 0; i < d2; i++) PUT(pcol + 1, row + d1 + i, 0.0, sm->vec[1][i]);
                for (int64_t i = 0; i < d; i++) PUT(pcol + 2, row + i, 0.0, sm->vec[2][i]);
                PUT(pcol, pcol, 0.0, sm->D[0]);
                PUT(pcol + 1, pcol + 1, 0.0, sm->D[1]);
                PUT(pcol + 2, pcol + 2, 0.0, sm->D[2]);
                pcol += 3;
            }
            row += d;
        }
    }
#undef PUT
    (void)sink;
    /* _kkt_backshift_colptrs, csc_assembly.jl:234-243 */
    for (int64_t i = N; i >= 1; i--) cp[i] = cp[i - 1];
    cp[0] = 0;
    /* directldl_kkt_assembly.jl:161-165: diagonal is last in each triu column */
    for (int64_t j = 0; j < N; j++) k->diag_full[j] = cp[j + 1] - 1;
    for (int64_t j = 0; j < n; j++) k->diagP[j] = cp[j + 1] - 1;

    /* _fill_Dsigns!, kktsolver_directldl.jl:112-126 */
    k->dsigns = (int64_t *)malloc(sizeof(int64_t) * (N ? N : 1));
    for (int64_t i = 0; i < N; i++) k->dsigns[i] = 1;
    for (int64_t i = n; i < n + m; i++) k->dsigns[i] = -1;
    {
        int64_t pp = n + m;
        for (int64_t s = 0; s < nsparse; s++) {
            if (k->smaps[s].kind == 1) { k->dsigns[pp] = -1; k->dsigns[pp + 1] = 1; pp += 2; }
            else { k->dsigns[pp] = -1; k->dsigns[pp + 1] = -1; k->dsigns[pp + 2] = 1; pp += 3; }
        }
    }
    k->x = (double *)calloc(N ? N : 1, sizeof(double));
    k->b = (double *)calloc(N ? N : 1, sizeof(double));
    k->work1 = (double *)calloc(N ? N : 1, sizeof(double));
    k->work2 = (double *)calloc(N ? N : 1, sizeof(double));
    k->hsbuf = (double *)calloc(nHs ? nHs : 1, sizeof(double));
    return k;
}

void oracle_kkt_free(oracle_kkt *k) {
    if (!k) return;
    free(k->colptr); free(k->rowval); free(k->nzval); free(k->mapP); free(k->mapA); free(k->mapHs);
    free(k->diagP); free(k->diag_full); free(k->dsigns);
    for (int64_t s = 0; s < k->nsparse; s++)
        for (int v = 0; v < 3; v++) free(k->smaps[s].vec[v]);
    free(k->smaps);
    free(k->x); free(k->b); free(k->work1); free(k->work2); free(k->hsbuf);
    qdldl_oracle_free(k->ldl);
    free(k);
}

int oracle_kkt_symbolic(oracle_kkt *k, const int64_t *perm, double dyn_eps, double dyn_delta) {
    qdldl_oracle_free(k->ldl);
    k->ldl = qdldl_oracle_new(k->N, k->colptr, k->rowval, k->nzval, perm, k->dsigns, dyn_eps, dyn_delta);
    return k->ldl != NULL;
}

void oracle_kkt_sizes(const oracle_kkt *k, int64_t *out) {
    out[0] = k->N; out[1] = k->n; out[2] = k->m; out[3] = k->p; out[4] = k->nnzK;
    out[5] = k->nHs; out[6] = k->nsparse; out[7] = k->ldl ? qdldl_oracle_nnzL(k->ldl) : 0;
}
const int64_t *oracle_kkt_colptr(const oracle_kkt *k) { return k->colptr; }
const int64_t *oracle_kkt_rowval(const oracle_kkt *k) { return k->rowval; }
const double *oracle_kkt_nzval(const oracle_kkt *k) { return k->nzval; }
const int64_t *oracle_kkt_map_P(const oracle_kkt *k) { return k->mapP; }
const int64_t *oracle_kkt_map_A(const oracle_kkt *k) { return k->mapA; }
const int64_t *oracle_kkt_map_Hs(const oracle_kkt *k) { return k->mapHs; }
const int64_t *oracle_kkt_map_diagP(const oracle_kkt *k) { return k->diagP; }
const int64_t *oracle_kkt_map_diag_full(const oracle_kkt *k) { return k->diag_full; }
const int64_t *oracle_kkt_dsigns(const oracle_kkt *k) { return k->dsigns; }
const int64_t *oracle_kkt_sparse_map(const oracle_kkt *k, int64_t i, int which, int64_t *len) {
    const sparse_map *sm = &k->smaps[i];
    if (which == 3) { *len = sm->pdim; return sm->D; }
    *len = sm->len[which];
    return sm->vec[which];
}

/* _update_values! / _scale_values!, kktsolver_directldl.jl:130-188: the solver's own
 * KKT copy first, then the LDL engine's permuted copy */
void oracle_kkt_update_values(oracle_kkt *k, const int64_t *index, const double *values, int64_t cnt) {
    for (int64_t i = 0; i < cnt; i++) k->nzval[index[i]] = values[i];
    if (k->ldl) qdldl_oracle_update_values(k->ldl, index, values, cnt);
}
void oracle_kkt_scale_values(oracle_kkt *k, const int64_t *index, int64_t cnt, double scale) {
    for (int64_t i = 0; i < cnt; i++) k->nzval[index[i]] *= scale;
    if (k->ldl) qdldl_oracle_scale_values(k->ldl, index, cnt, scale);
}

/* kktsolver_directldl.jl:223-228 */
void oracle_kkt_update_Hs(oracle_kkt *k, const double *hs) {
    for (int64_t i = 0; i < k->nHs; i++) k->hsbuf[i] = hs[i] * -1.0;
    oracle_kkt_update_values(k, k->mapHs, k->hsbuf, k->nHs);
}

/* _csc_update_sparsecone(SOC), directldl_datamaps.jl:61-79 */
void oracle_kkt_update_soc(oracle_kkt *k, int64_t s, double eta2, const double *u, const double *v) {
    sparse_map *sm = &k->smaps[s];
    oracle_kkt_update_values(k, sm->vec[0], u, sm->len[0]);
    oracle_kkt_update_values(k, sm->vec[1], v, sm->len[1]);
    oracle_kkt_scale_values(k, sm->vec[0], sm->len[0], -eta2);
    oracle_kkt_scale_values(k, sm->vec[1], sm->len[1], -eta2);
    double dv[2] = {-eta2, eta2};
    oracle_kkt_update_values(k, sm->D, dv, 2);
}

/* _csc_update_sparsecone(GenPow), directldl_datamaps.jl:146-167 */
void oracle_kkt_update_genpow(oracle_kkt *k, int64_t s, double sqrtmu,
                              const double *p, const double *q, const double *r) {
    sparse_map *sm = &k->smaps[s];
    oracle_kkt_update_values(k, sm->vec[0], q, sm->len[0]);
    oracle_kkt_update_values(k, sm->vec[1], r, sm->len[1]);
    oracle_kkt_update_values(k, sm->vec[2], p, sm->len[2]);
    oracle_kkt_scale_values(k, sm->vec[0], sm->len[0], -sqrtmu);
    oracle_kkt_scale_values(k, sm->vec[1], sm->len[1], -sqrtmu);
    oracle_kkt_scale_values(k, sm->vec[2], sm->len[2], -sqrtmu);
    double dv[3] = {-1.0, -1.0, 1.0};
    oracle_kkt_update_values(k, sm->D, dv, 3);
}

/* kktsolver_directldl.jl:374-386 */
void oracle_kkt_update_P(oracle_kkt *k, const double *Px) { oracle_kkt_update_values(k, k->mapP, Px, k->nnzP); }
void oracle_kkt_update_A(oracle_kkt *k, const double *Ax) { oracle_kkt_update_values(k, k->mapA, Ax, k->nnzA); }

/* _kktsolver_regularize_and_refactor!, kktsolver_directldl.jl:247-294 */
int oracle_kkt_regularize_and_refactor(oracle_kkt *k, int static_enable, double reg_const,
                                       double reg_prop, double *eps_used) {
    double *diag_kkt = k->work1, *diag_shifted = k->work2;
    const int64_t N = k->N;
    if (static_enable) {
        double maxdiag = 0.0; /* _compute_regularizer, :297-310 */
        for (int64_t i = 0; i < N; i++) {
            diag_kkt[i] = k->nzval[k->diag_full[i]];
            double a = fabs(diag_kkt[i]);
            if (a > maxdiag || a != a) maxdiag = a;
        }
        double eps = reg_const + reg_prop * maxdiag;
        for (int64_t i = 0; i < N; i++)
            diag_shifted[i] = (k->dsigns[i] == 1) ? diag_kkt[i] + eps : diag_kkt[i] - eps;
        oracle_kkt_update_values(k, k->diag_full, diag_shifted, N);
        k->diagonal_regularizer = eps;
        if (eps_used) *eps_used = eps;
    }
    int ok = qdldl_oracle_refactor(k->ldl);
    if (static_enable) /* restore the solver's own copy only, :285-291 */
        for (int64_t i = 0; i < N; i++) k->nzval[k->diag_full[i]] = diag_kkt[i];
    return ok;
}

/* kktsolver_setrhs!, kktsolver_directldl.jl:313-327 */
void oracle_kkt_setrhs(oracle_kkt *k, const double *rhsx, const double *rhsz) {
    memcpy(k->b, rhsx, sizeof(double) * k->n);
    memcpy(k->b + k->n, rhsz, sizeof(double) * k->m);
    for (int64_t i = k->n + k->m; i < k->N; i++) k->b[i] = 0.0;
}
void oracle_kkt_set_b(oracle_kkt *k, const double *b) { memcpy(k->b, b, sizeof(double) * k->N); }
void oracle_kkt_get_x(const oracle_kkt *k, double *x) { memcpy(x, k->x, sizeof(double) * k->N); }

/* y = Symmetric(K,:U) * x */
void oracle_kkt_symv(const oracle_kkt *k, const double *x, double *y) {
    for (int64_t i = 0; i < k->N; i++) y[i] = 0.0;
    for (int64_t j = 0; j < k->N; j++)
        for (int64_t q = k->colptr[j]; q < k->colptr[j + 1]; q++) {
            int64_t i = k->rowval[q];
            double v = k->nzval[q];
            y[i] += v * x[j];
            if (i != j) y[j] += v * x[i];
        }
}

static double norm_inf(const double *v, int64_t n) {
    double mx = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double a = fabs(v[i]);
        if (a != a) return a; /* NaN propagates, as Julia's norm(.,Inf) does */
        if (a > mx) mx = a;
    }
    return mx;
}

/* _get_refine_error!, kktsolver_directldl.jl:455-466: e = b - K*xi, returns ||e||_inf */
static double refine_error(const oracle_kkt *k, double *e, const double *b, const double *xi) {
    const int64_t N = k->N;
    for (int64_t i = 0; i < N; i++) e[i] = b[i];
    for (int64_t jj = 0; jj < N; jj++) {
        const int64_t j = k->resid_order ? N - 1 - jj : jj;   /* (test knob, see struct oracle_kkt; 0 = the reference's order) */
        for (int64_t q = k->colptr[j]; q < k->colptr[j + 1]; q++) {
            int64_t i = k->rowval[q];
            double v = k->nzval[q];
            e[i] -= v * xi[j];
            if (i != j) e[j] -= v * xi[i];
        }
    }
    return norm_inf(e, N);
}
void oracle_kkt_set_residual_order(oracle_kkt *k, int mode) { k->resid_order = mode; }
int oracle_kkt_last_norms(const oracle_kkt *k, double *out16) {
    for (int i = 0; i < k->n_last_norms; i++) out16[i] = k->last_norms[i];
    return k->n_last_norms;
}

void oracle_kkt_ldl_solve(const oracle_kkt *k, double *x, const double *b) {
    memcpy(x, b, sizeof(double) * k->N); /* directldl_qdldl.jl:93 */
    qdldl_oracle_solve(k->ldl, x);
}

/* _iterative_refinement, kktsolver_directldl.jl:389-449 */
static int iterative_refinement(oracle_kkt *k, double reltol, double abstol, int64_t max_iter,
                                double stop_ratio, int64_t *steps) {
    double *x = k->x, *b = k->b, *e = k->work1, *dx = k->work2;
    const int64_t N = k->N;
    double normb = norm_inf(b, N);
    double norme = refine_error(k, e, b, x);
    *steps = 0;
    k->n_last_norms = 0;
    k->last_norms[k->n_last_norms++] = norme;
    if (!isfinite(norme)) return 0;
    for (int64_t it = 0; it < max_iter; it++) {
        if (norme <= abstol + reltol * normb) break;
        double lastnorme = norme;
        oracle_kkt_ldl_solve(k, dx, e);
        (*steps)++;
        for (int64_t i = 0; i < N; i++) dx[i] += x[i];
        norme = refine_error(k, e, b, dx);
        if (k->n_last_norms < 16) k->last_norms[k->n_last_norms++] = norme;
        if (!isfinite(norme)) return 0;
        double improved = lastnorme / norme;
        if (improved < stop_ratio) {
            if (improved > 1.0) { double *t = x; x = dx; dx = t; }
            break;
        }
        { double *t = x; x = dx; dx = t; }
    }
    k->x = x; k->work2 = dx;
    return 1;
}

/* kktsolver_solve!, kktsolver_directldl.jl:346-371 */
int oracle_kkt_solve(oracle_kkt *k, double *lhsx, double *lhsz, int ir_enable, double reltol,
                     double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps) {
    int64_t steps = 0;
    oracle_kkt_ldl_solve(k, k->x, k->b);
    int ok;
    if (ir_enable) ok = iterative_refinement(k, reltol, abstol, max_iter, stop_ratio, &steps);
    else {
        ok = 1;
        for (int64_t i = 0; i < k->N; i++) if (!isfinite(k->x[i])) { ok = 0; break; }
    }
    if (ir_steps) *ir_steps = steps;
    if (ok) { /* kktsolver_getlhs!, :330-343 */
        if (lhsx) memcpy(lhsx, k->x, sizeof(double) * k->n);
        if (lhsz) memcpy(lhsz, k->x + k->n, sizeof(double) * k->m);
    }
    return ok;
}

/* D of the last numeric factorisation, permuted order (diagnostics: pivot-by-pivot comparison with the HIP path) */
const double *oracle_kkt_D(const oracle_kkt *k) { return k->ldl ? qdldl_oracle_D(k->ldl) : 0; }
int64_t oracle_kkt_nreg(const oracle_kkt *k) { return k->ldl ? qdldl_oracle_nreg(k->ldl) : 0; }
double oracle_kkt_sum_colcount_sq(const oracle_kkt *k) { return k->ldl ? qdldl_oracle_sum_colcount_sq(k->ldl) : 0.0; }
