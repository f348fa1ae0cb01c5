/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see qdldl_oracle.h).
 *
 * Restates the up-looking quasidefinite LDL^T that Clarabel.jl's `:qdldl` plugin
 * drives (reference call sites: src/kktsolvers/direct-ldl/directldl_qdldl.jl:18-25
 * ctor, :54 update_values!, :66 scale_values!, :77 refactor!, :94 solve!).
 * Algorithm = the published QDLDL method (Stellato et al., OSQP): elimination
 * tree + column counts, then for each k the sparse triangular solve that yields
 * row k of L, with the sign-driven pivot substitution
 *     if D[k]*Dsigns[k] < eps  then  D[k] = delta*Dsigns[k]
 * (SURVEY.md Appendix C).  Parity status: the reference pins this layer only
 * end-to-end — tests/ pin it against dense numpy solves and the
 * reference's known answers through the IPM driver.
 */
#include "qdldl_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define UNKNOWN (-1)

struct qdldl_oracle {
    int64_t n, nnzA;
    /* triu(P A P^T), rows unsorted within a column */
    int64_t *Ap, *Ai;
    double *Ax;
    int64_t *AtoPAPt; /* original nz index -> index in the permuted copy */
    int64_t *perm, *iperm;
    int64_t *dsigns; /* permuted order */
    /* factors */
    int64_t *etree, *Lnz, *Lp, *Li;
    double *Lx, *D, *Dinv;
    /* workspace */
    int64_t *iwork;   /* 3n: yIdx | elimBuffer | LNextSpaceInCol */
    unsigned char *bwork;
    double *fwork;
    double reg_eps, reg_delta;
    int64_t nreg;
};

/* elimination tree and column counts of L from the triu pattern */
static int64_t etree_and_counts(int64_t n, const int64_t *Ap, const int64_t *Ai,
                                int64_t *work, int64_t *Lnz, int64_t *etree) {
    for (int64_t i = 0; i < n; i++) { work[i] = 0; Lnz[i] = 0; etree[i] = UNKNOWN; }
    for (int64_t j = 0; j < n; j++) {
        work[j] = j;
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int64_t i = Ai[p];
            if (i > j) return -1; /* not upper triangular */
            while (work[i] != j) {
                if (etree[i] == UNKNOWN) etree[i] = j;
                Lnz[i]++;
                work[i] = j;
                i = etree[i];
            }
        }
    }
    int64_t sum = 0;
    for (int64_t i = 0; i < n; i++) sum += Lnz[i];
    return sum;
}

qdldl_oracle *qdldl_oracle_new(int64_t n, const int64_t *Ap, const int64_t *Ai,
                               const double *Ax, const int64_t *perm,
                               const int64_t *dsigns, double reg_eps, double reg_delta) {
    qdldl_oracle *F = (qdldl_oracle *)calloc(1, sizeof(*F));
    if (!F) return NULL;
    int64_t nnz = Ap[n];
    F->n = n; F->nnzA = nnz; F->reg_eps = reg_eps; F->reg_delta = reg_delta;
    F->perm = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    F->iperm = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    for (int64_t k = 0; k < n; k++) F->perm[k] = perm ? perm[k] : k;
    for (int64_t k = 0; k < n; k++) F->iperm[k] = -1;
    for (int64_t k = 0; k < n; k++) {
        int64_t o = F->perm[k];
        if (o < 0 || o >= n || F->iperm[o] != -1) { qdldl_oracle_free(F); return NULL; }
        F->iperm[o] = k;
    }
    F->dsigns = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    for (int64_t k = 0; k < n; k++) F->dsigns[k] = dsigns ? dsigns[F->perm[k]] : 1;

    /* symmetric permutation to triu, remembering where each entry went */
    F->Ap = (int64_t *)calloc(n + 1, sizeof(int64_t));
    F->Ai = (int64_t *)malloc(sizeof(int64_t) * (nnz ? nnz : 1));
    F->Ax = (double *)malloc(sizeof(double) * (nnz ? nnz : 1));
    F->AtoPAPt = (int64_t *)malloc(sizeof(int64_t) * (nnz ? nnz : 1));
    int64_t *cnt = (int64_t *)calloc(n + 1, sizeof(int64_t));
    for (int64_t j = 0; j < n; j++)
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int64_t pi = F->iperm[Ai[p]], pj = F->iperm[j];
            cnt[pi > pj ? pi : pj]++;
        }
    for (int64_t j = 0; j < n; j++) F->Ap[j + 1] = F->Ap[j] + cnt[j];
    for (int64_t j = 0; j < n; j++) cnt[j] = F->Ap[j];
    for (int64_t j = 0; j < n; j++)
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int64_t pi = F->iperm[Ai[p]], pj = F->iperm[j];
            int64_t col = pi > pj ? pi : pj, row = pi > pj ? pj : pi;
            int64_t dest = cnt[col]++;
            F->Ai[dest] = row;
            F->Ax[dest] = Ax ? Ax[p] : 0.0;
            F->AtoPAPt[p] = dest;
        }
    free(cnt);

    F->etree = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    F->Lnz = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
    F->iwork = (int64_t *)malloc(sizeof(int64_t) * (3 * n + 1));
    int64_t sumLnz = etree_and_counts(n, F->Ap, F->Ai, F->iwork, F->Lnz, F->etree);
    if (sumLnz < 0) { qdldl_oracle_free(F); return NULL; }
    F->Lp = (int64_t *)calloc(n + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) F->Lp[i + 1] = F->Lp[i] + F->Lnz[i];
    F->Li = (int64_t *)malloc(sizeof(int64_t) * (sumLnz ? sumLnz : 1));
    F->Lx = (double *)calloc(sumLnz ? sumLnz : 1, sizeof(double));
    F->D = (double *)calloc(n ? n : 1, sizeof(double));
    F->Dinv = (double *)calloc(n ? n : 1, sizeof(double));
    F->bwork = (unsigned char *)calloc(n ? n : 1, 1);
    F->fwork = (double *)calloc(n ? n : 1, sizeof(double));
    return F;
}

void qdldl_oracle_free(qdldl_oracle *F) {
    if (!F) return;
    free(F->Ap); free(F->Ai); free(F->Ax); free(F->AtoPAPt); free(F->perm); free(F->iperm);
    free(F->dsigns); free(F->etree); free(F->Lnz); free(F->Lp); free(F->Li); free(F->Lx);
    free(F->D); free(F->Dinv); free(F->iwork); free(F->bwork); free(F->fwork);
    free(F);
}

void qdldl_oracle_update_values(qdldl_oracle *F, const int64_t *index, const double *values, int64_t k) {
    for (int64_t i = 0; i < k; i++) F->Ax[F->AtoPAPt[index[i]]] = values[i];
}

void qdldl_oracle_scale_values(qdldl_oracle *F, const int64_t *index, int64_t k, double scale) {
    for (int64_t i = 0; i < k; i++) F->Ax[F->AtoPAPt[index[i]]] *= scale;
}

int qdldl_oracle_refactor(qdldl_oracle *F) {
    const int64_t n = F->n;
    const int64_t *Ap = F->Ap, *Ai = F->Ai, *Lp = F->Lp, *etree = F->etree;
    const double *Ax = F->Ax;
    int64_t *Li = F->Li;
    double *Lx = F->Lx, *D = F->D, *Dinv = F->Dinv, *yVals = F->fwork;
    unsigned char *yMark = F->bwork;
    int64_t *yIdx = F->iwork, *elim = F->iwork + n, *Lnext = F->iwork + 2 * n;
    F->nreg = 0;
    for (int64_t i = 0; i < n; i++) { yMark[i] = 0; yVals[i] = 0.0; D[i] = 0.0; Lnext[i] = Lp[i]; }

    for (int64_t k = 0; k < n; k++) {
        int64_t nnzY = 0;
        for (int64_t p = Ap[k]; p < Ap[k + 1]; p++) {
            int64_t b = Ai[p];
            if (b == k) { D[k] = Ax[p]; continue; }
            yVals[b] = Ax[p];
            if (!yMark[b]) {
                /* walk up the elimination tree: the reach of b below k */
                int64_t nnzE = 0, nx = b;
                yMark[nx] = 1; elim[nnzE++] = nx;
                nx = etree[nx];
                while (nx != UNKNOWN && nx < k) {
                    if (yMark[nx]) break;
                    yMark[nx] = 1; elim[nnzE++] = nx;
                    nx = etree[nx];
                }
                while (nnzE) yIdx[nnzY++] = elim[--nnzE];
            }
        }
        /* sparse triangular solve in topological order -> row k of L */
        for (int64_t i = nnzY - 1; i >= 0; i--) {
            int64_t c = yIdx[i];
            int64_t top = Lnext[c];
            double yc = yVals[c];
            for (int64_t j = Lp[c]; j < top; j++) yVals[Li[j]] -= Lx[j] * yc;
            Li[top] = k;
            Lx[top] = yc * Dinv[c];
            D[k] -= yc * Lx[top];
            Lnext[c]++;
            yVals[c] = 0.0;
            yMark[c] = 0;
        }
        /* dynamic regularisation, sign-driven */
        if (D[k] * (double)F->dsigns[k] < F->reg_eps) {
            D[k] = F->reg_delta * (double)F->dsigns[k];
            F->nreg++;
        }
        Dinv[k] = 1.0 / D[k];
    }
    for (int64_t k = 0; k < n; k++)
        if (!isfinite(Dinv[k])) return 0;
    return 1;
}

void qdldl_oracle_solve(const qdldl_oracle *F, double *x) {
    const int64_t n = F->n;
    double *t = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (int64_t k = 0; k < n; k++) t[k] = x[F->perm[k]];
    for (int64_t i = 0; i < n; i++) {
        double v = t[i];
        for (int64_t j = F->Lp[i]; j < F->Lp[i + 1]; j++) t[F->Li[j]] -= F->Lx[j] * v;
    }
    for (int64_t i = 0; i < n; i++) t[i] *= F->Dinv[i];
    for (int64_t i = n - 1; i >= 0; i--) {
        double v = t[i];
        for (int64_t j = F->Lp[i]; j < F->Lp[i + 1]; j++) v -= F->Lx[j] * t[F->Li[j]];
        t[i] = v;
    }
    for (int64_t k = 0; k < n; k++) x[F->perm[k]] = t[k];
    free(t);
}

int64_t qdldl_oracle_nnzL(const qdldl_oracle *F) { return F->Lp[F->n]; }
int64_t qdldl_oracle_nnzA(const qdldl_oracle *F) { return F->nnzA; }
int64_t qdldl_oracle_nreg(const qdldl_oracle *F) { return F->nreg; }
double qdldl_oracle_sum_colcount_sq(const qdldl_oracle *F) {
    double s = 0;
    for (int64_t i = 0; i < F->n; i++) s += (double)F->Lnz[i] * (double)F->Lnz[i];
    return s;
}
const double *qdldl_oracle_D(const qdldl_oracle *F) { return F->D; }
const int64_t *qdldl_oracle_perm(const qdldl_oracle *F) { return F->perm; }
