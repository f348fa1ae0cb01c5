/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the
 * product path (clarabel.jl_amd/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker / CPU baseline.
 *
 * CPU restatement of the LDL engine behind Clarabel.jl's `:qdldl` plugin
 * (reference: src/kktsolvers/direct-ldl/directldl_qdldl.jl:1-96).  The arithmetic
 * itself lives in the third-party package QDLDL.jl (compat "0.4.1",
 * /root/reference/Project.toml:38, sources NOT under /root/reference); this file
 * restates its published algorithm (the OSQP "QDLDL" up-looking quasidefinite
 * LDL^T) and is anchored on the reference's call sites.
 *
 * All indices 0-based, int64; values double.  Single-threaded, like the reference
 * (directldl_qdldl.jl:37 reports threads = 1).
 */
#ifndef QDLDL_ORACLE_H
#define QDLDL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qdldl_oracle qdldl_oracle;

/* Symbolic set-up only (the reference passes logical=true, directldl_qdldl.jl:18-25):
 *   A      : n x n upper-triangular CSC (Ap[n+1], Ai, Ax), diagonal present in every column
 *   perm   : fill-reducing order, perm[k] = original index eliminated k-th; NULL = natural
 *   dsigns : expected pivot signs (+1/-1), in ORIGINAL ordering
 * Returns NULL on malformed input. */
qdldl_oracle *qdldl_oracle_new(int64_t n, const int64_t *Ap, const int64_t *Ai,
                               const double *Ax, const int64_t *perm,
                               const int64_t *dsigns, double reg_eps, double reg_delta);
void qdldl_oracle_free(qdldl_oracle *F);

/* directldl_qdldl.jl:46-69 -> QDLDL.update_values!/scale_values!: index is into the
 * ORIGINAL (unpermuted) nzval; the permuted internal copy is what gets written. */
void qdldl_oracle_update_values(qdldl_oracle *F, const int64_t *index, const double *values, int64_t k);
void qdldl_oracle_scale_values(qdldl_oracle *F, const int64_t *index, int64_t k, double scale);

/* directldl_qdldl.jl:72-81: numeric factorisation; returns 1 iff every Dinv is finite. */
int qdldl_oracle_refactor(qdldl_oracle *F);

/* directldl_qdldl.jl:85-96: x <- K^{-1} x using the current factors (in place). */
void qdldl_oracle_solve(const qdldl_oracle *F, double *x);

int64_t qdldl_oracle_nnzL(const qdldl_oracle *F);
int64_t qdldl_oracle_nnzA(const qdldl_oracle *F);
int64_t qdldl_oracle_nreg(const qdldl_oracle *F);          /* pivots replaced in last refactor */
double  qdldl_oracle_sum_colcount_sq(const qdldl_oracle *F); /* sum_j c_j^2, c_j = nnz(L[:,j]) */
const double *qdldl_oracle_D(const qdldl_oracle *F);         /* permuted order */
const int64_t *qdldl_oracle_perm(const qdldl_oracle *F);

#ifdef __cplusplus
}
#endif
#endif
