/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or executed by the
 * product path (clarabel.jl_amd/).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker / CPU baseline.
 *
 * CPU restatement of Clarabel.jl's DirectLDLKKTSolver with the `:qdldl` engine:
 *   src/kktsolvers/kktsolver_directldl.jl            (update / regularise / refactor / solve / IR)
 *   src/kktsolvers/direct-ldl/directldl_kkt_assembly.jl   (:triu KKT assembly)
 *   src/kktsolvers/direct-ldl/directldl_datamaps.jl       (LDLDataMap, SOC / GenPow expansion maps)
 *   src/utils/csc_assembly.jl                             (count / fill primitives)
 * All indices here are 0-based (the reference is 1-based); values double.
 *
 * Parity status: the reference's tests pin this path only end-to-end
 * (SURVEY.md §8c).  tests/test_oracle_*.py pin (i) the assembly against the
 * hand-derived layout of SURVEY.md Appendix B, (ii) factor/solve against dense
 * numpy, (iii) the reference's known answers through the IPM driver.
 */
#ifndef KKT_ORACLE_H
#define KKT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_kkt oracle_kkt;

/* cone description = what the KKT structure needs to know about a cone
 *   numel       : rows the cone occupies (numel(cone))
 *   hs_dense    : 0 -> Hs block is diagonal (Hs_is_diagonal), 1 -> dense packed triu
 *   sparse_kind : 0 none, 1 SOC expansion (pdim 2, Dsigns (-1,+1)),
 *                 2 GenPow expansion (pdim 3, Dsigns (-1,-1,+1))
 *   dim1        : GenPow only: length of the q block (dim2 = numel - dim1)           */
oracle_kkt *oracle_kkt_assemble(int64_t n, int64_t m,
                                const int64_t *Pp, const int64_t *Pi, const double *Px,
                                const int64_t *Ap, const int64_t *Ai, const double *Ax,
                                int64_t ncones, const int64_t *cone_numel,
                                const int32_t *cone_hs_dense, const int32_t *cone_sparse_kind,
                                const int64_t *cone_dim1);
void oracle_kkt_free(oracle_kkt *k);

/* build the LDL engine (symbolic only; kktsolver_directldl.jl:86).  perm may be NULL. */
int oracle_kkt_symbolic(oracle_kkt *k, const int64_t *perm, double dyn_eps, double dyn_delta);

/* sizes: out[0..7] = N, n, m, p, nnzK, nHs, nsparse, nnzL(0 before symbolic) */
void oracle_kkt_sizes(const oracle_kkt *k, int64_t *out);
const int64_t *oracle_kkt_colptr(const oracle_kkt *k);
const int64_t *oracle_kkt_rowval(const oracle_kkt *k);
const double *oracle_kkt_nzval(const oracle_kkt *k);
const int64_t *oracle_kkt_map_P(const oracle_kkt *k);
const int64_t *oracle_kkt_map_A(const oracle_kkt *k);
const int64_t *oracle_kkt_map_Hs(const oracle_kkt *k);
const int64_t *oracle_kkt_map_diagP(const oracle_kkt *k);
const int64_t *oracle_kkt_map_diag_full(const oracle_kkt *k);
const int64_t *oracle_kkt_dsigns(const oracle_kkt *k);
/* sparse map i: which = 0 -> first vector (SOC u / GenPow q), 1 -> second (SOC v / GenPow r),
 * 2 -> third (GenPow p), 3 -> D.  *len receives its length. */
const int64_t *oracle_kkt_sparse_map(const oracle_kkt *k, int64_t i, int which, int64_t *len);

/* kktsolver_directldl.jl:223-228: Hs arrives as the cones produced it; negated here */
void oracle_kkt_update_Hs(oracle_kkt *k, const double *hs);
/* directldl_datamaps.jl:61-79 */
void oracle_kkt_update_soc(oracle_kkt *k, int64_t sparse_idx, double eta2, const double *u, const double *v);
/* directldl_datamaps.jl:146-167 */
void oracle_kkt_update_genpow(oracle_kkt *k, int64_t sparse_idx, double sqrtmu,
                              const double *p, const double *q, const double *r);
/* generic index/value forms (kktsolver_directldl.jl:130-188) */
void oracle_kkt_update_values(oracle_kkt *k, const int64_t *index, const double *values, int64_t cnt);
void oracle_kkt_scale_values(oracle_kkt *k, const int64_t *index, int64_t cnt, double scale);
/* kktsolver_directldl.jl:374-386 */
void oracle_kkt_update_P(oracle_kkt *k, const double *Px);
void oracle_kkt_update_A(oracle_kkt *k, const double *Ax);

/* kktsolver_directldl.jl:247-310; returns 1 on success.  eps_used may be NULL. */
int oracle_kkt_regularize_and_refactor(oracle_kkt *k, int static_enable, double reg_const,
                                       double reg_prop, double *eps_used);

/* kktsolver_directldl.jl:313-327 */
void oracle_kkt_setrhs(oracle_kkt *k, const double *rhsx, const double *rhsz);
/* kktsolver_directldl.jl:346-371 + :389-466.  lhsx / lhsz may be NULL.  Returns 1 on success. */
int oracle_kkt_solve(oracle_kkt *k, double *lhsx, double *lhsz, int ir_enable, double reltol,
                     double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps);
/* full-length access for layer tests: b and x are length N */
void oracle_kkt_set_b(oracle_kkt *k, const double *b);
void oracle_kkt_get_x(const oracle_kkt *k, double *x);
/* plain LDL solve without refinement (directldl_qdldl.jl:85-96), x,b length N */
void oracle_kkt_ldl_solve(const oracle_kkt *k, double *x, const double *b);
/* y = K*x with the symmetric view of the (unregularised) triu KKT */
void oracle_kkt_symv(const oracle_kkt *k, const double *x, double *y);
/* TEST KNOBS / diagnostics, not part of the restatement: association order of the refinement residual's sums (0 = the reference's),
 * residual norms of the last refined solve */
void oracle_kkt_set_residual_order(oracle_kkt *k, int mode);
int oracle_kkt_last_norms(const oracle_kkt *k, double *out16);

int64_t oracle_kkt_nreg(const oracle_kkt *k);
const double *oracle_kkt_D(const oracle_kkt *k);   /* D of the last factorisation, permuted order */
double oracle_kkt_sum_colcount_sq(const oracle_kkt *k);

#ifdef __cplusplus
}
#endif
#endif
