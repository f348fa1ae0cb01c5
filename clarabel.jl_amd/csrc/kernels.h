// launchers implemented in kernels.hip (all work is enqueued on the given stream)
#pragma once
#include <hip/hip_runtime.h>

#include "device_plan.h"

namespace hipkkt {
void launch_scatter_values(hipStream_t st, double *kval, const int64_t *idx, const double *vals, int64_t n, double scale);
void launch_scale_values(hipStream_t st, double *kval, const int64_t *idx, int64_t n, double scale);
void launch_soc_batch(hipStream_t st, double *kval, const int64_t *uidx, const int64_t *vidx, const int *cone_of,
                      const double *u, const double *v, const double *eta2, int64_t n, const int64_t *didx, int nsoc);
void launch_maxabs_gather(hipStream_t st, const double *v, const int64_t *idx, int64_t n, unsigned long long *slot);
void launch_init_panels(hipStream_t st, const DevPlan &P, int64_t nnz, int static_enable, double eps_const, double eps_prop);
void launch_factor_level(hipStream_t st, const DevPlan &P, int item_begin, int nitems, int wmax, double dyn_eps, double dyn_delta);
void launch_factor_panel(hipStream_t st, const DevPlan &P, int item_begin, int nitems, double dyn_eps, double dyn_delta);
void launch_update_stage(hipStream_t st, const DevPlan &P, int group_begin, int ngroups);
// full_k: the tiles of this launch carry whole panels of sources (>= 1.5 MFLOP per tile): a partial last round is cut into pieces
void launch_update_dense(hipStream_t st, const DevPlan &P, int group_begin, int ngroups, bool full_k = false);
// split-K: adds the partial tiles of `n` split target tiles to their targets (fixed order) and clears them (kernels.hip k_split_reduce)
void launch_split_reduce(hipStream_t st, const DevPlan &P, const SplitRec *recs, int n);
void launch_fwd_narrow(hipStream_t st, const DevPlan &P, int sn_begin, int n, int wmax, double *y, double *z);
void launch_bwd_narrow(hipStream_t st, const DevPlan &P, int sn_begin, int n, int wmax, const double *z, double *x, double *xout);
void launch_psd_hs(hipStream_t st, double *kval, const int64_t *map_hs, int64_t hs_off, const double *W, int n);
// front_block.hip
void launch_front_block(hipStream_t st, const DevPlan &P, const FrontBatch &B, int *sync_all, double *scratch_all, double *stream_all,
                        double dyn_eps, double dyn_delta, bool streamed, long long *trace = nullptr);
// front_block2.hip: the same batch with every tile transposed in the accumulators (round 5; the default)
void launch_front_block2(hipStream_t st, const DevPlan &P, const FrontBatch &B, int *sync_all, double *scratch_all, double *stream_all,
                         double dyn_eps, double dyn_delta, long long *trace = nullptr);
// zeroes the sync words of all front batches and fills the stream records of the streamed pivot chain with the "not yet written" sentinel
void launch_fb_reset(hipStream_t st, int *sync_all, int nsync, double *stream_all, int64_t nstream);
// front_sweep.hip: super-block sweeps over a front (FrontDesc::sb_g > 0) and the super-block inverses they need
void launch_front_fwd_sb(hipStream_t st, const DevPlan &P, const FrontDesc &F, double *y, double *z);
void launch_front_bwd_sb(hipStream_t st, const DevPlan &P, const FrontDesc &F, const double *z, double *x, double *xout);
void launch_invert_super(hipStream_t st, const DevPlan &P, const FrontDesc &F);
// scaling.hip (N1)
void launch_scaling_diag(hipStream_t st, const signed char *row_kind, const int64_t *row_hs, const int64_t *map_hs, const double *s,
                         const double *z, double *w, double *lam, double *kval, int64_t m);
void launch_scaling_soc(hipStream_t st, int nsoc, const int64_t *desc, const int64_t *map_hs, const double *s, const double *z,
                        double *w, double *lam, double *eta_out, double *soc_u, double *soc_v, double *soc_eta2, double *kval,
                        int *fail);
void launch_psd_rrt(hipStream_t st, const double *R, double *W, int n);
void launch_block_products(hipStream_t st, const DevPlan &P, const double *x, const double *z, double *Px, double *ATz,
                           double *Ax, int n, int m);
void launch_zero_words(hipStream_t st, void *p, int nwords);
void launch_update_gather(hipStream_t st, const DevPlan &P, int64_t ebegin, int64_t n, int64_t hbegin, int64_t nheavy, int max_blocks = 0);
void launch_invert_diag(hipStream_t st, const DevPlan &P, int n_small, int wmax_small, int n_wide);
void launch_mfma_probe(hipStream_t st, const double *A, const double *B, double *out);
void launch_permute_in(hipStream_t st, const double *b, const int *perm, double *y, int n, int *epoch, int *ticks, int nticks, int *zero = nullptr,
                       int nzero = 0);   // zero[0 .. nzero): words cleared by the solve's first kernel for the kernels behind it
void launch_fwd_level(hipStream_t st, const DevPlan &P, int item_begin, int nitems, double *y, double *z);
void launch_bwd_partial(hipStream_t st, const DevPlan &P, int item_begin, int nitems, const double *x);
void launch_bwd_final(hipStream_t st, const DevPlan &P, int sn_begin, int nsn, const double *z, double *x, double *xout);
void launch_fwd_seg(hipStream_t st, const DevPlan &P, int seg, int item_begin, int nitems, int nsuper, int first, double *y, double *z);
void launch_bwd_seg(hipStream_t st, const DevPlan &P, int seg, int item_begin, int nitems, int nsuper, int first, const double *z,
                    double *x, double *xout);
void launch_front_fwd(hipStream_t st, const DevPlan &P, const FrontDesc &F, double *y, double *z);
void launch_front_bwd(hipStream_t st, const DevPlan &P, const FrontDesc &F, const double *z, double *x, double *xout);
void launch_spmv_residual(hipStream_t st, const DevPlan &P, const double *b, const double *xi, double *e, int n,
                          unsigned long long *slot);
void launch_refine_decide(hipStream_t st, RefineState *rs, const double *scal, int phase, double reltol, double abstol, int max_iter,
                          double stop_ratio);
void launch_refine_add(hipStream_t st, const RefineState *rs, double *x0, double *x1, const double *corr, int n);
void launch_refine_copy_out(hipStream_t st, const RefineState *rs, const double *x0, const double *x1, double *out, int nm);
void launch_spmv_residual_cand(hipStream_t st, const DevPlan &P, const double *b, const RefineState *rs, const double *x0,
                               const double *x1, double *e, int n, unsigned long long *slot);
int residual_blocks(int n, int m);
int long_row_threshold();   // rows of the symmetric CSR view with more entries are taken by k_spmv_long
void launch_residuals(hipStream_t st, const DevPlan &P, const double *x, const double *z, const double *s, const double *q,
                      const double *b, double tau, double kappa, double *out, double *part, double *scal, int n, int m);
// N2: reduced-system algebra of kkt_solve! (part: 8 * residual_blocks(n, m) doubles, scal_out: 10 doubles)
void launch_reduced(hipStream_t st, const DevPlan &P, const double *s1, const double *s2, const double *xv, const double *q,
                    const double *b, double tau, double kappa, double rhs_tau, double rhs_kappa, double *part, double *scal_out,
                    double *lhs, int n, int m);
void launch_const_rhs(hipStream_t st, double *dst, const double *qb, int n, int nm, int N);
void launch_norm_inf(hipStream_t st, const double *v, int n, unsigned long long *slot);
void launch_add(hipStream_t st, double *dst, const double *a, int n);
void launch_set_rhs(hipStream_t st, double *b, const double *rhs, int nm, int n);
void launch_check_finite(hipStream_t st, const double *v, int n, int *flags);
}  // namespace hipkkt
