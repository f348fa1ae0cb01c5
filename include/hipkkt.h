/*
 * libclarabel_hipkkt.so — C ABI of the MI355X-native KKT linear-system path for Clarabel.jl.
 *
 * This is the drop-in boundary.  Every entry point replaces one
 * method of the reference's KKT-solver plugin contracts; the `ref:` tag on each declaration cites
 * the reference interface it stands in for (paths relative to /root/reference):
 *
 *   seam L1  AbstractKKTSolver        src/kktsolvers/kktsolver_defaults.jl:2-47
 *            (concrete reference impl src/kktsolvers/kktsolver_directldl.jl)
 *   seam L0  AbstractDirectLDLSolver  src/kktsolvers/direct-ldl/directldl_defaults.jl:1-72
 *            (concrete reference impls directldl_qdldl.jl, ext/directldl_pardiso.jl)
 *
 * Conventions
 *   - plain C types only; all pointers are HOST memory unless the name ends in `_dev`
 *     (device pointers on the handle's GPU, e.g. a torch tensor's data_ptr()).
 *   - integer arrays are int64_t holding indices of base `opts.index_base` (1 = exactly what Julia
 *     holds in SparseMatrixCSC.colptr/rowval and LDLDataMap; 0 = numpy/scipy).
 *   - the caller owns every array it passes; the library copies during the call and keeps no host
 *     pointers (Julia's GC may move or free them after ccall returns).
 *   - return value int32_t: 0 = ok; >0 = numerical failure (maps to Julia `false`);
 *     <0 = usage / device error (maps to `error()` in a constructor, `false` elsewhere).
 *     Nothing is thrown across the boundary.
 *   - every call is synchronous (returns after the handle's stream has drained), sets the device
 *     of the handle, and touches no global mutable state: distinct handles may be driven from
 *     distinct threads / processes.  Only Float64 is accelerated.
 */
#ifndef HIPKKT_H
#define HIPKKT_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hipkkt_solver *hipkkt_handle;

/* status codes */
#define HIPKKT_OK 0
#define HIPKKT_NUMERICAL_FAILURE 1   /* non-finite pivot / non-finite refinement residual */
#define HIPKKT_ERR_ARGUMENT (-1)
#define HIPKKT_ERR_DEVICE (-2)
#define HIPKKT_ERR_ALLOC (-3)
#define HIPKKT_ERR_INTERNAL (-4)

/* cone structure codes for hipkkt_create_from_parts (what the KKT pattern needs to know) */
#define HIPKKT_SPARSE_NONE 0
#define HIPKKT_SPARSE_SOC 1      /* SOCExpansionMap,    directldl_datamaps.jl:8-22  (pdim 2) */
#define HIPKKT_SPARSE_GENPOW 2   /* GenPowExpansionMap, directldl_datamaps.jl:81-99 (pdim 3) */

typedef struct hipkkt_opts {
    int32_t index_base;            /* 0 or 1 */
    int32_t supernode_max_width;   /* 0 = default (64) */
    int32_t relax_supernodes;      /* 1 = relaxed amalgamation (default), 0 = fundamental only */
    int32_t update_policy;         /* 0 = right-looking, 1 = left-looking, 2 = batched right-looking (default) */
    int32_t update_batch;          /* policy 2: #levels whose updates are applied together (0 = automatic: 4, or 5
                                      when a front of >= 64 panels dominates the factorisation) */
    int32_t front_min_panels;      /* chains of >= this many panels of one wide supernode are solved by the
                                      persistent front kernels; 0 = default (4), < 0 = never */
    double dynamic_reg_eps;        /* ref: settings.jl:123, passed at directldl_qdldl.jl:21 */
    double dynamic_reg_delta;      /* ref: settings.jl:124, passed at directldl_qdldl.jl:22 */
    double amd_dense_scale;        /* ref: directldl_qdldl.jl:24 (1.5); <=0 = default */
    const int64_t *user_perm;      /* optional fill-reducing order (index_base based); NULL = own AMD */
} hipkkt_opts;

/* defaults = the reference's Settings defaults (src/settings.jl:117-132) */
void hipkkt_default_opts(hipkkt_opts *opts);

/* ref: ldlsolver_is_available(::Val{:hip}) (pattern: ext/directldl_pardiso.jl:144,148).
 * Number of usable HIP devices (0 = not available).  Never fails. */
int32_t hipkkt_is_available(void);
/* Version of THIS interface.  A binding compares it with the HIPKKT_ABI_VERSION it was written against when it loads the library
 * and refuses a mismatch (the Julia glue and the ctypes mirror do): signatures may change between versions, never within one.
 * 4: hipkkt_get_profile / hipkkt_get_counters take the capacity of the caller's buffer (they wrote a fixed, growing number of values). */
#define HIPKKT_ABI_VERSION 4
int32_t hipkkt_abi_version(void);
/* releases the process-wide cache of device memory blocks the library keeps between handles (not in the reference: an embedding
 * host under memory pressure may call it at any time; live handles are unaffected) */
int32_t hipkkt_trim_cache(int32_t device_id);

/* ---- construction ------------------------------------------------------------------------ */

/* seam L0.  ref: ctor S{T}(KKT::SparseMatrixCSC{T},Dsigns,settings), directldl_qdldl.jl:6-28 /
 * ext/directldl_pardiso.jl:33-55.  KKT is the N x N :triu CSC image (diagonal present and LAST in
 * every column, as directldl_kkt_assembly.jl:161-165 guarantees).  Symbolic analysis only. */
int32_t hipkkt_create(int32_t device_id, int64_t N, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, const int64_t *dsigns, const hipkkt_opts *opts,
                      hipkkt_handle *out);

/* seam L1.  ref: DirectLDLKKTSolver{T}(P,A,cones,m,n,settings), kktsolver_directldl.jl:46-92:
 * assembles the :triu KKT image + LDLDataMap (directldl_kkt_assembly.jl:15-175,
 * directldl_datamaps.jl:170-214), Dsigns (kktsolver_directldl.jl:112-126), then the symbolic
 * analysis.  P is n x n :triu CSC, A is m x n CSC.  Per cone: numel, hs_dense (0 = diagonal Hs
 * block, 1 = dense packed-triu block), sparse_kind (HIPKKT_SPARSE_*), dim1 (GenPow only). */
int32_t hipkkt_create_from_parts(int32_t device_id, int64_t n, int64_t m,
                                 const int64_t *Pcolptr, const int64_t *Prowval, const double *Pnzval,
                                 const int64_t *Acolptr, const int64_t *Arowval, const double *Anzval,
                                 int64_t ncones, const int64_t *cone_numel, const int32_t *cone_hs_dense,
                                 const int32_t *cone_sparse_kind, const int64_t *cone_dim1,
                                 const hipkkt_opts *opts, hipkkt_handle *out);

/* ref: finalizer of the Julia wrapper struct (MOI.empty! calls finalize(solver),
 * src/MOI_wrapper/MOI_wrapper.jl:133).  Idempotent on NULL. */
void hipkkt_destroy(hipkkt_handle h);

/* ---- introspection ------------------------------------------------------------------------ */

/* out[0..15] = N, n, m, p, nnzK, nHs, nsparse, nnzP, nnzA, nnzL (strictly-lower entries of L,
 * structural), n_supernodes, n_levels, panel_doubles (supernodal storage incl. padding zeros),
 * n_update_tasks, etree_height (columns), ordering in use (0 minimum degree on K, 1 cone rows first /
 * variables last, 2 user, 3 nested dissection).  (n,m,p,nHs,nnzP,nnzA are 0 for L0 handles.) */
int32_t hipkkt_get_dims(hipkkt_handle h, int64_t *out16);

/* ref: linear_solver_info(ldlsolver) -> LinearSolverInfo(name,threads,direct,nnzA,nnzL),
 * directldl_qdldl.jl:35-42 / src/types.jl:198-206.  name = :hip, threads = 1 (host), direct = true. */
int32_t hipkkt_info(hipkkt_handle h, int64_t *nnzA, int64_t *nnzL);

/* flop / byte model of one numeric factorisation and one solve, from the symbolic factor actually
 * used:  out[0]=sum_j c_j^2+3c_j (factor flops), out[1]=executed factor flops incl.
 * supernode padding, out[2]=solve flops (4 nnzL + N), out[3]=algorithmic factor bytes,
 * out[4]=algorithmic solve bytes, out[5]=spmv bytes, out[6]=dense-update (MFMA) flops, out[7]=the part of out[6] executed by k_update_dense */
int32_t hipkkt_get_cost_model(hipkkt_handle h, double *out8);

/* copies of host-side structures (tests, Julia-side bookkeeping).  Any pointer may be NULL. */
int32_t hipkkt_get_kkt(hipkkt_handle h, int64_t *colptr, int64_t *rowval, double *nzval);
/* perm[k] = index eliminated k-th, of the factorisation IN USE: while the last factorisation lives in the robust-order twin
 * (hipkkt_get_counters out[4]) that is the twin's minimum-degree order, otherwise the handle's own */
int32_t hipkkt_get_perm(hipkkt_handle h, int64_t *perm);
int32_t hipkkt_get_dsigns(hipkkt_handle h, int64_t *dsigns);
/* which: 0 = map.P, 1 = map.A, 2 = map.Hsblocks, 3 = map.diagP, 4 = map.diag_full
 * (ref: LDLDataMap fields, directldl_datamaps.jl:170-181) */
int32_t hipkkt_get_map(hipkkt_handle h, int32_t which, int64_t *out);
/* sparse map i: which 0/1/2 = index vectors (SOC: u,v ; GenPow: q,r,p), 3 = D; *len receives length */
int32_t hipkkt_get_sparse_map(hipkkt_handle h, int64_t i, int32_t which, int64_t *out, int64_t *len);

/* ---- value updates ------------------------------------------------------------------------ */

/* ref: update_values!(ldlsolver,index,values) / scale_values!(ldlsolver,index,scale),
 * directldl_qdldl.jl:46-69, called from kktsolver_directldl.jl:130-188. */
int32_t hipkkt_update_values(hipkkt_handle h, const int64_t *index, const double *values, int64_t k);
int32_t hipkkt_scale_values(hipkkt_handle h, const int64_t *index, int64_t k, double scale);

/* ref: kktsolver_update! first half, kktsolver_directldl.jl:223-228: Hs as produced by
 * get_Hs!(cones,Hsblocks); negated and scattered through map.Hsblocks on the device. */
int32_t hipkkt_set_hs(hipkkt_handle h, const double *hs, int64_t nHs);
/* the _dev form does not synchronise with the host: hs_dev must stay valid until the next synchronising call on this handle
 * (hipkkt_refactor, any solve) has returned */
int32_t hipkkt_set_hs_dev(hipkkt_handle h, const double *hs_dev, int64_t nHs);

/* SURVEY section 8(f) row N1 (first step): the Hs block of PSD triangle cones formed ON THE DEVICE.
 * ref: get_Hs!(::PSDTriangleCone) = pack_triu(skron(R R^T)), coneops_psdtrianglecone.jl:153-161, 502-540.
 * For cone c (c = 0..npsd-1): dim[c] = matrix side n, w_all holds its dense symmetric W = R R^T (n x n, row-major,
 * cones concatenated), hs_off[c] = 0-based offset of the cone's block inside the Hs vector (as used by
 * hipkkt_set_hs).  The packed upper triangle of  W (x)_s W  (numel(numel+1)/2 values, numel = n(n+1)/2) is computed,
 * negated and scattered through map.Hsblocks like hipkkt_set_hs does -- with the same products, sums and rounding
 * as the reference's skron! loop, so K is bit-identical to the host path.  Saves the O(numel^2) host loop and the
 * upload of the block (813 450 doubles per 50 x 50 cone). */
int32_t hipkkt_set_hs_psd(hipkkt_handle h, int64_t npsd, const int64_t *hs_off, const int64_t *dim, const double *w_all);

/* SURVEY section 8(f) row N1 (completion): update_scaling! + get_Hs! of the symmetric cones ON THE DEVICE -- the caller ships the
 * iterate (s, z) (2 m doubles) instead of the Hs vector and the (u, v, eta) of every second-order cone.
 *   hipkkt_set_cone_types   once per handle: kinds[c] = 0 ZeroCone, 1 NonnegativeCone, 2 SecondOrderCone, 3 PSDTriangleCone
 *                           (cone_types.jl:6-67), anything else = a cone the caller keeps updating with hipkkt_set_hs / _set_genpow.
 *                           Checked against the (numel, hs_dense, sparse_kind) the handle was created with.
 *   hipkkt_update_scaling   per IPM iteration, replaces update_scaling!(cones,s,z,mu,PrimalDual) + get_Hs! + the Hs / sparse-cone
 *                           part of _kktsolver_update_inner! (kktsolver_directldl.jl:197-245):
 *      Zero          Hs = 0                                                      coneops_zerocone.jl:91
 *      Nonnegative   lambda = sqrt(s z), w = sqrt(s/z), Hs = w^2 (bit-exact)     coneops_nncone.jl:77-101
 *      SecondOrder   eta, w, lambda, and (d, u, v) of the sparse form or the dense block for dim <= 4, with the reference's
 *                    expressions and association (sums of squares are tree sums)  coneops_socone.jl:75-192
 *      PSDTriangle   psd_R = the cones' R factors (n x n, column-major, concatenated; the Cholesky / SVD of
 *                    coneops_psdtrianglecone.jl:78-143 stay with the caller): W = R R^T and triu(skron(W)) are formed on the
 *                    device (:145-161, :502-540).  psd_R = NULL leaves the PSD blocks to hipkkt_set_hs_psd.
 *    Outputs (any may be NULL): w_out, lambda_out of length m in cone order (NN: w, lambda; SOC: the normalised w and lambda; zero
 *    on other rows) -- what the host loop needs for mul_Hs! / step directions; soc_eta_out[k] = eta of the k-th second-order cone;
 *    *scaling_ok = 0 when a second-order cone's s or z is not interior (update_scaling! returns false, coneops_socone.jl:88-90).
 *    The resident KKT values, w_out, lambda_out and soc_eta_out are then UNSPECIFIED (the other cones have been rewritten, the
 *    failing cone partly): like the reference, which never reaches get_Hs! in that case, the caller must not factor or read them
 *    before a later call has succeeded.
 *   hipkkt_update_scaling_dev  the same with every pointer except scaling_ok in device memory (s, z resident: no PCIe at all). */
int32_t hipkkt_set_cone_types(hipkkt_handle h, int64_t ncones, const int32_t *kinds);
int32_t hipkkt_update_scaling(hipkkt_handle h, const double *s, const double *z, const double *psd_R, double *w_out,
                              double *lambda_out, double *soc_eta_out, int32_t *scaling_ok);
int32_t hipkkt_update_scaling_dev(hipkkt_handle h, const double *s_dev, const double *z_dev, const double *psd_R_dev,
                                  double *w_out_dev, double *lambda_out_dev, double *soc_eta_out_dev, int32_t *scaling_ok);
/* ref: _csc_update_sparsecone(::SecondOrderCone,...), directldl_datamaps.jl:61-79 */
int32_t hipkkt_set_soc(hipkkt_handle h, int64_t sparse_idx, double eta2, const double *u, const double *v,
                       int64_t dim);
/* all sparse SOC cones in one call (same arithmetic; one upload + one launch instead of 5 per cone):
 * eta2[nsoc], u_all / v_all = the cones' u and v vectors concatenated in sparse-map order */
int32_t hipkkt_set_soc_batch(hipkkt_handle h, int64_t nsoc, const double *eta2, const double *u_all,
                             const double *v_all, int64_t total);
/* ref: _csc_update_sparsecone(::GenPowerCone,...), directldl_datamaps.jl:146-167 */
int32_t hipkkt_set_genpow(hipkkt_handle h, int64_t sparse_idx, double sqrtmu, const double *p,
                          const double *q, const double *r);
/* ref: kktsolver_update_P!/A!, kktsolver_directldl.jl:374-386 */
int32_t hipkkt_update_P(hipkkt_handle h, const double *Pnzval, int64_t nnzP);
int32_t hipkkt_update_A(hipkkt_handle h, const double *Anzval, int64_t nnzA);

/* ---- factor ------------------------------------------------------------------------------- */

/* ref: _kktsolver_regularize_and_refactor!, kktsolver_directldl.jl:247-310 with
 * refactor!(ldlsolver,K), directldl_qdldl.jl:72-81.  eps = eps_const + eps_prop*max|diag K|;
 * K_fact = K + eps*diag(Dsigns); numeric LDL^T with the sign-driven dynamic pivot substitution;
 * the unregularised K stays resident for iterative refinement.  Returns 0, or
 * HIPKKT_NUMERICAL_FAILURE when some pivot inverse is non-finite.  eps_used / n_dynamic_reg may be NULL. */
int32_t hipkkt_refactor(hipkkt_handle h, int32_t static_reg_enable, double eps_const, double eps_prop,
                        double *eps_used, int64_t *n_dynamic_reg);

/* ---- solve -------------------------------------------------------------------------------- */

/* ref: kktsolver_setrhs!, kktsolver_directldl.jl:313-327: b = [rhsx; rhsz; 0_p] */
int32_t hipkkt_setrhs(hipkkt_handle h, const double *rhsx, const double *rhsz);
int32_t hipkkt_setrhs_dev(hipkkt_handle h, const double *rhs_dev /* n+m contiguous; not synchronised with the host: valid until the solve has returned */);
/* ref: kktsolver_solve!, kktsolver_directldl.jl:346-371 incl. _iterative_refinement :389-449 and
 * kktsolver_getlhs! :330-343.  lhsx / lhsz may be NULL (Julia `nothing`).  ir_steps may be NULL. */
int32_t hipkkt_solve(hipkkt_handle h, double *lhsx, double *lhsz, int32_t ir_enable, double reltol,
                     double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps);
/* same, result left on the device: lhs_dev receives n+m doubles (may be NULL to discard) */
int32_t hipkkt_solve_dev(hipkkt_handle h, double *lhs_dev, int32_t ir_enable, double reltol,
                         double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps);
/* SURVEY section 8(f) row N2.  nrhs right-hand sides on ONE factorisation, each refined exactly like hipkkt_solve does,
 * two at a time on concurrent solve contexts (the triangular sweeps are bound by dependency latency, so two of them
 * overlap almost completely).  The caller this serves: kkt_update! leaves the constant-rhs solve of kktsystem.jl:80-92
 * pending and the first kkt_solve! of the iteration (:135-215, affine step) solves [-q; b] and its own right-hand side
 * together.  rhsx = nrhs x n, rhsz = nrhs x m, lhsx / lhsz likewise (row-major, one right-hand side after the other;
 * lhs pointers may be NULL); ir_steps[nrhs] may be NULL.  Returns 0, or HIPKKT_NUMERICAL_FAILURE if any solve failed. */
int32_t hipkkt_solve_multi(hipkkt_handle h, int64_t nrhs, const double *rhsx, const double *rhsz, double *lhsx, double *lhsz,
                           int32_t ir_enable, double reltol, double abstol, int64_t max_iter, double stop_ratio,
                           int64_t *ir_steps);
/* same with everything resident: rhs_dev / lhs_dev = nrhs x (n+m) doubles on the device */
int32_t hipkkt_solve_multi_dev(hipkkt_handle h, int64_t nrhs, const double *rhs_dev, double *lhs_dev, int32_t ir_enable,
                               double reltol, double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps);
/* SURVEY section 8(f) row N2, second half.  ref: kkt_solve!(kktsystem, lhs, rhs, data, variables, cones, steptype),
 * src/kktsystem.jl:135-215 -- everything between the caller's cone algebra (the vector c of  Hs dz + ds = -c, :152-163) and
 * mul_Hs! (:203): the solve  K [x1; z1] = [rhs.x; c - rhs.z]  (:167-170), the numerator and denominator of d tau with
 * quad_form(., P, .) (:176-190, mathutils.jl:299-337) from the q, b made resident by hipkkt_set_qb and the resident P values,
 * and  dx = x1 + dtau x2,  dz = z1 + dtau z2  (:194-196).  ONE PCIe round trip and one host synchronisation per kkt_solve!.
 *   rhs_x[n], workz[m] (= c - rhs.z), var_x[n] (= variables.x) in;  scal_in = {variables.tau, variables.kappa, rhs.tau, rhs.kappa}
 *   const_pending != 0: the constant-rhs solve of _kkt_solve_constant_rhs! (:80-92), K [x2; z2] = [-q; b], left pending by
 *     kkt_update!, runs here next to (x1, z1) on the second solve context and its solution becomes the resident (x2, z2);
 *     const_pending == 0: (x2, z2) of the last such call (same factorisation) is used
 *   lhs_x[n], lhs_z[m] out (host; either may be NULL);  scal_out[10] = {dtau, tau_num, tau_den, q.x1, b.z1, xi'P x1, q.x2, b.z2,
 *     (xi - x2)'P(xi - x2), x2'P x2} (host, required);  ir_steps[2] = refinement steps of the (x1, z1) / (x2, z2) solve (may be NULL)
 * The _dev form takes in_dev = [rhs_x | workz | var_x] (2n + m doubles) and writes lhs_dev = [dx | dz] (n + m doubles, may be NULL:
 * the step then stays in the handle and only the scalars cross PCIe).  Returns 0, HIPKKT_NUMERICAL_FAILURE when a solve failed
 * (the reference returns is_success = false), < 0 on usage / device errors (hipkkt_set_qb not called: HIPKKT_ERR_ARGUMENT).
 * ON ANY NON-ZERO RETURN lhs_x / lhs_z / lhs_dev / scal_out ARE UNSPECIFIED: the reduction and the copies of the step are enqueued
 * behind the solves before their outcome is known (that is what makes it one synchronisation), so they may hold a rejected or
 * non-finite iterate.  This differs from hipkkt_solve, which like kktsolver_getlhs! writes lhs only on success; the reference's
 * caller returns at once on is_success = false (kktsystem.jl:172) and never reads lhs. */
int32_t hipkkt_kkt_solve_reduced(hipkkt_handle h, const double *rhs_x, const double *workz, const double *var_x, const double *scal_in4,
                                 int32_t const_pending, double *lhs_x, double *lhs_z, double *scal_out10, int32_t ir_enable,
                                 double reltol, double abstol, int64_t max_iter, double stop_ratio, int64_t *ir_steps2);
int32_t hipkkt_kkt_solve_reduced_dev(hipkkt_handle h, const double *in_dev, const double *scal_in4, int32_t const_pending,
                                     double *lhs_dev, double *scal_out10, int32_t ir_enable, double reltol, double abstol,
                                     int64_t max_iter, double stop_ratio, int64_t *ir_steps2);
/* seam L0.  ref: solve!(ldlsolver,K,x,b), directldl_qdldl.jl:85-96: x = K_fact^{-1} b, length N,
 * no refinement (the Julia-side DirectLDLKKTSolver refines).  x and b must not alias. */
int32_t hipkkt_ldl_solve(hipkkt_handle h, double *x, const double *b);

/* SURVEY section 8(f) row N4 (building block): the sparse products of residuals_update!, residuals.jl:12-25, from the
 * P and A values resident on the device (L1 handles): Px = Symmetric(P) x, ATz = A' z, Ax = A x.
 * x[n], z[m] in; Px[n], ATz[n], Ax[m] out (host memory; any of the outputs may be NULL). */
int32_t hipkkt_block_products(hipkkt_handle h, const double *x, const double *z, double *Px, double *ATz, double *Ax);

/* residuals_update!(residuals, variables, data), residuals.jl:1-37, computed on the device from
 * the resident P and A values (L1 handles).  hipkkt_set_qb makes q[n] and b[m] resident (once per problem, again after
 * update_q!/update_b!).  hipkkt_residuals: x[n], z[m], s[m], tau, kappa in; rx[n], rz[m], rx_inf[n], rz_inf[m], Px[n] out
 * (host memory, any may be NULL); scal5 = {dot_qx, dot_bz, dot_sz, dot_xPx, r_tau} (host, required).  The dot products
 * are summed in a fixed order (deterministic).  The _dev form takes xzs_dev = [x | z | s] and writes
 * out_dev = [rx | rz | rx_inf | rz_inf | Px] (3n + 2m doubles) without touching PCIe except for the five scalars. */
int32_t hipkkt_set_qb(hipkkt_handle h, const double *q, const double *b);
int32_t hipkkt_residuals(hipkkt_handle h, const double *x, const double *z, const double *s, double tau, double kappa,
                         double *rx, double *rz, double *rx_inf, double *rz_inf, double *Px, double *scal5);
int32_t hipkkt_residuals_dev(hipkkt_handle h, const double *xzs_dev, double tau, double kappa, double *out_dev, double *scal5);

/* ---- timing (device time on the handle's stream, HIP events) ------------------------------- */
/* out[0] = ms of last refactor (value scatter + numeric LDL), out[1] = ms of last solve call
 * (all LDL solves + SpMVs of the refinement), out[2] = accumulated refactor ms, out[3] = accumulated
 * solve ms, out[4] = #refactors, out[5] = #solve calls, out[6] = #LDL solves, out[7] = ms of the
 * dense-update kernels inside the last refactor (0 unless profiling was enabled) */
int32_t hipkkt_get_timing(hipkkt_handle h, double *out8);
int32_t hipkkt_reset_timing(hipkkt_handle h);
/* last refactorisation run with profiling enabled: out[0] = ms of all Schur-update kernels, out[1] = ms of the
 * k_update_dense<4,4> launches alone (one wavefront per 64x64 tile), out[2] = their algorithmic flops, out[3] = their
 * number; out[4] = ms of the k_front_block launches (one per update batch of a front), out[5] = their number, out[6] = the panels
 * they factor, out[7] = the Schur-update flops of the stages they absorb (not part of out[0]'s kernels), out[8] / out[9] = dense update
 * tiles / their flops that rode in those launches as extra workgroups instead of in their stage's own launch (the partial last round
 * of a front batch's far updates; not part of out[0] .. out[2] either); out[10] = the wide diagonal blocks (> 16 columns) of the last factorisation
 * OUTSIDE the fronts whose explicit inverse has an entry above 64 in magnitude: the solve kernels take one refinement step
 * y += Linv (b - L y)  on these (a product with such an inverse is several times less accurate than the reference's substitution; the
 * 64-column panels of a front are solved by the front sweeps, which never refine, and are not counted; testing build: switch
 * ACCURATE=<threshold>, 0 = never, negative = every wide block), out[11] = factorisations so far with at least one */
int32_t hipkkt_get_profile(hipkkt_handle h, double *out, int64_t cap);   /* writes min(cap, 12) values */
/* the k_update_dense<4,4> launches of that refactorisation one by one: ms[i], algorithmic flops[i], target tiles[i]
 * (any array may be NULL; at most cap entries are written, *count receives the number of launches) */
int32_t hipkkt_get_profile_launches(hipkkt_handle h, double *ms, double *flops, double *tiles, int64_t cap, int64_t *count);
/* 1 = time the update (MFMA) kernels and the k_front_block launches separately inside refactor (eager launches, adds event overhead);
 * 2 = the same with every far update tile kept in its stage's own launch (none riding in the next k_front_block launch): the
 * comparison figure for out[8] / out[9] of hipkkt_get_profile; 0 = off */
int32_t hipkkt_set_profiling(hipkkt_handle h, int32_t enable);

/* robustness counters: out[0] = persistent sweep time-outs seen so far (each one repeats the solve on the per-level
 * kernels and suspends the persistent kernels for a while), out[1] = persistent sweeps currently enabled (0/1),
 * out[2] = factorisations repeated on the robust-order twin, out[3] = twin exists, out[4] = the current factorisation
 * lives in the twin, out[5] = ordering in use (0 minimum degree on K, 1 cone rows first, 2 user), out[6] = #fronts,
 * out[7] = #segments, out[8] = update batches of fronts factored by one launch each (front_block.hip), out[9] = that path is
 * enabled (0 after one of its hand-offs timed out: the handle then keeps one launch per panel), out[10] / out[11] = symbolic plans
 * taken from / not found in the process-wide plan cache (same KKT pattern and options => the analysis of an earlier handle is reused;
 * testing build: switch PLAN_CACHE=0 disables it), out[12] = the pivot chain of the front batches is streamed block by block (front_block2.hip; 0 only in the testing build with
 * the switch FB_STREAM=0), out[13] = factorisations with refined block solves (hipkkt_get_profile out[11]), out[14] = target entries of the per-entry gather lists
 * that every factorisation applies on a side stream next to the levels of the following update batch (round 6).  Writes min(cap, 15) values. */
int32_t hipkkt_get_counters(hipkkt_handle h, int64_t *out, int64_t cap);

/* developer diagnostic, not part of the plugin contract: copies an internal vector of the last LDL solve (what = 0 the
 * permuted right-hand side, 1 z = D^-1 L^-1 b, 2 x in permuted order, 3 the forward update vectors, 4 the unregularised KKT values in nz order, 5 D and 6 1/D of the last factorisation, 7 / 8 the u / v vectors of the sparse second-order cones) or a plan table converted to doubles (10 supernode first
 * columns, 11 levels, 12 rows per supernode, 13 parents, 14 membership in the persistent sweeps; 15 / 16 the stream records / scratch
 * tiles of the front batches, 17 the refinement state of the last refined solve: ||e|| before its last step, ||b||, ||e|| after it, steps; 18 the residual b - K x of
 * the last SpMV on solve context 0 (original ordering), 19 the number of dense triangles of K outside the symmetric view, 20 six values per
 * dense update tile of the plan: stage, tasks, sum of source widths, tasks through a tile map, full-tile flag, sum of rows x columns x
 * width; 21 five values per supernode of the persistent segment sweeps: level, width, rows, longest and mean gather list of its row
 * slots -- tools/dense_stage_stats.py); *len receives the length, nothing is copied when cap is too small */
int32_t hipkkt_debug_dump(hipkkt_handle h, int32_t what, double *out, int64_t cap, int64_t *len);
/* developer diagnostic, host logic only (no device needed): how many of the `nd` dense tiles of a front batch's far stage -- the
 * first `ncrit` of them belong to the next batch's columns -- ride in the next k_front_block launch, which has `next_blk` workgroups
 * of its own; *per_wave receives the tiles per wavefront there (hipkkt_factor.cpp fb_extra_tiles_of_stage) */
int32_t hipkkt_debug_extra_tiles(int32_t nd, int32_t ncrit, int32_t next_blk, int32_t *per_wave);
/* test / experiment switches -- NOT part of the plugin contract, the Julia glue never calls it.  key = the switch's name ("FB_V2",
 * "SPLIT_K", "ACCURATE", "SPIN_LIMIT", ...: struct DebugOpts in csrc/hipkkt_internal.h lists them), value = its setting as text, NULL =
 * back to the default; key == NULL resets every switch.  A handle reads the switches when it is created.  Only the TESTING build of
 * the library (libclarabel_hipkkt_testing.so, compiled with -DHIPKKT_TESTING; what tests/ load) accepts a setting: the production
 * library returns HIPKKT_ERR_ARGUMENT for every key, contains neither the first form of the front-batch kernel nor the debug flags
 * of the sweeps, and reads no environment variable for any switch (HIPKKT_VERBOSE and HIPKKT_FB_TRACE, which change no result, are
 * the only two it reads).  Process-wide, not thread-safe against concurrent creates. */
int32_t hipkkt_debug_set(const char *key, const char *value);
/* 1 in the testing build, 0 in the production library */
int32_t hipkkt_debug_is_testing_build(void);

/* diagnostic: checks the FP64 matrix-core operand/result lane maps used by the update kernel
 * against a host product with an asymmetric B (returns 0 when they agree to 1e-12) */
int32_t hipkkt_selftest_mfma(int32_t device_id, double *max_err);

/* diagnostic, no handle needed: what the latency-bound kernels of the factorisation (k_front_block, k_factor_panel, the sweeps) depend
 * on and the matrix-core- / memory-bound ones do not.  out[0] = the shader clock (GHz) one busy wavefront really gets (shader cycles
 * against the 100 MHz constant clock), out[1] / out[2] = round trip (ns) of a flag between two workgroups on the SAME / on DIFFERENT
 * XCDs with relaxed agent-scope atomics (the hand-offs of the front kernels; -1 = no such pair), out[3] = XCDs seen, out[4] / out[5] =
 * the runtime's core / memory clock (MHz), out[6] = compute units, out[7] = reserved.  Writes min(cap, 8) values.  bench.py prints it
 * in its JSON line so that a slow line names its cause. */
int32_t hipkkt_box_probe(int32_t device_id, double *out, int64_t cap);

/* last error text for this handle (or for a failed create when h == NULL); never NULL */
const char *hipkkt_last_error(hipkkt_handle h);

#ifdef __cplusplus
}
#endif
#endif
