// Device-resident image of the symbolic plan + numeric state; passed BY VALUE to the kernels.
#pragma once
#include <stdint.h>

#include "symbolic.h"

namespace hipkkt {

enum { SC_MAXDIAG = 0, SC_NORMB = 1, SC_NORME = 2, SC_COUNT = 8 };  // 64-bit scalar slots
enum { FL_NONFINITE = 0, FL_NREG = 1, FL_FRONTFAIL = 2, FL_FACFAIL = 3, FL_NPOLISH = 4, FL_COUNT = 8 };   // int flags (FL_FACFAIL: k_front_block gave up)

// hand-off slot of the persistent sweeps: a value and its self-validating tag (kernels.hip front_slot_* / seg_slot_*)
struct __attribute__((aligned(16))) FrontSlot {
    double v;
    unsigned long long h;
};

// Device-resident state of one refined solve (kernels.hip k_refine_*): the control flow of the reference's
// _iterative_refinement (kktsolver_directldl.jl:389-449) is decided ON THE DEVICE, so a refined solve needs one host
// synchronisation (at its end) instead of one per refinement step.  The current iterate is xbuf[cur], the candidate
// xbuf[1 - cur] (the reference swaps x and dx).
struct RefineState {
    int32_t cur;        // index of the buffer holding the accepted iterate
    int32_t steps;      // refinement steps taken (LDL solves beyond the first)
    int32_t active;     // 1: the loop of :421-446 would go round again
    int32_t fail;       // 1: non-finite residual norm (solve reports a numerical failure)
    double lastnorme, normb, norme;
};

// One contribution to one target entry of the per-entry gather lists: sum_k L_s[i,k] d_k L_s[j,k] with everything the kernel
// needs in ONE 24-byte load (was: three index arrays, then two supernode tables, then the data: three dependent round trips).
struct GathPair {
    int64_t src;                  // Lx offset of L_s[i, 0]
    int32_t dj;                   // offset of L_s[j, 0] relative to src
    int32_t r;                    // rows of the source panel (column stride)
    int32_t K;                    // its width
    int32_t dfirst;               // first pivot of the source in D
};
// 64 rows of a dense triangle of K (k_spmv_dense_tri, kernels.hip)
struct DenseTriStrip {
    int c0, d, i0, pad;   // the triangle's first row / column, its dimension; first row of the strip inside it
    int64_t col0;         // the triangle's first entry in DevPlan::dtri_col
};
// One update batch of a front factored by ONE launch of k_front_block (front_block.hip): nb consecutive 64-column panels
constexpr int kFbMax = 5;         // panels per launch (= the longest update batch)
struct FrontBatch {
    int64_t fp_off;               // front_panels[fp_off + q], q = 0 .. nb-1
    int64_t scratch_off;          // doubles: kFbMax x (64 x 64 inverse + 64 pivots), then the L tiles of the diagonal workgroups
    int64_t stream_off;           // doubles: records of the streamed pivot chain, (8 j + Bk) * kFbRec for block Bk of the batch's panel j
    int32_t nb, nblk, r0;         // panels, 64-row blocks (= workgroups), rows of the first panel
    int32_t sync_off;             // ints: {ticket, error} | 16: second ticket | 32: Minv flags | 64 + 8 k + j: L(k,j) flags   (128 ints per batch)
    // a launch handles the row blocks [i_base, i_end) and takes its tickets from sync[tick]: one launch = (0, nblk, 0); the look-ahead
    // factorisation (hipkkt_factor.cpp) splits a batch into the launch of the row blocks the next batches need at once (0, R, 0) and
    // the launch of the rest (R, nblk, 16), which finds every hand-off flag already set
    int32_t i_base, i_end, tick, pad;   // pad: tiles per wavefront of the extra workgroups (0 = 1)
    // extra workgroups of the launch (blockIdx >= i_end - i_base): dense update tiles [x_begin, x_begin + x_count) of the PREVIOUS stage,
    // four per workgroup (one wavefront each), on compute units the panel kernel leaves idle (hipkkt_factor.cpp)
    int32_t x_begin, x_count;
};
constexpr int64_t kFbScratch = (int64_t)kFbMax * 4160 + (int64_t)(kFbMax * (kFbMax - 1) / 2) * 4096;   // doubles per batch
constexpr int kFbRec = 768;       // one streamed block of 8 pivots.  front_block2.hip: 12 chunks of 64 lanes (8 of raw columns a_rk = d_k l_rk in
                                  // operand order, 2 of T = L_bb^-T D_b^-1, 2 of the pivots); front_block.hip uses the first 528: [8][64] raw columns, d, 1/d
constexpr int64_t kFbStream = (int64_t)kFbMax * 8 * kFbRec;   // doubles per batch
constexpr int kGathHeavy = 24;    // target entries with more pairs than this get a wavefront of their own

struct DevPlan {
    // structure (read-only after setup)
    const int *sn_first;
    const int64_t *sn_rowptr;
    const int *sn_rows;
    const int64_t *sn_panel;
    const int64_t *sn_diag;
    const int64_t *u_off;
    const int64_t *p_off;
    const int64_t *lt_off;
    const int *inv_list;         // supernodes for k_invert_diag (narrow ones first) / k_invert_diag_wide
    const int *lvl_sn;
    const int *perm;
    const signed char *sgn_perm;
    const FacItem *fac_items;
    const FacRec *fac_recs;      // [fac_items] the same items, self-contained (k_factor_panel)
    const FacItem *slv_items;
    const FacItem *bwd_items;
    const int *rel;
    const UpdTask *upd_tasks;
    const UpdGroup *upd_groups;
    const DenseGroup *dgroups;   // [upd_groups] self-contained records of the dense groups (k_update_dense)
    const int16_t *upd_tmap;
    const DenseTask *dtasks;      // parallel to upd_tasks
    const int64_t *gath_tgt;
    const int64_t *gath_pptr;
    const GathPair *gath_pairs;   // one self-contained record per (target entry, source) pair (k_update_gather)
    const int64_t *gath_heavy;    // entries with more than kGathHeavy pairs: one wavefront each (k_update_gather_heavy)
    const int64_t *g_ptr;
    const int *g_idx;
    const int64_t *kmap;
    const signed char *kdiag_sign;
    const int64_t *sym_rowptr;
    const int *sym_col;
    const int64_t *sym_q;
    const int *long_rows;        // rows of that view with more than long_row_threshold() entries (one workgroup each in the SpMV)
    int n_long_rows;
    const DenseTriStrip *dtri_strips;   // dense triangles of K that left the view (symbolic.h HostPlan::dtri), 64 rows each
    const int64_t *dtri_col;
    int n_dtri_strips;
    double *dense_acc;           // [N] their part of K x, written by k_spmv_dense_tri and added by the view's kernels (per solve context)
    const FrontPanel *front_panels;
    const int64_t *front_gptr;
    const int *front_gidx;
    int *front_sync;
    // persistent level-free sweeps over the regular (non-front) supernodes: dependency counters and lists
    const FacItem *pbwd_items;   // backward order: per level (descending) partial items, then finals (blk = -1)
    const int *dep_total;        // [nsuper] forward items of all same-segment regular children (what fdone[s] must reach)
    const int *sn_nitems;        // forward items (64-row blocks) of a supernode
    const int *sn_bparent;       // same-segment regular parent, or -1
    int *seg_sync;               // [0,nseg) forward tickets, [nseg,2nseg) backward tickets, then per-supernode counters:
    int seg_ticket;              // bit 0 / bit 1: forward / backward segment-sweep items are atomic tickets (default 3); 0: blockIdx
    unsigned spin_limit;         // bound of every spin loop of the persistent sweeps (default 2^20; HIPKKT_SPIN_LIMIT)
    int *sn_polish;              // [nsuper] 1: the block solves with this supernode take one refinement step (written by k_invert_diag_wide per factorisation)
    double polish_tau;           // threshold on the largest |entry| of a wide block's explicit inverse (64; HIPKKT_ACCURATE)
    int dbg;                     // timing experiments only (HIPKKT_DEBUG_FLAGS; results are WRONG when set): bit 0 = the super-block sweeps skip
                                 // their L tile loads (what remains is the hand-off chain)
    int nseg;                    //   fdone[8][nsuper] (kernels.hip kSegSub); error word last   // per front: {ticket, error, flags[np]} (zeroed before every front kernel)
    // numeric state
    double *kval;    // resident, UNREGULARISED triu KKT values (original nz order)
    double *Lx;      // supernodal panels
    double *Ldiag;   // factored unit-lower diagonal blocks
    double *Linv;    // their inverses (column-major) and transposed inverses, for GEMV-style solves
    double *LinvT;
    double *LT;      // row-major copy of the off-diagonal panel rows (w contiguous values per row)
    double *SbInv;   // off-diagonal tiles of the super-block inverses of the fronts (symbolic.h FrontDesc::sbinv_off; front_sweep.hip)
    double *D;
    double *Dinv;
    double *ubuf;    // forward-solve update vectors
    // tagged hand-off of the backward segment sweep (kernels.hip seg_slot_*): 16-byte slots parallel to x / pbuf, the
    // solve epoch, and the copy of sn_rows that marks the rows whose x is produced inside the same launch
    FrontSlot *xseg, *pseg;
    int *seg_epoch;
    const int *rows_seg;         // sn_rows | kSegRowTag where the owning ancestor is solved inside the same launch
    double *pbuf;    // backward-solve partial dot products
    double *scal;    // SC_* slots (raw 64-bit)
    int *flags;      // FL_* slots
};

}  // namespace hipkkt
