// Host-side symbolic analysis for the supernodal LDL^T on MI355X.
// Produces the static "plan" every numeric kernel interprets: permutation, supernode partition,
// dense panel layout in HBM, per-level factor items, target-tile-owned update tasks, gather lists
// for the triangular solves and the symmetric SpMV index.  Runs once per problem on the host
// ("host C++ acceptable").
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace hipkkt {

constexpr int kMaxSnWidth = 64;   // supernode (panel) width cap = LDS-resident diagonal block
constexpr int kUpdRows = 64;      // target row-block owned by one workgroup in the update kernel
constexpr int kNarrowW = 8;       // solve: supernodes this narrow ...
constexpr int kNarrowR = 32;      // ... and with at most this many rows below the diagonal block: one thread per supernode
constexpr int kFacRows = 64;      // panel rows handled per workgroup in the TRSM part

// one contribution of a factored source panel to a target row-block (gather-GEMM-scatter)
struct UpdTask {
    int32_t src;        // source supernode
    int32_t row_lo;     // first source row (index into the source's row list) handled here
    int32_t nrows;      // #source rows (their target positions fall in one row-block)
    int32_t col_lo;     // first source row of the run that lies inside the target's columns
    int32_t ncols;      // length of that run
    int32_t rel_off;    // offset into rel[] of source row `col_lo` (rel[rel_off + (i - col_lo)])
    int32_t vt_begin;   // sparse groups: index (within the group) of this task's first wave-task (16-column
                        // strip); dense groups with bit 17 of geom set: index of this task's tile map
    int32_t geom;       // dense-tile geometry: bits 0-7 tile row of source row `row_lo`, bits 8-15 tile
                        // column of source row `col_lo`, bit 16 = rows AND columns land contiguously,
                        // bit 17 = gathered through upd_tmap[128 * vt_begin ..]: [0,64) tile row -> source row
                        // offset from row_lo (or -1), [64,128) tile column -> offset from col_lo (or -1)
};

// device-side packed record of one contribution to a dense tile: everything k_update_dense needs in ONE 48-byte
// load (built at set-up from UpdTask + the supernode arrays), so that the next task's record can be prefetched
struct DenseTask {
    int64_t panel_off;             // source panel in Lx
    int32_t r8, K, dfirst;         // source row count * 8 (byte stride of a column), width, first pivot
    int32_t row_lo, nrows, col_lo, ncols, geom, map, pad;
};

struct UpdGroup {
    int32_t tgt;        // target supernode
    int32_t row_base;   // first target panel row of the block
    int32_t task_begin, task_end;
    int32_t nvt;        // number of wave-tasks (sum over tasks of ceil(ncols/16))
    int32_t dense;      // 0: relative-index scatter (k_update_stage); 1: register tile, all sources accumulated
                        // (k_update_dense, tile maps where a source does not land contiguously); 2: tiny scattered
                        // contributions, summed per target entry (k_update_gather)
};

struct FacItem {
    int32_t sn;
    int32_t blk;        // row-chunk index inside the panel's off-diagonal part
};

// Device-side record of a dense update group (target tile) for k_update_dense: everything in one 32-byte load.
struct DenseGroup {
    int64_t tile_off;                      // Lx offset of the tile's first row in the target panel
    int32_t rt, nrt, wt;                   // target panel rows, rows of this tile (<= 64), target width
    int32_t task_begin, task_end, pad;
};

// Split-K of a dense target tile (hipkkt_setup.cpp plan_split_k): the tile's contributions are accumulated by `nparts` wavefronts into
// partial tiles (64 x 64, column-major, at Lx + scratch_off + q * 4096) and added to the target in a fixed order by k_split_reduce
struct SplitRec {
    int64_t tile_off, scratch_off;
    int32_t rt, nrt, wt, nparts;
};

// Device-side record of a factor item for k_factor_panel: everything the kernel needs in one load.
struct FacRec {
    int64_t panel_off, diag_off, lt_off;   // offsets into Lx / Ldiag / LT
    int32_t f, w, r, blk;                  // first column, width, rows, row-chunk index
};

// A "front" = one wide (fundamental) supernode that the width cap split into a chain of np panels of
// cw columns (the last may be narrower).  Its panels have nested row structures (panel p holds the
// front rows [cw*p, rF)), so the triangular solves over the chain are done by ONE persistent kernel
// per sweep (k_front_fwd / k_front_bwd) instead of np dependent launches.
struct FrontPanel {
    int64_t panel_off, lt_off, diag_off;   // offsets into Lx / LT / Ldiag(Linv)
    int32_t r, w, f, sn;                   // rows, width, first (permuted) column, supernode id
};
struct FrontDesc {
    int32_t s0, np, cw, W;        // first panel (supernode id), #panels, panel width, own columns
    int32_t rF, nb;               // rows of the front (= rows of panel 0), #row blocks (np + ceil((rF-W)/64))
    int32_t level_first, level_last;
    int64_t gptr_off;             // front_gptr[gptr_off + i .. +1]: external children's ubuf entries on front row i
    int64_t fp_off;               // front_panels[fp_off + p]
    int64_t ubelow_off;           // u_off of the last panel (its off-diagonal rows = the rows below the front)
    int64_t rows_off;             // sn_rowptr[s0]: global (permuted) index of every front row
    int32_t sync_off, sync_blk;   // sync area of this front: two blocks of sync_blk ints {ticket, error, flags[np], slots},
                                  // one per sweep; each sweep's kernel re-zeroes the OTHER block for the next solve
    // super-block sweeps (kernels.hip k_front_fwd_sb / k_front_bwd_sb): the np panels are grouped into nsb super-blocks of sb_g
    // consecutive panels; the off-diagonal 64 x 64 tiles of every super-block's INVERSE are formed after each factorisation
    // (k_invert_super) at sbinv_off: tile (B, bl > cl) at ((B * sb_g (sb_g - 1) / 2 + bl (bl - 1) / 2 + cl) * 8192: 4096 doubles
    // column-major (forward sweep), then 4096 row-major (backward sweep).  sb_g = 0: one hop per panel (k_front_fwd / _bwd).
    int32_t sb_g, nsb;
    int64_t sbinv_off;
};
constexpr int kSbG = 8;           // panels per super-block of the front sweeps
constexpr int kSbMinPanels = 10;  // fronts with fewer panels always keep one hop per panel (a second super-block must exist)
constexpr int kSbMaxPanels = 1024; // ... and so do fronts with more (the sweeps keep a panel table in LDS)

struct DenseTri {
    int c0, d;        // first row / column (original ordering), dimension
    int64_t col0;     // first entry of its columns' offsets in HostPlan::dtri_col
};

struct PlanOptions {
    int max_width = kMaxSnWidth;
    bool relax = true;
    int update_policy = 2;  // 0 right-looking, 1 left-looking, 2 batched right-looking
    int update_batch = 0;   // levels per batch for policy 2 (0 = automatic: 4, or 5 for fronts of >= 64 panels)
    double amd_dense_scale = 1.5;
    double dense_min_cover = 64.0;   // a target tile takes the matrix-core path when its sources cover at least this
                                     // many entries each on average (else: per-entry gather lists)
    int n_hold = 0;            // > 0: also try the "variables last" order (nodes < n_hold held back) and keep
                               // whichever order predicts fewer factor flops
    int front_block_min_width = 1024;   // supernodes at least this wide are cut into full 64-column panels (remainder last) so that
                               // k_front_block can take their update batches; narrower ones keep balanced panel widths (measured:
                               // cfg 1's 710-column root is 15 % slower to factor with 11 x 64 + 6 than with 12 x 60)
    int front_min_panels = 4;  // chains at least this long are solved by the persistent front kernels (0 = never)
    int superhop = 16;         // fronts of at least max(this, kSbMinPanels) panels are swept super-block by super-block (2 hand-offs per kSbG
                               // panels); 0 = never.  Below ~14 panels the hops saved per unit (8 sweeps on the critical path) cost less than
                               // the 0.11-0.13 ms of k_invert_super per factorisation (measured on cfg 1's 12-panel root: 917 -> 869 units/s)
    int nd_mode = 1;           // nested dissection candidate: 0 never, 1 when the latency + throughput model predicts a
                               // >= 20 % cheaper KKT iteration than minimum degree, 2 always
    int nd_leaf = 256;         // subgraphs of at most this many nodes are ordered by minimum degree
    int dense_tri_first_col = -1;   // >= 0: dense triangles of K at or right of this column leave the symmetric view (HostPlan::dtri); the
                               // L1 seam passes n: the kernels that walk the view for P and A (residuals, reduced system) never need Hs.
                               // Needs sorted row indices inside the columns of K (the assembled image has them)
    int dense_tri_min_dim = 128;
    // called (synchronously, from build_plan) with the minimum-degree order on K just BEFORE the "cone rows first" candidate is
    // evaluated against it: the caller may start preparing the robust fallback (a twin analysed in that order) speculatively
    // while this analysis goes on; HostPlan::ordering_used tells afterwards whether it will ever be needed
    std::function<void(const std::vector<int> &)> on_alternative_order;
    // polled at every phase boundary: build_plan returns "cancelled" once it reads true
    const std::atomic<bool> *cancel = nullptr;
};

struct HostPlan {
    int N = 0;
    int64_t nnzK = 0;
    std::vector<int> perm, iperm;  // perm[k] = original index eliminated k-th

    int nsuper = 0;
    std::vector<int> sn_first;       // [nsuper+1] first (permuted) column
    std::vector<int> sn_of_col;      // [N]
    std::vector<int64_t> sn_rowptr;  // [nsuper+1] into sn_rows
    std::vector<int> sn_rows;        // sorted row structure, own columns first
    std::vector<int64_t> sn_panel;   // [nsuper+1] offset of the r x w column-major panel in Lx
    std::vector<int64_t> sn_diag;    // [nsuper+1] offset of the factored w x w diagonal block in Ldiag
    std::vector<int> sn_parent, sn_level;
    int nlevels = 0;
    std::vector<int> lvl_ptr, lvl_sn;

    std::vector<int64_t> kmap;      // [nnzK] original nz -> Lx offset
    std::vector<int64_t> diag_dst;  // [N] permuted column k -> Lx offset of its diagonal entry

    std::vector<FacItem> fac_items;
    std::vector<int> fac_lvl_ptr;  // [nlevels+1]
    std::vector<int> fac_lvl_maxw; // [nlevels] widest supernode of the level (LDS sizing)

    std::vector<int> rel;
    std::vector<UpdTask> upd_tasks;
    std::vector<UpdGroup> upd_groups;
    std::vector<int> upd_stage_ptr;  // [nlevels+1] groups executed after factor(level)
    std::vector<int16_t> upd_tmap;
    std::vector<int> upd_stage_ndense;  // [nlevels] the first ndense groups of a stage are dense tiles,
    std::vector<int> upd_stage_ngather; // [nlevels] the next ngather groups go through the per-entry gather lists
    int update_batch_used = 4;          // the batch length the schedule was built with
    // per-entry gather lists (groups of kind 2): entries of one stage are contiguous
    std::vector<int64_t> gath_stage_ptr;   // [nlevels+1] into gath_tgt
    std::vector<int64_t> gath_tgt;         // Lx offset of the target entry
    std::vector<int64_t> gath_pptr;        // [nentries+1] into the pair arrays
    std::vector<int64_t> gath_src;         // Lx offset of L_s[i, 0]
    std::vector<int32_t> gath_dj;          // offset of L_s[j, 0] relative to gath_src
    std::vector<int32_t> gath_sn;          // source supernode (stride, width, pivots)
    std::vector<double> upd_stage_flops_dense;   // [nlevels] flops of the dense tiles of a stage
    double flops_update_dense = 0;   // part of flops_update executed by the dense-tile kernel

    std::vector<int64_t> u_off;  // [nsuper+1] offsets of each panel's off-diagonal rows in ubuf
    std::vector<int64_t> lt_off; // [nsuper+1] offsets of the row-major copy of L21 (w x (r-w)) in LT
    std::vector<int64_t> g_ptr;  // [|sn_rows|+1] per panel row slot: update-vector entries of the children landing there
    std::vector<int> g_idx;      // ubuf positions

    std::vector<FrontDesc> fronts;
    std::vector<FrontPanel> front_panels;
    std::vector<int64_t> front_gptr;
    std::vector<int> front_gidx;
    std::vector<int> sn_front;        // [nsuper] front index of a panel handled by the front kernels, else -1
    int front_sync_ints = 0;
    int64_t sbinv_doubles = 0;        // storage of the super-block inverse tiles of all fronts (FrontDesc::sbinv_off)

    std::vector<int64_t> sym_rowptr;  // symmetric CSR view of K (original ordering) WITHOUT the entries of the dense triangles below
    std::vector<int> sym_col;
    std::vector<int64_t> sym_q;       // index into Kval
    // Dense triangles of K (PlanOptions::dense_tri_first_col): runs of >= dense_tri_min_dim consecutive columns c0 .. c0 + d - 1 whose
    // column c0 + j ends with the rows c0 .. c0 + j -- the packed upper triangle a PSD cone's Hs block is stored as
    // (directldl_kkt_assembly.jl:49-57, _csc_colcount_dense_triangle).  The residual SpMV of the refinement takes them from the values
    // alone (k_spmv_dense_tri: no index traffic, coalesced) instead of through 2 x 12 bytes of indices per entry of the view.
    std::vector<DenseTri> dtri;
    std::vector<int64_t> dtri_col;    // per triangle column (DenseTri::col0 + j): index into Kval of its entry in row c0

    // statistics / cost model
    int64_t nnzL = 0;            // strictly-lower structural nonzeros of L (column counts)
    int64_t panel_doubles = 0;   // supernodal storage incl. diagonal blocks and relaxation zeros
    int64_t diag_doubles = 0;
    int64_t ubuf_len = 0;
    int etree_height = 0;
    int ordering_used = 0;       // 0 = minimum degree on K, 1 = cone rows first / variables last, 2 = user, 3 = nested dissection
    double cost_md_seconds = 0, cost_nd_seconds = 0;   // predicted seconds per KKT iteration unit of the two candidates
    int cost_md_levels = 0, cost_nd_levels = 0;
    std::string timing_note;     // milliseconds per phase of build_plan (HIPKKT_VERBOSE)
    double flops_colcount = 0;   // sum_j c_j^2 + 3 c_j
    double flops_update = 0;     // executed flops of the dense update tasks (2*rows*cols*k)
    double flops_exec = 0;       // update + diagonal-block + TRSM flops actually executed
};

// The panels of a front that share an update batch (levels with the same floor(level / batch)) can be factored by ONE launch
// (front_block.hip) when every such level holds nothing but its front panel, all of them are 64 wide and the stages between them
// carry nothing but the batch's own just-in-time updates.  Returns those batches (at least 2 and at most max_nb panels each).
struct FrontBatchHost { int front, p0, nb, level_first, level_last; };
std::vector<FrontBatchHost> front_batches(const HostPlan &P, int update_policy, int max_nb);

// Ap/Ai: upper-triangular CSC pattern (diagonal present), 0-based.
// user_perm: optional (size N) or nullptr.  Returns empty string on success.
std::string build_plan(int N, const int64_t *Ap, const int64_t *Ai, const int64_t *user_perm,
                       const PlanOptions &opt, HostPlan &plan);

void amd_order(int n, const int64_t *Ap, const int64_t *Ai, double dense_scale, std::vector<int> &perm,
               const char *hold = nullptr);
// nested dissection by BFS level structures, minimum degree on leaves of <= leaf_size nodes (ordering.cpp)
void nd_order(int n, const int64_t *Ap, const int64_t *Ai, double dense_scale, int leaf_size, std::vector<int> &perm);

}  // namespace hipkkt
