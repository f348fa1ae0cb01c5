// Internal declarations shared by the translation units of libclarabel_hipkkt.so's host side:
//   hipkkt_abi.cpp     C entry points: creation / getters / value updates / residuals / scaling / diagnostics
//   hipkkt_setup.cpp   runtime objects, device residency of a plan, handle creation (finish_create)
//   hipkkt_factor.cpp  the factorisation's launch sequence, its graph, hipkkt_refactor and the robust-order twin
//   hipkkt_solve.cpp   LDL solves, device-side iterative refinement, the solve entry points
// Host orchestration only; all numeric work is in kernels.hip / front_block.hip / assemble_dev.hip / scaling.hip.
#pragma once
#include "../../include/hipkkt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "assemble.h"
#include "device_plan.h"
#include "kernels.h"
#include "runtime_pool.h"
#include "symbolic.h"

namespace hipkkt { int box_probe(int device, double *out); }   // probe.hip
namespace hipkkt_host {
using namespace hipkkt;

extern thread_local std::string g_create_error;

struct DeviceError {
    std::string msg;
};

#define HK_CHECK(expr)                                                                            \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            throw DeviceError{std::string(#expr) + ": " + hipGetErrorString(e_)};                 \
    } while (0)

// Synchronous copy / fill on a handle's OWN stream.  The legacy (NULL) stream is never used: an operation on it from
// one host thread fails while another thread captures a hipGraph ("would make the legacy stream depend on a capturing
// stream"), and handles are meant to be driven concurrently from several threads.
inline void copy_sync(hipStream_t st, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (!bytes) return;
    HK_CHECK(hipMemcpyAsync(dst, src, bytes, kind, st));
    HK_CHECK(hipStreamSynchronize(st));
}
inline void fill_async(hipStream_t st, void *dst, int value, size_t bytes) {
    if (bytes) HK_CHECK(hipMemsetAsync(dst, value, bytes, st));
}

// Test / experiment switches (hipkkt_debug_set, include/hipkkt.h).  The library reads NO environment variable for them: a production
// build (libclarabel_hipkkt.so, what the Julia glue and bench.py load) keeps these defaults for ever -- hipkkt_debug_set refuses -- and
// does not even contain the code some of them select (the first form of the front-batch kernel, the debug flags of the sweeps).
// The -DHIPKKT_TESTING build (libclarabel_hipkkt_testing.so, what tests/ load) accepts them; a handle reads them when it is created.
// (Until round 5 every one of them was an environment read inside the product: a HIPKKT_* variable left over in a caller's environment
// silently changed the arithmetic order, or -- DEBUG_FLAGS -- the results.)
struct DebugOpts {
    bool plan_cache = true;        // PLAN_CACHE=0: no process-wide plan cache
    bool fb_extra = true;          // FB_EXTRA=0: every far stage applies all of its tiles in its own launch
    bool fb_stream = true;         // FB_STREAM=0: the round-3 pivot chain (whole-tile hand-off; first form of the kernel only)
    bool fb_v2 = true;             // FB_V2=0: the first form of the front-batch kernel (front_block.hip)
    bool force_twin = false;       // FORCE_TWIN=1: every successful factorisation in the cheap order counts as broken down
    bool no_graph = false;         // NO_GRAPH=1: eager launches
    bool no_persist = false;       // NO_PERSIST=1: per-level solve kernels only
    bool full_tiles = true;        // FULL_TILES=0: the general core for every dense tile
    bool front_block = true;       // FRONT_BLOCK=0: one launch per panel
    bool split_k = true;           // SPLIT_K=0
    bool gather_overlap = true;    // GATHER_OVERLAP=0: big gather stages run whole, in line (no side stream)
    bool gather_sort = true;       // GATHER_SORT=0: the entries of a gather stage stay in target order
    int gather_side_blocks = 512;  // GATHER_SIDE_BLOCKS=<n>: grid bound of the side-stream gather (2 workgroups per compute unit; 0 = unbounded)
    bool dense_tri = true;         // DENSE_TRI=0: dense Hs triangles stay in the symmetric view
    bool ordering_amd = false;     // ORDERING=amd: minimum degree on K only
    bool no_front = false;         // NO_FRONT=1: no persistent front kernels
    bool host_assembly = false;    // HOST_ASSEMBLY=1: the host twin of the assembly kernels
    int front_block_min_rows = -1; // FRONT_BLOCK_MIN_ROWS=<n> (-1: the plan's default)
    int superhop = -1;             // SUPERHOP=<n> (-1: the plan's default)
    int debug_flags = 0;           // DEBUG_FLAGS=<bits>: timing experiments, results are WRONG when set
    long long spin_limit = -1;     // SPIN_LIMIT=<n> (-1: 2^20)
    long long persist_retry = -1;  // PERSIST_RETRY=<n> (-1: 64)
    double accurate = 64.0;        // ACCURATE=<tau>: 0 = never, negative = every wide block
    int fb_extra_pw = 1;           // FB_EXTRA_PW=<tiles per wavefront>,<penalty 1>,<penalty 2>
    double fb_pen1 = 10.0, fb_pen2 = 28.0;
};
DebugOpts &debug_opts();
bool verbose();                    // HIPKKT_VERBOSE in the environment: diagnostic lines on stderr (changes no result)

struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    bool valid = false;
    // parameters baked into the captured kernel arguments
    int static_enable = -1;
    double eps_const = 0, eps_prop = 0;
};

}  // namespace hipkkt_host

using namespace hipkkt;        // internal header of one library: the host files all work inside these two namespaces
using namespace hipkkt_host;

// One solve context = everything a refined KKT solve mutates: its stream, work vectors, the hand-off / counter words
// of the persistent sweeps (a private copy of those DevPlan pointers), the device-side refinement state and its
// captured graphs.  Two contexts let two right-hand sides be solved CONCURRENTLY on one factorisation (SURVEY
// section 8(f) row N2: the constant-rhs solve and the affine solve of an IPM iteration): the sweeps are bound by
// dependency latency, not by throughput, so two of them overlap almost perfectly.
constexpr int kNumCtx = 2;
struct SolveCtx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DevPlan dp{};                 // S->dp with this context's ubuf / pbuf / slots / sync words / scal / flags
    double *d_b = nullptr, *d_x0 = nullptr, *d_x1 = nullptr, *d_e = nullptr, *d_corr = nullptr;
    double *d_y = nullptr, *d_z = nullptr, *d_xp = nullptr;
    RefineState *d_rs = nullptr, *h_rs = nullptr;   // device state, pinned host copy
    int *h_flags = nullptr;                         // pinned copy of dp.flags
    hipkkt_host::GraphSlot g_ldl, g_first, g_step;               // plain LDL solve (d_b -> d_x0) / refined solve incl. its first step / one more step
    double g_reltol = -1, g_abstol = -1, g_stop = -1;   // parameters baked into g_first / g_step
    int64_t g_maxit = -1;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    double last_ms = 0;
    int64_t last_steps = 0;
    bool ir_used = false;
    bool extra_steps = false;     // the last solve_finish ran refinement steps beyond the one inside the solve's graph
    const double *result() const { return (ir_used && h_rs->cur) ? d_x1 : d_x0; }
};

struct hipkkt_solver {
    int device = 0;
    SolveCtx ctx[kNumCtx];
    hipStream_t stream = nullptr;
    hipkkt_opts opts{};
    bool l1 = false;
    KKTImage img;      // L1: assembled image; L0: colptr/rowval/nzval/dsigns copied in
    HostPlan plan;
    DevPlan dp{};
    std::vector<std::pair<void *, size_t>> allocs;   // device slabs (RuntimePool blocks: pointer, capacity)
    std::string err;

    int N = 0;
    int64_t nnzK = 0;
    // solve item lists (256-row blocks) per level
    std::vector<FacItem> slv_items, bwd_items;
    std::vector<int> slv_lvl_ptr, bwd_lvl_ptr;
    std::vector<int> reg_lvl_sn, reg_lvl_ptr;   // supernodes of every level that are NOT front panels
    std::vector<int> lvl_wnarrow, all_lvl_wnarrow;   // [nlevels] widest narrow supernode of the level
    std::vector<int> lvl_nnarrow;                // [nlevels] the first lvl_nnarrow[l] supernodes of a level's list are narrow (one thread each)
    // the same level lists over ALL supernodes (front panels included), appended to the same item arrays: the path without
    // any persistent kernel, taken after a sweep time-out
    std::vector<int> all_slv_lvl_ptr, all_bwd_lvl_ptr, all_reg_lvl_ptr;
    std::vector<int> all_lvl_nnarrow;
    // persistent sweeps over the regular supernodes: segments = level ranges between front kernels
    bool use_persist = true;
    double t_init_runtime = 0;           // seconds (HIPKKT_VERBOSE)
    std::chrono::steady_clock::time_point t_created{};   // when the assembly returned
    // front batches factored by one launch each (front_block.hip); a time-out inside one downgrades the handle to one launch per panel
    bool use_front_block = true;
    std::vector<FrontBatch> fbatches;
    std::vector<int> lvl_fb;             // [nlevels] index of the batch that starts at this level, -2 inside a batch, -1 otherwise
    std::vector<int> fb_last_level;      // per batch
    int *d_fb_sync = nullptr;
    double *d_fb_scratch = nullptr;
    // streamed pivot chain of k_front_block (front_block.hip): per (batch, panel, 8-pivot block) one record of 528 doubles that the
    // diagonal workgroup publishes as it eliminates and the next diagonal workgroup polls (sentinel-filled before every factorisation)
    double *d_fb_stream = nullptr;
    hipStream_t idle_stream = nullptr;   // never carries work: see init_runtime (hipkkt_setup.cpp)
    int64_t fb_stream_doubles = 0;
    bool fb_streamed = true;             // HIPKKT_FB_STREAM=0: the round-3 chain (hand-off of the explicit inverse after all 64 pivots)
    // per front batch: what the next batch of the same front needs from this batch's far stage ([columns of the next batch | rest]
    // order of the stage's dense tiles, hipkkt_setup.cpp order_far_stages) -- the rest may ride in the next k_front_block launch
    struct NextBatch {
        int next_blk = 0;                               // workgroups of the next batch's panel kernel
        bool has_next = false;                          // the next batch of the same front follows at once: [chain | rest] order valid
        int ncrit = 0;                                  // far stage: the first ncrit dense groups are the next batch's columns
    };
    std::vector<NextBatch> next_batch;
    int64_t split_scratch_doubles = 0;   // split-K partial tiles behind the panels in Lx (cleared with them before every factorisation)
    // refined block solves (kernels.hip k_invert_diag_wide): wide diagonal blocks whose explicit inverse has an entry above the threshold
    double accurate_threshold = 64.0;    // HIPKKT_ACCURATE=<threshold> ("0" = never, "-1" = every wide block)
    int last_npolish = 0;                // marked blocks of the last factorisation
    int64_t n_accurate_factorisations = 0;   // factorisations with at least one
    bool fb_v2 = true;                   // front batches by front_block2.hip (tiles transposed in the accumulators, round 5); HIPKKT_FB_V2=0: front_block.hip
    bool force_twin = false;             // HIPKKT_FORCE_TWIN=1 at create (tests): see hipkkt_refactor
    bool fb_extra = true;                // the partial last round of a batch's far updates rides in the next k_front_block launch (HIPKKT_FB_EXTRA=0: off)
    long long *d_fb_trace = nullptr;     // HIPKKT_FB_TRACE=1: wall-clock stamps of the first 8 workgroups of every batch (debug_dump 9)
    bool persist_allowed = true;         // false: HIPKKT_NO_PERSIST (never tried)
    int64_t persist_retry_at = -1;       // after a sweep time-out: the LDL-solve count at which the persistent kernels are tried again
    int64_t persist_backoff = 0;         // doubles with every time-out (64, 128, ...)
    int64_t n_sweep_timeouts = 0;
    int64_t n_twin_refactors = 0;        // factorisations repeated on the robust-order twin
    int nseg = 0;
    std::vector<int> seg_of_level;               // [nlevels]
    std::vector<int> fseg_ptr, bseg_ptr;         // [nseg+1] into slv_items / pbwd_items
    std::vector<int> seg_lo, seg_hi, seg_lstar;  // per segment: level range [lo, hi] and the first level handled by the
                                                 // persistent kernels (wide bottom levels keep one launch per level)
    std::vector<FacItem> pbwd_items;
    int wmax_all = 1;
    int inv_nsmall = 0, inv_wsmall = 1, inv_nwide = 0;   // split of the diagonal-block inversions (kernels.hip)
    std::vector<int64_t> p_off;
    // split-K of stages with few target tiles and long contribution lists (hipkkt_setup.cpp plan_split_k): per level, the range of the
    // sub-groups appended to the dense-group records and of the reduce records
    std::vector<int> split_group_begin, split_group_count, split_rec_ptr;
    SplitRec *d_split_recs = nullptr;
    std::vector<int64_t> gath_heavy_ptr;   // [nlevels+1] into the list of heavy gather entries (kernels.hip k_update_gather_heavy)
    // Deferred part of a big gather stage (hipkkt_setup.cpp split_gather_stages): the entries of stage l are ordered [targets the next
    // update batch factors or updates | targets beyond it]; the second part runs on the side stream NEXT TO the levels of the next
    // batch and is joined before that batch's far stage (hipkkt_factor.cpp enqueue_gather / join_side).  gath_split[l] = entries of
    // the first part (-1: the stage is not split), gath_heavy_split[l] = how many of the stage's heavy entries belong to it.
    std::vector<int64_t> gath_split, gath_heavy_split;
    int side_join_level = -1;              // >= 0 while a side-stream gather is in flight: the first level whose update stage must wait for it
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    // device index arrays for value updates
    int64_t *d_mapHs = nullptr, *d_mapP = nullptr, *d_mapA = nullptr, *d_diag_full = nullptr;
    // all sparse SOC cones concatenated
    int64_t soc_total = 0;
    int nsoc = 0;
    std::vector<int64_t> soc_off;  // per sparse map (SOC only) offset into the concatenated arrays
    std::vector<int> soc_of_sparse; // sparse-map index -> soc ordinal or -1
    int64_t *d_soc_uidx = nullptr, *d_soc_vidx = nullptr, *d_soc_didx = nullptr;
    int *d_soc_cone = nullptr;
    double *d_soc_u = nullptr, *d_soc_v = nullptr, *d_soc_eta2 = nullptr;
    // N1: update_scaling! / get_Hs! on the device (hipkkt_set_cone_types + hipkkt_update_scaling, scaling.hip)
    std::vector<int64_t> cone_numel;                       // as given to hipkkt_create_from_parts
    std::vector<int32_t> cone_hs_dense, cone_sparse_kind;
    bool sc_ready = false;
    int sc_nsoc = 0;                                       // ALL second-order cones (sparse and dense), cone order
    std::vector<int64_t> sc_psd_hs, sc_psd_n;              // per PSD cone: first Hs entry, matrix dimension
    int64_t sc_psd_total = 0;                              // sum of n * n
    int64_t sc_cap_socdesc = 0, sc_cap_psd = 0;            // capacities of the d_sc_* buffers (allocated once, re-used by later calls)
    signed char *d_sc_kind = nullptr;
    int64_t *d_sc_rowhs = nullptr, *d_sc_socdesc = nullptr;
    double *d_sc_sz = nullptr, *d_sc_wl = nullptr, *d_sc_eta = nullptr, *d_sc_R = nullptr, *d_sc_W = nullptr;
    int *d_sc_fail = nullptr;

    // vectors
    double *d_b = nullptr, *d_x = nullptr, *d_dx = nullptr, *d_e = nullptr;
    double *d_sin = nullptr, *d_sout = nullptr, *d_y = nullptr, *d_z = nullptr, *d_xp = nullptr;
    double *d_qb = nullptr, *d_res_in = nullptr, *d_res_out = nullptr, *d_res_part = nullptr;   // residuals_update! on the device (N4)
    double *d_red = nullptr, *d_red_part = nullptr;   // reduced-system algebra of kkt_solve! (N2): [x1;z1] | [x2;z2] | step | variables.x | scalars
    double *h_scal_red = nullptr;                     // pinned read-back of its ten scalars
    bool red_have_const = false;                      // (x2, z2) of the current factorisation is resident
    double *d_stage = nullptr;   // staging for host-supplied values
    int64_t *d_stage_idx = nullptr;
    int64_t stage_cap = 0;
    double *h_scal = nullptr;    // pinned read-back area
    int *h_flags = nullptr;

    hipkkt_host::GraphSlot g_factor;
    bool use_graph = true;
    bool runtime_ready = false;   // init_runtime done (streams, events, pinned areas)
    PlanOptions plan_opts;       // as used for the current plan
    // robust-order twin (minimum degree on K), created on the first factorisation that fails in the
    // "variables last" order; every later factorisation still tries the fast order first
    hipkkt_solver *fallback = nullptr;
    // the twin's symbolic analysis runs on a host thread from the moment the cheap order is chosen (finish_create)
    bool device_ready = false;      // (a twin:) init_runtime + setup_device already ran on the speculation thread
    std::unique_ptr<hipkkt_solver> twin_pending;
    std::future<std::string> twin_future;
    std::shared_ptr<std::atomic<bool>> twin_cancel;   // set when the twin turns out not to be needed
    std::shared_ptr<std::atomic<int>> twin_go;        // 0: the owner has not chosen its order yet, 1: cheap order chosen (the twin may take device memory), 2: not needed
    bool using_fallback = false;
    bool profiling = false;
    bool profiling_no_extra = false;     // hipkkt_set_profiling(h, 2): the profiled refactorisations keep every far tile in its stage's own launch
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    bool kval_event_pending = false;     // ev3 marks an asynchronous value update of the resident K on the main stream (hipkkt_set_hs_dev)
    double t_last_factor = 0, t_last_solve = 0, t_acc_factor = 0, t_acc_solve = 0, t_last_update = 0;
    int64_t n_factor = 0, n_solvecalls = 0, n_ldlsolves = 0, n_rhs_solved = 0;
    double last_eps = 0;
    double prof_fb_flops = 0;
    double prof_fb_ms = 0;               // last profiled refactorisation: the k_front_block launches
    double prof_extra_tiles = 0, prof_extra_flops = 0;   // ... dense update tiles / flops that rode in them (HIPKKT_FB_EXTRA)
    int prof_fb_launches = 0, prof_fb_panels = 0;
    double prof_dense4_ms = 0, prof_dense4_flops = 0;   // last profiled refactorisation: k_update_dense<4,4> alone
    int prof_dense4_launches = 0;
    std::vector<double> prof_launch_ms, prof_launch_flops, prof_launch_tiles;   // per k_update_dense<4,4> launch of that refactorisation
    int64_t last_nreg = 0;

    // Device memory comes from a few slabs (bump allocation, 256-byte aligned) instead of one hipMalloc per array:
    // a handle owns ~80 arrays, and on the small problems of a batch the ~160 hipMalloc / hipFree calls were a
    // third of the set-up + tear-down time.
    char *slab_cur = nullptr;
    size_t slab_left = 0;
    template <class T>
    T *dalloc(size_t n) {
        if (n == 0) n = 1;
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (bytes > slab_left) {
            size_t slab = std::max<size_t>(bytes, (size_t)8 << 20);
            void *p = RuntimePool::get().dev_alloc(device, slab, &slab);
            if (!p) throw std::bad_alloc();
            allocs.push_back({p, slab});
            slab_cur = (char *)p;
            slab_left = slab;
        }
        void *p = slab_cur;
        slab_cur += bytes;
        slab_left -= bytes;
        return (T *)p;
    }
    template <class T>
    T *upload(const std::vector<T> &v) {
        T *p = dalloc<T>(v.size());
        if (!v.empty()) hipkkt_host::copy_sync(stream, p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return p;
    }
    void ensure_stage(int64_t n) {
        if (n <= stage_cap) return;
        int64_t cap = std::max<int64_t>(n, 2 * stage_cap);
        d_stage = dalloc<double>(cap);
        d_stage_idx = dalloc<int64_t>(cap);
        stage_cap = cap;
    }
    ~hipkkt_solver() {
        if (twin_cancel) twin_cancel->store(true);
        if (twin_go && twin_go->load() == 0) twin_go->store(2);
        if (twin_future.valid()) twin_future.wait();     // the thread reads twin_pending's image (a cancelled one ends at its next phase)
        twin_pending.reset();
        delete fallback;
        (void)hipSetDevice(device);
        // everything below goes back to the process-wide cache (runtime_pool.h): the streams must be idle first
        RuntimePool &rp = RuntimePool::get();
        if (stream) (void)hipStreamSynchronize(stream);
        for (SolveCtx &C : ctx)
            if (C.own_stream && C.stream) (void)hipStreamSynchronize(C.stream);
        if (g_factor.exec) (void)hipGraphExecDestroy(g_factor.exec);
        for (SolveCtx &C : ctx) {
            for (hipkkt_host::GraphSlot *g : {&C.g_ldl, &C.g_first, &C.g_step})
                if (g->exec) (void)hipGraphExecDestroy(g->exec);
            rp.pinned_free(device, C.h_rs);
            rp.pinned_free(device, C.h_flags);
            for (hipEvent_t e : {C.ev_a, C.ev_b}) rp.event_put(device, e);
            if (C.own_stream) rp.stream_put(device, 2, C.stream);
        }
        for (auto &a : allocs) rp.dev_free(device, a.first, a.second);
        rp.pinned_free(device, h_scal);
        rp.pinned_free(device, h_flags);
        for (hipEvent_t e : {ev0, ev1, ev2, ev3, ev_fork, ev_join}) rp.event_put(device, e);
        rp.stream_put(device, 0, stream);
        rp.stream_put(device, 1, idle_stream);
    }
};

namespace hipkkt_host {

// seg_sync = [forward tickets | backward tickets] padded to whole 128-byte lines, then fdone [kSegSub = 8 arrays of nsuper],
// then the error word (kernels.hip seg_sync())
inline size_t seg_sync_ints(int nseg, int nsuper) { return (size_t)((2 * nseg + 31) & ~31) + 8 * (size_t)nsuper + 16; }

// hipkkt_setup.cpp
void plan_cache_counts(int64_t *hits, int64_t *misses);   // process-wide cache of symbolic plans (same pattern + options)
void init_runtime(hipkkt_solver *S);
void setup_device(hipkkt_solver *S);
int32_t finish_create(hipkkt_solver *S, const hipkkt_opts *opts, hipkkt_handle *out);
// hipkkt_factor.cpp
void enqueue_factor(hipkkt_solver *S, int static_enable, double eps_const, double eps_prop);
// hipkkt_solve.cpp
void enqueue_ldl_solve(hipkkt_solver *S, SolveCtx &C, const double *in, double *out, int *zero = nullptr, int nzero = 0);
int32_t solve_many(hipkkt_solver *S, int nrhs, int ir_enable, double reltol, double abstol, int64_t max_iter, double stop_ratio,
                   int64_t *ir_steps, double *const *out_dev, int nm);
hipkkt_solver *solve_target(hipkkt_solver *S);
void account_fallback_solve(hipkkt_solver *S, hipkkt_solver *T);

template <class F>
void run_graphed(hipkkt_solver *S, hipStream_t stream, GraphSlot &slot, bool reusable, F &&enqueue) {
    if (!S->use_graph) { enqueue(); return; }
    if (!(slot.valid && reusable)) {
        if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; slot.valid = false; }
        hipGraph_t graph = nullptr;
        HK_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
        try {
            enqueue();
        } catch (...) {
            (void)hipStreamEndCapture(stream, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        HK_CHECK(hipStreamEndCapture(stream, &graph));
        hipError_t e = hipGraphInstantiate(&slot.exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { slot.exec = nullptr; S->use_graph = false; enqueue(); return; }
        slot.valid = true;
    }
    HK_CHECK(hipGraphLaunch(slot.exec, stream));
}

inline double slot_value(const hipkkt_solver *S, int slot) {
    double v;
    memcpy(&v, (const char *)S->h_scal + slot * sizeof(double), sizeof(double));
    return v;  // the slot holds the raw bit pattern of a non-negative double (or NaN)
}

inline void read_scalars(hipkkt_solver *S) {
    HK_CHECK(hipMemcpyAsync(S->h_flags, S->dp.flags, FL_COUNT * sizeof(int), hipMemcpyDeviceToHost, S->stream));
    HK_CHECK(hipMemcpyAsync(S->h_scal, S->dp.scal, SC_COUNT * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    HK_CHECK(hipStreamSynchronize(S->stream));
}

}  // namespace hipkkt_host

#define HK_ENTER(h)                                                      \
    if (!(h)) return HIPKKT_ERR_ARGUMENT;                                \
    hipkkt_solver *S = (h);                                              \
    try {                                                                \
        if (hipSetDevice(S->device) != hipSuccess) { S->err = "hipSetDevice failed"; return HIPKKT_ERR_DEVICE; }

#define HK_LEAVE                                                         \
    }                                                                    \
    catch (const DeviceError &e) { S->err = e.msg; return HIPKKT_ERR_DEVICE; } \
    catch (const std::bad_alloc &) { S->err = "out of memory"; return HIPKKT_ERR_ALLOC; } \
    catch (...) { S->err = "internal error"; return HIPKKT_ERR_INTERNAL; }
