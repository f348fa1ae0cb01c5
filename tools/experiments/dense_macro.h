// The dense Schur-update MACRO tile (round 6): one WORKGROUP owns 2 x 2 target tiles of 64 x 64 -- row blocks I0, I1 of the target
// panels J0, J1 -- whose contributions come from the same source panels (the far stage of a front batch: every tile is fed by the
// batch's five panels).  A k-step of the four tiles needs four row ranges of the source panel (the rows of I0, I1, J0, J1): each is
// fetched ONCE per workgroup (thread = row, wavefront = one of the step's 4 columns: four 512-byte segments per wavefront), P k-steps
// ahead of its use, laid down in LDS and read back by the two wavefronts that need it.  Against one wavefront per tile with operands
// straight from the panels (dense_tile.h dense_tile_core_full): half the operand traffic out of the L2s (4 instead of 8 row ranges
// per 64 matrix-core instructions), and the memory latency is covered by a ring of P staged k-steps per THREAD (5 values each)
// instead of two sets of 9 operands per wavefront.  Every wavefront still owns one tile in 16 accumulators and issues the same
// products in the same order as the single-tile code: bit-identical results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dense_tile.h"   // (clarabel.jl_amd/csrc)

namespace hipkkt {

// one source panel's contribution to a macro tile: everything in one 48-byte record
struct MacroTask {
    int64_t panel_off;             // source panel in Lx
    int32_t r8, K, dfirst;         // source row count * 8 (byte stride of a column), width (multiple of 4), first pivot
    int32_t row_off[4];            // first source row of the target's row blocks I0, I1 and of the target panels' column blocks J0, J1
    int32_t pad[3];
};
static_assert(sizeof(MacroTask) == 48, "MacroTask is copied to LDS word by word");
// tile w = 2 wi + wj = (row block I_wi, target panel J_wj)
struct MacroGroup {
    int64_t tile_off[4];           // Lx offset of each tile's first row in its target panel
    int32_t rt[2];                 // rows (column stride) of the target panels J0, J1
    int32_t task_begin, task_end;  // into the MacroTask records
    int32_t nsteps, pad;           // sum of K / 4 over the tasks
};
struct DevPlanLite { double *Lx; const double *D; };   // (tools/ubench_macro.hip)

constexpr int kMacroMaxTasks = 32;  // source panels per macro tile (their records are staged in LDS)
constexpr int kMacroKS = 80;       // LDS stride (doubles) between the 4 columns of a staged k-step: 64 rows + 16 (the four 16-row
                                   // segments a half-wave reads for one operand then sit in different banks)
// LDS-only workgroup barrier (__syncthreads() would also wait for the global loads of the prefetch ring)
__device__ __forceinline__ void macro_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int P>
__device__ __forceinline__ void dense_macro_tile(double *Lx, const double *D, const MacroGroup *Gp, const MacroTask *mt) {
    static_assert(P >= 2 && (P & 1) == 0, "the LDS buffer of a ring slot is static: P even");
    __shared__ double sbuf[2][4][4 * kMacroKS];
    __shared__ MacroTask stask[kMacroMaxTasks];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lk = lane >> 4;
    const int wave = rfl(tid >> 6), wi = wave >> 1, wj = wave & 1;
    const MacroGroup G = *Gp;
    const int task_end = rfl(G.task_end), S = rfl(G.nsteps);
    double *tp = rfl_ptr(Lx + (wave == 0 ? G.tile_off[0] : wave == 1 ? G.tile_off[1] : wave == 2 ? G.tile_off[2] : G.tile_off[3]));
    const int rt = rfl(wj ? G.rt[1] : G.rt[0]);

    // ---- the fetch side runs P k-steps ahead of the matrix-core side and crosses task boundaries on its own.  Everything that
    //      changes from step to step is wave-uniform and lives in scalar registers: the base of a row range is panel + first row +
    //      (k0 + wave) * rows, the lane offset (lane * 8) never changes.  The task records wait in LDS (a task switch must not touch
    //      the vector-memory counter: a record fetched from global memory in the middle of the stream made every switch -- and, through
    //      the register copies at the join of the branch, every STEP -- wait for the whole prefetch ring).
    const int ntask = task_end - rfl(G.task_begin);
    for (int i = tid; i < ntask * (int)(sizeof(MacroTask) / 4); i += 256) ((int *)stask)[i] = ((const int *)(mt + rfl(G.task_begin)))[i];
    macro_bar();
    int lq = 0, k0l = 0, Kl = 0;
    unsigned stepb = 0;                                             // bytes between two k-steps of the current source panel
    const char *rb0, *rb1, *rb2, *rb3, *db;                         // uniform bases of the step the fetch side is at
    auto setup = [&](int q) {
        const MacroTask T = stask[q];
        const char *sp = (const char *)rfl_ptr(Lx + T.panel_off);
        const unsigned r8 = (unsigned)rfl(T.r8);
        Kl = rfl(T.K);
        stepb = 4u * r8;
        const char *w0 = sp + (size_t)wave * r8;                    // this wavefront fetches column k0 + wave of all four row ranges
        rb0 = w0 + (size_t)rfl(T.row_off[0]) * 8; rb1 = w0 + (size_t)rfl(T.row_off[1]) * 8;
        rb2 = w0 + (size_t)rfl(T.row_off[2]) * 8; rb3 = w0 + (size_t)rfl(T.row_off[3]) * 8;
        db = (const char *)rfl_ptr(D + T.dfirst);
        k0l = 0;
    };
    setup(0);
    const unsigned lane8 = 8u * (unsigned)lane, lk8 = 8u * (unsigned)lk;
    struct Stage { double v[4], d; };
    Stage g[P];
    auto issue = [&](Stage &st) {
        st.v[0] = ld_off((const double *)rb0, lane8);
        st.v[1] = ld_off((const double *)rb1, lane8);
        st.v[2] = ld_off((const double *)rb2, lane8);
        st.v[3] = ld_off((const double *)rb3, lane8);
        st.d = ld_off((const double *)db, lk8);                     // the pivots of this lane's k (operand layout: k = lane >> 4)
        // next step (scalar side only); past the last step the fetch side stays where it is (those loads are never used)
        if (k0l + 4 < Kl) {
            k0l += 4; rb0 += stepb; rb1 += stepb; rb2 += stepb; rb3 += stepb; db += 32;
        } else if (lq + 1 < ntask) {
            lq++;
            setup(lq);
        }
    };
#pragma unroll
    for (int u = 0; u < P; u++) issue(g[u]);
    // the accumulators are requested AFTER the first operands (loads return in order: the first k-step only waits for its own row
    // ranges and for the first quarter of the tile, not for all 32 KB of it)
    v4f64 acc[4][4];               // acc[tj][ti][reg]: column tj * 16 + lk + 4 reg, row ti * 16 + l15 (dense_tile.h)
#pragma unroll
    for (int tj = 0; tj < 4; tj++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) acc[tj][ti][reg] = ld_off(tp, (unsigned)(ti * 16 + l15 + (tj * 16 + lk + 4 * reg) * rt) * 8u);
        __builtin_amdgcn_sched_barrier(0);
    }

    // Three stages behind the fetch: ring slot -> LDS (one k-step ahead of its use) -> operand registers (read back right after the
    // barrier, while the matrix core works on the step before) -> matrix core.  One barrier per k-step; a slot's LDS buffer is
    // static (P even), and nobody can still be reading the buffer that is written: its reads were waited for at the barrier before.
    double ra[2][4], rb[2][4], rnd[2];
    auto to_lds = [&](int u) {
        double *B = &sbuf[u & 1][0][0];
#pragma unroll
        for (int q = 0; q < 4; q++) B[q * 4 * kMacroKS + wave * kMacroKS + lane] = g[u].v[q];
    };
    auto from_lds = [&](int u) {                                    // raw values: nothing here waits for the reads
        const double *B = &sbuf[u & 1][0][0];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            ra[u & 1][t] = B[(2 + wj) * 4 * kMacroKS + lk * kMacroKS + 16 * t + l15];
            rb[u & 1][t] = B[wi * 4 * kMacroKS + lk * kMacroKS + 16 * t + l15];
        }
    };
    {
        to_lds(0);
        rnd[0] = -g[0].d;
        issue(g[0]);
        macro_bar();
        from_lds(0);
    }
    for (int s = 0; s < S; s += P) {
#pragma unroll
        for (int u = 0; u < P; u++) {
            if (s + u < S) {                                        // workgroup-uniform
                const int un = (u + 1) % P;                         // ring slot of the next step
                const bool more = s + u + 1 < S;
                double a[4];
#pragma unroll
                for (int t = 0; t < 4; t++) a[t] = ra[u & 1][t] * rnd[u & 1];
                // the next step's values go to LDS BEFORE this step's matrix-core work, the barrier sits in the MIDDLE of it (the
                // first eight instructions are still in the pipe while the wavefront waits for its LDS stores and for the others),
                // the read-back of the next step's operands follows, then the second eight
                if (more) {
                    to_lds(un);
                    rnd[un & 1] = -g[un].d;
                    issue(g[un]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tj = 0; tj < 2; tj++)
#pragma unroll
                    for (int ti = 0; ti < 4; ti++) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], rb[u & 1][ti], acc[tj][ti], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                macro_bar();
                if (more) from_lds(un);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tj = 2; tj < 4; tj++)
#pragma unroll
                    for (int ti = 0; ti < 4; ti++) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], rb[u & 1][ti], acc[tj][ti], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) st_off(tp, (unsigned)(ti * 16 + l15 + (tj * 16 + lk + 4 * reg) * rt) * 8u, acc[tj][ti][reg]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace hipkkt
