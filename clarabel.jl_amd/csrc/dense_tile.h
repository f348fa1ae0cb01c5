// The dense Schur-update tile (k_update_dense, kernels.hip): one wavefront keeps NT x NR blocks of 16 x 16 of a target tile in FP64
// matrix-core accumulators while it sweeps all contributing source panels.  In a header because two translation units run it: the
// update kernels (kernels.hip) and the extra workgroups of a k_front_block launch (front_block.hip), which apply tiles of the previous
// update stage on compute units the panel kernel leaves idle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_plan.h"

namespace hipkkt {

#ifndef HIPKKT_V4F64_DEFINED
#define HIPKKT_V4F64_DEFINED
typedef double v4f64 __attribute__((ext_vector_type(4)));
#endif
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <class T>
__device__ __forceinline__ T *rfl_ptr(T *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
#define HK_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ double ld_off(const double *base, unsigned byte_off) {
    // uniform base + 32-bit lane offset, explicitly in the global address space (global_load ... saddr)
    return *(const HK_GLOBAL double *)((const HK_GLOBAL char *)base + byte_off);
}
__device__ __forceinline__ void st_off(double *base, unsigned byte_off, double v) {
    *(HK_GLOBAL double *)((HK_GLOBAL char *)base + byte_off) = v;
}

// NT 16-column strips x NR 16-row blocks of a tile per wavefront
template <int NT, int NR>
struct DenseRaw {
    double a[NT], b[NR], d;
};

// issue the loads of one k-step (4 k's): operand rows are contiguous for a fixed k
template <int NT, int NR>
__device__ __forceinline__ void dense_load(DenseRaw<NT, NR> &f, const double *sp, const double *dv, const unsigned (&coff)[NT],
                                           const unsigned (&roff)[NR], unsigned r8, int K, int k0, int lk) {
    int kk = k0 + lk;
    kk = kk < K ? kk : K - 1;
    const unsigned ko = (unsigned)kk * r8;
#ifdef HIPKKT_EXPERIMENT_NOLOAD
    f.d = 1.0 + ko * 1e-9;
#pragma unroll
    for (int t = 0; t < NT; t++) f.a[t] = 1e-3 * (coff[t] + 1);
#pragma unroll
    for (int t = 0; t < NR; t++) f.b[t] = 1e-3 * (roff[t] + 1);
#else
    f.d = ld_off(dv, (unsigned)kk * 8u);
#pragma unroll
    for (int t = 0; t < NT; t++) f.a[t] = ld_off(sp, coff[t] + ko);
#pragma unroll
    for (int t = 0; t < NR; t++) f.b[t] = ld_off(sp, roff[t] + ko);
#endif
}

template <int NT, int NR>
__device__ __forceinline__ void dense_mma(const DenseRaw<NT, NR> &f, v4f64 (&acc)[NT][NR], unsigned mbits, int K, int k0, int lk) {
    const double dk = (k0 + lk < K) ? -f.d : 0.0;     // negated: acc = C - sum
    double a[NT], b[NR];
#pragma unroll
    for (int t = 0; t < NT; t++)    // mbits: bit t = column operand t valid, bit 4+t = row operand t valid
        a[t] = ((mbits >> t) & 1u) ? f.a[t] * dk : 0.0;
#pragma unroll
    for (int t = 0; t < NR; t++) b[t] = ((mbits >> (4 + t)) & 1u) ? f.b[t] : 0.0;
#pragma unroll
    for (int tj = 0; tj < NT; tj++)
#pragma unroll
        for (int ti = 0; ti < NR; ti++)
            acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], b[ti], acc[tj][ti], 0, 0, 0);
}

// NT = 4: one wavefront per tile (4 tiles per workgroup): highest operand reuse, for launches with
//         thousands of tiles.   NT = 1: one wavefront per 16-column strip (one tile per workgroup):
//         4x shorter critical path, for the just-in-time updates of the next panel (<= ~100 tiles); these may
//         also split the tile's rows over 4/NR workgroups (NR 16-row blocks per wavefront).
// DEEP: four k-steps of operands in flight also for NT > 1 (72 instead of 36 operand registers) -- for a wavefront that is ALONE on its
//       SIMD (the extra workgroups of a k_front_block launch: 512 registers available, no second wavefront to cover an L2 round trip)
template <int NT, int NR, bool DEEP = false>
__device__ __forceinline__ void dense_tile_core(const DevPlan &P, double *tp, int rt, int nrt, int wt, int task_begin,
                                                int task_end, int lane, int tj0, int ti0) {
    const int l15 = lane & 15, lk = lane >> 4;

    // accumulators <- the target tile.  acc[tj][ti][reg]: column (tj0+tj)*16 + lk + 4*reg, row (ti0+ti)*16 + l15
    v4f64 acc[NT][NR];
#pragma unroll
    for (int tj = 0; tj < NT; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int jj = (tj0 + tj) * 16 + lk + 4 * reg;
#pragma unroll
            for (int ti = 0; ti < NR; ti++) {
                const int ii = (ti0 + ti) * 16 + l15;
                const bool ok = ii < nrt && jj < wt;
                const double v = ld_off(tp, ok ? (unsigned)(ii + jj * rt) * 8u : 0u);
                acc[tj][ti][reg] = ok ? v : 0.0;
            }
        }

    // Task records are 48-byte packed structs; the NEXT record is requested at the top of an iteration and only made wave-uniform
    // (readfirstlane = the wait) at its end, so its latency hides under this task's MFMAs.  Round 6: the tile MAP of the next task
    // (which rows / columns of the tile a source that does not land contiguously feeds: 84 % of the tasks of the sparse tree's launch
    // into a front) is requested one task ahead as well -- its index comes from a two-word look-ahead (geom, map) two tasks ahead --
    // so that a task starts with ONE dependent round trip (its operands) instead of two (map, then operands).  The map loads are
    // unconditional (an unmapped task reads map 0, which always exists: hipkkt_setup.cpp) -- loads under a condition would make the
    // compiler wait for everything in flight at the next use of an earlier load.
    const int last = task_end - 1;
    auto gm_of = [&](int q, int &g_, int &m_) { const DenseTask *t = P.dtasks + (q < last ? q : last); g_ = t->geom; m_ = t->map; };
    auto req_maps = [&](int geom_u, int map_u, int (&mr_)[NR], int (&mc_)[NT]) {     // (geom_u, map_u: wave-uniform)
        const int16_t *tm = P.upd_tmap + (int64_t)((geom_u & (1 << 17)) ? map_u : 0) * 128;
#pragma unroll
        for (int x = 0; x < NR; x++) mr_[x] = tm[(ti0 + x) * 16 + l15];
#pragma unroll
        for (int x = 0; x < NT; x++) mc_[x] = tm[64 + (tj0 + x) * 16 + l15];
    };
    DenseTask Tc = P.dtasks[task_begin];
    int mr[NR], mc[NT], g1, m1;
    req_maps(rfl(Tc.geom), rfl(Tc.map), mr, mc);
    gm_of(task_begin + 1, g1, m1);
    for (int q = task_begin; q < task_end; q++) {
        const DenseTask Tn = P.dtasks[q + 1 < task_end ? q + 1 : q];
        int g2, m2, mrn[NR], mcn[NT];
        gm_of(q + 2, g2, m2);
        req_maps(rfl(g1), rfl(m1), mrn, mcn);
        const int row_lo = rfl(Tc.row_lo), nrows = rfl(Tc.nrows), col_lo = rfl(Tc.col_lo), ncols = rfl(Tc.ncols),
                  geom = rfl(Tc.geom), K = rfl(Tc.K);
        const unsigned r8 = (unsigned)rfl(Tc.r8);
        const double *sp = rfl_ptr(P.Lx + Tc.panel_off);
        const double *dv = rfl_ptr(P.D + Tc.dfirst);
        const int c_r = geom & 255, c_c = (geom >> 8) & 255;
        unsigned roff[NR], coff[NT], mbits = 0;
        if (geom & (1 << 17)) {      // wave-uniform: operands gathered through the task's tile maps
#pragma unroll
            for (int x = 0; x < NR; x++) {
                const int m = mr[x];
                roff[x] = (unsigned)(row_lo + (m >= 0 ? m : 0)) * 8u;
                mbits |= m >= 0 ? 16u << x : 0u;
            }
#pragma unroll
            for (int x = 0; x < NT; x++) {
                const int m = mc[x];
                coff[x] = (unsigned)(col_lo + (m >= 0 ? m : 0)) * 8u;
                mbits |= m >= 0 ? 1u << x : 0u;
            }
        } else {
#pragma unroll
            for (int x = 0; x < NR; x++) {
                const int ii = (ti0 + x) * 16 + l15 - c_r;
                const bool okr = ii >= 0 && ii < nrows;
                roff[x] = (unsigned)(row_lo + (okr ? ii : 0)) * 8u;
                mbits |= okr ? 16u << x : 0u;
            }
#pragma unroll
            for (int x = 0; x < NT; x++) {
                const int jj = (tj0 + x) * 16 + l15 - c_c;
                const bool okc = jj >= 0 && jj < ncols;
                coff[x] = (unsigned)(col_lo + (okc ? jj : 0)) * 8u;
                mbits |= okc ? 1u << x : 0u;
            }
        }
        // register double buffering: the loads of step k+1 (clamped past the end) are in flight during
        // the MFMAs of step k
        if (NT == 1 || DEEP) {
            // a 16-column strip has only 4 MFMAs (256 clocks) per k-step, less than one memory latency: keep FOUR
            // k-steps of operands in flight (ring of 4 register buffers; steps past K contribute zeros)
            DenseRaw<NT, NR> f0, f1, f2, f3;
            dense_load<NT, NR>(f0, sp, dv, coff, roff, r8, K, 0, lk);
            dense_load<NT, NR>(f1, sp, dv, coff, roff, r8, K, 4, lk);
            dense_load<NT, NR>(f2, sp, dv, coff, roff, r8, K, 8, lk);
            dense_load<NT, NR>(f3, sp, dv, coff, roff, r8, K, 12, lk);
            for (int k0 = 0; k0 < K; k0 += 16) {
                __builtin_amdgcn_sched_barrier(0);
                dense_mma<NT, NR>(f0, acc, mbits, K, k0, lk);
                dense_load<NT, NR>(f0, sp, dv, coff, roff, r8, K, k0 + 16, lk);
                __builtin_amdgcn_sched_barrier(0);
                dense_mma<NT, NR>(f1, acc, mbits, K, k0 + 4, lk);
                dense_load<NT, NR>(f1, sp, dv, coff, roff, r8, K, k0 + 20, lk);
                __builtin_amdgcn_sched_barrier(0);
                dense_mma<NT, NR>(f2, acc, mbits, K, k0 + 8, lk);
                dense_load<NT, NR>(f2, sp, dv, coff, roff, r8, K, k0 + 24, lk);
                __builtin_amdgcn_sched_barrier(0);
                dense_mma<NT, NR>(f3, acc, mbits, K, k0 + 12, lk);
                dense_load<NT, NR>(f3, sp, dv, coff, roff, r8, K, k0 + 28, lk);
            }
        } else {
            DenseRaw<NT, NR> fa, fb;
            dense_load<NT, NR>(fa, sp, dv, coff, roff, r8, K, 0, lk);
            for (int k0 = 0; k0 < K; k0 += 8) {     // steps past K contribute zeros (dk = 0)
                dense_load<NT, NR>(fb, sp, dv, coff, roff, r8, K, k0 + 4, lk);
                __builtin_amdgcn_sched_barrier(0);
                dense_mma<NT, NR>(fa, acc, mbits, K, k0, lk);
                __builtin_amdgcn_sched_barrier(0);
                dense_load<NT, NR>(fa, sp, dv, coff, roff, r8, K, k0 + 8, lk);
                __builtin_amdgcn_sched_barrier(0);
                if (k0 + 4 < K) dense_mma<NT, NR>(fb, acc, mbits, K, k0 + 4, lk);   // narrow sources (K <= 4): one step
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        Tc = Tn;
        g1 = g2; m1 = m2;
#pragma unroll
        for (int x = 0; x < NR; x++) mr[x] = mrn[x];
#pragma unroll
        for (int x = 0; x < NT; x++) mc[x] = mcn[x];
    }
#pragma unroll
    for (int tj = 0; tj < NT; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int jj = (tj0 + tj) * 16 + lk + 4 * reg;
#pragma unroll
            for (int ti = 0; ti < NR; ti++) {
                const int ii = (ti0 + ti) * 16 + l15;
                if (ii < nrt && jj < wt) st_off(tp, (unsigned)(ii + jj * rt) * 8u, acc[tj][ti][reg]);
            }
        }
}

// [Round 6, measured and removed: the general tile as ONE stream over all its tasks -- the next task's record / tile map / lane offsets
// prepared and its first operands requested while the current task is multiplied, instead of three dependent round trips per task.
// Bit-identical, 26 more registers spilled, and SLOWER: cfg 2a's factorisation 3.66 -> 3.77 ms, cfg 5's 4.55 -> 4.96 ms.  With two
// wavefronts per SIMD the other wavefront already covers those round trips.  A second attempt at the end of the round, on top of the
// map look-ahead of dense_tile_core and WITHOUT a second set of operand registers (the next task's first operands in place of the
// wasted load one step past the end): 14 more spills, 3.54 -> 3.68 ms / 4.20 -> 4.58 ms.  What did pay is the map look-ahead alone.]
// FULL tiles (DenseGroup::pad & 1, set when the plan is uploaded: a 64 x 64 tile whose every task covers all 64 rows and columns
// contiguously with K a multiple of 8 -- all but the edge tiles of a front's big launches): nothing to mask, the lane offsets are
// loop-invariant, and the k-steps advance the UNIFORM base pointers in scalar registers: 9 loads + 4 multiplies per 16 matrix-core
// instructions instead of 9 + 35 vector instructions (address arithmetic with a clamp and a 32-bit multiply, 18 selects), which
// cost matrix-core time one for one (the two pipes of a SIMD do not overlap for FP64).  Same products, same order of accumulation
// as dense_tile_core: bit-identical.  A function of its own (not a branch inside the task loop of dense_tile_core: that spilled
// 227 registers) so that its live ranges and the general path's are allocated separately.
__device__ __forceinline__ void dense_tile_core_full(const DevPlan &P, double *tp, int rt, int task_begin, int task_end, int lane) {
    const int l15 = lane & 15, lk = lane >> 4;
    v4f64 acc[4][4];
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int jj = tj * 16 + lk + 4 * reg;
#pragma unroll
            for (int ti = 0; ti < 4; ti++) acc[tj][ti][reg] = ld_off(tp, (unsigned)(ti * 16 + l15 + jj * rt) * 8u);
        }
    // One stream of k-steps over ALL tasks of the tile: the first operands of task q+1 are requested before the last matrix-core
    // instructions of task q (a task is one 64-column source panel = 16 k-steps; restarting the load pipeline per task exposed a
    // memory round trip five times per tile).  The record of task q+2 is requested at the same point.
    const double *sp, *dv;
    unsigned ca[4], rb[4], dl, step;
    int K;
    auto setup = [&](const DenseTask &T) {
        const int row_lo = rfl(T.row_lo), col_lo = rfl(T.col_lo);
        const unsigned r8 = (unsigned)rfl(T.r8);
        K = rfl(T.K);
        sp = rfl_ptr(P.Lx + T.panel_off);      // wave-uniform bases (scalar registers) + 32-bit lane offsets
        dv = rfl_ptr(P.D + T.dfirst);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            ca[t] = (unsigned)(col_lo + t * 16 + l15) * 8u + (unsigned)lk * r8;
            rb[t] = (unsigned)(row_lo + t * 16 + l15) * 8u + (unsigned)lk * r8;
        }
        dl = (unsigned)lk * 8u;
        step = 4u * r8;
    };
    DenseRaw<4, 4> fa, fb;
    auto load = [&](DenseRaw<4, 4> &f) {
        f.d = ld_off(dv, dl);
#pragma unroll
        for (int t = 0; t < 4; t++) f.a[t] = ld_off(sp, ca[t]);
#pragma unroll
        for (int t = 0; t < 4; t++) f.b[t] = ld_off(sp, rb[t]);
    };
    auto advance = [&]() {          // the lane offsets move on by one k-step (4 columns of the source panel): 9 adds
        dl += 32u;
#pragma unroll
        for (int t = 0; t < 4; t++) { ca[t] += step; rb[t] += step; }
    };
    auto mma = [&](const DenseRaw<4, 4> &f) {
        const double nd = -f.d;
        double a[4];
#pragma unroll
        for (int t = 0; t < 4; t++) a[t] = f.a[t] * nd;
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], f.b[ti], acc[tj][ti], 0, 0, 0);
    };
    DenseTask Tn = P.dtasks[task_begin + 1 < task_end ? task_begin + 1 : task_begin];
    setup(P.dtasks[task_begin]);
    load(fa);
    advance();
    for (int q = task_begin; q < task_end; q++) {
        const int Kq = K;
        for (int k0 = 0; k0 < Kq; k0 += 8) {
            load(fb);
            __builtin_amdgcn_sched_barrier(0);
            mma(fa);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 8 < Kq) {
                advance();
                load(fa);
                advance();
            } else if (q + 1 < task_end) {                       // the next task's first k-step, and the record after it
                const DenseTask Tc = Tn;
                Tn = P.dtasks[q + 2 < task_end ? q + 2 : q + 1];
                setup(Tc);
                load(fa);
                advance();
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(fb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int jj = tj * 16 + lk + 4 * reg;
#pragma unroll
            for (int ti = 0; ti < 4; ti++) st_off(tp, (unsigned)(ti * 16 + l15 + jj * rt) * 8u, acc[tj][ti][reg]);
        }
}

template <int NT, int NR, bool DEEP = false>
__device__ __forceinline__ void dense_tile(const DevPlan &P, const DenseGroup *Gp, int lane, int tj0, int ti0) {
    // one self-contained record per tile (no group -> supernode tables -> panel chain of dependent loads)
    const DenseGroup G = *Gp;
    const int task_begin = rfl(G.task_begin), task_end = rfl(G.task_end);
    const int wt = rfl(G.wt);
    if (tj0 * 16 >= wt) return;
    const int rt = rfl(G.rt);
    double *tp = rfl_ptr(P.Lx + G.tile_off);
    const int nrt = rfl(G.nrt);
    if (ti0 * 16 >= nrt) return;
    if (NT == 4 && NR == 4 && (rfl(G.pad) & 1)) {       // wave-uniform
        dense_tile_core_full(P, tp, rt, task_begin, task_end, lane);
        return;
    }
    dense_tile_core<NT, NR, DEEP>(P, tp, rt, nrt, wt, task_begin, task_end, lane, tj0, ti0);
}

}  // namespace hipkkt
