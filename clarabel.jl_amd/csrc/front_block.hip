// K4 over a FRONT, one launch per update batch: the nb <= 5 consecutive 64-column panels of a front that share
// an update batch are factored by ONE kernel instead of nb x (panel kernel + just-in-time update kernel).  Workgroup b owns the b-th
// 64-row block of the batch's first panel (tickets in arrival order, like the front sweeps of the solves); it keeps its rows of all nb
// column blocks in matrix-core accumulators and walks the columns left to right:
//     step j:   X_j = A_j * (L_jj^-T D_j^-1)              16x16x4 FP64 MFMA against the published inverse of the diagonal block
//               A_k -= (X_j D_j) * L(k,j)^T   for k > j    MFMA against the published rows of the diagonal workgroups
// The workgroups of the first nb blocks ("diagonal" workgroups) stop at their own diagonal tile, eliminate it with the 8-pivot-blocked
// in-register LDL^T of k_factor_panel (same pivot rule, kernels.hip), invert the unit-lower factor and publish  L^-T D^-1  and their
// L(k,j) tiles through a scratch area of whole, 128-byte aligned tiles (write-through stores + drained flag; consumers: one relaxed
// poll, one agent acquire, sc1 loads -- MI355X_MICROARCH.md "inter-workgroup visibility").  Nothing another workgroup reads is read
// before its flag, no scratch line is shared between producers, and a workgroup only ever waits for lower tickets (already running):
// no deadlock at any grid size; every spin is bounded and aborts the whole factorisation through FL_FACFAIL (the host then repeats it
// with one launch per panel).
//
// STREAMED PIVOT CHAIN (round 4, template parameter STREAM).  The critical path of a front is the chain of its diagonal tiles.  In the
// round-3 form the diagonal workgroup i waited for ALL 64 pivots of tile i-1, its explicit inverse and the publish before it could form
// X = A(i,i-1) L^-T D^-1, update its own tile and start its pivots: 22.7 us per panel of which 9.9 us are pivots.  Streamed: workgroup
// i-1 publishes every finished block of 8 pivots (the raw columns a_rk = d_k l_rk of its tile, d_k, 1/d_k: one 528-double record,
// write-through stores, no flag -- the record area is filled with a NaN sentinel before every factorisation and the consumer polls the
// data itself) and workgroup i treats its rows of panel i-1 as what they are, 64 more rows of that panel: per record it eliminates the
// block's 8 columns of A(i,i-1) by substitution (lane = row, the arithmetic of k_factor_panel's chunk rows), then applies the rank-8
// update to the rest of A(i,i-1) and to its own diagonal tile on the matrix core.  It trails the producer by one hand-off and starts
// its own pivots one block time after the producer's last pivot: the inverse, its publish and both 64^3 products leave the chain
// (the inverse is still formed and published, for the workgroups below the diagonal, while the next tile is already being eliminated).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "dense_tile.h"

namespace hipkkt {

constexpr int FLD = 65;   // LDS row stride of a 64 x 64 tile

__device__ __forceinline__ int fb_ldi(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double fb_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fb_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double fb_readlane(double x, int l) {   // l wave-uniform
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// 1 / d on the pivot chain: the hardware reciprocal r (2^-27 or better), then r (1 + e + e^2) with e = 1 - d r -- three dependent
// operations instead of the four of two Newton steps, error ~ e^3 + one rounding
__device__ __forceinline__ double fb_rcp3(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    const double e2 = fma(e, e, e);
    return fma(r, e2, r);
}
__device__ __forceinline__ double fb_rcp(double d) {   // kernels.hip pivot_rcp
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double2 fb_unpack(v4u r) {
    double2 d;
    d.x = __longlong_as_double((long long)(((unsigned long long)r[1] << 32) | r[0]));
    d.y = __longlong_as_double((long long)(((unsigned long long)r[3] << 32) | r[2]));
    return d;
}
// four 16-byte sc1 loads in flight per lane (issue and wait in ONE asm statement: kernels.hip, inline-asm lessons)
__device__ __forceinline__ void fb_ld16x4(const double *p0, const double *p1, const double *p2, const double *p3, double2 &d0,
                                          double2 &d1, double2 &d2, double2 &d3) {
    v4u r0, r1, r2, r3;
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
    d0 = fb_unpack(r0); d1 = fb_unpack(r1); d2 = fb_unpack(r2); d3 = fb_unpack(r3);
}
__device__ __forceinline__ void fb_st16(double *p, double a, double b) {
    const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    v4u r = {(unsigned)ua, (unsigned)(ua >> 32), (unsigned)ub, (unsigned)(ub >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(r) : "memory");
}
// LDS-only workgroup barrier: __syncthreads() also waits for this wave's outstanding global stores (vmcnt), which would put the
// write-through latency of every streamed record on the pivot chain.  Nothing in the loops that use it communicates through global
// memory inside the workgroup.
__device__ __forceinline__ void fb_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// An exec-masked block (s_cbranch_execz) between a matrix-core instruction and the first read of its accumulators (v_accvgpr_read)
// is a trap: the waves that skip the block arrive at the read with fewer wait states than the compiler's hazard recogniser counted
// along the fall-through path, and read accumulators that are still being written (measured in round 4: deterministic 1e-4 errors
// of the factorisation with a store block of two of the four waves behind the rank-8 update).  The loops below keep such blocks
// away from that path (they sit behind a barrier, next to wave 0's eliminations); where one cannot be avoided, this pins 24 wait
// states (enough for the 16-pass FP64 MFMA) between the matrix-core instructions before it and whatever follows.
__device__ __forceinline__ void fb_mfma_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// "not written yet" in a stream record: a NaN no arithmetic produces (both halves equal, so a 32-bit fill pattern would do too)
constexpr unsigned kFbSentHalf = 0xFFFFFFFFu;               // (all ones: front_block2.hip tests a whole record with unsigned maxima of the high words)
constexpr unsigned long long kFbSentinel = 0xFFFFFFFFFFFFFFFFull;
__device__ __forceinline__ bool fb_fresh(v4u r) {
    return !(r[0] == kFbSentHalf && r[1] == kFbSentHalf) && !(r[2] == kFbSentHalf && r[3] == kFbSentHalf);
}
// polls two 16-byte chunks of a stream record until all four doubles have been written (per lane; bounded)
__device__ __forceinline__ bool fb_poll2(const double *p0, const double *p1, double2 &d0, double2 &d1, int *err, int *failflag, unsigned lim) {
    for (unsigned spins = 0;; spins++) {
        v4u r0, r1;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(r0), "=&v"(r1)
                     : "v"(p0), "v"(p1)
                     : "memory");
        if (fb_fresh(r0) && fb_fresh(r1)) { d0 = fb_unpack(r0); d1 = fb_unpack(r1); return true; }
        if ((spins & 63u) == 63u || lim < 64u) {
            if (spins > lim) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(failflag, 1);
                return false;
            }
            if (fb_ldi(err) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// a 64 x 64 row-major scratch tile <-> an LDS tile [row * FLD + col]: 8 chunks of 16 bytes per thread
__device__ __forceinline__ void fb_tile_load(const double *tl, double *S, int tid) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        double2 d[4];
        const int i0 = tid + 1024 * h;
        fb_ld16x4(tl + 2 * i0, tl + 2 * (i0 + 256), tl + 2 * (i0 + 512), tl + 2 * (i0 + 768), d[0], d[1], d[2], d[3]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = i0 + 256 * q, row = idx >> 5, c = 2 * (idx & 31);
            S[row * FLD + c] = d[q].x;
            S[row * FLD + c + 1] = d[q].y;
        }
    }
}
__device__ __forceinline__ void fb_tile_store(double *tl, const double *S, int tid) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int idx = tid + 256 * q, row = idx >> 5, c = 2 * (idx & 31);
        fb_st16(tl + 2 * idx, S[row * FLD + c], S[row * FLD + c + 1]);
    }
}

// whole-workgroup wait for a flag of another workgroup; false = timed out / somebody else failed (uniform over the workgroup)
__device__ __forceinline__ bool fb_wait(const int *flag, int *err, int *failflag, unsigned lim, int *sres) {
    if (threadIdx.x == 0) {
        int ok = 1;
        for (unsigned spins = 0; fb_ldi(flag) == 0; spins++) {
            if ((spins & 63u) == 63u || lim < 64u) {
                if (spins > lim) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicOr(failflag, 1);
                    ok = 0;
                    break;
                }
                if (fb_ldi(err) != 0) { ok = 0; break; }
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *sres = ok;
    }
    __syncthreads();
    const bool ok = *sres != 0;           // no acquire fence: every hand-off payload is stored AND loaded with sc1 (bypasses this CU's L1)
    __syncthreads();                      // *sres may be rewritten by the next wait
    return ok;
}
// all payload stores of the workgroup drained (write-through sc1 stores: no L2 write-back needed), then the flag
__device__ __forceinline__ void fb_publish(int *flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one 16 x 16 product on the matrix core, operands in LDS: C = sum_k A[ar + l15][ac + k] B[br + k][bc + l15], k < 4 * nk
__device__ __forceinline__ v4f64 fb_mm16(const double *A, int ar, int ac, const double *Bm, int br, int bc, int nk, int l15, int lk) {
    v4f64 c = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < nk; kk++)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(ar + l15) * FLD + ac + 4 * kk + lk], Bm[(br + 4 * kk + lk) * FLD + bc + l15], c, 0, 0, 0);
    return c;
}
// panel rows of this block (column-major) and their row-major copy for the backward solves, from an LDS tile
__device__ __forceinline__ void fb_store_panel(const DevPlan &P, const FrontPanel &pj, int roff, int nr, const double *S, int tid) {
    double *dst = P.Lx + pj.panel_off + roff;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int row = idx & 63, col = idx >> 6;
        if (row < nr) dst[row + (int64_t)col * pj.r] = S[row * FLD + col];
    }
    double *lt = P.LT + pj.lt_off + (int64_t)(roff - 64) * 64;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int col = idx & 63, row = idx >> 6;
        if (row < nr) lt[idx] = S[row * FLD + col];
    }
}

// kept out of line: the register allocation of the panel path must not depend on it
// Each wavefront is alone on its SIMD (100 KB of LDS per workgroup): tiles with four k-steps of operands in flight, kFbExtraPerWave
// of them one after the other -- the launch lasts as long as its panel chain anyway.
__device__ __forceinline__ void fb_extra_tiles(const DevPlan &P, int begin, int count, int xb, int per_wave) {
    const int lane = threadIdx.x & 63;
    int idx = rfl((xb * 4 + (int)(threadIdx.x >> 6)) * per_wave);
    for (int q = 0; q < per_wave && idx < count; q++, idx++) dense_tile<4, 4, true>(P, P.dgroups + begin + idx, lane, 0, 0);
}

#define FB_T(slot) do { if (trace && tid == 0 && i < 5) trace[(B.sync_off / 128 * 8 + i) * 16 + (slot)] = (long long)wall_clock64(); } while (0)
// shader-clock stamps of workgroup 0's pivot loop, 4 per block of 8 pivots (trace words 80 ..): after the first barrier, after wave 0's
// eliminations, after the second barrier, after the rank-8 update
#define FB_TB(ph) do { if (trace && tid == 0 && i == 0) trace[(B.sync_off / 128 * 8 + 5) * 16 + 4 * Bk + (ph)] = (long long)clock64(); } while (0)
template <bool STREAM>
__global__ void __launch_bounds__(256)
k_front_block(DevPlan P, FrontBatch B, int *sync_all, double *scratch_all, double *stream_all, double dyn_eps, double dyn_delta,
              long long *trace) {
    __shared__ double Sa[64 * FLD];     // this workgroup's 64 x 64 strip: A_j, then X_j; the pivot loop's small buffers; the inverse
    __shared__ double Sb[64 * FLD];     // the other operand: Minv_j, then L(k,j); the pivot loop's result L11
    __shared__ double St[64 * FLD];     // products of the blocked inverse
    __shared__ double Sd[64];           // D_j
    __shared__ int sblk, sres;
    int *sync = sync_all + B.sync_off;
    int *err = sync + 1, *fl_minv = sync + 32, *fl_L = sync + 64;
    double *scratch = scratch_all + B.scratch_off;
    double *ltiles = scratch + (int64_t)kFbMax * 4160;
    double *stream = stream_all + B.stream_off;            // STREAM: record (panel j, block Bk) at ((8 j + Bk) * kFbRec)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, lk = lane >> 4;
    if ((int)blockIdx.x >= B.i_end - B.i_base) {          // extra workgroup: four tiles of the previous update stage (no hand-off, no LDS)
        fb_extra_tiles(P, B.x_begin, B.x_count, (int)blockIdx.x - (B.i_end - B.i_base), B.pad > 0 ? B.pad : 1);
        return;
    }
    if (tid == 0) sblk = atomicAdd(sync + B.tick, 1);
    __syncthreads();
    const int i = B.i_base + sblk;                        // row block of this workgroup
    if (i >= B.i_end) return;
    FB_T(0);
    const FrontPanel *fp = P.front_panels + B.fp_off;
    const int nb = B.nb;
    const bool diag = i < nb;
    const int ncb = diag ? i + 1 : nb;                    // column blocks held here
    // STREAM: a diagonal workgroup's last step (the panel just left of its own tile) is the streamed one after this loop
    const int nsteps = diag ? (STREAM ? i - 1 : i) : nb;
    const int nr = min(64, B.r0 - 64 * i);
    // expected pivot signs of this workgroup's own columns: requested now, used by the pivots (a diagonal workgroup's critical path)
    const int f = diag ? fp[i].f : 0;
    const signed char sgn_l = diag ? P.sgn_perm[f + lane] : (signed char)0;
    v4f64 acc[kFbMax][4];
    // ---- load the row block (coalesced over rows, through LDS into the accumulator layout: lane (col l15, row lk + 4 reg))
#pragma unroll
    for (int k = 0; k < kFbMax; k++) {
        if (k < ncb) {
            const FrontPanel pk = fp[k];
            const double *src = P.Lx + pk.panel_off + 64 * (i - k);
            double tmp[16];
#pragma unroll
            for (int q = 0; q < 16; q++) tmp[q] = lane < nr ? src[lane + (int64_t)(wv + 4 * q) * pk.r] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; q++) Sa[lane * FLD + wv + 4 * q] = tmp[q];
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) acc[k][sub][reg] = Sa[(16 * wv + lk + 4 * reg) * FLD + 16 * sub + l15];
            __syncthreads();
        }
    }
    FB_T(1);
    // A diagonal workgroup's own tile lives in accumulators of its own from the start: its update by panel j needs nothing from another
    // workgroup, so it is applied FIRST in every step and the waits for the tiles L(k, j) of the workgroups above overlap with it
    // (round 4: the next diagonal workgroup's streamed step starts 2 us earlier; same updates in the same order: bit-identical).
    v4f64 tacc[4];
#pragma unroll
    for (int sub = 0; sub < 4; sub++) {
        tacc[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < kFbMax; k++) if (k == i) tacc[sub] = acc[k][sub];
    }
    // ---- the columns left of this block's own tile
#pragma unroll
    for (int j = 0; j < kFbMax; j++) {
        if (j < nsteps) {                                  // workgroup-uniform
            const bool last = !STREAM && diag && j == nsteps - 1;     // next: this workgroup's own diagonal tile (critical path of the front)
            if (!fb_wait(fl_minv + j, err, P.flags + FL_FACFAIL, P.spin_limit, &sres)) return;
            if (j == nsteps - 1) FB_T(2);
            const double *mv = scratch + (int64_t)j * 4160;
            fb_tile_load(mv, Sb, tid);                                                                        // Minv[k][c]
            if (tid < 64) Sd[tid] = fb_ld(mv + 4096 + tid);
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Sa[(16 * wv + lk + 4 * reg) * FLD + 16 * sub + l15] = acc[j][sub][reg];
            __syncthreads();
            if (j == nsteps - 1) FB_T(3);
            // X_j = A_j Minv_j (Minv upper triangular: k-steps below a 16-column strip's diagonal are skipped)
            v4f64 x[4];
#pragma unroll
            for (int sub = 0; sub < 4; sub++) x[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                const double a = Sa[(16 * wv + l15) * FLD + 4 * kk + lk];
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
                    if (4 * kk <= 16 * sub + 15)
                        x[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Sb[(4 * kk + lk) * FLD + 16 * sub + l15], x[sub], 0, 0, 0);
            }
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Sa[(16 * wv + lk + 4 * reg) * FLD + 16 * sub + l15] = x[sub][reg];
            __syncthreads();
            if (j == nsteps - 1) FB_T(4);
            // L(i,j): diagonal workgroups hand their tile to the workgroups below through the scratch area; the panel (column-major)
            // and its row-major copy for the backward solves are written now, or -- before a diagonal tile -- after the pivots
            if (diag) fb_tile_store(ltiles + (int64_t)(i * (i - 1) / 2 + j) * 4096, Sa, tid);
            if (!last) {
                fb_store_panel(P, fp[j], 64 * (i - j), nr, Sa, tid);
                if (diag) fb_publish(fl_L + 8 * i + j);
            }   // (last: the panel copy is made from the scratch tile after this workgroup's own pivots)
            if (j == nsteps - 1) FB_T(5);
            // A_k -= (X_j D_j) L(k,j)^T: the diagonal tile first (L(i,j) is this workgroup's own X_j) ...
            if (diag) {
#pragma unroll
                for (int kk = 0; kk < 16; kk++) {
                    const double a = -(Sa[(16 * wv + l15) * FLD + 4 * kk + lk] * Sd[4 * kk + lk]);
#pragma unroll
                    for (int sub = 0; sub < 4; sub++)
                        tacc[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Sa[(16 * sub + l15) * FLD + 4 * kk + lk], tacc[sub], 0, 0, 0);
                }
            }
            // ... then the tiles that need the L(k,j) of the diagonal workgroups above
#pragma unroll
            for (int k = j + 1; k < kFbMax; k++) {
                if (k < ncb && !(diag && k == i)) {
                    if (!fb_wait(fl_L + 8 * k + j, err, P.flags + FL_FACFAIL, P.spin_limit, &sres)) return;
                    fb_tile_load(ltiles + (int64_t)(k * (k - 1) / 2 + j) * 4096, Sb, tid);
                    __syncthreads();
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) {
                        const double a = -(Sa[(16 * wv + l15) * FLD + 4 * kk + lk] * Sd[4 * kk + lk]);
#pragma unroll
                        for (int sub = 0; sub < 4; sub++)
                            acc[k][sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Sb[(16 * sub + l15) * FLD + 4 * kk + lk], acc[k][sub], 0, 0, 0);
                    }
                    __syncthreads();                      // Sb is overwritten by the next tile / the next step
                }
            }
            if (diag) __syncthreads();                    // Sa (X_j, read by the diagonal tile's update) is rewritten by the next step
        }
    }
    // ---- the diagonal tile stays in its accumulators tacc (rows 16 wv + lk + 4 reg, columns 16 sub + l15).
    double *Pc = Sa;                                                 // [64][9]   the block's columns, by row
    double (*colL)[64] = (double (*)[64])(Sa + 64 * 9);              // [8][64]   l_ik
    double (*colC)[64] = (double (*)[64])(Sa + 64 * 9 + 512);        // [8][64]   raw a_ik = d_k l_ik
    double *dsave = Sa + 64 * 9 + 1024;                              // [64]
    double *dinvs = Sa + 64 * 9 + 1024 + 64;                         // [64]      1/d_k as used on the chain (fb_rcp)
    double (*colC2)[64] = (double (*)[64])(Sa + 64 * 9 + 1024 + 128);   // [8][64] second buffer of colC for the pivot loop (STREAM)
    double lrow[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};       // STREAM, wave 3: the block of L(i, i-1) whose store is pending
    if (STREAM && diag && i > 0) {
        // ---- streamed step: this workgroup's rows of panel i-1, block by block behind the workgroup that eliminates tile i-1
        v4f64 xacc[4];
#pragma unroll
        for (int sub = 0; sub < 4; sub++) {
            xacc[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < kFbMax; k++) if (k == i - 1) xacc[sub] = acc[k][sub];
        }
        const double *rec0 = stream + (int64_t)(i - 1) * 8 * kFbRec;
        double *ltile = ltiles + (int64_t)(i * (i - 1) / 2 + i - 1) * 4096;       // L(i, i-1), row-major, for the workgroups below
        if (tid == 0) sres = 1;
        __syncthreads();                                   // Sa / Sb (operands of the last regular step) are free
        // Waves 0 and 1 fetch the records (wave 3 has write-through stores in flight: it must not wait on the vector-memory counter).
        // The record of block Bk + 1 is REQUESTED while block Bk is processed (plain agent-scope loads, the compiler places the wait at
        // their first use): a consumer that has fallen behind finds it complete and pays no memory round trip per block; one that is
        // level with the producer sees the sentinel and polls.
        const int pc = wv < 2 ? 2 * tid : 0, pe = (tid >= 64 && tid < 72) ? 512 + 2 * (tid - 64) : 0;
        double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0, n4 = 0.0, n5 = 0.0;
        auto request = [&](const double *rec) {
            n0 = fb_ld(rec + pc); n1 = fb_ld(rec + pc + 1); n2 = fb_ld(rec + pc + 256); n3 = fb_ld(rec + pc + 257);
            n4 = fb_ld(rec + pe); n5 = fb_ld(rec + pe + 1);
        };
        auto fresh = [&](double v) { return (unsigned long long)__double_as_longlong(v) != kFbSentinel; };
        if (wv < 2) request(rec0);
#pragma unroll
        for (int Bk = 0; Bk < 8; Bk++) {
            {
                const int sub = Bk >> 1, c0 = 8 * (Bk & 1);
                if (l15 >= c0 && l15 < c0 + 8) {
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) Pc[(16 * wv + lk + 4 * reg) * 9 + l15 - c0] = xacc[sub][reg];
                }
            }
            double *CR = Sb + (Bk & 1) * kFbRec;           // the record: [8][64] raw columns of tile i-1, 8 pivots, 8 reciprocals
            if (wv < 2) {
                const double *rec = rec0 + (int64_t)Bk * kFbRec;
                bool ok = true;
                if (!(fresh(n0) && fresh(n1) && fresh(n2) && fresh(n3) && fresh(n4) && fresh(n5))) {      // not there yet: poll
                    double2 d0, d1, e0, e1;
                    ok = fb_poll2(rec + pc, rec + pc + 256, d0, d1, err, P.flags + FL_FACFAIL, P.spin_limit) &&
                         fb_poll2(rec + pe, rec + pe, e0, e1, err, P.flags + FL_FACFAIL, P.spin_limit);
                    n0 = d0.x; n1 = d0.y; n2 = d1.x; n3 = d1.y; n4 = e0.x; n5 = e0.y;
                }
                CR[pc] = n0; CR[pc + 1] = n1; CR[pc + 256] = n2; CR[pc + 257] = n3;
                if (pe) { CR[pe] = n4; CR[pe + 1] = n5; }
                if (!ok) sres = 0;
            }
            fb_bar();
            if (sres == 0) return;                         // (uniform: nobody writes sres after this point)
            if (Bk == 0) FB_T(12);
            if (wv < 2 && Bk < 7) request(rec0 + (int64_t)(Bk + 1) * kFbRec);
            if (wv == 0) {
                double p[8], cr[28], dv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { p[q] = Pc[lane * 9 + q]; dv[q] = CR[520 + q]; }
#pragma unroll
                for (int kk = 0, t = 0; kk < 8; kk++)
#pragma unroll
                    for (int jj = kk + 1; jj < 8; jj++, t++) cr[t] = CR[kk * 64 + 8 * Bk + jj];       // a_{jj,kk} of tile i-1 (wave-uniform)
#pragma unroll
                for (int kk = 0, t = 0; kk < 8; kk++) {
                    const double li = p[kk] * dv[kk];
                    colL[kk][lane] = li;
                    colC[kk][lane] = p[kk];
#pragma unroll
                    for (int jj = kk + 1; jj < 8; jj++, t++) p[jj] = fma(-li, cr[t], p[jj]);
                }
            } else if (wv == 3 && Bk > 0) {                // next to wave 0's elimination: L(i, i-1)[row = lane][8 (Bk-1) ..], 64 bytes per lane
#pragma unroll
                for (int q = 0; q < 4; q++) fb_st16(ltile + lane * 64 + 8 * (Bk - 1) + 2 * q, lrow[2 * q], lrow[2 * q + 1]);
            }
            fb_bar();
            if (wv == 3) {                                 // this block's l values, stored during the next block's elimination
#pragma unroll
                for (int q = 0; q < 8; q++) lrow[q] = colL[q][lane];
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const double av = -colL[4 * ks + lk][16 * wv + l15];
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
                    if (sub >= ((8 * Bk + 8) >> 4))
                        xacc[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, CR[(4 * ks + lk) * 64 + 16 * sub + l15], xacc[sub], 0, 0, 0);
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
                    tacc[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, colC[4 * ks + lk][16 * sub + l15], tacc[sub], 0, 0, 0);
            }
            // straight from the matrix-core instructions to the next iteration's accumulator reads (fb_mfma_settle's comment); Pc is
            // rewritten at once (only wave 0 read it, before the barrier above); colL / colC by wave 0 after the next iteration's
            // first barrier; the record buffer alternates
        }
        fb_mfma_settle();                                  // the trace stamps / the ballot below are exec-masked blocks in front of the pivots' reads of tacc
        FB_T(13);
    }
    FB_T(6);
    if (!diag) return;
    // ---- Per block of 8 pivots: the waves hand the block's 8 columns to wave 0 through LDS, wave 0 eliminates them without leaving
    //      the wavefront (lane = row; the pivot rule and arithmetic of k_factor_panel, kernels.hip), and every wave applies the rank-8
    //      update to its 16 rows on the matrix core.  L11 is collected in Sb for the inverse and the solves.
    //      STREAM: the loop's barriers are LDS-only; the raw columns alternate between two buffers so that waves 1 and 2 can publish
    //      the record of block Bk - 1 (write-through stores, no flag) WHILE wave 0 eliminates block Bk -- they idle there anyway, and
    //      nothing is added between the two barriers that bracket the matrix-core update.
    FB_T(7);
    const unsigned long long spos = __ballot(sgn_l > 0);
    const double dyn_delta_inv = 1.0 / dyn_delta;
    int nreg = 0;
    const bool pub = STREAM && i + 1 < nb && (wv == 1 || wv == 2);
    auto publish = [&](int Bp) {                          // record of block Bp for the next diagonal workgroup
        const int c = tid - 64;
        const double *flat = (Bp & 1) ? &colC2[0][0] : &colC[0][0];
        double *rec = stream + ((int64_t)i * 8 + Bp) * kFbRec;
        fb_st16(rec + 2 * c, flat[2 * c], flat[2 * c + 1]);
        fb_st16(rec + 2 * c + 256, flat[2 * c + 256], flat[2 * c + 257]);
        if (c >= 64 && c < 68) fb_st16(rec + 512 + 2 * (c - 64), dsave[8 * Bp + 2 * (c - 64)], dsave[8 * Bp + 2 * (c - 64) + 1]);
        if (c >= 68 && c < 72) fb_st16(rec + 520 + 2 * (c - 68), dinvs[8 * Bp + 2 * (c - 68)], dinvs[8 * Bp + 2 * (c - 68) + 1]);
    };
    if (STREAM) fb_bar(); else __syncthreads();           // Sa (X_j) and Sb (the last operand tile / record) are free
#pragma unroll
    for (int Bk = 0; Bk < 8; Bk++) {
        double (*cC)[64] = (STREAM && (Bk & 1)) ? colC2 : colC;
        {
            const int sub = Bk >> 1, c0 = 8 * (Bk & 1);
            if (l15 >= c0 && l15 < c0 + 8) {
#pragma unroll
                for (int reg = 0; reg < 4; reg++) Pc[(16 * wv + lk + 4 * reg) * 9 + l15 - c0] = tacc[sub][reg];
            }
        }
        if (STREAM) {
            fb_bar();
        } else {
            if (Bk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the L tile stores issued before the pivots have drained ...
            __syncthreads();
            if (Bk == 0 && i > 0 && tid == 0)                                // ... for every thread: hand the tile over
                __hip_atomic_store(fl_L + 8 * i + (i - 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        FB_TB(0);
        if (wv == 0) {
            // Wave 0 is ISSUE-bound here (one FP64 instruction per 8 cycles, one 32-bit one per 4; measured with pieces compiled out:
            // DESIGN.md), so the loop is written for few instructions AND a short pivot-to-pivot chain:
            //   d_k = a_kk - c_{k,k-1}^2 / d_{k-1}  ->  1 / d_k   is one fma, the hardware reciprocal and three more fma (fb_rcp3);
            //   a_kk and c_{k,k-1} are fetched from the lanes that hold them one pivot EARLIER (with the updates of the pivots before
            //   k-1 applied; what pivot k-1 contributes is the fma on the chain);
            //   the pivot rule is evaluated next to the reciprocal and applied by register selects -- written so that the compiler
            //   can neither branch on it (vector compare -> scalar branch -> reciprocal) nor mask the exec register for it (that costs
            //   ~85 cycles per pivot: the scalar unit waits for the vector compare);
            //   the pivots / reciprocals are written by one lane ONCE per block.
            double pcol[8], dk[8], dik[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pcol[q] = Pc[lane * 9 + q];
            double akk = fb_readlane(pcol[0], 8 * Bk), csq = 0.0, dinv_prev = 0.0;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int k = 8 * Bk + kk;
                double d = fma(-csq, dinv_prev, akk);
                const int sm = ((spos >> k) & 1ull) ? 0 : (int)0x80000000;               // expected sign negative: test -d, substitute -delta
                const bool bad = __hiloint2double(__double2hiint(d) ^ sm, __double2loint(d)) < dyn_eps;
                double dinv = fb_rcp3(d);
                double dsub = __hiloint2double(__double2hiint(dyn_delta) ^ sm, __double2loint(dyn_delta));
                double isub = __hiloint2double(__double2hiint(dyn_delta_inv) ^ sm, __double2loint(dyn_delta_inv));
                asm volatile("" : "+v"(dinv), "+v"(dsub), "+v"(isub));                   // all three in vector registers, unconditionally
                d = bad ? dsub : d;
                dinv = bad ? isub : dinv;
                nreg += bad ? 1 : 0;
                dk[kk] = d;
                dik[kk] = dinv;
                const double reg = pcol[kk];
                if (kk < 7) {
                    akk = fb_readlane(pcol[kk + 1], k + 1);
                    const double cn = fb_readlane(reg, k + 1);
                    csq = cn * cn;
                }
                dinv_prev = dinv;
                const double li = reg * dinv;
                colL[kk][lane] = li;
                cC[kk][lane] = reg;
                Sb[lane * FLD + k] = li;
#pragma unroll
                for (int jj = kk + 1; jj < 8; jj++) {
                    const double cj = fb_readlane(reg, 8 * Bk + jj);
                    pcol[jj] = fma(-li, cj, pcol[jj]);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 8; q++) { dsave[8 * Bk + q] = dk[q]; dinvs[8 * Bk + q] = dik[q]; }
            }
        } else if (STREAM) {                              // next to wave 0's elimination
            if (pub && Bk > 0) publish(Bk - 1);
            if (wv == 3 && i > 0) {
                if (Bk == 0) {                            // the last block of L(i, i-1) from the streamed step
#pragma unroll
                    for (int q = 0; q < 4; q++) fb_st16(ltiles + (int64_t)(i * (i - 1) / 2 + i - 1) * 4096 + lane * 64 + 56 + 2 * q, lrow[2 * q], lrow[2 * q + 1]);
                }
                if (Bk == 1) {                            // ... drained by now (issued one block = 1.4 us ago): hand the tile to the workgroups below
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(fl_L + 8 * i + (i - 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (wv == 0) FB_TB(1);
        if (STREAM) fb_bar(); else __syncthreads();
        FB_TB(2);
        if (Bk < 7) {
            // a_ij -= sum_k l_ik a_jk over the block's 8 pivots, for the 16-column strips that still hold live columns
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const double av = -colL[4 * ks + lk][16 * wv + l15];
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
                    if (sub >= ((8 * Bk + 8) >> 4))
                        tacc[sub] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, cC[4 * ks + lk][16 * sub + l15], tacc[sub], 0, 0, 0);
            }
        }
        fb_mfma_settle();                                 // the stamp below is an exec-masked block between the matrix-core instructions above and the
        FB_TB(3);                                         // next iteration's reads of their accumulators (this form is not the default any more)
        // Pc / colL are rewritten after the next iteration's first barrier / by wave 0 after it: every wave is past its reads
    }
    if (pub) publish(7);
    __syncthreads();
    FB_T(8);
    // ---- L11^-1 (blocked: 16 x 16 diagonal blocks by substitution, the rest on the matrix core), then  Minv = L11^-T D^-1
    double dkeep = 0.0;
    if (tid < 64) { dkeep = dsave[tid]; Sd[tid] = 1.0 / dkeep; }
    __syncthreads();
    for (int idx = tid; idx < 64 * FLD; idx += 256) Sa[idx] = 0.0;
    __syncthreads();
    if (tid < 64) {   // thread = column j of diagonal block bq
        const int bq = tid >> 4, jc = tid & 15, o = 16 * bq;
        double xx[16];
#pragma unroll
        for (int c = 0; c < 16; c++) xx[c] = c == jc ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++)
#pragma unroll
            for (int r_ = k + 1; r_ < 16; r_++) xx[r_] = fma(-Sb[(o + r_) * FLD + o + k], xx[k], xx[r_]);
#pragma unroll
        for (int c = 0; c < 16; c++) Sa[(o + c) * FLD + o + jc] = xx[c];
    }
    __syncthreads();
    if (wv < 2) {     // block size 16, pairs (0,1) and (2,3):  T = L21 X11
        const int o = 32 * wv;
        const v4f64 c = fb_mm16(Sb, o + 16, o, Sa, o, o, 4, l15, lk);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) St[(o + 16 + lk + 4 * reg) * FLD + o + l15] = c[reg];
    }
    __syncthreads();
    if (wv < 2) {     // X21 = -X22 T
        const int o = 32 * wv;
        const v4f64 c = fb_mm16(Sa, o + 16, o + 16, St, o + 16, o, 4, l15, lk);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) Sa[(o + 16 + lk + 4 * reg) * FLD + o + l15] = -c[reg];
    }
    __syncthreads();
    {                 // block size 32:  T = L21 X11, one 16 x 16 piece per wave
        const int ti = wv >> 1, tj = wv & 1;
        const v4f64 c = fb_mm16(Sb, 32 + 16 * ti, 0, Sa, 0, 16 * tj, 8, l15, lk);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) St[(32 + 16 * ti + lk + 4 * reg) * FLD + 16 * tj + l15] = c[reg];
    }
    __syncthreads();
    {                 // X21 = -X22 T
        const int ti = wv >> 1, tj = wv & 1;
        const v4f64 c = fb_mm16(Sa, 32 + 16 * ti, 32, St, 32, 16 * tj, 8, l15, lk);
        __syncthreads();                                  // every wave has read the X22 rows it needs before X21 is written
#pragma unroll
        for (int reg = 0; reg < 4; reg++) Sa[(32 + 16 * ti + lk + 4 * reg) * FLD + 16 * tj + l15] = -c[reg];
    }
    __syncthreads();
    FB_T(9);
    {
        // Minv[k][c] = W[c][k] / d_c  (upper triangular), row-major in the scratch area, then the pivots
        double *mv = scratch + (int64_t)i * 4160;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = tid + 256 * q, k = idx >> 5, c = 2 * (idx & 31);
            fb_st16(mv + 2 * idx, c >= k ? Sa[c * FLD + k] * Sd[c] : 0.0, c + 1 >= k ? Sa[(c + 1) * FLD + k] * Sd[c + 1] : 0.0);
        }
        if (tid < 64) fb_st(mv + 4096 + tid, dkeep);
        fb_publish(fl_minv + i);
    }
    FB_T(10);
    // ---- off the critical path: the factored block for the solves' explicit inverses, D and 1/D (exact division), the last X_j
    {
        double *ld = P.Ldiag + fp[i].diag_off;
        for (int idx = tid; idx < 4096; idx += 256) {
            const int r_ = idx & 63, k = idx >> 6;
            ld[idx] = r_ > k ? Sb[r_ * FLD + k] : (r_ == k ? 1.0 : 0.0);
        }
        if (tid < 64) {
            P.D[f + tid] = dkeep;
            P.Dinv[f + tid] = Sd[tid];
            if (!isfinite(Sd[tid])) atomicOr(P.flags + FL_NONFINITE, 1);
        }
        if (lane == 0 && nreg) atomicAdd(P.flags + FL_NREG, nreg);
    }
    if (i > 0) {
        __syncthreads();
        fb_tile_load(ltiles + (int64_t)(i * (i - 1) / 2 + i - 1) * 4096, Sa, tid);   // this workgroup's own L(i, i-1)
        __syncthreads();
        fb_store_panel(P, fp[i - 1], 64, nr, Sa, tid);
    }
    FB_T(11);
}

void launch_front_block(hipStream_t st, const DevPlan &P, const FrontBatch &B, int *sync_all, double *scratch_all, double *stream_all,
                        double dyn_eps, double dyn_delta, bool streamed, long long *trace) {
    if (B.i_end <= B.i_base) return;
    const dim3 grid(B.i_end - B.i_base + (B.x_count + 4 * std::max(B.pad, 1) - 1) / (4 * std::max(B.pad, 1)));
    if (streamed) hipLaunchKernelGGL(k_front_block<true>, grid, dim3(256), 0, st, P, B, sync_all, scratch_all, stream_all, dyn_eps, dyn_delta, trace);
    else hipLaunchKernelGGL(k_front_block<false>, grid, dim3(256), 0, st, P, B, sync_all, scratch_all, stream_all, dyn_eps, dyn_delta, trace);
}

}  // namespace hipkkt
