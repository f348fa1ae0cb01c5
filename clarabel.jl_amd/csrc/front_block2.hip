// K4 over a FRONT, one launch per update batch -- second form of the kernel (round 5; the first form is front_block.hip, kept behind
// HIPKKT_FB_V2=0 for comparison).  Same contract, same work split (workgroup b owns the b-th 64-row block of the batch's first panel,
// the first nb workgroups are the "diagonal" workgroups whose tiles form the critical chain), same hand-off protocol (write-through
// payload, flag / NaN sentinel, bounded spins that abort through FL_FACFAIL) -- but every tile lives TRANSPOSED in the FP64
// matrix-core accumulators: lane (l15, lk) of wavefront w holds the entries (row 16 w + l15, column 16 sub + lk + 4 reg) of a tile.
// With D[m][n] += A[m][k] B[k][n], lane (l15, lk): A[l15][lk], B[lk][l15], D[lk + 4 reg][l15], a product  C^T = S^T R^T  takes its
// B operand -- the workgroup's own rows R -- STRAIGHT FROM THE ACCUMULATOR REGISTERS (k-step kk <-> sub = kk / 4, reg = kk % 4) and
// its A operand -- data shared by all rows: the inverse of a diagonal block, the rows of a diagonal workgroup above, a streamed block
// of pivots -- in "operand order" (one coalesced 512-byte load per k-step and 16-row strip).  Consequences:
//   * a workgroup's own tiles are never transposed through LDS: they are loaded from / stored to the panels directly in the
//     accumulator layout.  The workgroups BELOW the diagonal stage only the shared operand of a step through LDS (all 256 threads
//     fetch 16 values each in one round trip -- per-wave loads straight into registers were four times the traffic and, as relaxed
//     atomic loads next to their use, a memory round trip per k-step);
//   * the DIAGONAL workgroups take EVERY panel left of their tile from the stream of 8-pivot records (the later readers trail the
//     first): per record 2 + <= 8 (+ 8 for the panel next to the own tile) matrix-core instructions, all operands in registers or
//     prefetched: l = p T with T = L_bb^-T D_b^-1 of the block's 8 x 8 diagonal part (formed by an idle wavefront of the producer),
//     then the rank-8 updates.  No step of the chain waits for an explicit 64 x 64 inverse (round 4: the chain waited 4 - 5 us for
//     it, then 14 us for the step with it); the successor finishes its last record 3 - 6 us behind the producer's last pivot block;
//   * no global store inside the record loops (one in-order memory counter on gfx9: a store there made every record wait for a
//     write-through acknowledgement), finished rows stay in the tiles' registers until the end.
// LDS carries only what crosses wavefronts: the shared operand of a regular step, the diagonal tile's update (X D / l d of all 64
// rows), the pivot loop (block columns -> wave 0, its l / raw columns -> everybody) and the blocked inverse of L11.
// Measured: 121 -> 94 us per launch of 5 panels, chain of diagonal tiles 20.4 -> 15.9 us per panel.
//
// Scratch layouts of a batch (private to this kernel; same sizes as the first form):
//   operand order of a 64 x 64 tile M, value M(mu, kappa):  offset (16 (mu / 16) + kappa / 4) * 64 + 16 (kappa % 4) + mu % 16
//   scratch + j * 4160            Minv_j^T in operand order, i.e. value (c, k) = (L_jj^-T D_j^-1)[k][c]; then the 64 pivots d
//   ltiles + tri(k, j) * 4096     L(k,j) D_j in operand order (mu = row of block k, kappa = column of panel j)
//   stream + (8 j + Bk) * kFbRec  record of pivot block Bk of tile j, 12 chunks of 64 lanes: chunks 2 q + e (q < 4, e < 2): raw column
//                                 a[c = 16 q + l15][k' = 4 e + lk] = d_k' l_ck' of the tile's rows; 8 + e: T[4 e + lk][l15] (0 for
//                                 l15 >= 8); 10 + e: d[4 e + lk]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels.h"
#include "dense_tile.h"

namespace hipkkt {
namespace {

constexpr int LDT = 65;   // LDS row stride of the 64 x 64 tiles of the inverse
constexpr int CS = 68;    // row stride of the pivot loop's column buffers (the four k-rows a lane quad reads sit in different banks)

__device__ __forceinline__ int f2_ldi(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double f2_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void f2_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double f2_readlane(double x, int l) {   // l wave-uniform
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double f2_rcp3(double d) {   // front_block.hip fb_rcp3
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    const double e2 = fma(e, e, e);
    return fma(r, e2, r);
}
// LDS-only workgroup barrier (front_block.hip fb_bar: __syncthreads() would also wait for the write-through stores in flight)
__device__ __forceinline__ void f2_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// front_block.hip fb_mfma_settle: wait states between matrix-core instructions and an exec-masked block / a loop back-edge
__device__ __forceinline__ void f2_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// "not written yet" in a stream record = all ones (k_fb_reset, front_block.hip): a NaN no arithmetic produces.  A record is complete
// when the unsigned maximum of the high words of its values is below 0xFFFFFFFF (two v_max3_u32 per six values).
__device__ __forceinline__ unsigned f2_hi(double v) { return (unsigned)__double2hiint(v); }
__device__ __forceinline__ unsigned f2_max3(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }
// uniform base + 32-bit lane offset (+ immediate): one instruction per access, no 64-bit address arithmetic on the vector side
__device__ __forceinline__ double f2_ldo(const double *base, unsigned byte_off) {
    return __hip_atomic_load((const HK_GLOBAL double *)((const HK_GLOBAL char *)base + byte_off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void f2_sto(double *base, unsigned byte_off, double v) {
    __hip_atomic_store((HK_GLOBAL double *)((HK_GLOBAL char *)base + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave-level wait for a counter of another workgroup: every lane polls the same word and the value is made wave-uniform, so the
// loop is a scalar branch (no exec-masked block next to matrix-core code).  false = timed out / somebody else failed.
__device__ __forceinline__ bool f2_wait(const int *flag, int want, int *err, int *failflag, unsigned lim) {
    for (unsigned spins = 0;; spins++) {
        if (__builtin_amdgcn_readfirstlane(f2_ldi(flag)) >= want) return true;
        if ((spins & 63u) == 63u || lim < 64u) {
            if (spins > lim) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((threadIdx.x & 63) == 0) atomicOr(failflag, 1);
                return false;
            }
            if (__builtin_amdgcn_readfirstlane(f2_ldi(err)) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// this wavefront's write-through stores have drained: count it in (consumers wait for all four wavefronts)
__device__ __forceinline__ void f2_wave_done(int *flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void f2_extra_tiles(const DevPlan &P, int begin, int count, int xb, int per_wave) {   // front_block.hip fb_extra_tiles
    const int lane = threadIdx.x & 63;
    int idx = rfl((xb * 4 + (int)(threadIdx.x >> 6)) * per_wave);
    for (int q = 0; q < per_wave && idx < count; q++, idx++) dense_tile<4, 4, true>(P, P.dgroups + begin + idx, lane, 0, 0);
}

#define F2_T(slot) do { if (TRACE) { if (trace && tid == 0 && i < 5) trace[(B.sync_off / 128 * 8 + i) * 16 + (slot)] = (long long)wall_clock64(); } } while (0)
#define F2_TB(ph) do { if (TRACE) { if (trace && tid == 0 && i == 0) trace[(B.sync_off / 128 * 8 + 5) * 16 + 4 * Bk + (ph)] = (long long)clock64(); } } while (0)

#ifndef FB_EARLY_STORE
#define FB_EARLY_STORE 1
#endif
template <bool TRACE>
__global__ void __launch_bounds__(256)
k_front_block2(DevPlan P, FrontBatch B, int *sync_all, double *scratch_all, double *stream_all, double dyn_eps, double dyn_delta,
               long long *trace) {
    // S0: regular steps: two buffers [4 waves][16 k-steps][64 lanes] of X D (diagonal workgroups); streamed step: the l d exchange at
    //     S0 + 4096; pivot loop: Pc, colL, cC (two buffers), dsave, dinvs at S0; inverse: X at S0, products at S0 + 4160
    __shared__ double S0[8320];
    __shared__ double S1[64 * LDT];     // L11 of the diagonal tile (unit lower, row-major stride LDT)
    __shared__ double Sd[64];
    __shared__ int sblk;
    int *sync = sync_all + B.sync_off;
    int *err = sync + 1, *fl_minv = sync + 32, *fl_L = sync + 64;
    double *scratch = scratch_all + B.scratch_off;
    double *ltiles = scratch + (int64_t)kFbMax * 4160;
    double *stream = stream_all + B.stream_off;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lk = lane >> 4;
    const int wv = rfl(tid >> 6);
    if ((int)blockIdx.x >= B.i_end - B.i_base) {          // extra workgroup: tiles of the previous update stage (no hand-off, no LDS)
        f2_extra_tiles(P, B.x_begin, B.x_count, (int)blockIdx.x - (B.i_end - B.i_base), B.pad > 0 ? B.pad : 1);
        return;
    }
    if (tid == 0) sblk = atomicAdd(sync + B.tick, 1);
    __syncthreads();
    const int i = rfl(B.i_base + sblk);                   // row block of this workgroup
    if (i >= B.i_end) return;
    F2_T(0);
    const FrontPanel *fp = P.front_panels + B.fp_off;
    const int nb = B.nb;
    // the two kinds of workgroup are compiled separately (diagonal: streamed steps, then the pivots; below the diagonal: nb - 1 steps
    // with the published inverses, the last panel from the stream, nothing else) -- one body with run-time flags spills registers
    auto body = [&](auto diag_tag) {
    constexpr bool diag = decltype(diag_tag)::value;
    const int ncb = diag ? i + 1 : nb;                    // column blocks held here
    const int nsteps = diag ? 0 : nb - 1;                 // regular steps (with the published inverses): the workgroups below the diagonal,
                                                          // all panels but the batch's last (taken from the stream: the launch ends sooner);
                                                          // a diagonal workgroup takes every panel left of its tile from the stream
    const int nr = min(64, B.r0 - 64 * i);
    const int n = 16 * wv + l15;                          // this lane's row of the block
    const bool rowok = n < nr;
    const int f = diag ? fp[i].f : 0;
    const signed char sgn_l = diag ? P.sgn_perm[f + lane] : (signed char)0;
    int *failflag = P.flags + FL_FACFAIL;
    const unsigned lane8 = 8u * (unsigned)lane;
    const unsigned lim = P.spin_limit;

    // ---- the row block, straight into the accumulator layout (16 loads per tile and lane, 128-byte segments)
    v4f64 acc[kFbMax][4];
#pragma unroll
    for (int k = 0; k < kFbMax; k++) {
        if (k < ncb) {
            const FrontPanel pk = fp[k];
            // (rows past the front's last one: the load goes to the last valid row and is zeroed afterwards -- a conditional load
            // would put every one of the 16 loads into an exec-masked block of its own with a wait in front of it)
            const double *src = P.Lx + pk.panel_off + 64 * (i - k) + min(n, nr - 1);
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) acc[k][sub][reg] = src[(int64_t)(16 * sub + lk + 4 * reg) * pk.r];
            if (nr < 64) {
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) acc[k][sub][reg] = rowok ? acc[k][sub][reg] : 0.0;
            }
        }                                                 // (tiles k >= ncb are never touched: every use below is guarded the same way)
    }
    // a diagonal workgroup's own tile lives in registers of its own from the start (it is updated in every step)
    v4f64 tr[4];
#pragma unroll
    for (int sub = 0; sub < 4; sub++) {
        tr[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < kFbMax; k++)
            if (k == i) tr[sub] = acc[k][sub];
    }
    F2_T(1);

    // ---- consumer of the streamed pivot chain: the eight records of tile jt turn this workgroup's tile t of panel jt into its rows of
    //      L (l = p T per block of 8 columns, then the rank-8 update of the columns right of it), wave by wave, everything in
    //      registers.  WITH_TR (the panel next to the own tile, jt = i - 1): every record also updates the diagonal tile (l d of all
    //      64 rows through LDS, one barrier per record) and goes out as L(i, jt) D for the workgroups below.  Without (every other reader):
    //      no LDS, no barrier; the pivots of the lane's columns are returned in dcol for the step that follows.
    //      Ring of three records in registers: while record Bk is processed, records Bk + 1 and Bk + 2 are in flight (a fourth
    //      slot changed nothing: the loop is bound by its 18 FP64 matrix-core instructions of 64 cycles each per record) (a consumer that
    //      has fallen behind finds them complete and pays no memory round trip per record; one that is level with the producer sees
    //      the sentinel and polls).  Fully unrolled: the strip / register of a block's columns is static.  The loop is instruction-
    //      issue bound (one wavefront issues every ~5 cycles): uniform base + lane offset addressing, one negation per record (the B
    //      operand), the freshness test by unsigned maxima, the panel stores elsewhere.
    auto consume = [&](auto with_tr_tag, int jt, v4f64 (&t)[4], double (&dcol)[4][4]) -> bool {
        constexpr bool WITH_TR = decltype(with_tr_tag)::value;
        const double *rec0 = stream + (int64_t)jt * 8 * kFbRec;
        double *ldx = S0 + 4096;                          // [2 buffers][4 waves][2][64 lanes]
        double rc[3][4][2], rt[3][2], rd[3][2];
        double pn0 = 0.0, pn1 = 0.0;                      // -l of the record before (WITH_TR: its diagonal-tile update trails by one record)
#define F2_REQUEST(slot, Bq) do { \
            const double *rq_ = rec0 + (int64_t)(Bq) * kFbRec; \
            _Pragma("unroll") for (int q = (Bq) >> 1; q < 4; q++) { \
                rc[slot][q][0] = f2_ldo(rq_, lane8 + (q * 2) * 512); rc[slot][q][1] = f2_ldo(rq_, lane8 + (q * 2 + 1) * 512); } \
            rt[slot][0] = f2_ldo(rq_ + 512, lane8); rt[slot][1] = f2_ldo(rq_ + 512, lane8 + 512); \
            rd[slot][0] = f2_ldo(rq_ + 512, lane8 + 1024); rd[slot][1] = f2_ldo(rq_ + 512, lane8 + 1536); } while (0)
        F2_REQUEST(0, 0); F2_REQUEST(1, 1); F2_REQUEST(2, 2);
#pragma unroll
        for (int Bk = 0; Bk < 8; Bk++) {
            const int sb = Bk >> 1, par = Bk & 1, sl = Bk % 3;
            for (unsigned spins = 0;; spins++) {               // level with the producer: poll (bounded)
                unsigned m = f2_max3(f2_hi(rt[sl][0]), f2_hi(rt[sl][1]), f2_hi(rd[sl][0]));
                m = max(m, f2_hi(rd[sl][1]));
#pragma unroll
                for (int q = sb; q < 4; q++) m = f2_max3(m, f2_hi(rc[sl][q][0]), f2_hi(rc[sl][q][1]));
                if (__builtin_amdgcn_readfirstlane((int)(__ballot(m != 0xFFFFFFFFu) == ~0ull)) != 0) break;
                if ((spins & 63u) == 63u || lim < 64u) {
                    if (spins > lim) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (lane == 0) atomicOr(failflag, 1);
                        return false;
                    }
                    if (__builtin_amdgcn_readfirstlane(f2_ldi(err)) != 0) return false;
                }
                __builtin_amdgcn_s_sleep(1);
                F2_REQUEST(sl, Bk);
            }
            if (WITH_TR && Bk == 0) F2_T(3);
            double cr[4][2], tv[2], dv[2];
#pragma unroll
            for (int q = sb; q < 4; q++) { cr[q][0] = rc[sl][q][0]; cr[q][1] = rc[sl][q][1]; }
            tv[0] = rt[sl][0]; tv[1] = rt[sl][1]; dv[0] = rd[sl][0]; dv[1] = rd[sl][1];
            if (Bk + 3 < 8) F2_REQUEST(sl, Bk + 3);
            // l^T = T^T p^T: the block's 8 columns of the strip are the B operand as they stand
            v4f64 lT = {0.0, 0.0, 0.0, 0.0};
            lT = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[0], t[sb][2 * par], lT, 0, 0, 0);
            lT = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[1], t[sb][2 * par + 1], lT, 0, 0, 0);
            if (WITH_TR && Bk > 0) {
                // while l of this record is on its way through the matrix core: the diagonal tile's update by the record BEFORE
                // (its l d of all 64 rows went to LDS in the previous iteration; the barrier closes that exchange).  The strips right
                // of a wave's own rows are the upper triangle, but the slowest wave -- the last, all four strips -- sets the pace at
                // the barrier: no per-wave branches here.
                f2_bar();
                const double *lp = ldx + ((Bk - 1) & 1) * 512;
                double la[4][2];
#pragma unroll
                for (int so = 0; so < 4; so++) { la[so][0] = lp[(so * 2) * 64 + lane]; la[so][1] = lp[(so * 2 + 1) * 64 + lane]; }
#pragma unroll
                for (int so = 0; so < 4; so++) {
                    tr[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[so][0], pn0, tr[so], 0, 0, 0);
                    tr[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[so][1], pn1, tr[so], 0, 0, 0);
                }
            }
            const double l0 = lT[0], l1 = lT[1], nl0 = -l0, nl1 = -l1;
            if (WITH_TR) {
                // l d for the other wavefronts (diagonal tile, next iteration) and for the workgroups below
                const double ld0 = l0 * dv[0], ld1 = l1 * dv[1];
                double *lx = ldx + par * 512;
                lx[(wv * 2) * 64 + lane] = ld0;
                lx[(wv * 2 + 1) * 64 + lane] = ld1;
                pn0 = nl0; pn1 = nl1;
            }
            // (no global store in this loop: on gfx9 loads and stores share one in-order counter, and a write-through store takes
            // a microsecond to be acknowledged -- a wave that stores here waits for its stores whenever it waits for a record)
            dcol[sb][2 * par] = dv[0];
            dcol[sb][2 * par + 1] = dv[1];
            // the finished columns stay in the tile's registers; the record's raw columns are zero on the rows of its own block, so
            // the update of the block's own strip leaves them alone
            t[sb][2 * par] = l0;
            t[sb][2 * par + 1] = l1;
            // rank-8 update of the columns right of the block, the strip of the NEXT block's columns first (the block's own strip:
            // only when its second half is still to come)
#pragma unroll
            for (int q = sb + par; q < 4; q++) {
                t[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cr[q][0], nl0, t[q], 0, 0, 0);
                t[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cr[q][1], nl1, t[q], 0, 0, 0);
            }
        }
        if (WITH_TR) {                                    // the diagonal tile's update by the last record
            f2_bar();
            const double *lp = ldx + 512;
            double la[4][2];
#pragma unroll
            for (int so = 0; so < 4; so++) { la[so][0] = lp[(so * 2) * 64 + lane]; la[so][1] = lp[(so * 2 + 1) * 64 + lane]; }
#pragma unroll
            for (int so = 0; so < 4; so++) {
                tr[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[so][0], pn0, tr[so], 0, 0, 0);
                tr[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[so][1], pn1, tr[so], 0, 0, 0);
            }
        }
#undef F2_REQUEST
        return true;
    };

    // ---- the panels left of this block: X_j^T = Minv_j^T A_j^T, then A_k^T -= (L(k,j) D_j) X_j^T for the tiles right of it
    double *STG = S1;                                     // staging tile of the shared operand (S1 holds L11 only from the pivots on)
    const unsigned tid8 = 8u * (unsigned)tid;
    auto stage_tile = [&](const double *tile) {           // 64 x 64 tile in operand order, global -> LDS, all four waves
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = f2_ldo(tile + (q >> 1) * 512, tid8 + (q & 1) * 2048);
        f2_bar();                                         // everybody is done with the tile before
#pragma unroll
        for (int q = 0; q < 16; q++) STG[tid + 256 * q] = v[q];
        f2_bar();
    };
#pragma unroll
    for (int j = 0; j < (diag ? 0 : kFbMax); j++) {
        if (j < nsteps) {                                 // workgroup-uniform (only the workgroups below the diagonal run these steps)
            const FrontPanel pj = fp[j];
            const double *mv = scratch + (int64_t)j * 4160;
            v4f64 x[4];
            double dj[4][4];                              // pivots of this lane's columns m = 16 sub + lk + 4 reg (diagonal workgroups)
            if (!f2_wait(fl_minv + j, 1, err, failflag, lim)) return;
            if (diag) {
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) dj[sub][reg] = f2_ld(mv + 4096 + 16 * sub + lk + 4 * reg);
            }
            // The A operand -- the same for all four waves -- comes through LDS: every thread fetches 16 values of the tile (all in
            // flight at once, one memory round trip), the tile is laid down in operand order, each wave reads its k-steps back
            // (lane-contiguous: no bank conflicts).  [Direct loads per wave, first version of this kernel: four times the traffic, and
            // a relaxed atomic load next to its use makes every k-step a round trip -- 30 us per step.]
            stage_tile(mv);
#pragma unroll
            for (int so = 0; so < 4; so++) {              // Minv is upper triangular: k <= c, 4 so + 4 k-steps per strip
                double am[16];
#pragma unroll
                for (int kk = 0; kk < 4 * so + 4; kk++) am[kk] = STG[(so * 16 + kk) * 64 + lane];
                x[so] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4 * so + 4; kk++)
                    x[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[kk], acc[j][kk >> 2][kk & 3], x[so], 0, 0, 0);
            }
            if (diag) {
                // L(i,j) D_j for the workgroups below (operand order: this lane's entry is exactly its own slot) and, through LDS, for
                // the other wavefronts of this workgroup
                double *lt = ltiles + (int64_t)(i * (i - 1) / 2 + j) * 4096 + (wv * 16) * 64;
                double *xd = S0 + (j & 1) * 4096;
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const double v = x[sub][reg] * dj[sub][reg];
                        f2_sto(lt + (4 * sub + (reg & 2)) * 64, lane8 + (reg & 1) * 512, v);
                        xd[(wv * 16 + 4 * sub + reg) * 64 + lane] = v;
                    }
            }
#pragma unroll
            for (int so = 0; so < 4; so++) x[so] = -x[so];    // from here on X is only the B operand of updates:  A -= (L D) X^T
            if (diag) {
                // the diagonal tile  T_ii -= (X D) X^T
                const double *xd = S0 + (j & 1) * 4096;
                f2_bar();
#pragma unroll
                for (int so = 0; so < 4; so++) {
                    double xa[16];
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) xa[kk] = xd[(so * 16 + kk) * 64 + lane];
#pragma unroll
                    for (int kk = 0; kk < 16; kk++)
                        tr[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[kk], x[kk >> 2][kk & 3], tr[so], 0, 0, 0);
                }
                f2_settle();
                f2_wave_done(fl_L + 8 * i + j);
            }
            // the panel (column-major) and its row-major copy for the backward solves -- after the hand-off: the flag waits for this
            // wave's outstanding stores, and these 32 strided ones are slow to drain (x holds -X by now)
            f2_settle();
            if (rowok) {
                double *dst = P.Lx + pj.panel_off + 64 * (i - j) + n;
                double *lt2 = P.LT + pj.lt_off + (int64_t)(64 * (i - j) - 64 + n) * 64;
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        dst[(int64_t)(16 * sub + lk + 4 * reg) * pj.r] = -x[sub][reg];
                        lt2[16 * sub + lk + 4 * reg] = -x[sub][reg];
                    }
            }
            // the tiles that need the rows of the diagonal workgroups above
#pragma unroll
            for (int k = j + 1; k < kFbMax; k++) {
                if (k < ncb && !(diag && k == i)) {
                    if (!f2_wait(fl_L + 8 * k + j, 4, err, failflag, lim)) return;
                    const double *lt = ltiles + (int64_t)(k * (k - 1) / 2 + j) * 4096;
                    stage_tile(lt);
#pragma unroll
                    for (int so = 0; so < 4; so++) {
                        double al[16];
#pragma unroll
                        for (int kk = 0; kk < 16; kk++) al[kk] = STG[(so * 16 + kk) * 64 + lane];
#pragma unroll
                        for (int kk = 0; kk < 16; kk++)
                            acc[k][so] = __builtin_amdgcn_mfma_f64_16x16x4f64(al[kk], x[kk >> 2][kk & 3], acc[k][so], 0, 0, 0);
                    }
                }
            }
        }
    }
    F2_T(2);
    if constexpr (!diag) {
        // the batch's last panel: behind its pivot blocks instead of behind its inverse (published 4 - 5 us after the last pivot, then a
        // staged product): every workgroup below the diagonal ends ~6 us earlier, and with it the launch
        v4f64 x[4];
#pragma unroll
        for (int sub = 0; sub < 4; sub++) {
            x[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < kFbMax; k++)
                if (k == nb - 1) x[sub] = acc[k][sub];
        }
        double dj[4][4];
        if (!consume(std::false_type{}, nb - 1, x, dj)) return;
        f2_settle();
        if (rowok) {
            const FrontPanel pj = fp[nb - 1];
            double *dst = P.Lx + pj.panel_off + 64 * (i - (nb - 1)) + n;
            double *lt2 = P.LT + pj.lt_off + (int64_t)(64 * (i - (nb - 1)) - 64 + n) * 64;
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    dst[(int64_t)(16 * sub + lk + 4 * reg) * pj.r] = x[sub][reg];
                    lt2[16 * sub + lk + 4 * reg] = x[sub][reg];
                }
        }
        return;
    } else {

    // ---- a diagonal workgroup's panels j2 < i - 1 from the stream (a later reader of tile j2's records, behind the workgroups i' < i):
    //      no step waits for the inverse of a tile, which is published 4 - 5 us after its pivots.  Then the step's hand-off and
    //      updates as in the regular steps.  One copy of the code, rolled over j2; the tiles are selected by run-time index.
    const bool early_store = FB_EARLY_STORE && i == nb - 1 && B.nblk > nb;
#pragma unroll 1
    for (int j2 = 0; j2 < i - 1; j2++) {
        v4f64 x[4];
#pragma unroll
        for (int sub = 0; sub < 4; sub++) {
            x[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < kFbMax; k++)
                if (k == j2) x[sub] = acc[k][sub];
        }
        double dj[4][4];
        if (!consume(std::false_type{}, j2, x, dj)) return;
        if (j2 == i - 2) F2_T(9);
        {
            double *lt = ltiles + (int64_t)(i * (i - 1) / 2 + j2) * 4096 + (wv * 16) * 64;
            double *xd = S0 + (j2 & 1) * 4096;
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const double v = x[sub][reg] * dj[sub][reg];
                    f2_sto(lt + (4 * sub + (reg & 2)) * 64, lane8 + (reg & 1) * 512, v);
                    xd[(wv * 16 + 4 * sub + reg) * 64 + lane] = v;
                }
#pragma unroll
            for (int so = 0; so < 4; so++) x[so] = -x[so];
            f2_bar();
#pragma unroll
            for (int so = 0; so < 4; so++) {
                double xa[16];
#pragma unroll
                for (int kk = 0; kk < 16; kk++) xa[kk] = xd[(so * 16 + kk) * 64 + lane];
#pragma unroll
                for (int kk = 0; kk < 16; kk++)
                    tr[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[kk], x[kk >> 2][kk & 3], tr[so], 0, 0, 0);
            }
            f2_settle();
            if (j2 == i - 2) F2_T(11);
            f2_wave_done(fl_L + 8 * i + j2);
        }
        // the tiles between panel j2 and the own one need L(k, j2) D of the diagonal workgroups above (raised after THEIR step j2)
#pragma unroll 1
        for (int k = j2 + 1; k < i; k++) {
            if (!f2_wait(fl_L + 8 * k + j2, 4, err, failflag, lim)) return;
            if (j2 == i - 2) F2_T(13);
            stage_tile(ltiles + (int64_t)(k * (k - 1) / 2 + j2) * 4096);
            v4f64 t[4];
#pragma unroll
            for (int sub = 0; sub < 4; sub++) {
                t[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < kFbMax; kk++)
                    if (kk == k) t[sub] = acc[kk][sub];
            }
#pragma unroll
            for (int so = 0; so < 4; so++) {
                double al[16];
#pragma unroll
                for (int kk = 0; kk < 16; kk++) al[kk] = STG[(so * 16 + kk) * 64 + lane];
#pragma unroll
                for (int kk = 0; kk < 16; kk++)
                    t[so] = __builtin_amdgcn_mfma_f64_16x16x4f64(al[kk], x[kk >> 2][kk & 3], t[so], 0, 0, 0);
            }
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int kk = 0; kk < kFbMax; kk++)
                    if (kk == k) acc[kk][sub] = t[sub];
        }
        // the finished rows stay in the tile's registers (x holds -X); they go to the panel at the very end -- a store here would
        // sit in front of the next step's record loads on the memory counter
#pragma unroll
        for (int sub = 0; sub < 4; sub++)
#pragma unroll
            for (int kk = 0; kk < kFbMax; kk++)
                if (kk == j2) acc[kk][sub] = -x[sub];
        // ... except in the batch's LAST diagonal workgroup (round 6): the launch ends with ITS panel stores (four tiles, ~10 us after
        // its last pivot), and it spends most of the launch waiting for the workgroups before it -- its stores of a finished panel
        // wait in front of flag polls that are not due yet
        if (early_store && rowok) {
            const FrontPanel pk = fp[j2];
            double *dst = P.Lx + pk.panel_off + 64 * (i - j2);
            double *lt2 = P.LT + pk.lt_off + (int64_t)(64 * (i - j2) - 64) * 64;
            const unsigned o1 = 8u * (unsigned)(n + lk * pk.r), o2 = 8u * (unsigned)(n * 64 + lk);
#pragma unroll
            for (int sub = 0; sub < 4; sub++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const double v = -x[sub][reg];
                    st_off(dst + (int64_t)(16 * sub + 4 * reg) * pk.r, o1, v);
                    st_off(lt2, o2 + 8u * (unsigned)(16 * sub + 4 * reg), v);
                }
        }
    }
    v4f64 xr[4];                                          // the tile of panel i - 1, L(i, i-1) after the streamed step
#pragma unroll
    for (int sub = 0; sub < 4; sub++) {
        xr[sub] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < kFbMax; k++)
            if (k == i - 1) xr[sub] = acc[k][sub];
    }
    // ---- streamed step: this workgroup's rows of panel i - 1, record by record behind the workgroup that eliminates tile i - 1.
    if (i > 0) {
        double dl[4][4];
        f2_bar();                                         // the X D buffers of the steps before are free
        if (!consume(std::true_type{}, i - 1, xr, dl)) return;
        // L(i, i-1) D for the workgroups below (operand order); the flag follows during the first pivot block
        double *lt = ltiles + (int64_t)(i * (i - 1) / 2 + i - 1) * 4096 + (wv * 16) * 64;
#pragma unroll
        for (int sub = 0; sub < 4; sub++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) f2_sto(lt + (4 * sub + (reg & 2)) * 64, lane8 + (reg & 1) * 512, xr[sub][reg] * dl[sub][reg]);
        f2_settle();
        F2_T(4);
    }

    // ---- the diagonal tile.  Per block of 8 pivots: every wave hands the block's 8 columns of its rows to wave 0 through LDS, wave 0
    //      eliminates them without leaving the wavefront (lane = row; the pivot rule and arithmetic of k_factor_panel / front_block.hip),
    //      and the rank-8 update goes to the matrix core -- split in two: behind the elimination only the strip that holds the NEXT
    //      block's columns is updated (2 instructions per wave, then the columns go back to wave 0); the other live strips are
    //      updated by waves 1 - 3 WHILE wave 0 eliminates the next block (they would idle there), followed by the record of the
    //      finished block for the next diagonal workgroup (wave 1 forms T of that block first).  A wave's rows need the strips up to
    //      its own only (the rest is the upper triangle), so wave 0 takes part in the update of block 0 alone.
    double *Pc = S0;                                      // [64][9]   the block's columns, by row
    double *colLa = S0 + 576, *colLb = colLa + 8 * CS;    // [8][CS]   l_ik, two buffers
    double *cCa = colLb + 8 * CS, *cCb = cCa + 8 * CS;    // [8][CS]   raw a_ik = d_k l_ik, two buffers
    double *dsave = cCb + 8 * CS, *dinvs = dsave + 64;    // [64] d_k, [64] 1 / d_k as used on the chain
    double *L11 = S1;
    const unsigned long long spos = __ballot(sgn_l > 0);
    const double dyn_delta_inv = 1.0 / dyn_delta;
    int nreg = 0;
    const bool pub = i + 1 < nb || B.nblk > nb;          // (the last tile's records are read by the workgroups below the diagonal)
    auto publish = [&](int Bp) {                          // waves 1 - 3: record of block Bp for the next diagonal workgroup
        const double *cP = (Bp & 1) ? cCb : cCa;
        double *rec = stream + ((int64_t)i * 8 + Bp) * kFbRec;
        if (wv == 1) {
            // row kq of X = L_bb^-1 by back substitution (x_j = -sum_{m > j} x_m L_mj for j < kq), every lane the full recurrence
            // with uniform LDS reads; T[k][k'] = X[k'][k] / d_k'
            const int kq = l15 & 7, o = 8 * Bp;
            double xs[8];
#pragma unroll
            for (int m = 0; m < 8; m++) xs[m] = m == kq ? 1.0 : 0.0;
#pragma unroll
            for (int j = 6; j >= 0; j--) {
                double v = 0.0;
#pragma unroll
                for (int m = j + 1; m < 8; m++) v = fma(-xs[m], L11[(o + m) * LDT + o + j], v);
                xs[j] = j < kq ? v : xs[j];
            }
            const double di = dinvs[o + kq];
            const double s0 = lk == 0 ? xs[0] : (lk == 1 ? xs[1] : (lk == 2 ? xs[2] : xs[3]));
            const double s1 = lk == 0 ? xs[4] : (lk == 1 ? xs[5] : (lk == 2 ? xs[6] : xs[7]));
            f2_sto(rec + 512, lane8, l15 < 8 ? s0 * di : 0.0);
            f2_sto(rec + 512, lane8 + 512, l15 < 8 ? s1 * di : 0.0);
            f2_sto(rec + 512, lane8 + 1024, dsave[o + lk]);
            f2_sto(rec + 512, lane8 + 1536, dsave[o + 4 + lk]);
        } else if (wv >= 2) {
            // the raw columns, zero on the rows up to the block's own (the consumer keeps its finished columns in the same registers)
            const int q0 = 2 * (wv - 2);
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const int c = 16 * (q0 + q) + l15;
                    f2_sto(rec, lane8 + ((q0 + q) * 2 + e) * 512, c > 8 * Bp + 7 ? cP[(4 * e + lk) * CS + c] : 0.0);
                }
        }
    };
    // wave 0's elimination of one block.  RULE = false: the dynamic-regularisation rule (D_k s_k < eps => D_k = delta s_k) is only
    // TESTED, at the end of the block from the eight pivots (wave 0 is instruction-issue bound: selects, sign masks and the count
    // cost a quarter of its instructions); a block that needs it is repeated with RULE = true from the same input.  Same arithmetic
    // in both forms, so a factorisation without substituted pivots is bit-identical to one computed by the RULE form alone.
    auto eliminate = [&](auto rule_tag, int Bk, double *cL, double *cC) -> bool {
        constexpr bool RULE = decltype(rule_tag)::value;
        double pcol[8], dk[8], dik[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pcol[q] = Pc[lane * 9 + q];
        double akk = f2_readlane(pcol[0], 8 * Bk), csq = 0.0, dinv_prev = 0.0;
        int nr_ = 0;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const int k = 8 * Bk + kk;
            double d = fma(-csq, dinv_prev, akk);
            double dinv = f2_rcp3(d);
            if (RULE) {
                const int sm = ((spos >> k) & 1ull) ? 0 : (int)0x80000000;               // expected sign negative: test -d, substitute -delta
                const bool bad = __hiloint2double(__double2hiint(d) ^ sm, __double2loint(d)) < dyn_eps;
                double dsub = __hiloint2double(__double2hiint(dyn_delta) ^ sm, __double2loint(dyn_delta));
                double isub = __hiloint2double(__double2hiint(dyn_delta_inv) ^ sm, __double2loint(dyn_delta_inv));
                asm volatile("" : "+v"(dinv), "+v"(dsub), "+v"(isub));                   // all three in vector registers, unconditionally
                d = bad ? dsub : d;
                dinv = bad ? isub : dinv;
                nr_ += bad ? 1 : 0;
            }
            dk[kk] = d;
            dik[kk] = dinv;
            const double reg = pcol[kk];
            if (kk < 7) {
                akk = f2_readlane(pcol[kk + 1], k + 1);
                const double cn = f2_readlane(reg, k + 1);
                csq = cn * cn;
            }
            dinv_prev = dinv;
            const double li = reg * dinv;
            cL[kk * CS + lane] = li;
            cC[kk * CS + lane] = reg;
            L11[lane * LDT + k] = li;
#pragma unroll
            for (int jj = kk + 1; jj < 8; jj++) {
                const double cj = f2_readlane(reg, 8 * Bk + jj);
                pcol[jj] = fma(-li, cj, pcol[jj]);
            }
        }
        if (!RULE) {
            bool anybad = false;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int sm = ((spos >> (8 * Bk + kk)) & 1ull) ? 0 : (int)0x80000000;
                anybad = anybad || __hiloint2double(__double2hiint(dk[kk]) ^ sm, __double2loint(dk[kk])) < dyn_eps;
            }
            if (__builtin_amdgcn_readfirstlane((int)anybad)) return false;
        }
        nreg += nr_;
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 8; q++) { dsave[8 * Bk + q] = dk[q]; dinvs[8 * Bk + q] = dik[q]; }
        }
        return true;
    };
    // rank-8 update of strip q of this wave's rows with block Bq (buffers of its parity)
    auto update_strip = [&](int Bq, int q) {
        const double *cL = (Bq & 1) ? colLb : colLa, *cC = (Bq & 1) ? cCb : cCa;
        const double b0 = -cL[lk * CS + n], b1 = -cL[(4 + lk) * CS + n];
        tr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cC[lk * CS + 16 * q + l15], b0, tr[q], 0, 0, 0);
        tr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(cC[(4 + lk) * CS + 16 * q + l15], b1, tr[q], 0, 0, 0);
    };
    // this wave's rows of the panels left of the tile (kept in the tiles' registers since their steps): the panels (column-major) and
    // their row-major copies for the backward solves, after the pivots (stores during the second pivot block by the idle waves were
    // measured: they arrive late at the block's barrier, 11.6 -> 15.8 us per tile).
    auto store_panels = [&]() {
        if (!rowok) return;
#pragma unroll
        for (int k = 0; k < kFbMax - 1; k++) {
            if (k < i && !(early_store && k < i - 1)) {
                // uniform base + 32-bit lane offset per store (no 64-bit address arithmetic on the vector side: these 32 stores per
                // tile are the tail of the launch for the batch's last diagonal workgroup)
                const FrontPanel pk = fp[k];
                double *dst = P.Lx + pk.panel_off + 64 * (i - k);
                double *lt2 = P.LT + pk.lt_off + (int64_t)(64 * (i - k) - 64) * 64;
                const unsigned o1 = 8u * (unsigned)(n + lk * pk.r), o2 = 8u * (unsigned)(n * 64 + lk);
#pragma unroll
                for (int sub = 0; sub < 4; sub++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        const double v = k == i - 1 ? xr[sub][reg] : acc[k][sub][reg];
                        st_off(dst + (int64_t)(16 * sub + 4 * reg) * pk.r, o1, v);
                        st_off(lt2, o2 + 8u * (unsigned)(16 * sub + 4 * reg), v);
                    }
            }
        }
    };
    f2_bar();                                             // the l d buffers of the streamed step are free
    F2_T(5);
#pragma unroll
    for (int Bk = 0; Bk < 8; Bk++) {
        const int sb = Bk >> 1, par = Bk & 1;
        if (wv >= sb) {                                   // (the rows of the waves before are dead)
            Pc[n * 9 + lk] = tr[sb][2 * par];
            Pc[n * 9 + 4 + lk] = tr[sb][2 * par + 1];
        }
        f2_bar();
        F2_TB(0);
        if (wv == 0) {
            double *cL = par ? colLb : colLa, *cC = par ? cCb : cCa;
            if (!eliminate(std::false_type{}, Bk, cL, cC)) eliminate(std::true_type{}, Bk, cL, cC);
        } else {                                          // next to wave 0's elimination
            if (Bk == 0 && i > 0) f2_wave_done(fl_L + 8 * i + (i - 1));   // L(i, i-1) D of the streamed step: these waves idle here anyway while their last stores drain
            if (Bk > 0) {
                // the rest of block Bk - 1's update: strips right of the one that was updated at once, up to this wave's own
#pragma unroll
                for (int q = sb + 1; q < 4; q++)
                    if (q <= wv) update_strip(Bk - 1, q);
                if (pub) publish(Bk - 1);
            }
        }
        F2_TB(1);
        f2_bar();
        F2_TB(2);
        if (wv == 0 && Bk == 0 && i > 0) f2_wave_done(fl_L + 8 * i + (i - 1));   // (wave 0: after its first elimination -- its stores are a block time old)
        if (Bk < 7) {
            // the strip of the next block's columns, at once
            const int u = (Bk + 1) >> 1;
            if (u <= wv) update_strip(Bk, u);
        }
        f2_settle();
        F2_TB(3);
    }
    if (pub && wv > 0) publish(7);
    __syncthreads();
    F2_T(6);
    // ---- L11^-1 (blocked: 16 x 16 diagonal blocks by substitution, the rest on the matrix core), then  Minv = L11^-T D^-1
    double *Sa = S0, *St = S0 + 4160;
    const double *Sb = S1;
    auto mm16 = [&](const double *A, int ar, int ac, const double *Bm, int br, int bc, int nk) {
        v4f64 c = {0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < nk; kk++)
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(ar + l15) * LDT + ac + 4 * kk + lk], Bm[(br + 4 * kk + lk) * LDT + bc + l15], c, 0, 0, 0);
        return c;
    };
    double dkeep = 0.0;
    if (tid < 64) { dkeep = dsave[tid]; Sd[tid] = 1.0 / dkeep; }
    __syncthreads();
    // the inverse is read by the workgroups below the diagonal in their steps with the published inverses -- all panels but the
    // batch's last (their last step, like every step of a diagonal workgroup, follows the stream)
    const bool need_minv = i + 1 < nb && B.nblk > nb;
    if (need_minv) {
    for (int idx = tid; idx < 64 * LDT; idx += 256) Sa[idx] = 0.0;
    __syncthreads();
    if (tid < 64) {   // thread = column j of diagonal block bq
        const int bq = tid >> 4, jc = tid & 15, o = 16 * bq;
        double xx[16];
#pragma unroll
        for (int c = 0; c < 16; c++) xx[c] = c == jc ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++)
#pragma unroll
            for (int r_ = k + 1; r_ < 16; r_++) xx[r_] = fma(-Sb[(o + r_) * LDT + o + k], xx[k], xx[r_]);
#pragma unroll
        for (int c = 0; c < 16; c++) Sa[(o + c) * LDT + o + jc] = xx[c];
    }
    __syncthreads();
    if (wv < 2) {     // block size 16, pairs (0,1) and (2,3):  T = L21 X11
        const int o = 32 * wv;
        const v4f64 c = mm16(Sb, o + 16, o, Sa, o, o, 4);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) St[(o + 16 + lk + 4 * reg) * LDT + o + l15] = c[reg];
    }
    __syncthreads();
    if (wv < 2) {     // X21 = -X22 T
        const int o = 32 * wv;
        const v4f64 c = mm16(Sa, o + 16, o + 16, St, o + 16, o, 4);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) Sa[(o + 16 + lk + 4 * reg) * LDT + o + l15] = -c[reg];
    }
    __syncthreads();
    {                 // block size 32:  T = L21 X11, one 16 x 16 piece per wave
        const int ti = wv >> 1, tj = wv & 1;
        const v4f64 c = mm16(Sb, 32 + 16 * ti, 0, Sa, 0, 16 * tj, 8);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) St[(32 + 16 * ti + lk + 4 * reg) * LDT + 16 * tj + l15] = c[reg];
    }
    __syncthreads();
    {                 // X21 = -X22 T
        const int ti = wv >> 1, tj = wv & 1;
        const v4f64 c = mm16(Sa, 32 + 16 * ti, 32, St, 32, 16 * tj, 8);
        __syncthreads();                                  // every wave has read the X22 rows it needs before X21 is written
#pragma unroll
        for (int reg = 0; reg < 4; reg++) Sa[(32 + 16 * ti + lk + 4 * reg) * LDT + 16 * tj + l15] = -c[reg];
    }
    __syncthreads();
    {
        // Minv^T in operand order: value (c, k) = Minv[k][c] = W[c][k] / d_c for c >= k (W = L11^-1), then the pivots
        double *mv = scratch + (int64_t)i * 4160;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int g = wv + 4 * q;                     // group = 16 (c / 16) + k / 4; this wave's lanes are the group's 64 slots
            const int c = 16 * (g >> 4) + l15, k = 4 * (g & 15) + lk;
            f2_st(mv + g * 64 + lane, c >= k ? Sa[c * LDT + k] * Sd[c] : 0.0);
        }
        if (tid < 64) f2_st(mv + 4096 + tid, dkeep);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(fl_minv + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }   // need_minv
    F2_T(7);
    // ---- off the critical path: the factored block for the solves' explicit inverses, D and 1/D (exact division)
    {
        double *ld = P.Ldiag + fp[i].diag_off;
        for (int idx = tid; idx < 4096; idx += 256) {
            const int r_ = idx & 63, k = idx >> 6;
            ld[idx] = r_ > k ? Sb[r_ * LDT + k] : (r_ == k ? 1.0 : 0.0);
        }
        if (tid < 64) {
            P.D[f + tid] = dkeep;
            P.Dinv[f + tid] = Sd[tid];
            if (!isfinite(Sd[tid])) atomicOr(P.flags + FL_NONFINITE, 1);
        }
        if (lane == 0 && nreg) atomicAdd(P.flags + FL_NREG, nreg);
    }
    store_panels();
    F2_T(8);
    }
    };   // body
    if (i < nb) body(std::true_type{});
    else body(std::false_type{});
}

}  // namespace

void launch_front_block2(hipStream_t st, const DevPlan &P, const FrontBatch &B, int *sync_all, double *scratch_all, double *stream_all,
                         double dyn_eps, double dyn_delta, long long *trace) {
    if (B.i_end <= B.i_base) return;
    const dim3 grid(B.i_end - B.i_base + (B.x_count + 4 * std::max(B.pad, 1) - 1) / (4 * std::max(B.pad, 1)));
    if (trace) hipLaunchKernelGGL(k_front_block2<true>, grid, dim3(256), 0, st, P, B, sync_all, scratch_all, stream_all, dyn_eps, dyn_delta, trace);
    else hipLaunchKernelGGL(k_front_block2<false>, grid, dim3(256), 0, st, P, B, sync_all, scratch_all, stream_all, dyn_eps, dyn_delta, trace);
}

// before every factorisation: the sync words of all front batches to zero, the stream records to "not written yet"
__global__ void k_fb_reset(int *sync_all, int nsync, unsigned long long *stream_all, long long nstream) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    for (long long q = t; q < nsync; q += step) sync_all[q] = 0;
    for (long long q = t; q < nstream; q += step) stream_all[q] = 0xFFFFFFFFFFFFFFFFull;   // the "not written yet" sentinel (f2_hi)
}
void launch_fb_reset(hipStream_t st, int *sync_all, int nsync, double *stream_all, int64_t nstream) {
    const long long n = std::max<long long>(nsync, nstream);
    if (n <= 0) return;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(k_fb_reset, dim3(blocks), dim3(256), 0, st, sync_all, nsync, (unsigned long long *)stream_all, (long long)nstream);
}


}  // namespace hipkkt
