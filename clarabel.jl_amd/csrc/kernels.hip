// HIP kernels of the KKT path for gfx950 (MI355X / CDNA4).  See DESIGN.md section 5 for the roofline that bounds each
// one.  Everything is Float64; indices are int32 except panel offsets (int64).  (KKT assembly: assemble_dev.hip.)
//
//   value updates     k_scatter_values / k_scale_values / k_soc_batch / k_psd_hs                (K1-K2, HBM / latency)
//   regulariser       k_maxabs_gather + k_init_panels (+-eps folded into the K -> panel scatter)  (K3)
//   numeric LDL^T     k_factor_panel (w > 8: register-resident, 8-pivot blocks), k_factor_level   (K4, dependency latency)
//                     k_update_dense<NT,NR> (FP64 matrix-core tiles: <4,4> far, <1,2> just in time), k_update_gather
//                     (per-entry lists of the leaves), k_update_stage (relative-index scatter, fallback)   (K4, MFMA)
//                     k_invert_diag / k_invert_diag_wide (explicit inverses of the diagonal blocks for the solves)
//   triangular solves k_front_fwd / k_front_bwd (persistent sweeps over a front), k_fwd_seg / k_bwd_seg (persistent,
//                     ticket-ordered sweeps over the regular supernodes), k_fwd_level / k_bwd_partial / k_bwd_final and
//                     k_fwd_narrow / k_bwd_narrow (one launch per level: wide bottom levels, time-out fallback)  (K5)
//   refinement        k_spmv_residual(_cand), k_norm_inf, k_refine_decide / _add / _copy_out (decided on the device)  (K6-K8)
//   residuals (N4)    k_residuals + k_residuals_finish;  k_block_products
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "device_plan.h"
#include "kernels.h"
#include "sweep_common.h"
#include "dense_tile.h"

namespace hipkkt {


// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_max_abs(unsigned long long *slot, double v) {
    // |v| as an ordered integer; NaN (all-ones exponent, non-zero mantissa) compares largest,
    // so a NaN anywhere surfaces as a non-finite maximum
    unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(v));
    atomicMax(slot, bits);
}

__device__ __forceinline__ double wave_max(double v);
// one same-address atomic per BLOCK (same-word atomics serialise at ~88/us on this chip)
__device__ __forceinline__ void block_atomic_max_abs(unsigned long long *slot, double a);

__device__ __forceinline__ double wave_max(double v) {
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = (o > v || o != o) ? o : v;
    }
    return v;
}

__device__ __forceinline__ void block_atomic_max_abs(unsigned long long *slot, double a) {
    __shared__ double wm_[16];
    a = wave_max(a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) wm_[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = wm_[0];
        for (int i = 1; i < nw; i++) m = (wm_[i] > m || wm_[i] != wm_[i]) ? wm_[i] : m;
        atomic_max_abs(slot, m);
    }
}

// ------------------------------------------------------------------------------------------
// K1/K2: value updates on the resident KKT image (ref: kktsolver_directldl.jl:130-188)
// ------------------------------------------------------------------------------------------
__global__ void k_scatter_values(double *__restrict__ kval, const int64_t *__restrict__ idx,
                                 const double *__restrict__ vals, int64_t n, double scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) kval[idx[i]] = vals[i] * scale;
}

__global__ void k_scale_values(double *__restrict__ kval, const int64_t *__restrict__ idx, int64_t n,
                               double scale) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) kval[idx[i]] *= scale;
}

// all sparse SOC cones in one launch (ref: _csc_update_sparsecone, directldl_datamaps.jl:61-79):
// K[u_idx] = u * (-eta^2), K[v_idx] = v * (-eta^2), K[D_idx] = (-eta^2, +eta^2)
__global__ void k_soc_batch(double *__restrict__ kval, const int64_t *__restrict__ uidx, const int64_t *__restrict__ vidx,
                            const int *__restrict__ cone_of, const double *__restrict__ u, const double *__restrict__ v,
                            const double *__restrict__ eta2, int64_t n, const int64_t *__restrict__ didx, int nsoc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double sc = -eta2[cone_of[i]];
        kval[uidx[i]] = u[i] * sc;
        kval[vidx[i]] = v[i] * sc;
    }
    if (i < nsoc) {
        kval[didx[2 * i]] = -eta2[i];
        kval[didx[2 * i + 1]] = eta2[i];
    }
}

__global__ void k_fill(double *__restrict__ p, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ------------------------------------------------------------------------------------------
// K3: static regulariser eps = c + p * max|diag K|  (ref: kktsolver_directldl.jl:297-310)
// scal[SC_MAXDIAG] must be zeroed before the launch
// ------------------------------------------------------------------------------------------
__global__ void k_maxabs_gather(const double *__restrict__ v, const int64_t *__restrict__ idx, int64_t n,
                                unsigned long long *__restrict__ slot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0.0;
    if (i < n) a = fabs(idx ? v[idx[i]] : v[i]);
    block_atomic_max_abs(slot, a);
}

// Lx <- scatter(K) with the +-eps shift on the diagonal (the resident Kval stays unregularised:
// kktsolver_directldl.jl:285-291 restores the diagonal for the refinement step)
__global__ void k_init_panels(double *__restrict__ Lx, const double *__restrict__ kval,
                              const int64_t *__restrict__ kmap, const signed char *__restrict__ kdiag_sign,
                              int64_t nnz, const double *__restrict__ scal, int static_enable,
                              double eps_const, double eps_prop) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    double v = kval[q];
    int sg = kdiag_sign[q];
    if (static_enable && sg != 0) {
        double maxdiag = __longlong_as_double((long long)((const unsigned long long *)scal)[SC_MAXDIAG]);
        double eps = eps_const + eps_prop * maxdiag;
        v += sg > 0 ? eps : -eps;
    }
    Lx[kmap[q]] = v;
}

// ------------------------------------------------------------------------------------------
// K4a: per-level panel factorisation.  One workgroup per (supernode, 64-row chunk).
// Every workgroup of a supernode refactors the w x w diagonal block redundantly in LDS
// (bit-identical results, no inter-workgroup hand-off inside the launch); chunk 0 publishes
// L11 / D / Dinv, every chunk solves its own rows  L21 = A21 L11^-T D^-1  in place.
// Pivot rule = QDLDL's: if D_k * sign_k < eps then D_k = delta * sign_k   (SURVEY.md App. C).
// ------------------------------------------------------------------------------------------
constexpr int LDT = kFacRows + 8;      // LDS leading dimension of the TRSM tile [k][row]

// dynamic LDS is sized for the widest supernode of the level (wmax): Aw[wmax][wmax+1], dd[wmax], T[wmax][LDT]
__global__ void __launch_bounds__(256)
k_factor_level(DevPlan P, int item_begin, int wmax, double dyn_eps, double dyn_delta) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int LDW = wmax + 1;                  // LDS leading dimension of the diagonal block
    double *Aw = smem;                         // [wmax * LDW]
    double *dd = Aw + wmax * LDW;              // [wmax] pivots
    double *T = dd + wmax;                     // [wmax * LDT]
    const FacItem it = P.fac_items[item_begin + blockIdx.x];
    const int s = it.sn;
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
    double *pan = P.Lx + P.sn_panel[s];
    const int tid = threadIdx.x;

    for (int idx = tid; idx < w * w; idx += 256) {
        int i = idx % w, j = idx / w;
        Aw[i + j * LDW] = pan[i + (int64_t)j * r];
    }
    __syncthreads();
    // right-looking LDL^T of the lower triangle; thread (ti,tj): row k+1+ti, columns k+1+tj (+4...)
    const int ti = tid & 63, tj = tid >> 6;
    int nreg = 0;
    for (int k = 0; k < w; k++) {
        double d = Aw[k + k * LDW];
        const double sg = (double)P.sgn_perm[f + k];
        if (d * sg < dyn_eps) { d = dyn_delta * sg; nreg++; }
        const double dinv = 1.0 / d;
        if (tid == 0) dd[k] = d;
        const int i = k + 1 + ti;
        if (i < w) {
            const double aik = Aw[i + k * LDW];
            for (int j = k + 1 + tj; j <= i; j += 4) Aw[i + j * LDW] -= aik * (Aw[j + k * LDW] * dinv);
        }
        __syncthreads();
    }
    // scale columns: L11 = A11_lower * D^-1
    for (int idx = tid; idx < w * w; idx += 256) {
        int i = idx % w, k = idx / w;
        if (i > k) Aw[i + k * LDW] /= dd[k];
    }
    __syncthreads();
    if (it.blk == 0) {
        double *ld = P.Ldiag + P.sn_diag[s];
        for (int idx = tid; idx < w * w; idx += 256) {
            int i = idx % w, k = idx / w;
            ld[idx] = i > k ? Aw[i + k * LDW] : (i == k ? 1.0 : 0.0);
        }
        if (tid < w) {
            P.D[f + tid] = dd[tid];
            P.Dinv[f + tid] = 1.0 / dd[tid];
            if (!isfinite(1.0 / dd[tid])) atomicOr(P.flags + FL_NONFINITE, 1);
        }
        if (tid == 0 && nreg) atomicAdd(P.flags + FL_NREG, nreg);
    }
    // TRSM on this chunk's rows: 4 lanes per row, y_k = a_k - sum_{j<k} y_j L11[k][j]
    const int lo = w + it.blk * kFacRows;
    const int nr = min(kFacRows, r - lo);
    if (nr <= 0) return;
    for (int idx = tid; idx < nr * w; idx += 256) {
        int row = idx % nr, k = idx / nr;
        T[k * LDT + row] = pan[(lo + row) + (int64_t)k * r];
    }
    __syncthreads();
    {
        const int q = tid & 3;
        for (int row = tid >> 2; row < nr; row += 64) {
            for (int k = 0; k < w; k++) {
                double acc = 0.0;
                for (int j = q; j < k; j += 4) acc += T[j * LDT + row] * Aw[k + j * LDW];
                acc += __shfl_xor(acc, 1, 64);
                acc += __shfl_xor(acc, 2, 64);
                const double yk = T[k * LDT + row] - acc;
                T[k * LDT + row] = yk;  // all 4 lanes store the same value
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < nr * w; idx += 256) {
        int row = idx % nr, k = idx / nr;
        const double v = T[k * LDT + row] / dd[k];
        T[k * LDT + row] = v;
        pan[(lo + row) + (int64_t)k * r] = v;
    }
    __syncthreads();
    // row-major copy (w contiguous doubles per row) for the backward solve's L21^T x
    double *lt = P.LT + P.lt_off[s] + (int64_t)(lo - w) * w;
    for (int idx = tid; idx < nr * w; idx += 256) {
        int k = idx % w, row = idx / w;
        lt[idx] = T[k * LDT + row];
    }
}

// ------------------------------------------------------------------------------------------
// K4a, wide panels (w > 8): register-resident right-looking LDL^T of the whole panel chunk.
// 8 wavefronts: group D (waves 0-3) holds the w x w diagonal block, group O (waves 4-7) the chunk's
// 64 off-diagonal rows; lane = row, wave v of a group owns the columns j = v (mod 4) (16 registers).
// Step k: the owner waves publish column k through LDS (double buffered: one barrier per step),
// every wave forms l_ik = a_ik / d_k for its row and applies  a_ij -= l_ik * a_jk  to its own
// columns j > k.  The panel rows are eliminated with the same steps, so the TRSM costs no extra
// pass: L21[:,k] leaves the kernel at step k.  The critical path per pivot is
// LDS write -> barrier -> LDS read -> 1/d (112 clk) -> mul -> fma  (~400 clk; tools/ubench.hip).
// Pivot rule = QDLDL's.  Every chunk repeats the diagonal block (bit-identical).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double x, int l) {   // l wave-uniform
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// 1/d on the pivot-to-pivot critical path: hardware reciprocal + two Newton steps (~50 clocks) instead of the IEEE
// division sequence (~112 clocks, tools/ubench.hip).  |relative error| <= 1 ulp; the D and 1/D that are stored for
// the solves are formed with the exact division after the loop.  d is never zero or subnormal here (the pivot rule
// has replaced such values by +-delta); an infinite or NaN d propagates and is caught by the non-finite check.
__device__ __forceinline__ double pivot_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// r (1 + e + e^2), e = 1 - d r: three dependent operations after the hardware reciprocal instead of the four of two Newton steps
// (error ~ e^3 + one rounding: 1.9 units of 2^-53 measured over 6.7e7 values, tools/ubench_pivot.hip)
__device__ __forceinline__ double pivot_rcp3(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    const double e2 = fma(e, e, e);
    return fma(r, e2, r);
}

// Blocked by 8 pivots: wave v of a group owns the column blocks {v, v+4} (8 columns each).  The owner of block B
// eliminates its 8 columns WITHOUT leaving the wavefront (pivot and the entries a_jk come from the lanes that hold
// them: v_readlane), publishes the block's L columns / raw columns / 1/d through LDS, and after ONE barrier every
// wave applies the rank-8 update to its own live blocks; the chunk rows (group O) follow with the published
// values and a second barrier.  16 barriers per 64-column panel instead of 64, and only the 8x8 triangle of the
// current block sits on the pivot-to-pivot critical path.

// Round 6: TWO workgroups per compute unit (62 KB of LDS instead of 96, <= 128 registers).  A level of many panels -- the 20 cone
// chains of an SDP: 400 items per level on cfg 5 -- ran in two rounds of one workgroup per compute unit, 31 us per level against the
// 15 us one round takes.  What went: the staging tile of the factored diagonal block (its columns now go from the published L columns
// in LDS straight to Ldiag, by a wavefront that is not on the pivot chain, inside the loop -- whose barrier therefore no longer waits
// for global memory).
__device__ __forceinline__ void fp_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // LDS-only workgroup barrier

__global__ void __launch_bounds__(512, 4)                 // (HIP: the second figure is wavefronts per SIMD: 4 = two of these workgroups per compute unit)
k_factor_panel(DevPlan P, int item_begin, double dyn_eps, double dyn_delta) {
    __shared__ double colL[2][8][64];     // l_ik of the diagonal-block rows of block B          (parity B & 1)
    __shared__ double colC[3][8][64];     // raw a_ik = d_k l_ik of the diagonal-block rows        (B % 3: read for two iterations)
    __shared__ double colLO[2][8][64];    // l_ik of the chunk rows                                (parity B & 1)
    __shared__ double dinvs[3][8];
    __shared__ double Yt[64 * 65];        // [row * 65 + k]: L21 of the chunk rows (group O), staged for the panel and its row-major copy
    __shared__ double dsave[64];
    // one self-contained record per item: the panel kernels are the critical path of the factorisation, and the
    // chain  item -> supernode tables -> panel  cost two extra dependent memory round trips per launch
    const FacRec it = P.fac_recs[item_begin + blockIdx.x];
    const int f = it.f, w = it.w, r = it.r;
    double *pan = P.Lx + it.panel_off;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int grp = wv >> 2, v = wv & 3;
    const int lo = w + it.blk * kFacRows;
    const int nr = min(kFacRows, r - lo);
    const int prow = grp ? lo + lane : lane;              // panel row of this lane
    const bool rvalid = grp ? lane < nr : lane < w;
    double a[16];                                         // a[8*h + jj] = column 8*(v + 4h) + jj of this lane's row
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const int j = 8 * (v + 4 * (c >> 3)) + (c & 7);
        a[c] = (rvalid && j < w) ? pan[prow + (int64_t)j * r] : 0.0;
    }
    const unsigned long long spos = __ballot(lane < w && P.sgn_perm[f + (lane < w ? lane : 0)] > 0);
    double *myY = &Yt[lane * 65];                        // (group O only)
    int nreg = 0;
    const int nB = (w + 7) >> 3;
    // Software pipeline with ONE barrier per block: in iteration B the diagonal-row owner eliminates block B while
    // the chunk-row owner eliminates block B-1 with the values published one barrier earlier; after the barrier
    // the diagonal-row waves apply block B and the chunk-row waves apply block B-1 to their live blocks.
    // No global stores inside the loop: __syncthreads() would wait for them every step.
#pragma unroll
    for (int B = 0; B <= 8; B++) {
        if (B <= nB) {                                    // workgroup-uniform
            if (grp == 0 && B < 8 && B < nB && v == (B & 3)) {   // diagonal rows, block B: 8 pivots in-wave
                // Round 5: the lean chain of the front-batch kernels (front_block.hip, front_block2.hip) here as well --
                //   d_k = a_kk - c_{k,k-1}^2 / d_{k-1}  ->  1 / d_k   is one fma, the hardware reciprocal and three more fma;
                //   a_kk and c_{k,k-1} are fetched from the lanes that hold them one pivot EARLIER;
                //   the pivot rule is only TESTED, at the end of the block from the eight pivots (no compare / select / branch on the
                //   chain); a block that needs a substituted pivot is repeated from its saved input with the rule applied;
                //   the pivots / reciprocals are published once per block.
                const int rb = 8 * (B >> 2), pb = B & 1, p3 = B % 3;
                double a_in[8];
#pragma unroll
                for (int q = 0; q < 8; q++) a_in[q] = a[rb + q];
                auto eliminate = [&](auto rule_tag) -> bool {
                    constexpr bool RULE = decltype(rule_tag)::value;
                    double dk[8], dik[8];
                    int nr_ = 0;
                    double akk = readlane_f64(a[rb], 8 * B), csq = 0.0, dinv_prev = 0.0;
#pragma unroll
                    for (int kk = 0; kk < 8; kk++) {
                        const int k = 8 * B + kk;
                        double d = fma(-csq, dinv_prev, akk);
                        double dinv = pivot_rcp3(d);
                        if (RULE) {
                            const double sg = ((spos >> k) & 1ull) ? 1.0 : -1.0;
                            const bool bad = k < w && d * sg < dyn_eps;
                            d = bad ? dyn_delta * sg : d;
                            dinv = bad ? sg / dyn_delta : dinv;
                            nr_ += bad ? 1 : 0;
                        }
                        dinv = k < w ? dinv : 0.0;          // (columns past the panel's width: padding)
                        dk[kk] = d;
                        dik[kk] = dinv;
                        const double reg = a[rb + kk];
                        if (kk < 7) {
                            akk = readlane_f64(a[rb + kk + 1], k + 1);
                            const double cn = readlane_f64(reg, k + 1);
                            csq = cn * cn;
                        }
                        dinv_prev = dinv;
                        const double li = reg * dinv;
                        colL[pb][kk][lane] = li;
                        colC[p3][kk][lane] = k < w ? reg : 0.0;
#pragma unroll
                        for (int jj = kk + 1; jj < 8; jj++) {
                            const double cj = readlane_f64(reg, 8 * B + jj);
                            a[rb + jj] = fma(-li, cj, a[rb + jj]);
                        }
                    }
                    if (!RULE) {
                        bool anybad = false;
#pragma unroll
                        for (int kk = 0; kk < 8; kk++) {
                            const int k = 8 * B + kk;
                            const double sg = ((spos >> k) & 1ull) ? 1.0 : -1.0;
                            anybad = anybad || (k < w && dk[kk] * sg < dyn_eps);
                        }
                        if (__builtin_amdgcn_readfirstlane((int)anybad)) return false;
                    }
                    nreg += nr_;
                    if (lane == 0) {
#pragma unroll
                        for (int kk = 0; kk < 8; kk++) { dsave[8 * B + kk] = dk[kk]; dinvs[p3][kk] = dik[kk]; }
                    }
                    return true;
                };
                if (!eliminate(std::false_type{})) {
#pragma unroll
                    for (int q = 0; q < 8; q++) a[rb + q] = a_in[q];
                    eliminate(std::true_type{});
                }
            }
            if (grp == 1 && B >= 1 && v == ((B - 1) & 3)) {       // chunk rows, block B-1: published operands
                const int Bo = B - 1, rb = 8 * (Bo >> 2), pb = Bo & 1, p3 = Bo % 3;
#pragma unroll
                for (int kk = 0; kk < 8; kk++) {
                    const int k = 8 * Bo + kk;
                    const double li = a[rb + kk] * dinvs[p3][kk];
                    colLO[pb][kk][lane] = li;
                    myY[k] = li;
#pragma unroll
                    for (int jj = kk + 1; jj < 8; jj++) a[rb + jj] = fma(-li, colC[p3][kk][8 * Bo + jj], a[rb + jj]);
                }
            }
            fp_bar();
            // the factored diagonal block goes to Ldiag from the published columns, by the diagonal-row wavefront that eliminated two
            // blocks ago (not the one that eliminates next); colL[B & 1] stays valid until block B + 2 is published
            if (grp == 0 && it.blk == 0 && B < nB && B < 8 && v == ((B + 2) & 3) && lane < w) {
                double *ld = P.Ldiag + it.diag_off;
#pragma unroll
                for (int kk = 0; kk < 8; kk++) {
                    const int k = 8 * B + kk;
                    if (k < w) ld[lane + k * w] = lane > k ? colL[B & 1][kk][lane] : (lane == k ? 1.0 : 0.0);
                }
            }
            const int Bu = grp == 0 ? B : B - 1;          // block this wave applies now
            if (Bu >= 0 && Bu < nB && Bu < 8) {
                const double(*Lsrc)[64] = grp == 0 ? colL[Bu & 1] : colLO[Bu & 1];
                const int p3 = Bu % 3;
                double lk_[8];
#pragma unroll
                for (int kk = 0; kk < 8; kk++) lk_[kk] = Lsrc[kk][lane];
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (v + 4 * h > Bu) {
#pragma unroll
                        for (int jj = 0; jj < 8; jj++) {
                            const int j = 8 * (v + 4 * h) + jj;
#pragma unroll
                            for (int kk = 0; kk < 8; kk++) a[8 * h + jj] = fma(-lk_[kk], colC[p3][kk][j], a[8 * h + jj]);
                        }
                    }
            }
        }
    }
    __syncthreads();
    if (it.blk == 0) {
        if (tid < w) {
            const double d = dsave[tid], dinv = 1.0 / d;
            P.D[f + tid] = d;
            P.Dinv[f + tid] = dinv;
            if (!isfinite(dinv)) atomicOr(P.flags + FL_NONFINITE, 1);
        }
        if (grp == 0 && lane == 0 && nreg) atomicAdd(P.flags + FL_NREG, nreg);   // each owner wave counted its own blocks
    }
    if (nr > 0) {
        for (int idx = tid; idx < nr * w; idx += 512) {   // column-major panel rows
            const int row = idx % nr, k = idx / nr;
            pan[(lo + row) + (int64_t)k * r] = Yt[row * 65 + k];
        }
        // row-major copy (w contiguous doubles per row) for the backward solve's L21^T x
        double *lt = P.LT + it.lt_off + (int64_t)(lo - w) * w;
        for (int idx = tid; idx < nr * w; idx += 512) {
            const int k = idx % w, row = idx / w;
            lt[idx] = Yt[row * 65 + k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// after the factorisation: explicit inverses of the unit-lower diagonal blocks (and their
// transposes) so that the solves do small GEMVs instead of w-step substitutions.  One launch.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_invert_diag(DevPlan P, int list_begin, int n) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x >= n) return;
    const int s = P.inv_list[list_begin + blockIdx.x];
    const int w = P.sn_first[s + 1] - P.sn_first[s];
    const int LDL = w | 1;
    double *Ls = smem;              // [w * LDL]
    double *Xs = Ls + w * LDL;      // [w * LDL]
    const double *ld = P.Ldiag + P.sn_diag[s];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < w * w; idx += 64) {
        int i = idx % w, k = idx / w;
        Ls[i + k * LDL] = ld[idx];
        Xs[i + k * LDL] = (i == k) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (tid < w) {
        const int j = tid;  // column j of the inverse: solve L x = e_j
        for (int k = j; k < w; k++) {
            const double xk = Xs[k + j * LDL];
            for (int i = k + 1; i < w; i++) Xs[i + j * LDL] -= Ls[i + k * LDL] * xk;
        }
    }
    __syncthreads();
    double *li = P.Linv + P.sn_diag[s];
    double *lit = P.LinvT + P.sn_diag[s];
    for (int idx = tid; idx < w * w; idx += 64) {
        int i = idx % w, j = idx / w;
        li[idx] = Xs[i + j * LDL];     // Linv[i][j], column-major
        lit[idx] = Xs[j + i * LDL];    // LinvT stored so that element (i,j) of Linv sits at [j + i*w]
    }
}

// Wide diagonal blocks (16 < w <= 64, the panels of the fronts): the column-by-column substitution above is a
// 2048-step read-modify-write chain through LDS (~200 us, after the last panel of the factorisation, with nothing
// to overlap it).  Blocked instead, 256 threads: the four 16x16 diagonal blocks are inverted in registers (one
// thread per column), then  X21 = -X22 (L21 X11)  is applied at block size 16 and at block size 32 -- small dense
// products with all operands in LDS.  The block is padded to 64 with the identity.
__global__ void __launch_bounds__(256)
k_invert_diag_wide(DevPlan P, int list_begin) {
    __shared__ double Ls[64 * 65], Xs[64 * 65], Ts[64 * 65];   // [row * 65 + col]
    const int s = P.inv_list[list_begin + blockIdx.x];
    const int w = P.sn_first[s + 1] - P.sn_first[s];
    const double *ld = P.Ldiag + P.sn_diag[s];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx & 63, k = idx >> 6;
        Ls[i * 65 + k] = (i < w && k < w) ? ld[i + k * w] : (i == k ? 1.0 : 0.0);
        Xs[i * 65 + k] = 0.0;
    }
    __syncthreads();
    if (tid < 64) {   // 16x16 diagonal blocks: thread = column j of block bq, forward substitution in registers
        const int bq = tid >> 4, j = tid & 15, o = 16 * bq;
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; c++) x[c] = c == j ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++)
#pragma unroll
            for (int i = k + 1; i < 16; i++) x[i] = fma(-Ls[(o + i) * 65 + o + k], x[k], x[i]);
#pragma unroll
        for (int c = 0; c < 16; c++) Xs[(o + c) * 65 + o + j] = x[c];
    }
    __syncthreads();
    // block size 16: pairs (0,1) and (2,3).  T = L21 X11, then X21 = -X22 T
    for (int e = tid; e < 512; e += 256) {
        const int pr = e >> 8, i = (e >> 4) & 15, j = e & 15, o = 32 * pr;
        double a = 0.0;
#pragma unroll
        for (int m = 0; m < 16; m++) a = fma(Ls[(o + 16 + i) * 65 + o + m], Xs[(o + m) * 65 + o + j], a);
        Ts[(o + 16 + i) * 65 + o + j] = a;
    }
    __syncthreads();
    for (int e = tid; e < 512; e += 256) {
        const int pr = e >> 8, i = (e >> 4) & 15, j = e & 15, o = 32 * pr;
        double a = 0.0;
#pragma unroll
        for (int m = 0; m < 16; m++) a = fma(Xs[(o + 16 + i) * 65 + o + 16 + m], Ts[(o + 16 + m) * 65 + o + j], a);
        Xs[(o + 16 + i) * 65 + o + j] = -a;
    }
    __syncthreads();
    // block size 32: T = L21 X11 (32x32 each, X11 lower triangular), X21 = -X22 T
    for (int e = tid; e < 1024; e += 256) {
        const int i = e >> 5, j = e & 31;
        double a = 0.0;
        for (int m = j; m < 32; m++) a = fma(Ls[(32 + i) * 65 + m], Xs[m * 65 + j], a);
        Ts[(32 + i) * 65 + j] = a;
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
        const int i = e >> 5, j = e & 31;
        double a = 0.0;
        for (int m = 0; m <= i; m++) a = fma(Xs[(32 + i) * 65 + 32 + m], Ts[(32 + m) * 65 + j], a);
        Xs[(32 + i) * 65 + j] = -a;
    }
    __syncthreads();
    double *li = P.Linv + P.sn_diag[s];
    double *lit = P.LinvT + P.sn_diag[s];
    double mx = 0.0;
    for (int idx = tid; idx < w * w; idx += 256) {
        const int i = idx % w, j = idx / w;
        li[idx] = Xs[i * 65 + j];      // Linv[i][j], column-major
        lit[idx] = Xs[j * 65 + i];     // LinvT: element (i,j) of Linv at [j + i*w]
        mx = fmax(mx, fabs(Xs[i * 65 + j]));
    }
    // REFINED BLOCK SOLVES.  The solve kernels multiply by this explicit inverse; where it has large entries the product is several
    // times less accurate than the reference's substitution (forward error ~ eps |Linv| |b| against ~ eps |L^-1| |L| |y|) -- on
    // batch seed 324 enough to flip the stop-ratio branch of the iterative refinement and to end the IPM one iteration apart, 1.3e-5
    // off in the objective (round 4).  Established on the CPU with the host interpreter of the plan
    // (tools/seed324_inverse_vs_substitution.py): the first-solve residual is 5.5e-5 with the inverses against substitution's
    // 1.46e-5; one step  y += Linv (b - L y)  on the WIDE blocks (> 16 columns: 16 of that problem's 1361 supernodes) restores
    // 1.46e-5 exactly, on the narrow ones it changes nothing.  So a wide block whose inverse has an entry above DevPlan::polish_tau
    // is marked, and the kernels that solve with regular supernodes take that step on it (k_fwd_level / k_fwd_seg / k_bwd_final /
    // k_bwd_seg).  The fronts' sweeps do not (their panels come in chains whose super-block inverses are a different construction).
    {
        __shared__ double wmx[4];
        mx = wave_max(mx);
        if ((tid & 63) == 0) wmx[tid >> 6] = mx;
        __syncthreads();
        if (tid == 0) {
            const double m = fmax(fmax(wmx[0], wmx[1]), fmax(wmx[2], wmx[3]));
            // every wide block is marked (after a sweep time-out the per-level kernels solve the fronts' panels too, and refine them);
            // COUNTED are the blocks the regular kernels solve (sn_nitems > 0: not a panel of a front) -- the front sweeps never
            // refine, so counting their 64-column panels overstated what is refined (review of round 5)
            const bool big = w > 16 && !(m <= P.polish_tau);
            P.sn_polish[s] = big ? 1 : 0;
            if (big && P.sn_nitems[s] > 0) atomicAdd(P.flags + FL_NPOLISH, 1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K4b: Schur-complement updates of one stage.  One workgroup owns one 64-row block of one
// target panel (exclusive ownership => deterministic, no atomics) and applies its task list in
// order:   C[rel(i), col(j)] -= sum_k L_s[i,k] * d_k * L_s[j,k]
// The contraction runs on the FP64 matrix core: v_mfma_f64_16x16x4_f64, one 16x16 output tile
// per wave step, operands gathered straight from the source panel (rows are contiguous per k).
// ------------------------------------------------------------------------------------------
constexpr int LDC = kUpdRows + 1;
constexpr int kUpdWaves = 8;   // wavefronts per workgroup in the update kernel

template <int MT>
__device__ __forceinline__ void upd_strip(const double *__restrict__ sp, const double *__restrict__ dv, int r, int K,
                                          int row_lo, int nrows, int tm0, int col_row, int ncols_left, int l15, int lk,
                                          v4f64 (&acc)[4]) {
    // operands gathered straight from the source panel: for a fixed k the 16 rows of a tile are
    // contiguous.  Out-of-range rows are clamped (their outputs are discarded at the scatter);
    // out-of-range k contributes zero through the B operand.
    const double *ap[MT];
#pragma unroll
    for (int t = 0; t < MT; t++) {
        int ai = (tm0 + t) * 16 + l15;
        ai = ai < nrows ? ai : nrows - 1;
        ap[t] = sp + (row_lo + ai);
    }
    int bj = l15 < ncols_left ? l15 : ncols_left - 1;
    const double *bp = sp + (col_row + bj);
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 4) {
        int kk = k0 + lk;
        const double km = kk < K ? 1.0 : 0.0;
        kk = kk < K ? kk : K - 1;
        const int64_t off = (int64_t)kk * r;
        const double b = bp[off] * (dv[kk] * km);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[t][off], b, acc[t], 0, 0, 0);
    }
}

// One workgroup owns one 64-row block of one target panel.  Its task list is cut into wave-tasks
// (one source row range x one 16-column strip); wavefronts take wave-tasks round-robin, run the
// contraction on the FP64 matrix core independently (their load latencies overlap) and then apply
// their results to the LDS-resident target tile strictly in list order (LDS turn counter), so the
// summation order - and therefore every bit of the factor - is fixed.
__global__ void __launch_bounds__(kUpdWaves * 64)
k_update_stage(DevPlan P, int group_begin) {
    __shared__ double Ct[kMaxSnWidth * LDC];
    __shared__ int turn;
    const UpdGroup G = P.upd_groups[group_begin + blockIdx.x];
    const int t = G.tgt;
    const int ft = P.sn_first[t];
    const int wt = P.sn_first[t + 1] - ft;
    const int rt = (int)(P.sn_rowptr[t + 1] - P.sn_rowptr[t]);
    double *tp = P.Lx + P.sn_panel[t];
    const int tid = threadIdx.x;
    const int nrt = min(kUpdRows, rt - G.row_base);
    for (int idx = tid; idx < nrt * wt; idx += kUpdWaves * 64) {
        int row = idx % nrt, col = idx / nrt;
        Ct[col * LDC + row] = tp[(G.row_base + row) + (int64_t)col * rt];
    }
    if (tid == 0) turn = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    int q = G.task_begin;                 // task that contains the current wave-task
    UpdTask T = P.upd_tasks[q];
    for (int v = wave; v < G.nvt; v += kUpdWaves) {
        while (q + 1 < G.task_end) {       // advance to the task holding wave-task v
            const UpdTask Tn = P.upd_tasks[q + 1];
            if (Tn.vt_begin > v) break;
            T = Tn;
            q++;
        }
        const int tn = v - T.vt_begin;     // 16-column strip inside the task
        const int s = T.src;
        const int fs = P.sn_first[s];
        const int K = P.sn_first[s + 1] - fs;
        const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
        const double *sp = P.Lx + P.sn_panel[s];
        const double *dv = P.D + fs;
        const int mt = (T.nrows + 15) >> 4;
        const int ncl = T.ncols - tn * 16;
        v4f64 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
        if (mt == 1) upd_strip<1>(sp, dv, r, K, T.row_lo, T.nrows, 0, T.col_lo + tn * 16, ncl, l15, lk, acc);
        else if (mt == 2) upd_strip<2>(sp, dv, r, K, T.row_lo, T.nrows, 0, T.col_lo + tn * 16, ncl, l15, lk, acc);
        else upd_strip<4>(sp, dv, r, K, T.row_lo, T.nrows, 0, T.col_lo + tn * 16, ncl, l15, lk, acc);
        // target coordinates of this lane's results (C/D layout of the f64 16x16x4 form:
        // col = lane & 15, row = (lane >> 4) + 4 * reg)
        const int *srows = P.sn_rows + P.sn_rowptr[s];
        const int *rel = P.rel + T.rel_off;
        const int jj = tn * 16 + l15;
        const int cp = jj < T.ncols ? srows[T.col_lo + jj] - ft : -1;
        int rp[16];
#pragma unroll
        for (int tmi = 0; tmi < 4; tmi++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int ii = tmi * 16 + lk + 4 * reg;
                rp[tmi * 4 + reg] = (ii < T.nrows && cp >= 0) ? rel[(T.row_lo + ii) - T.col_lo] - G.row_base : -1;
            }
        // ordered application
        while (__atomic_load_n(&turn, __ATOMIC_RELAXED) != v) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int tmi = 0; tmi < 4; tmi++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int rr = rp[tmi * 4 + reg];
                if (rr >= 0) Ct[cp * LDC + rr] -= acc[tmi][reg];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __atomic_store_n(&turn, v + 1, __ATOMIC_RELAXED);
    }
    __syncthreads();
    for (int idx = tid; idx < nrt * wt; idx += kUpdWaves * 64) {
        int row = idx % nrt, col = idx / nrt;
        tp[(G.row_base + row) + (int64_t)col * rt] = Ct[col * LDC + row];
    }
}

// ------------------------------------------------------------------------------------------
// K4b (dense tiles): one WAVEFRONT owns one 64-row x <=64-column tile of a target panel and keeps it
// in 16 FP64 accumulators (4 x 4 tiles of v_mfma_f64_16x16x4_f64) while it sweeps ALL contributing
// source panels (K = sum of their widths, 256 for the batched updates of a dense front).  The tile is
// loaded once into the accumulators, the product is accumulated with a negated A operand
// (C - sum_k L[j,k] d_k L[i,k]), and stored once.  Operands come straight from the source panels:
// for a fixed k the 16 rows of an MFMA operand are contiguous (128 B), 4 k's per instruction.
// The transposed product is computed (A operand = target COLUMNS, B operand = target ROWS) so that
// the C/D layout (col = lane&15, row = (lane>>4)+4*reg) puts 16 consecutive panel rows in
// consecutive lanes: tile loads / stores are 128-B segments too.
// No LDS, no barriers, no inter-wave communication: 4 independent wavefronts per workgroup.
// ------------------------------------------------------------------------------------------
// Grid-stride over the tiles of a launch (a bounded grid is used by the look-ahead experiments, hipkkt_factor.cpp).
template <int NT, int NR>
__global__ void __launch_bounds__(256, NT == 2 ? 4 : 2)
k_update_dense(DevPlan P, int group_begin, int ngroups) {
    static_assert(NR == 4 || NT == 1, "rows are split only between the workgroups of a one-strip-per-wave launch");
    const int lane = threadIdx.x & 63;
    const int wave = rfl(threadIdx.x >> 6);
    // NT 16-column strips per wavefront: 4/NT wavefronts share a tile, NT tiles per workgroup;
    // NR < 4: 4/NR consecutive workgroups share a tile, each takes NR of its four 16-row blocks
    const int tj0 = (wave % (4 / NT)) * NT;           // first 16-column strip of this wavefront
    if (NR == 4) {
        for (int g = rfl(blockIdx.x * NT + wave / (4 / NT)); g < ngroups; g += gridDim.x * NT)
            dense_tile<NT, NR>(P, P.dgroups + group_begin + g, lane, tj0, 0);
    } else {
        constexpr int RS = 4 / NR;
        for (int u = blockIdx.x; u < ngroups * RS; u += gridDim.x)
            dense_tile<NT, NR>(P, P.dgroups + group_begin + u / RS, lane, tj0, (u % RS) * NR);
    }
}

// A launch of T tiles runs in ceil(T / 1024) rounds of one 64 x 64 tile per SIMD (~60 us each at K = 320): 1128 tiles cost two
// rounds, 2278 three.  Here the r tiles beyond the last FULL round are cut into pieces, one wavefront each, so that the
// partial round costs a fraction of a whole one: four 16-column strips per tile when r <= 256 (one sub-round of ~22 us),
// two 32-column halves when r <= 512 (one sub-round of ~35 us).  Workgroups [0, nfull / 4) take four whole tiles, every
// later workgroup one tile (PIECES = 4) or two tiles (PIECES = 2).  Measured on cfg 2a: 11 launches 1.49 -> 1.33 ms.
template <int PIECES>
__global__ void __launch_bounds__(256, 2)
k_update_dense_tail(DevPlan P, int group_begin, int nfull, int ngroups) {
    const int lane = threadIdx.x & 63;
    const int wave = rfl(threadIdx.x >> 6);
    const int nfw = nfull >> 2;
    if ((int)blockIdx.x < nfw) {
        dense_tile<4, 4>(P, P.dgroups + group_begin + rfl(blockIdx.x * 4 + wave), lane, 0, 0);
    } else if (PIECES == 4) {
        const int g = nfull + ((int)blockIdx.x - nfw);
        if (g < ngroups) dense_tile<1, 4>(P, P.dgroups + group_begin + g, lane, wave, 0);
    } else {
        const int g = nfull + 2 * ((int)blockIdx.x - nfw) + (wave >> 1);
        if (g < ngroups) dense_tile<2, 4>(P, P.dgroups + group_begin + g, lane, (wave & 1) * 2, 0);
    }
}

// ------------------------------------------------------------------------------------------
// K4b (tiny contributions): the leaves of the elimination tree are 1-2 column supernodes whose few
// rows land on scattered single entries of far ancestors (the dense root front).  One thread per
// TARGET ENTRY sums its (source, row i, row j) pairs  sum_k L_s[i,k] d_k L_s[j,k]  in the fixed order
// of the plan's gather list and subtracts once: no atomics, deterministic, fully parallel.
// ------------------------------------------------------------------------------------------
// sum_k L_s[i,k] d_k L_s[j,k] of one pair, added to acc in k order.  The 12 operands of four k's are requested TOGETHER and the
// dependent fma chain runs afterwards (a plain loop waits for the memory round trip of every k).  Same products, same order of
// accumulation as the plain loop.  [Round 6, measured: this does NOT shorten cfg 2a's big gather launch (355 us for 4.0e6 entries,
// 5.5e6 pairs): that launch is bound by the number of scattered memory requests -- adjacent target entries take their operands from
// different small source panels -- not by the dependent chain of its longest thread.]
__device__ __forceinline__ double gath_pair_sum(const DevPlan &P, const GathPair &G, double acc) {
    const double *li = P.Lx + G.src;
    const double *lj = li + G.dj;
    const double *dv = P.D + G.dfirst;
    const int64_t r = G.r;
    const int K = G.K;
    for (int k0 = 0; k0 < K; k0 += 4) {
        double a[4], b[4], d[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kk = k0 + q < K ? k0 + q : K - 1;      // clamped: loads past the end repeat the last column and are not used
            a[q] = li[kk * r];
            b[q] = lj[kk * r];
            d[q] = dv[kk];
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (k0 + q < K) acc = fma(a[q] * d[q], b[q], acc);
    }
    return acc;
}
__global__ void __launch_bounds__(256)
k_update_gather(DevPlan P, int64_t ebegin, int64_t n) {
    // grid-stride: a launch on the side stream (hipkkt_factor.cpp enqueue_gather) is given a bounded grid so that it leaves wavefront
    // slots and registers on every compute unit to the small kernels of the main stream it runs next to
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p0 = P.gath_pptr[ebegin + e], p1 = P.gath_pptr[ebegin + e + 1];
        if (p1 - p0 > kGathHeavy || p1 <= p0) continue;        // k_update_gather_heavy
        double *tp = P.Lx + P.gath_tgt[ebegin + e];
        const double t0 = *tp;                                 // (requested next to the first record)
        double acc = 0.0;
        GathPair G = P.gath_pairs[p0];
        for (int64_t p = p0; p < p1; p++) {
            const GathPair Gn = P.gath_pairs[p + 1 < p1 ? p + 1 : p];   // the next record is on its way while this pair is summed
            acc = gath_pair_sum(P, G, acc);
            G = Gn;
        }
        *tp = t0 - acc;
    }
}
// target entries with long pair lists (a dense row / column of the root that every leaf touches: 1189 pairs on cfg 3) kept
// ONE thread busy for a millisecond while the rest of the launch had finished: one wavefront each, lane l takes the
// pairs l, l + 64, ... in order and the 64 partial sums are added in a fixed tree -- deterministic, like the thread version
__global__ void __launch_bounds__(256)
k_update_gather_heavy(DevPlan P, int64_t hbegin, int64_t n) {
    const int64_t h = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= n) return;
    const int lane = threadIdx.x & 63;
    const int64_t e = P.gath_heavy[hbegin + h];
    const int64_t p0 = P.gath_pptr[e], p1 = P.gath_pptr[e + 1];
    double acc = 0.0;
    for (int64_t p = p0 + lane; p < p1; p += 64) {
        const GathPair G = P.gath_pairs[p];
        acc = gath_pair_sum(P, G, acc);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) P.Lx[P.gath_tgt[e]] -= acc;
}

// probe used by hipkkt's self test: D = A(16x4) * B(4x16) through the same MFMA form and the
// same lane maps as k_update_stage; out[i*16+j] receives D[i][j].
__global__ void k_mfma_probe(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, lk = lane >> 4;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[l15 * 4 + lk], B[lk * 16 + l15], acc, 0, 0, 0);
    for (int reg = 0; reg < 4; reg++) out[(lk + 4 * reg) * 16 + l15] = acc[reg];
}

// ------------------------------------------------------------------------------------------
// K5: triangular solves, level-scheduled ("wavefront") over the supernodal elimination tree.
// forward:  each supernode gathers the updates its descendants left in ubuf (fixed order),
//           solves with the unit-lower L11 inside one wavefront, publishes y_J and z_J = y_J/D,
//           and leaves its own update vector  u_s = L21 * y_J  in ubuf.
// backward: partial dot-products  L21^T x_R  per 256-row block, then a per-supernode finaliser
//           (fixed summation order) + unit-upper solve with L11^T.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_permute_in(const double *__restrict__ b, const int *__restrict__ perm, double *__restrict__ y, int n, int *epoch,
             int *ticks, int nticks, int *zero, int nzero) {
    // first kernel of every LDL solve: a new epoch invalidates the tagged hand-off values of the previous solve
    if (blockIdx.x == 0 && threadIdx.x == 0 && epoch) ((unsigned *)epoch)[0] += 1u;   // wraps after 2^32 solves: any
                                                                                        // two consecutive epochs differ
    // ... and the ticket words of the segment sweeps start from zero.  They are cleared HERE, by a kernel that runs
    // no atomics, and live in cache lines of their own (seg_sync layout): a plain store next to a word that other
    // workgroups are incrementing atomically can lose increments when the writer's L2 writes the line back (seen on
    // MI355X: duplicate tickets => the last items of a launch silently never ran).
    if (blockIdx.x == 0) {
        for (int q = threadIdx.x; q < nticks; q += blockDim.x) ticks[q] = 0;
        // the norm words the refinement kernels BEHIND this solve accumulate into with atomicMax (a launch of their own until round 6)
        for (int q = threadIdx.x; q < nzero; q += blockDim.x) zero[q] = 0;
    }
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) y[k] = b[perm[k]];
}

// One refinement step of a marked block's forward solve against the factored block itself (k_invert_diag_wide explains):
// yv += Linv (rhs - L yv); rhs, yv, scratch in LDS ([w] each), every thread of the workgroup calls it.
__device__ __forceinline__ void polish_fwd(const DevPlan &P, int s, int w, const double *rhs, double *yv, double *scratch) {
    const double *ldg = P.Ldiag + P.sn_diag[s], *li = P.Linv + P.sn_diag[s];
    const int tid = threadIdx.x;
    if (tid < w) {
        double a = rhs[tid] - yv[tid];
        for (int k2 = 0; k2 < tid; k2++) a -= ldg[tid + k2 * w] * yv[k2];
        scratch[tid] = a;
    }
    __syncthreads();
    double c = 0.0;
    if (tid < w)
        for (int k2 = 0; k2 <= tid; k2++) c += li[tid + k2 * w] * scratch[k2];
    __syncthreads();
    if (tid < w) yv[tid] += c;
    __syncthreads();
}
// the same for the backward solve: xv += Linv^T (tv - L^T xv)
__device__ __forceinline__ void polish_bwd(const DevPlan &P, int s, int w, const double *tv, double *xv, double *scratch) {
    const double *ldg = P.Ldiag + P.sn_diag[s], *lit = P.LinvT + P.sn_diag[s];
    const int tid = threadIdx.x;
    if (tid < w) {
        double a = tv[tid] - xv[tid];
        for (int i2 = tid + 1; i2 < w; i2++) a -= ldg[i2 + tid * w] * xv[i2];
        scratch[tid] = a;
    }
    __syncthreads();
    double c = 0.0;
    if (tid < w)
        for (int i2 = tid; i2 < w; i2++) c += lit[tid + i2 * w] * scratch[i2];
    __syncthreads();
    if (tid < w) xv[tid] += c;
    __syncthreads();
}

// Latency model behind these kernels (measured on MI355X, tools/ubench.hip): a dependent f64 FMA
// costs 32 cycles, an LDS round trip 60, an L2 hit ~220, HBM/MALL ~650, and one CU pulls only
// ~10 B/clk from HBM.  Hence: every dot product runs on 4 independent accumulators, and a panel is
// spread over as many workgroups as possible (64 rows each) instead of a few 256-row blocks.
constexpr int kSlvRows = 64;

__global__ void __launch_bounds__(256)
k_fwd_level(DevPlan P, int item_begin, double *__restrict__ y, double *__restrict__ z) {
    __shared__ double rhs[kMaxSnWidth];
    __shared__ double part[4][kMaxSnWidth];
    __shared__ double yv[kMaxSnWidth];
    const FacItem it = P.slv_items[item_begin + blockIdx.x];
    const int s = it.sn;
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int64_t slot0 = P.sn_rowptr[s];
    const int r = (int)(P.sn_rowptr[s + 1] - slot0);
    const double *pan = P.Lx + P.sn_panel[s];
    const double *li = P.Linv + P.sn_diag[s];
    const int tid = threadIdx.x;
    const int i = tid & 63, pq = tid >> 6;
    // right-hand side of the diagonal block: b_J minus what the children pushed onto these rows
    if (tid < w) {
        double acc = 0.0;
        const int64_t g0 = P.g_ptr[slot0 + tid], g1 = P.g_ptr[slot0 + tid + 1];
        for (int64_t g = g0; g < g1; g++) acc += P.ubuf[P.g_idx[g]];
        rhs[tid] = y[f + tid] - acc;
    }
    // prefetch this thread's share of the row GEMV while the diagonal block is being solved
    const int row = w + it.blk * kSlvRows + i;
    double pv[16];
    double gsum = 0.0;
    if (row < r) {
        const double *pr = pan + row;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int k = pq + 4 * t;
            pv[t] = k < w ? pr[(int64_t)k * r] : 0.0;
        }
        if (pq == 0) {
            const int64_t g0 = P.g_ptr[slot0 + row], g1 = P.g_ptr[slot0 + row + 1];
            for (int64_t g = g0; g < g1; g++) gsum += P.ubuf[P.g_idx[g]];
        }
    }
    __syncthreads();
    {   // y_J = L11^-1 rhs as a small lower-triangular GEMV (explicit inverse from k_invert_diag)
        double a0 = 0.0, a1 = 0.0;
        if (i < w) {
            int k = pq;
            for (; k + 4 <= i; k += 8) {
                a0 += li[i + k * w] * rhs[k];
                a1 += li[i + (k + 4) * w] * rhs[k + 4];
            }
            if (k <= i) a0 += li[i + k * w] * rhs[k];
        }
        part[pq][i] = a0 + a1;
    }
    __syncthreads();
    if (tid < w) {
        yv[tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    } else if (tid < kMaxSnWidth) {
        yv[tid] = 0.0;   // padded columns multiply prefetched zeros: keep them finite
    }
    __syncthreads();
    if (P.sn_polish[s]) polish_fwd(P, s, w, rhs, yv, part[0]);           // (workgroup-uniform)
    if (tid < w && it.blk == 0) z[f + tid] = yv[tid] * P.Dinv[f + tid];   // y keeps the right-hand side: the other block items of
                                                                          // this supernode (same launch) may still have to read it
    // this panel's update vector = children's contributions passed through + L21 * y_J
    {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (row < r) {
#pragma unroll
            for (int t = 0; t < 16; t += 4) {
                a0 += pv[t] * yv[(pq + 4 * t) & 63];
                a1 += pv[t + 1] * yv[(pq + 4 * t + 4) & 63];
                a2 += pv[t + 2] * yv[(pq + 4 * t + 8) & 63];
                a3 += pv[t + 3] * yv[(pq + 4 * t + 12) & 63];
            }
        }
        __syncthreads();
        part[pq][i] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (pq == 0 && row < r)
        P.ubuf[P.u_off[s] + (row - w)] = gsum + (((part[0][i] + part[1][i]) + part[2][i]) + part[3][i]);
}

// partial L21^T x over one block of <=64 off-diagonal rows, from the row-major copy LT:
// lane = column (coalesced), the row's x value is a wave-uniform broadcast.  red[4][64] is LDS.
__device__ __forceinline__ double bwd_block_dot(const DevPlan &P, int s, int w, int r, int blk,
                                                const double *__restrict__ x, double (*red)[kMaxSnWidth]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int *rows = P.sn_rows + P.sn_rowptr[s];
    const double *lt = P.LT + P.lt_off[s];
    const int lo = w + blk * kSlvRows + wave * 16;
    const int hi = min(min(r, lo + 16), w + (blk + 1) * kSlvRows);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (lane < w) {
        int i = lo;
        for (; i + 4 <= hi; i += 4) {
            a0 += lt[(int64_t)(i - w) * w + lane] * x[rows[i]];
            a1 += lt[(int64_t)(i + 1 - w) * w + lane] * x[rows[i + 1]];
            a2 += lt[(int64_t)(i + 2 - w) * w + lane] * x[rows[i + 2]];
            a3 += lt[(int64_t)(i + 3 - w) * w + lane] * x[rows[i + 3]];
        }
        for (; i < hi; i++) a0 += lt[(int64_t)(i - w) * w + lane] * x[rows[i]];
    }
    red[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    return ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// K5 on a level of NARROW supernodes (w <= kNarrowW columns, <= kNarrowR rows below the diagonal block; the
// thousands of 1-2 column leaves of a sparse QP): one THREAD per supernode instead of one 256-thread workgroup --
// a level-0 launch of cfg 2a drops from 17884 workgroups (39 us) to 70.  Same arithmetic as k_fwd_level /
// k_bwd_final, sums taken in index order.
template <int NW>   // NW >= widest supernode of the launch (register arrays)
__global__ void __launch_bounds__(256)
k_fwd_narrow(DevPlan P, int sn_begin, int n, double *__restrict__ y, double *__restrict__ z) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int s = P.lvl_sn[sn_begin + t];
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int64_t slot0 = P.sn_rowptr[s];
    const int r = (int)(P.sn_rowptr[s + 1] - slot0);
    const double *pan = P.Lx + P.sn_panel[s];
    const double *li = P.Linv + P.sn_diag[s];
    double rhs[NW], yv[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) {
        rhs[k] = 0.0;
        if (k < w) {
            double acc = 0.0;
            for (int64_t g = P.g_ptr[slot0 + k]; g < P.g_ptr[slot0 + k + 1]; g++) acc += P.ubuf[P.g_idx[g]];
            rhs[k] = y[f + k] - acc;
        }
    }
#pragma unroll
    for (int i = 0; i < NW; i++) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k <= i; k++)
            if (i < w) v += li[i + k * w] * rhs[k];
        yv[i] = v;
        if (i < w) z[f + i] = v * P.Dinv[f + i];
    }
    double *u = P.ubuf + P.u_off[s];
    for (int row = w; row < r; row++) {
        double a = 0.0;
        for (int64_t g = P.g_ptr[slot0 + row]; g < P.g_ptr[slot0 + row + 1]; g++) a += P.ubuf[P.g_idx[g]];
#pragma unroll
        for (int k = 0; k < NW; k++)
            if (k < w) a += pan[row + (int64_t)k * r] * yv[k];
        u[row - w] = a;
    }
}

template <int NW>
__global__ void __launch_bounds__(256)
k_bwd_narrow(DevPlan P, int sn_begin, int n, const double *__restrict__ z, double *__restrict__ x, double *__restrict__ xout) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int s = P.lvl_sn[sn_begin + t];
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int64_t slot0 = P.sn_rowptr[s];
    const int r = (int)(P.sn_rowptr[s + 1] - slot0);
    const int *rows = P.sn_rows + slot0;
    const double *lt = P.LT + P.lt_off[s];          // row-major: lt[(i - w) * w + k]
    const double *lit = P.LinvT + P.sn_diag[s];     // Linv[i][k] at [k + i * w]
    double tv[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) tv[k] = k < w ? z[f + k] : 0.0;
    for (int i = w; i < r; i++) {
        const double xi = x[rows[i]];
#pragma unroll
        for (int k = 0; k < NW; k++)
            if (k < w) tv[k] -= lt[(int64_t)(i - w) * w + k] * xi;
    }
#pragma unroll
    for (int k = 0; k < NW; k++) {
        double v = 0.0;
#pragma unroll
        for (int i = k; i < NW; i++)
            if (i < w) v += lit[k + i * w] * tv[i];
        if (k < w) {
            x[f + k] = v;
            xout[P.perm[f + k]] = v;
        }
    }
}

// panels with more than one 64-row block: per-block partial sums
__global__ void __launch_bounds__(256)
k_bwd_partial(DevPlan P, int item_begin, const double *__restrict__ x) {
    __shared__ double red[4][kMaxSnWidth];
    const FacItem it = P.bwd_items[item_begin + blockIdx.x];
    const int s = it.sn;
    const int w = P.sn_first[s + 1] - P.sn_first[s];
    const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
    const double v = bwd_block_dot(P, s, w, r, it.blk, x, red);
    if (threadIdx.x < w) P.pbuf[P.p_off[s] + (int64_t)it.blk * w + threadIdx.x] = v;
}

__global__ void __launch_bounds__(256)
k_bwd_final(DevPlan P, int sn_begin, const double *__restrict__ z, double *__restrict__ x,
            double *__restrict__ xout) {
    __shared__ double red[4][kMaxSnWidth];
    __shared__ double tv[kMaxSnWidth];
    const int s = P.lvl_sn[sn_begin + blockIdx.x];
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
    const double *lit = P.LinvT + P.sn_diag[s];
    const int tid = threadIdx.x;
    const int k = tid & 63, pq = tid >> 6;
    const int nblk = (r - w + kSlvRows - 1) / kSlvRows;
    double acc = 0.0;
    if (nblk == 1) {
        acc = bwd_block_dot(P, s, w, r, 0, x, red);   // short panel: fused
        __syncthreads();
    } else if (nblk > 1) {
        double a = 0.0;
        if (k < w) {
            const double *pb = P.pbuf + P.p_off[s] + k;
            for (int b = pq; b < nblk; b += 4) a += pb[(int64_t)b * w];
        }
        red[pq][k] = a;
        __syncthreads();
        acc = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
        __syncthreads();
    }
    if (tid < w) tv[tid] = z[f + tid] - acc;
    __syncthreads();
    {   // x_J = L11^-T t :  x_k = sum_{i>=k} Linv[i][k] t_i ; LinvT holds Linv[i][k] at [k + i*w]
        double a0 = 0.0, a1 = 0.0;
        if (k < w) {
            int i = k + pq;
            for (; i + 4 < w; i += 8) {
                a0 += lit[k + i * w] * tv[i];
                a1 += lit[k + (i + 4) * w] * tv[i + 4];
            }
            if (i < w) a0 += lit[k + i * w] * tv[i];
        }
        red[pq][k] = a0 + a1;
    }
    __syncthreads();
    if (P.sn_polish[s]) {                                                 // (workgroup-uniform)
        __shared__ double xv[kMaxSnWidth];
        if (tid < w) xv[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        __syncthreads();
        polish_bwd(P, s, w, tv, xv, red[0]);
        if (tid < w) {
            x[f + tid] = xv[tid];
            xout[P.perm[f + tid]] = xv[tid];
        }
        return;
    }
    if (tid < w) {
        const double v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        x[f + tid] = v;
        xout[P.perm[f + tid]] = v;
    }
}

// ------------------------------------------------------------------------------------------
// K5 over a FRONT (a chain of np panels cut from one wide supernode, e.g. the dense root of a random
// sparse QP): ONE persistent launch per sweep instead of np dependent launches.  Workgroup b owns the
// b-th row block of the front, accumulates  sum_q L[b,q] y_q  over the panels q in order while the
// y_q are published by their owners through self-validating hand-off slots (below), then solves its
// own diagonal block and publishes.  The L blocks of step q+1 are prefetched while the workgroup
// waits for y_q, so a hop costs one hand-off + two 64x64 GEMVs (1.6 us measured; the sweep reads L
// once, 126 MB on cfg 2a, i.e. it also runs at ~0.9 TB/s of HBM traffic).
// Deadlock freedom: block indices are tickets taken in arrival order (a workgroup only ever waits for
// lower tickets, whose owners are already running); every spin is bounded and aborts through the
// front's error word, which makes the solve report a failure instead of hanging.
// Summation order is fixed (panel order, then a fixed 4-way tree): results are deterministic.
// ------------------------------------------------------------------------------------------
// (front_ld_flag / front_ld / front_st, the self-validating hand-off slots and front_slots(): sweep_common.h)

__global__ void __launch_bounds__(256)
k_front_fwd(DevPlan P, FrontDesc F, double *__restrict__ y, double *__restrict__ z) {
    __shared__ double red[4][64];
    __shared__ double tv[64];
    __shared__ double ybuf[2][64];
    __shared__ int sb, okflag;
    int *sync = P.front_sync + F.sync_off;
    if (threadIdx.x == 0) sb = atomicAdd(sync, 1);
    __syncthreads();
    const int b = sb;
    if (b >= F.nb) return;
    // re-arm the backward sweep's block (idle during this launch), every workgroup a share
    for (int q = b * 256 + threadIdx.x; q < F.sync_blk; q += F.nb * 256) sync[F.sync_blk + q] = 0;
    FrontSlot *slots = front_slots(sync, F.np);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const FrontPanel *fps = P.front_panels + F.fp_off;
    const bool own = b < F.np;
    FrontPanel me = fps[own ? b : 0];
    const int i0 = own ? F.cw * b : F.W + 64 * (b - F.np);
    const int nrows = own ? me.w : min(64, F.rF - i0);
    const int i = i0 + lane;
    const bool valid = lane < nrows;
    // this row's start value: own rows  b_i - (external children), rows below the front  + (external children)
    double base = 0.0;
    if (wv == 0 && valid) {
        double G = 0.0;
        const int64_t g0 = P.front_gptr[F.gptr_off + i], g1 = P.front_gptr[F.gptr_off + i + 1];
        for (int64_t g = g0; g < g1; g++) G += P.ubuf[P.front_gidx[g]];
        base = own ? y[me.f + lane] - G : G;
    }
    const double dinv_own = (own && valid) ? P.Dinv[me.f + lane] : 0.0;   // static: fetched before any wait
    // own diagonal block: row `lane` of Linv, columns j = wv + 4t (lower triangle)
    double li[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int j = wv + 4 * t;
        li[t] = (own && valid && j <= lane) ? P.Linv[me.diag_off + lane + j * me.w] : 0.0;
    }
    const int nq = own ? b : F.np;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    FrontPanel fq = fps[0];
    double lv[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int k = wv + 4 * t;
        lv[t] = (nq > 0 && valid && k < fq.w) ? P.Lx[fq.panel_off + i + (int64_t)k * fq.r] : 0.0;
    }
    bool ok = true;
    for (int q = 0; q < nq; q++) {
        FrontPanel fn = fq;
        double ln[16];
        if (q + 1 < nq) {                      // prefetch the next panel's block before waiting
            fn = fps[q + 1];
            const int jl = i - F.cw * (q + 1);
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int k = wv + 4 * t;
                ln[t] = (valid && k < fn.w) ? P.Lx[fn.panel_off + jl + (int64_t)k * fn.r] : 0.0;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; t++) ln[t] = 0.0;
        }
        // ONE wave polls the panel's hand-off slots (value and validity in one 16-byte load per lane); the
        // workgroup reads y_q from LDS (double buffered by hop parity: one barrier per hop)
        double *yb = ybuf[q & 1];
        if (wv == 0) {
            double yv = 0.0;
            ok = front_slot_wait(slots + q * 64 + lane, yv, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
            yb[lane] = ok ? yv : 0.0;
            if (lane == 0) okflag = ok ? 1 : 0;
        }
        __syncthreads();
        if (!okflag) { ok = false; break; }
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
            a0 = fma(lv[t], yb[(wv + 4 * t) & 63], a0);
            a1 = fma(lv[t + 1], yb[(wv + 4 * t + 4) & 63], a1);
            a2 = fma(lv[t + 2], yb[(wv + 4 * t + 8) & 63], a2);
            a3 = fma(lv[t + 3], yb[(wv + 4 * t + 12) & 63], a3);
        }
        fq = fn;
#pragma unroll
        for (int t = 0; t < 16; t++) lv[t] = ln[t];
    }
    red[wv][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    const double tot = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (!own) {
        if (wv == 0 && valid && ok) P.ubuf[F.ubelow_off + (i - F.W)] = base + tot;
        return;
    }
    if (wv == 0) tv[lane] = valid ? base - tot : 0.0;
    __syncthreads();
    {   // y_J = Linv * t  (lower-triangular GEMV, 4-way split over j)
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            s0 = fma(li[t], tv[wv + 4 * t], s0);
            s1 = fma(li[t + 1], tv[wv + 4 * t + 4], s1);
        }
        red[wv][lane] = s0 + s1;
    }
    __syncthreads();
    if (wv == 0) {
        const double v = valid ? ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane] : 0.0;
        if (ok) front_slot_st(slots + b * 64 + lane, v);     // first: the next panel's owner is waiting for it
        if (valid && ok) z[me.f + lane] = v * dinv_own;
    }
}

// backward sweep over a front: ticket t owns panel p = np-1-t:  x_J = L_JJ^-T (z_J - L_RJ^T x_R), where R
// = the rows below the front (final before the launch) followed by the later panels' columns in
// descending panel order (published inside the launch).  lane = column, wave = 16-row slice.
__global__ void __launch_bounds__(256)
k_front_bwd(DevPlan P, FrontDesc F, const double *__restrict__ z, double *__restrict__ x, double *__restrict__ xout) {
    __shared__ double red[4][64];
    __shared__ double tv[64];
    __shared__ double xbuf[2][64];
    __shared__ int sb, okflag;
    int *sync = P.front_sync + F.sync_off + F.sync_blk;
    if (threadIdx.x == 0) sb = atomicAdd(sync, 1);
    __syncthreads();
    if (sb >= F.np) return;
    // re-arm the forward sweep's block for the next solve (idle during this launch), every workgroup a share
    for (int q = sb * 256 + threadIdx.x; q < F.sync_blk; q += F.np * 256) sync[q - F.sync_blk] = 0;
    FrontSlot *slots = front_slots(sync, F.np);
    const int p = F.np - 1 - sb;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const FrontPanel *fps = P.front_panels + F.fp_off;
    const FrontPanel me = fps[p];
    const int w = me.w;
    const bool cvalid = lane < w;
    const double *lt = P.LT + me.lt_off;            // row-major: lt[(j - w) * w + k], j = local panel row
    const int perm_own = cvalid ? P.perm[me.f + lane] : 0;                // static: fetched before any wait
    // own diagonal block: column `lane` of Linv (as stored transposed), rows i2 = wv + 4t >= lane
    double lit[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int i2 = wv + 4 * t;
        lit[t] = (cvalid && i2 >= lane && i2 < w) ? P.LinvT[me.diag_off + lane + i2 * w] : 0.0;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    // (A) rows below the front: x is final
    {
        const int *rows = P.sn_rows + F.rows_off;
        for (int ib = F.W; ib < F.rF; ib += 64) {
            const int lo = ib + 16 * wv;
            double xv[16], l[16];
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int i = lo + t;
                const bool in = i < F.rF && i < ib + 64;
                xv[t] = in ? x[rows[in ? i : F.W]] : 0.0;
                l[t] = (in && cvalid) ? lt[(int64_t)(i - F.cw * p - w) * w + lane] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 16; t += 4) {
                a0 = fma(l[t], xv[t], a0);
                a1 = fma(l[t + 1], xv[t + 1], a1);
                a2 = fma(l[t + 2], xv[t + 2], a2);
                a3 = fma(l[t + 3], xv[t + 3], a3);
            }
        }
    }
    // (B) later panels, descending
    bool ok = true;
    double lv[16];
    {
        const int q = F.np - 1;
        const FrontPanel fq = fps[q];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int jr = 16 * wv + t;
            lv[t] = (q > p && cvalid && jr < fq.w) ? lt[(int64_t)(F.cw * (q - p) + jr - w) * w + lane] : 0.0;
        }
    }
    for (int q = F.np - 1; q > p; q--) {
        double ln[16];
        if (q - 1 > p) {
            const FrontPanel fn = fps[q - 1];
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const int jr = 16 * wv + t;
                ln[t] = (cvalid && jr < fn.w) ? lt[(int64_t)(F.cw * (q - 1 - p) + jr - w) * w + lane] : 0.0;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; t++) ln[t] = 0.0;
        }
        double *xb = xbuf[q & 1];
        if (wv == 0) {   // one polling wave, broadcast through LDS (see k_front_fwd)
            double xv = 0.0;
            ok = front_slot_wait(slots + q * 64 + lane, xv, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
            xb[lane] = ok ? xv : 0.0;
            if (lane == 0) okflag = ok ? 1 : 0;
        }
        __syncthreads();
        if (!okflag) { ok = false; break; }
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
            a0 = fma(lv[t], xb[16 * wv + t], a0);
            a1 = fma(lv[t + 1], xb[16 * wv + t + 1], a1);
            a2 = fma(lv[t + 2], xb[16 * wv + t + 2], a2);
            a3 = fma(lv[t + 3], xb[16 * wv + t + 3], a3);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) lv[t] = ln[t];
    }
    red[wv][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wv == 0) tv[lane] = cvalid ? z[me.f + lane] - (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) : 0.0;
    __syncthreads();
    {   // x_J = Linv^T t :  x_k = sum_{i >= k} Linv[i][k] t_i
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            s0 = fma(lit[t], tv[wv + 4 * t], s0);
            s1 = fma(lit[t + 1], tv[wv + 4 * t + 4], s1);
        }
        red[wv][lane] = s0 + s1;
    }
    __syncthreads();
    if (wv == 0) {
        const double v = cvalid ? ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane] : 0.0;
        if (ok) front_slot_st(slots + p * 64 + lane, v);     // first: the previous panel's owner is waiting for it
        if (cvalid && ok) {
            x[me.f + lane] = v;
            xout[perm_own] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K5 over the REGULAR supernodes (everything that is not a front panel): level-free persistent sweeps.
// One launch per segment (= the levels between two front kernels) instead of one launch per level
// and kernel: workgroup tickets are handed out in level order, an item first waits for the completion
// counters of the supernodes it depends on (forward: its same-segment children; backward: its parent and,
// for the finaliser, the partial blocks of its own panel), then runs the same arithmetic as the
// level-scheduled kernels.  Data produced inside the launch (update vectors, partial sums, x) is written
// with agent-scope stores and read with agent-scope loads; counters are device-scope atomics.
// Deadlock freedom and bounded spins: as for the front kernels.  The forward launch re-arms the backward
// counters and vice versa.
// ------------------------------------------------------------------------------------------
// fdone: kSegSub arrays of nsuper counters.  The items of a child count themselves into its parent's counters, block b into array
// b mod kSegSub: the device serialises atomics on one word at ~11 ns each, and a supernode high in the tree of a random QP waits for
// the ~300 blocks of its children (3 us per level on one word).  The waiting thread adds the kSegSub words up.  Only for supernodes
// that wait for at least kSegSubMin blocks: eight polled words instead of one made the sweeps of the banded QP (2400 waiting
// workgroups, a few blocks per child) 5 % slower.
constexpr int kSegSub = 8;
constexpr int kSegSubMin = 64;
struct SegSync {
    int *ftick, *btick, *fdone, *err;
};
__device__ __forceinline__ SegSync seg_sync(const DevPlan &P, int nsuper) {
    SegSync s;
    s.ftick = P.seg_sync;                                 // ticket words: [0, 2 nseg), padded to whole 128-byte lines
    s.btick = P.seg_sync + P.nseg;
    s.fdone = P.seg_sync + ((2 * P.nseg + 31) & ~31);     // (mirrored by seg_sync_ints() in hipkkt_internal.h)
    s.err = s.fdone + (int64_t)kSegSub * nsuper;
    return s;
}
// thread 0 waits until the kSegSub counters of supernode s add up to `want` (relaxed polls); false on time-out / foreign failure
__device__ __forceinline__ bool seg_wait_sum(int *ctr, int stride, int want, int *err, int *failflag, unsigned lim) {
    for (unsigned spins = 0;; spins++) {
        int v[kSegSub];
#pragma unroll
        for (int k = 0; k < kSegSub; k++) v[k] = front_ld_flag(ctr + (int64_t)k * stride);
        int sum = 0;
#pragma unroll
        for (int k = 0; k < kSegSub; k++) sum += v[k];
        if (sum >= want) return true;
        if ((spins & 127u) == 127u || lim < 128u) {
            if (spins > lim) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(failflag, 1);
                return false;
            }
            if (front_ld_flag(err) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// thread 0 waits until *ctr >= want (relaxed polls); false on time-out / foreign failure
__device__ __forceinline__ bool seg_wait(int *ctr, int want, int *err, int *failflag, unsigned lim) {
    for (unsigned spins = 0;; spins++) {
        if (front_ld_flag(ctr) >= want) return true;
        if ((spins & 127u) == 127u || lim < 128u) {
            if (spins > lim) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(failflag, 1);
                return false;
            }
            if (front_ld_flag(err) != 0) return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// Tagged hand-off inside a segment sweep: a value produced by one item and consumed by a later item of the SAME
// launch travels as a self-validating 16-byte slot (see FrontSlot) next to its plain copy, keyed with the solve's
// epoch so that the previous solve's slots are invalid without any re-arming.  The consumer polls the very values
// it needs -- no dependency counter, no "stores complete" wait before an atomic, no second round trip.
__device__ __forceinline__ unsigned long long seg_key(const DevPlan &P) {
    return kSlotKey ^ ((unsigned long long)(((const unsigned *)P.seg_epoch)[0] + 1u) * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ void seg_slot_st(FrontSlot *p, double v, unsigned long long key) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), h = b ^ key;
    v4u r = {(unsigned)b, (unsigned)(b >> 32), (unsigned)h, (unsigned)(h >> 32)};
    // s_nop: the data VGPRs of a store wider than 64 bits must not be overwritten in the next wait states (the
    // compiler's hazard recogniser does not look inside inline asm and re-used them for an address right away)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(r) : "memory");
}
// one lane waits for one slot; false = timed out / another workgroup failed (bounded like every other spin)
__device__ __forceinline__ bool seg_slot_poll(const FrontSlot *p, unsigned long long key, double &v, int *err, int *failflag, unsigned lim) {
    for (unsigned spins = 0;; spins++) {
        const FrontSlot s = front_slot_ld(p);
        if (((unsigned long long)__double_as_longlong(s.v) ^ s.h) == key) { v = s.v; return true; }
        if ((spins & 127u) == 127u || lim < 128u) {
            if (spins > lim) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(failflag, 1);
                return false;
            }
            if (front_ld_flag(err) != 0) return false;
        }
    }
}
__device__ __forceinline__ bool seg_spin_check(unsigned &spins, int *err, int *failflag, unsigned lim) {   // false = give up
    if ((++spins & 127u) == 0u || lim < 128u) {
        if (spins > lim) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(failflag, 1);
            return false;
        }
        if (front_ld_flag(err) != 0) return false;
    }
    return true;
}
constexpr int kSegRowTag = 0x40000000;   // rows_seg: the row's x comes from an item of the same launch

__global__ void __launch_bounds__(256)
k_fwd_seg(DevPlan P, int seg, int item_begin, int nitems, int nsuper, int first_launch, double *__restrict__ y,
          double *__restrict__ z) {
    __shared__ double rhs[kMaxSnWidth];
    __shared__ double part[4][kMaxSnWidth];
    __shared__ double yv[kMaxSnWidth];
    __shared__ int sb;
    const SegSync Y = seg_sync(P, nsuper);
    const int tid = threadIdx.x;
    // item = a ticket taken in arrival order (level order): an item only ever waits for items with lower tickets, whose
    // owners are therefore already running -- forward progress does not depend on the order in which the hardware
    // dispatches workgroups.  One atomic per workgroup (~11 ns each on one word); the wide bottom levels, where that
    // would add up, are not part of the persistent launches (hipkkt_setup.cpp kPersistMaxItems).  seg_ticket bit 0 clear selects
    // item = blockIdx (in-order dispatch assumed; kept for A/B timing only: HIPKKT_SEG_TICKET=0; bit 1 = backward sweep).
    if (P.seg_ticket & 1) {
        if (tid == 0) sb = atomicAdd(Y.ftick + seg, 1);
        __syncthreads();
    }
    const int t = (P.seg_ticket & 1) ? sb : (int)blockIdx.x;
    if (P.seg_ticket & 1) __syncthreads();   // sb is reused below
    if (t >= nitems) return;
    const FacItem it = P.slv_items[item_begin + t];
    const int s = it.sn;
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int64_t slot0 = P.sn_rowptr[s];
    const int r = (int)(P.sn_rowptr[s + 1] - slot0);
    const double *pan = P.Lx + P.sn_panel[s];
    const double *li = P.Linv + P.sn_diag[s];
    const int i = tid & 63, pq = tid >> 6;
    const int row = w + it.blk * kSlvRows + i;
    // ---- everything that does not depend on other workgroups is fetched BEFORE the dependency wait:
    //      Linv row, panel row, gather-list bounds and first indices, right-hand side
    double liv[16], pv[16];
#pragma unroll
    for (int t2 = 0; t2 < 16; t2++) {
        const int k = pq + 4 * t2;
        liv[t2] = (i < w && k <= i) ? li[i + k * w] : 0.0;
        pv[t2] = (row < r && k < w) ? pan[row + (int64_t)k * r] : 0.0;
    }
    // (kSegGath indices per slot: a row of a supernode high in the tree collects one entry per child that reaches it -- up to 12 on
    // the random QPs of the benchmark -- and every entry past the prefetched ones costs two dependent round trips, index then value,
    // on the critical path of the level)
    constexpr int kSegGath = 12;
    int64_t ga0 = 0, ga1 = 0, gb0 = 0, gb1 = 0;
    int ia[kSegGath], ib[kSegGath];                   // the first gather indices of this thread's slots
#pragma unroll
    for (int q = 0; q < kSegGath; q++) { ia[q] = 0; ib[q] = 0; }
    double yin = 0.0, dinv_own = 0.0;
    if (tid < w) {
        dinv_own = P.Dinv[f + tid];
        ga0 = P.g_ptr[slot0 + tid];
        ga1 = P.g_ptr[slot0 + tid + 1];
#pragma unroll
        for (int q = 0; q < kSegGath; q++)
            if (ga0 + q < ga1) ia[q] = P.g_idx[ga0 + q];
        yin = y[f + tid];
    }
    if (pq == 0 && row < r) {
        gb0 = P.g_ptr[slot0 + row];
        gb1 = P.g_ptr[slot0 + row + 1];
#pragma unroll
        for (int q = 0; q < kSegGath; q++)
            if (gb0 + q < gb1) ib[q] = P.g_idx[gb0 + q];
    }
    // ---- dependencies: every block of every same-segment child has been published; the children count
    //      themselves into THIS supernode's counter, so the wait is one poll loop on one word
    const int want = P.dep_total[s], fpar = P.sn_bparent[s];
    const int fsub = (fpar >= 0 && P.dep_total[fpar] >= kSegSubMin) ? (it.blk & (kSegSub - 1)) : 0;   // (read before the wait)
    if (tid == 0) {
        sb = (want == 0 || (want >= kSegSubMin ? seg_wait_sum(Y.fdone + s, nsuper, want, Y.err, P.flags + FL_FRONTFAIL, P.spin_limit)
                                               : seg_wait(Y.fdone + s, want, Y.err, P.flags + FL_FRONTFAIL, P.spin_limit))) ? 1 : 0;
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (!sb) return;
    // the sums run left to right over the list (the first four always take part, absent entries as 0.0: the order of rounds 2 - 5)
    if (tid < w) {
        double u[kSegGath];
#pragma unroll
        for (int q = 0; q < kSegGath; q++) u[q] = ga0 + q < ga1 ? front_ld(P.ubuf + ia[q]) : 0.0;   // in flight together
        double acc = ((u[0] + u[1]) + u[2]) + u[3];
#pragma unroll
        for (int q = 4; q < kSegGath; q++)
            if (ga0 + q < ga1) acc += u[q];
        for (int64_t g = ga0 + kSegGath; g < ga1; g++) acc += front_ld(P.ubuf + P.g_idx[g]);
        rhs[tid] = yin - acc;
    } else if (tid < kMaxSnWidth) {
        rhs[tid] = 0.0;   // padded columns meet zero weights: keep them finite
    }
    double gsum = 0.0;
    if (pq == 0 && row < r) {
        double u[kSegGath];
#pragma unroll
        for (int q = 0; q < kSegGath; q++) u[q] = gb0 + q < gb1 ? front_ld(P.ubuf + ib[q]) : 0.0;
        gsum = ((u[0] + u[1]) + u[2]) + u[3];
#pragma unroll
        for (int q = 4; q < kSegGath; q++)
            if (gb0 + q < gb1) gsum += u[q];
        for (int64_t g = gb0 + kSegGath; g < gb1; g++) gsum += front_ld(P.ubuf + P.g_idx[g]);
    }
    __syncthreads();
    {   // y_J = L11^-1 rhs (explicit inverse; thread (i, pq) owns the columns k = pq + 4t <= i)
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int t2 = 0; t2 < 16; t2 += 2) {
            a0 += liv[t2] * rhs[(pq + 4 * t2) & 63];
            a1 += liv[t2 + 1] * rhs[(pq + 4 * t2 + 4) & 63];
        }
        part[pq][i] = a0 + a1;
    }
    __syncthreads();
    if (tid < w) {
        yv[tid] = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    } else if (tid < kMaxSnWidth) {
        yv[tid] = 0.0;
    }
    __syncthreads();
    if (P.sn_polish[s]) polish_fwd(P, s, w, rhs, yv, part[0]);   // (workgroup-uniform)
    if (tid < w && it.blk == 0) z[f + tid] = yv[tid] * dinv_own;   // y is never overwritten (see k_fwd_level)
    {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int t2 = 0; t2 < 16; t2 += 4) {
            a0 += pv[t2] * yv[(pq + 4 * t2) & 63];
            a1 += pv[t2 + 1] * yv[(pq + 4 * t2 + 4) & 63];
            a2 += pv[t2 + 2] * yv[(pq + 4 * t2 + 8) & 63];
            a3 += pv[t2 + 3] * yv[(pq + 4 * t2 + 12) & 63];
        }
        __syncthreads();
        part[pq][i] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (pq == 0 && row < r)
        front_st(P.ubuf + P.u_off[s] + (row - w), gsum + (((part[0][i] + part[1][i]) + part[2][i]) + part[3][i]));
    // publish: the storing wave drains its stores, then one device-scope increment
    if (pq == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && fpar >= 0) atomicAdd(Y.fdone + (int64_t)fsub * nsuper + fpar, 1);
}

// backward: item.blk >= 0 = partial dot products of one 64-row block of a long panel, item.blk == -1 = the
// supernode's finaliser (short panels: fused dot product)
__global__ void __launch_bounds__(256)
k_bwd_seg(DevPlan P, int seg, int item_begin, int nitems, int nsuper, int first_launch, const double *__restrict__ z,
          double *__restrict__ x, double *__restrict__ xout) {
    __shared__ double red[4][kMaxSnWidth];
    __shared__ double tv[kMaxSnWidth];
    __shared__ int bad;
    const SegSync Y = seg_sync(P, nsuper);
    const int tid = threadIdx.x;
    __shared__ int tick;
    if (tid == 0) { bad = 0; tick = (P.seg_ticket & 2) ? atomicAdd(Y.btick + seg, 1) : (int)blockIdx.x; }   // see k_fwd_seg
    __syncthreads();                // `bad` may be set by any wave from here on
    const int t = tick;
    if (t >= nitems) return;
    if (first_launch) {             // re-arm the forward sweep's counters for the next solve, every workgroup a share
        for (int q = t * 256 + tid; q < kSegSub * nsuper; q += nitems * 256) Y.fdone[q] = 0;
    }
    const unsigned long long key = seg_key(P);
    const FacItem it = P.pbwd_items[item_begin + t];
    const int s = it.sn;
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
    const int nblk = (r - w + kSlvRows - 1) / kSlvRows;
    const int lane = tid & 63, wave = tid >> 6;
    const int *rows = P.rows_seg + P.sn_rowptr[s];   // sn_rows with kSegRowTag on the rows solved inside this launch
    const double *lt = P.LT + P.lt_off[s];
    // ---- static data first: the 16 rows x 1 column of L21^T this thread multiplies, their row indices (and the x
    //      values that were final before this launch), the column of Linv^T and z (finaliser)
    const bool dot_here = it.blk >= 0 || nblk == 1;
    const int blk = it.blk >= 0 ? it.blk : 0;
    const int lo = w + blk * kSlvRows + wave * 16;
    double lv[16], xv[16];
    int ri[16];
#pragma unroll
    for (int t2 = 0; t2 < 16; t2++) {
        const int i = lo + t2;
        const bool in = dot_here && i < r && i < w + (blk + 1) * kSlvRows;
        ri[t2] = in ? rows[i] : -1;
        lv[t2] = (in && lane < w) ? lt[(int64_t)(i - w) * w + lane] : 0.0;
        xv[t2] = (ri[t2] >= 0 && !(ri[t2] & kSegRowTag)) ? x[ri[t2]] : 0.0;
    }
    int ri_lane = -1;   // row lo + (lane & 15) of this wave's 16 rows, for the tagged polls
    {
        const int i = lo + (lane & 15);
        if (dot_here && i < r && i < w + (blk + 1) * kSlvRows) ri_lane = rows[i];
    }
    double litv[16];
    double zin = 0.0;
    int perm_own = 0;
    if (it.blk < 0) {
        if (tid < w) perm_own = P.perm[f + tid];
        const double *lit = P.LinvT + P.sn_diag[s];
#pragma unroll
        for (int t2 = 0; t2 < 16; t2++) {
            const int i2 = wave + 4 * t2;
            litv[t2] = (lane < w && i2 >= lane && i2 < w) ? lit[lane + i2 * w] : 0.0;
        }
        if (tid < w) zin = z[f + tid];
    } else {
#pragma unroll
        for (int t2 = 0; t2 < 16; t2++) litv[t2] = 0.0;
    }
    // ---- dependencies = the tagged x values (ancestors solved inside this launch) and partial sums themselves
    bool ok = true;
    double dotv = 0.0;
    if (dot_here) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        {   // lane l < 16 polls the slot of row lo + l (ONE load instruction per round for the wave's 16 rows); the
            // values are then broadcast to all lanes
            const int rl = ri_lane;                       // this lane's row (tagged index) or -1
            const bool mine = lane < 16 && rl >= 0 && (rl & kSegRowTag);
            double myv = 0.0;
            bool got = !mine;
            unsigned spins = 0;
            while (ok) {
                if (!got) {
                    const FrontSlot sl = front_slot_ld(P.xseg + (rl & ~kSegRowTag));
                    if (((unsigned long long)__double_as_longlong(sl.v) ^ sl.h) == key) { myv = sl.v; got = true; }
                }
                if (__ballot(!got) == 0ull) break;
                ok = seg_spin_check(spins, Y.err, P.flags + FL_FRONTFAIL, P.spin_limit);
            }
#pragma unroll
            for (int t2 = 0; t2 < 16; t2++) {
                const double bv = readlane_f64(myv, t2);
                if (ri[t2] >= 0 && (ri[t2] & kSegRowTag)) xv[t2] = bv;
            }
        }
        if (!ok) { bad = 1; atomicOr(P.flags + FL_FRONTFAIL, 4); }   // bit 2: the backward segment sweep gave up
#pragma unroll
        for (int t2 = 0; t2 < 16; t2 += 4) {
            a0 = fma(lv[t2], xv[t2], a0);
            a1 = fma(lv[t2 + 1], xv[t2 + 1], a1);
            a2 = fma(lv[t2 + 2], xv[t2 + 2], a2);
            a3 = fma(lv[t2 + 3], xv[t2 + 3], a3);
        }
        red[wave][lane] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (bad) return;
        dotv = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        __syncthreads();
    }
    if (it.blk >= 0) {   // partial item: its w sums travel to the finaliser as tagged slots
        if (tid < w) seg_slot_st(P.pseg + P.p_off[s] + (int64_t)it.blk * w + tid, dotv, key);
        return;
    }
    double acc = dotv;
    if (nblk > 1) {
        double a = 0.0;
        if (lane < w) {
            // kSegPoll slots per wavefront and round trip (round 6: one dependent poll per block made the finaliser of a 4500-row
            // panel wait 18 round trips after the last partial sum had arrived); the sum still runs over b2 = wave, wave + 4, ... in order
            constexpr int kSegPoll = 8;
            const FrontSlot *pb = P.pseg + P.p_off[s] + lane;
            for (int b0 = wave; b0 < nblk && ok; b0 += 4 * kSegPoll) {
                double v[kSegPoll];
                bool got[kSegPoll];
#pragma unroll
                for (int u = 0; u < kSegPoll; u++) { v[u] = 0.0; got[u] = b0 + 4 * u >= nblk; }
                unsigned spins = 0;
                while (true) {
                    bool all = true;
#pragma unroll
                    for (int u = 0; u < kSegPoll; u++) {
                        if (!got[u]) {
                            const FrontSlot sl = front_slot_ld(pb + (int64_t)(b0 + 4 * u) * w);
                            if (((unsigned long long)__double_as_longlong(sl.v) ^ sl.h) == key) { v[u] = sl.v; got[u] = true; }
                            else all = false;
                        }
                    }
                    if (all) break;
                    ok = seg_spin_check(spins, Y.err, P.flags + FL_FRONTFAIL, P.spin_limit);
                    if (!ok) break;
                }
#pragma unroll
                for (int u = 0; u < kSegPoll; u++)
                    if (b0 + 4 * u < nblk) a += v[u];
            }
        }
        if (!ok) { bad = 1; atomicOr(P.flags + FL_FRONTFAIL, 4); }   // bit 2: the backward segment sweep gave up
        red[wave][lane] = a;
        __syncthreads();
        if (bad) return;
        acc = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        __syncthreads();
    }
    if (tid < kMaxSnWidth) tv[tid] = tid < w ? zin - acc : 0.0;
    __syncthreads();
    {   // x_J = Linv^T t
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int t2 = 0; t2 < 16; t2 += 2) {
            s0 = fma(litv[t2], tv[(wave + 4 * t2) & 63], s0);
            s1 = fma(litv[t2 + 1], tv[(wave + 4 * t2 + 4) & 63], s1);
        }
        red[wave][lane] = s0 + s1;
    }
    __syncthreads();
    if (P.sn_polish[s]) {                                                 // (workgroup-uniform; k_invert_diag_wide explains)
        __shared__ double xv[kMaxSnWidth];
        if (tid < w) xv[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        __syncthreads();
        polish_bwd(P, s, w, tv, xv, red[0]);
        if (tid < w) {
            seg_slot_st(P.xseg + f + tid, xv[tid], key);
            x[f + tid] = xv[tid];
            xout[perm_own] = xv[tid];
        }
        return;
    }
    if (tid < w) {   // publish: tagged slot for the descendants solved in this launch, plain copy for everybody else
        const double v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        seg_slot_st(P.xseg + f + tid, v, key);
        x[f + tid] = v;
        xout[perm_own] = v;
    }
}

// ------------------------------------------------------------------------------------------
// K6-K8: iterative refinement pieces (ref: kktsolver_directldl.jl:389-466)
// e = b - K*xi with the symmetric CSR view of the unregularised K; 8 lanes per row
// ------------------------------------------------------------------------------------------
// Rows longer than kLongRow entries (a budget row 1'x = 1, the expansion columns of big cones, dense PSD blocks) would keep
// their 4 lanes busy for thousands of sequential iterations (cfg 3: 0.3 ms per SpMV for 70 000 nonzeros): they are left
// out here and taken by k_spmv_long, one workgroup each.
constexpr int kLongRow = 192;
__device__ __forceinline__ void spmv_short_rows(const int64_t *__restrict__ rowptr, const int *__restrict__ col,
                                                const int64_t *__restrict__ qidx, const double *__restrict__ kval,
                                                const double *__restrict__ b, const double *__restrict__ xi, double *__restrict__ e,
                                                int n, unsigned long long *__restrict__ norm_slot, const double *__restrict__ dacc) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gid >> 2, sub = gid & 3;
    double acc = 0.0;
    bool mine = false;
    if (row < n) {
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        mine = p1 - p0 <= kLongRow;
        if (mine)
            for (int64_t p = p0 + sub; p < p1; p += 4) acc += kval[qidx[p]] * xi[col[p]];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    double a = 0.0;
    if (mine && sub == 0) {
        if (dacc) acc += dacc[row];         // the row's part inside a dense triangle (k_spmv_dense_tri; 0 for every other row)
        const double ev = b[row] - acc;
        e[row] = ev;
        a = fabs(ev);
    }
    block_atomic_max_abs(norm_slot, a);
}
// one workgroup per long row: fixed partition of the entries over the 256 threads, fixed reduction tree (deterministic)
__device__ __forceinline__ void spmv_long_row(int row, const int64_t *__restrict__ rowptr, const int *__restrict__ col,
                                              const int64_t *__restrict__ qidx, const double *__restrict__ kval,
                                              const double *__restrict__ b, const double *__restrict__ xi, double *__restrict__ e,
                                              unsigned long long *__restrict__ norm_slot, const double *__restrict__ dacc) {
    __shared__ double red[4];
    const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
    double acc = 0.0;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) acc += kval[qidx[p]] * xi[col[p]];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ev = b[row] - ((((red[0] + red[1]) + red[2]) + red[3]) + (dacc ? dacc[row] : 0.0));
        e[row] = ev;
        atomic_max_abs(norm_slot, ev);
    }
}
__global__ void __launch_bounds__(256)
k_spmv_residual(const int64_t *__restrict__ rowptr, const int *__restrict__ col, const int64_t *__restrict__ qidx,
                const double *__restrict__ kval, const double *__restrict__ b, const double *__restrict__ xi,
                double *__restrict__ e, int n, unsigned long long *__restrict__ norm_slot, const double *__restrict__ dacc) {
    spmv_short_rows(rowptr, col, qidx, kval, b, xi, e, n, norm_slot, dacc);
}
__global__ void __launch_bounds__(256)
k_spmv_long(const int *__restrict__ long_rows, const int64_t *__restrict__ rowptr, const int *__restrict__ col,
            const int64_t *__restrict__ qidx, const double *__restrict__ kval, const double *__restrict__ b,
            const double *__restrict__ xi, double *__restrict__ e, unsigned long long *__restrict__ norm_slot,
            const double *__restrict__ dacc) {
    spmv_long_row(long_rows[blockIdx.x], rowptr, col, qidx, kval, b, xi, e, norm_slot, dacc);
}

// The dense triangles of K that left the symmetric view (symbolic.h HostPlan::dtri: the packed upper triangles of PSD-cone Hs blocks,
// column c0 + j = rows c0 .. c0 + j, contiguous in kval from col_off[j]): acc_out[c0 + i] = sum_j H(i, j) x[c0 + j] of the symmetric H.
// One workgroup (8 wavefronts) per 64 rows i0 .. i0 + 63 of a triangle; every stored entry is read twice, both times coalesced:
//   row part     sum_{j >= i} H(i, j) x_j: lane = row, the wavefronts take the columns j = i0 + w, i0 + w + 8, ...  (entry (i, j) sits at
//                col_off[j] + i: 64 consecutive doubles per column and wavefront)
//   column part  sum_{r < i} H(r, i) x_r: column i's rows 0 .. i - 1 are contiguous; every wavefront takes 8 of the strip's columns,
//                lanes stride over the rows, fixed reduction tree
// against 2 x (8 + 4 + 8) bytes per entry through the view (one of the two gathers strided: one value per cache line).  Fixed
// partition and summation order: deterministic.  cfg 5 (20 triangles of 1275): 345 -> 86 us per SpMV (2 x 130 MB at 3.4 TB/s, fabric side).
__global__ void __launch_bounds__(512)
k_spmv_dense_tri(const DenseTriStrip *__restrict__ strips, const int64_t *__restrict__ col_off, const double *__restrict__ kval,
                 const RefineState *st, const double *__restrict__ x0, const double *__restrict__ x1, double *__restrict__ acc_out) {
    if (st && !st->active) return;
    const double *__restrict__ x = st ? (st->cur ? x0 : x1) : x0;       // st: the candidate iterate of a refinement step (k_spmv_residual_cand)
    __shared__ double part[8][64];
    __shared__ double csum[64];
    const DenseTriStrip T = strips[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = T.i0 + lane;
    const int64_t *__restrict__ co = col_off + T.col0;
    const double *__restrict__ xb = x + T.c0;
    double a = 0.0;
#pragma unroll 4
    for (int j = T.i0 + w; j < T.d; j += 8) {
        const double v = kval[co[j] + (i <= j ? i : j)];      // (lanes right of the diagonal re-read the diagonal entry: never out of the column)
        a = fma(i <= j ? v : 0.0, xb[j], a);
    }
    part[w][lane] = a;
    for (int c = 0; c < 8; c++) {
        const int ci = T.i0 + 8 * w + c;                      // wave-uniform
        double s = 0.0;
        if (ci < T.d) {
            const double *__restrict__ colv = kval + co[ci];
#pragma unroll 4
            for (int r = lane; r < ci; r += 64) s = fma(colv[r], xb[r], s);
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) csum[8 * w + c] = s;
    }
    __syncthreads();
    if (w == 0 && i < T.d) {
        double t = csum[lane];
#pragma unroll
        for (int q = 0; q < 8; q++) t += part[q][lane];
        acc_out[T.c0 + i] = t;
    }
}

// the three sparse products of residuals_update! (residuals.jl:12-25) from the RESIDENT KKT
// values -- K = [P A'; A -Hs] in the original ordering, symmetric CSR view, 4 lanes per row:
//   rows i < n:        Px_i  = sum_{c < n} K_ic x_c ,   ATz_i = sum_{n <= c < n+m} K_ic z_{c-n}
//   rows n <= i < n+m: Ax_{i-n} = sum_{c < n} K_ic x_c          (the -Hs block and the expansion columns are skipped)
__global__ void __launch_bounds__(256)
k_block_products(const int64_t *__restrict__ rowptr, const int *__restrict__ col, const int64_t *__restrict__ qidx,
                 const double *__restrict__ kval, const double *__restrict__ x, const double *__restrict__ z,
                 double *__restrict__ Px, double *__restrict__ ATz, double *__restrict__ Ax, int n, int m) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gid >> 2, sub = gid & 3;
    double ax = 0.0, az = 0.0;
    if (row < n + m) {
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        for (int64_t p = p0 + sub; p < p1; p += 4) {
            const int c = col[p];
            const double v = kval[qidx[p]];
            if (c < n) ax += v * x[c];
            else if (row < n && c < n + m) az += v * z[c - n];
        }
    }
    ax += __shfl_xor(ax, 1, 64);
    ax += __shfl_xor(ax, 2, 64);
    az += __shfl_xor(az, 1, 64);
    az += __shfl_xor(az, 2, 64);
    if (sub == 0 && row < n + m) {
        if (row < n) { Px[row] = ax; ATz[row] = az; }
        else Ax[row - n] = ax;
    }
}

// residuals_update! (residuals.jl:1-37) from the RESIDENT P and A values -- rows of the symmetric
// CSR view of K, 4 lanes per row (see k_block_products):
//   rows i < n :  Px_i = (P x)_i, rx_inf_i = -(A' z)_i, rx_i = rx_inf_i - Px_i - q_i tau,  partial sums of q.x and x.Px
//   rows n+j   :  rz_inf_j = s_j + (A x)_j,             rz_j = rz_inf_j - b_j tau,         partial sums of b.z and s.z
// The four dot products are reduced per workgroup into part[blockIdx][4] and summed in block order by k_residuals_finish
// (deterministic).  out = [rx | rz | rx_inf | rz_inf | Px]  (n + m + n + m + n doubles).
__global__ void __launch_bounds__(256)
k_residuals(const int64_t *__restrict__ rowptr, const int *__restrict__ col, const int64_t *__restrict__ qidx,
            const double *__restrict__ kval, const double *__restrict__ x, const double *__restrict__ z,
            const double *__restrict__ sv, const double *__restrict__ q, const double *__restrict__ b, double tau,
            double *__restrict__ out, double *__restrict__ part, int n, int m) {
    __shared__ double red[4][4];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gid >> 2, sub = gid & 3;
    double ax = 0.0, az = 0.0;
    if (row < n + m) {
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        for (int64_t p = p0 + sub; p < p1; p += 4) {
            const int c = col[p];
            const double v = kval[qidx[p]];
            if (c < n) ax += v * x[c];
            else if (row < n && c < n + m) az += v * z[c - n];
        }
    }
    ax += __shfl_xor(ax, 1, 64);
    ax += __shfl_xor(ax, 2, 64);
    az += __shfl_xor(az, 1, 64);
    az += __shfl_xor(az, 2, 64);
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;      // q.x, b.z, s.z, x.Px contributions of this row
    if (sub == 0 && row < n + m) {
        double *rx = out, *rz = out + n, *rxi = rz + m, *rzi = rxi + n, *Px = rzi + m;
        if (row < n) {
            const double xi = x[row], rinf = -az;
            Px[row] = ax;
            rxi[row] = rinf;
            rx[row] = (rinf - ax) - q[row] * tau;
            d0 = q[row] * xi;
            d3 = xi * ax;
        } else {
            const int j = row - n;
            const double rinf = sv[j] + ax;
            rzi[j] = rinf;
            rz[j] = rinf - b[j] * tau;
            d1 = b[j] * z[j];
            d2 = sv[j] * z[j];
        }
    }
    // workgroup reduction in a fixed order
    for (int off = 32; off > 0; off >>= 1) {
        d0 += __shfl_down(d0, off, 64);
        d1 += __shfl_down(d1, off, 64);
        d2 += __shfl_down(d2, off, 64);
        d3 += __shfl_down(d3, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = d0; red[wave][1] = d1; red[wave][2] = d2; red[wave][3] = d3; }
    __syncthreads();
    if (threadIdx.x < 4)
        part[(int64_t)blockIdx.x * 4 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
// scal = [q.x, b.z, s.z, x.Px, r_tau = q.x + b.z + kappa + x.Px / tau]   (residuals.jl:27-34)
__global__ void __launch_bounds__(256)
k_residuals_finish(const double *__restrict__ part, int nblocks, double tau, double kappa, double *__restrict__ scal) {
    __shared__ double buf[4][64];
    const int which = threadIdx.x & 3, slot = threadIdx.x >> 2;      // 64 partial sums per dot product, fixed order
    double a = 0.0;
    for (int bq = slot; bq < nblocks; bq += 64) a += part[(int64_t)bq * 4 + which];
    buf[which][slot] = a;
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0.0;
        for (int i = 0; i < 64; i++) t += buf[threadIdx.x][i];
        buf[threadIdx.x][0] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double qx = buf[0][0], bz = buf[1][0], sz = buf[2][0], xPx = buf[3][0];
        scal[0] = qx; scal[1] = bz; scal[2] = sz; scal[3] = xPx;
        scal[4] = qx + bz + kappa + xPx / tau;
    }
}

// ------------------------------------------------------------------------------------------
// SURVEY section 8(f) row N2, second half: the reduced-system algebra of kkt_solve! (kktsystem.jl:170-196) on the device.
// Given the two resident solutions  [x1; z1] (this step's right-hand side)  and  [x2; z2] (the constant right-hand side [-q; b]),
// xi = x / tau and the resident q, b, P:
//   tau_num = rhs.tau - rhs.kappa / tau + q.x1 + b.z1 + 2 xi'P x1
//   tau_den = kappa / tau - q.x2 - b.z2 + (xi - x2)'P(xi - x2) - x2'P x2          (quad_form of mathutils.jl:299-337 = x'(P y), P symmetric)
//   dtau = tau_num / tau_den ;  lhs = [x1; z1] + dtau [x2; z2]
// Rows of the symmetric CSR view of K (4 lanes per row, like k_residuals); the seven dot products are reduced per workgroup into
// part[blockIdx][8] and summed in block order by k_reduced_finish (deterministic).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_reduced_rows(const int64_t *__restrict__ rowptr, const int *__restrict__ col, const int64_t *__restrict__ qidx,
               const double *__restrict__ kval, const double *__restrict__ s1, const double *__restrict__ s2,
               const double *__restrict__ xv, const double *__restrict__ q, const double *__restrict__ b, double tau,
               double *__restrict__ part, int n, int m) {
    __shared__ double red[4][8];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gid >> 2, sub = gid & 3;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (row < n) {
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        for (int64_t p = p0 + sub; p < p1; p += 4) {
            const int c = col[p];
            if (c < n) {
                const double v = kval[qidx[p]];
                a1 += v * s1[c];
                a2 += v * s2[c];
                a3 += v * (xv[c] / tau);
            }
        }
    }
    a1 += __shfl_xor(a1, 1, 64); a1 += __shfl_xor(a1, 2, 64);
    a2 += __shfl_xor(a2, 1, 64); a2 += __shfl_xor(a2, 2, 64);
    a3 += __shfl_xor(a3, 1, 64); a3 += __shfl_xor(a3, 2, 64);
    double d[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // q.x1, b.z1, xi'P x1, q.x2, b.z2, (xi-x2)'P(xi-x2), x2'P x2
    if (sub == 0 && row < n + m) {
        if (row < n) {
            const double xi = xv[row] / tau, x1 = s1[row], x2 = s2[row];
            d[0] = q[row] * x1;
            d[2] = xi * a1;
            d[3] = q[row] * x2;
            d[5] = (xi - x2) * (a3 - a2);
            d[6] = x2 * a2;
        } else {
            const int j = row - n;
            d[1] = b[j] * s1[row];
            d[4] = b[j] * s2[row];
        }
    }
#pragma unroll
    for (int c = 0; c < 7; c++)
        for (int off = 32; off > 0; off >>= 1) d[c] += __shfl_down(d[c], off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < 8; c++) red[wave][c] = d[c];
    __syncthreads();
    if (threadIdx.x < 8)
        part[(int64_t)blockIdx.x * 8 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
// scal_in = [tau, kappa, rhs.tau, rhs.kappa];  scal_out = [dtau, tau_num, tau_den, q.x1, b.z1, xi'Px1, q.x2, b.z2, (xi-x2)'P(xi-x2), x2'Px2]
__global__ void __launch_bounds__(256)
k_reduced_finish(const double *__restrict__ part, int nblocks, double tau, double kappa, double rhs_tau, double rhs_kappa,
                 double *__restrict__ scal_out) {
    __shared__ double buf[8][32];
    const int which = threadIdx.x & 7, slot = threadIdx.x >> 3;       // 32 partial sums per dot product, fixed order
    double a = 0.0;
    for (int bq = slot; bq < nblocks; bq += 32) a += part[(int64_t)bq * 8 + which];
    buf[which][slot] = a;
    __syncthreads();
    if (threadIdx.x < 8) {
        double t = 0.0;
        for (int i = 0; i < 32; i++) t += buf[threadIdx.x][i];
        buf[threadIdx.x][0] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double qx1 = buf[0][0], bz1 = buf[1][0], xPx1 = buf[2][0], qx2 = buf[3][0], bz2 = buf[4][0], dPd = buf[5][0], x2Px2 = buf[6][0];
        const double num = rhs_tau - rhs_kappa / tau + qx1 + bz1 + 2.0 * xPx1;         // kktsystem.jl:183
        double den = kappa / tau - qx2 - bz2;                                           // :189
        den += dPd - x2Px2;                                                             // :190
        scal_out[0] = num / den; scal_out[1] = num; scal_out[2] = den;
        scal_out[3] = qx1; scal_out[4] = bz1; scal_out[5] = xPx1; scal_out[6] = qx2; scal_out[7] = bz2; scal_out[8] = dPd; scal_out[9] = x2Px2;
    }
}
// lhs = [x1; z1] + dtau [x2; z2]   (kktsystem.jl:195-196), dtau read from the device
__global__ void k_reduced_axpy(const double *__restrict__ s1, const double *__restrict__ s2, const double *__restrict__ scal,
                               double *__restrict__ lhs, int nm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nm) lhs[i] = s1[i] + scal[0] * s2[i];
}
// the constant right-hand side of _kkt_solve_constant_rhs! (kktsystem.jl:80-92): [-q; b; 0]
__global__ void k_const_rhs(double *__restrict__ dst, const double *__restrict__ qb, int n, int nm, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) dst[i] = i < n ? -qb[i] : (i < nm ? qb[i] : 0.0);
}

__global__ void __launch_bounds__(256)
k_norm_inf(const double *__restrict__ v, int n, unsigned long long *__restrict__ slot) {
    double a = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double t = fabs(v[i]);
        a = (t > a || t != t) ? t : a;
    }
    block_atomic_max_abs(slot, a);
}

// ---- refinement decided on the device (one host synchronisation per refined solve instead of one per step).
// k_refine_decide<phase 0> runs after the first solve and its residual, <phase 1> after every refinement step; both are
// single-thread kernels that replay the branches of kktsolver_directldl.jl:418-446 on the RefineState.
__global__ void k_refine_decide(RefineState *st, const unsigned long long *scal, int phase, double reltol, double abstol,
                                int max_iter, double stop_ratio) {
    if (blockIdx.x || threadIdx.x) return;
    const double norme = __longlong_as_double((long long)scal[SC_NORME]);
    if (phase == 0) {
        const double normb = __longlong_as_double((long long)scal[SC_NORMB]);
        st->cur = 0;
        st->steps = 0;
        st->normb = normb;
        st->norme = norme;
        st->lastnorme = norme;
        st->fail = isfinite(norme) ? 0 : 1;                                        // :414-416
        st->active = (!st->fail && max_iter > 0 && !(norme <= abstol + reltol * normb)) ? 1 : 0;   // :421-426
        return;
    }
    if (!st->active) return;
    st->steps += 1;
    st->norme = norme;
    if (!isfinite(norme)) { st->fail = 1; st->active = 0; return; }               // :433-435
    const double improved = st->lastnorme / norme;                                // :437
    if (improved < stop_ratio) {                                                  // :438-444: insufficient improvement
        if (improved > 1.0) st->cur ^= 1;                                         //           (keep the better of the two)
        st->active = 0;
        return;
    }
    st->cur ^= 1;                                                                 // :445 swap x, dx
    st->lastnorme = norme;
    st->active = (st->steps < max_iter && !(norme <= abstol + reltol * st->normb)) ? 1 : 0;
}
// candidate = iterate + correction  (:430-431: dx = K \ e, dx += x); inactive states leave the candidate alone
__global__ void k_refine_add(const RefineState *st, double *__restrict__ x0, double *__restrict__ x1,
                             const double *__restrict__ corr, int n) {
    if (!st->active) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double *src = st->cur ? x1 : x0;
    double *dst = st->cur ? x0 : x1;
    if (i < n) dst[i] = src[i] + corr[i];
}
// out = the accepted iterate (first nm entries)
__global__ void k_refine_copy_out(const RefineState *st, const double *__restrict__ x0, const double *__restrict__ x1,
                                  double *__restrict__ out, int nm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double *src = st->cur ? x1 : x0;
    if (i < nm) out[i] = src[i];
}
// e = b - K * (candidate), ||e||_inf; the candidate is xbuf[1 - cur] (after a step) -- see k_spmv_residual
__global__ void __launch_bounds__(256)
k_spmv_residual_cand(const int64_t *__restrict__ rowptr, const int *__restrict__ col, const int64_t *__restrict__ qidx,
                     const double *__restrict__ kval, const double *__restrict__ b, const RefineState *st,
                     const double *__restrict__ x0, const double *__restrict__ x1, double *__restrict__ e, int n,
                     unsigned long long *__restrict__ norm_slot, const double *__restrict__ dacc) {
    if (!st->active) return;           // the norm slot keeps its previous value; k_refine_decide ignores it when inactive
    spmv_short_rows(rowptr, col, qidx, kval, b, st->cur ? x0 : x1, e, n, norm_slot, dacc);
}
__global__ void __launch_bounds__(256)
k_spmv_long_cand(const int *__restrict__ long_rows, const int64_t *__restrict__ rowptr, const int *__restrict__ col,
                 const int64_t *__restrict__ qidx, const double *__restrict__ kval, const double *__restrict__ b,
                 const RefineState *st, const double *__restrict__ x0, const double *__restrict__ x1, double *__restrict__ e,
                 unsigned long long *__restrict__ norm_slot, const double *__restrict__ dacc) {
    if (!st->active) return;
    spmv_long_row(long_rows[blockIdx.x], rowptr, col, qidx, kval, b, st->cur ? x0 : x1, e, norm_slot, dacc);
}

__global__ void k_add(double *__restrict__ dst, const double *__restrict__ a, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += a[i];
}

__global__ void k_set_rhs(double *__restrict__ b, const double *__restrict__ rhs, int nm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = i < nm ? rhs[i] : 0.0;
}

__global__ void k_check_finite(const double *__restrict__ v, int n, int *__restrict__ flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !isfinite(v[i])) atomicOr(flags + FL_NONFINITE, 1);
}

// ------------------------------------------------------------------------------------------
// host-callable launchers (used by the hipkkt_*.cpp host files; all launches go to the handle's stream)
// ------------------------------------------------------------------------------------------
static inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

void launch_scatter_values(hipStream_t st, double *kval, const int64_t *idx, const double *vals, int64_t n, double scale) {
    if (n > 0) hipLaunchKernelGGL(k_scatter_values, dim3(nblk(n)), dim3(256), 0, st, kval, idx, vals, n, scale);
}
void launch_scale_values(hipStream_t st, double *kval, const int64_t *idx, int64_t n, double scale) {
    if (n > 0) hipLaunchKernelGGL(k_scale_values, dim3(nblk(n)), dim3(256), 0, st, kval, idx, n, scale);
}
void launch_soc_batch(hipStream_t st, double *kval, const int64_t *uidx, const int64_t *vidx, const int *cone_of,
                      const double *u, const double *v, const double *eta2, int64_t n, const int64_t *didx, int nsoc) {
    int64_t m = n > nsoc ? n : nsoc;
    if (m > 0)
        hipLaunchKernelGGL(k_soc_batch, dim3(nblk(m)), dim3(256), 0, st, kval, uidx, vidx, cone_of, u, v, eta2, n, didx, nsoc);
}
void launch_maxabs_gather(hipStream_t st, const double *v, const int64_t *idx, int64_t n, unsigned long long *slot) {
    if (n > 0) hipLaunchKernelGGL(k_maxabs_gather, dim3(nblk(n)), dim3(256), 0, st, v, idx, n, slot);
}
void launch_init_panels(hipStream_t st, const DevPlan &P, int64_t nnz, int static_enable, double eps_const, double eps_prop) {
    if (nnz > 0)
        hipLaunchKernelGGL(k_init_panels, dim3(nblk(nnz)), dim3(256), 0, st, P.Lx, P.kval, P.kmap, P.kdiag_sign, nnz,
                           P.scal, static_enable, eps_const, eps_prop);
}
static size_t factor_lds_bytes(int wmax) { return sizeof(double) * (size_t)(wmax * (wmax + 1) + wmax + wmax * LDT); }
void launch_factor_level(hipStream_t st, const DevPlan &P, int item_begin, int nitems, int wmax, double dyn_eps, double dyn_delta) {
    if (nitems > 0)
        hipLaunchKernelGGL(k_factor_level, dim3(nitems), dim3(256), factor_lds_bytes(wmax), st, P, item_begin, wmax, dyn_eps,
                           dyn_delta);
}
void launch_factor_panel(hipStream_t st, const DevPlan &P, int item_begin, int nitems, double dyn_eps, double dyn_delta) {
    if (nitems > 0) hipLaunchKernelGGL(k_factor_panel, dim3(nitems), dim3(512), 0, st, P, item_begin, dyn_eps, dyn_delta);
}
void launch_update_stage(hipStream_t st, const DevPlan &P, int group_begin, int ngroups) {
    if (ngroups > 0) hipLaunchKernelGGL(k_update_stage, dim3(ngroups), dim3(kUpdWaves * 64), 0, st, P, group_begin);
}
// Few tiles (<= 384: the just-in-time updates on the critical path): 16-column strips x 32 rows per wavefront, two workgroups per tile
// (measured on cfg 2a, factorisation: 5.81 / 5.65 / 5.65 ms with 64 / 32 / 16 rows per wavefront).  Plenty of tiles: one wavefront per
// tile; full_k launches (whole panels of sources, >= 1.5 MFLOP per tile) cut the tiles of a partial last round into strips / halves.
void launch_update_dense(hipStream_t st, const DevPlan &P, int group_begin, int ngroups, bool full_k) {
    if (ngroups <= 0) return;
    if (ngroups > 384) {
        const int round = 1024, r = ngroups % round, nfull = ngroups - r;   // nfull: multiple of 1024, hence of 4
        if (full_k && r > 0 && (r <= 256 || (nfull == 0 && r <= 768)))      // (a lone partial round: three strip sub-rounds still beat it)
            hipLaunchKernelGGL(k_update_dense_tail<4>, dim3(nfull / 4 + r), dim3(256), 0, st, P, group_begin, nfull, ngroups);
        else if (full_k && r > 0 && r <= 512)
            hipLaunchKernelGGL(k_update_dense_tail<2>, dim3(nfull / 4 + (r + 1) / 2), dim3(256), 0, st, P, group_begin, nfull, ngroups);
        else
            hipLaunchKernelGGL((k_update_dense<4, 4>), dim3((ngroups + 3) / 4), dim3(256), 0, st, P, group_begin, ngroups);
    } else {
        hipLaunchKernelGGL((k_update_dense<1, 2>), dim3(ngroups * 2), dim3(256), 0, st, P, group_begin, ngroups);
    }
}
// Split-K (hipkkt_setup.cpp plan_split_k): a stage with FEW target tiles and MANY contributions per tile (cfg 5: the 1000 x 1000 block
// of the variables = 136 tiles, each fed by 20 cones x 5 panels per update batch, K = 6400) kept 136 of the 1024 SIMDs busy for 0.7 ms
// per batch.  The contributions of such a tile are cut into up to 16 chunks, each accumulated by a wavefront of its own into a partial
// tile that starts from zero (the unchanged tile code of dense_tile.h on a scratch tile); this kernel adds the partial tiles to the
// target in a FIXED order (deterministic) and clears them for the next stage.
__global__ void __launch_bounds__(256)
k_split_reduce(DevPlan P, const SplitRec *recs) {
    const SplitRec R = recs[blockIdx.x];
    double *tp = P.Lx + R.tile_off, *sc = P.Lx + R.scratch_off;
    for (int idx = threadIdx.x; idx < 4096; idx += 256) {
        const int row = idx & 63, col = idx >> 6;
        double s = 0.0;
        for (int q = 0; q < R.nparts; q++) {
            s += sc[(int64_t)q * 4096 + idx];
            sc[(int64_t)q * 4096 + idx] = 0.0;
        }
        if (row < R.nrt && col < R.wt) tp[row + (int64_t)col * R.rt] += s;
    }
}
void launch_split_reduce(hipStream_t st, const DevPlan &P, const SplitRec *recs, int n) {
    if (n > 0) hipLaunchKernelGGL(k_split_reduce, dim3(n), dim3(256), 0, st, P, recs);
}
void launch_fwd_narrow(hipStream_t st, const DevPlan &P, int sn_begin, int n, int wmax, double *y, double *z) {
    if (n <= 0) return;
    if (wmax <= 4) hipLaunchKernelGGL(k_fwd_narrow<4>, dim3(nblk(n)), dim3(256), 0, st, P, sn_begin, n, y, z);
    else hipLaunchKernelGGL(k_fwd_narrow<kNarrowW>, dim3(nblk(n)), dim3(256), 0, st, P, sn_begin, n, y, z);
}
void launch_bwd_narrow(hipStream_t st, const DevPlan &P, int sn_begin, int n, int wmax, const double *z, double *x, double *xout) {
    if (n <= 0) return;
    if (wmax <= 4) hipLaunchKernelGGL(k_bwd_narrow<4>, dim3(nblk(n)), dim3(256), 0, st, P, sn_begin, n, z, x, xout);
    else hipLaunchKernelGGL(k_bwd_narrow<kNarrowW>, dim3(nblk(n)), dim3(256), 0, st, P, sn_begin, n, z, x, xout);
}
// zero a few 32-bit words.  Small hipMemsetAsync nodes inside a captured hipGraph are not reliable in every ROCm
// set-up (under rocprofv3 the 16-byte memset of the flag words was seen to write its own arguments instead of
// zeros; under PyTorch's bundled runtime a 56-byte one was seen to do nothing), so the factor / solve graphs
// clear their flag and scalar words with this kernel.
__global__ void k_zero_words(int *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
// Hs block of one PSD triangle cone on the device: entry e of the packed (column-major) upper triangle of
// W (x)_s W, e <-> (a <= b), a <-> (i <= j), b <-> (k <= l) in the svec ordering.  The four cases of the reference's
// skron! (coneops_psdtrianglecone.jl:502-540), with its products, association and rounding:
//   i != j, k != l :  W_ik W_jl + W_il W_jk          i == j, k != l :  (sqrt2 W_jl) W_jk
//   i != j, k == l :  (sqrt2 W_il) W_jk               i == j, k == l :  W_jl W_jl
// Explicitly rounded products and sums (no FMA contraction).  The value is negated and scattered through
// map.Hsblocks (kktsolver_directldl.jl:225-228).
__device__ __forceinline__ long long tri_root(long long e) {   // largest t with t(t+1)/2 <= e
    long long t = (long long)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while (t * (t + 1) / 2 > e) t--;
    while ((t + 1) * (t + 2) / 2 <= e) t++;
    return t;
}
__global__ void __launch_bounds__(256)
k_psd_hs(double *__restrict__ kval, const int64_t *__restrict__ map_hs, int64_t hs_off, const double *__restrict__ W, int n,
         int64_t nent) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nent) return;
    const long long b = tri_root(e), a = e - b * (b + 1) / 2;
    const long long j = tri_root(a), i = a - j * (j + 1) / 2;
    const long long l = tri_root(b), k = b - l * (l + 1) / 2;
    const double sqrt2 = sqrt(2.0);
    double v;
    if (i != j && k != l) {
        // products and the sum are rounded one by one like the reference's loop: HIP's `*` followed by `+` is contracted
        // into an FMA by the compiler, so the products pass through an opaque asm before they are added
        double p1 = W[i * n + k] * W[j * n + l];
        double p2 = W[i * n + l] * W[j * n + k];
        asm volatile("" : "+v"(p1), "+v"(p2));
        v = p1 + p2;
    } else if (i == j && k != l) {
        v = (sqrt2 * W[j * n + l]) * W[j * n + k];
    } else if (i != j) {
        v = (sqrt2 * W[i * n + l]) * W[j * n + k];
    } else {
        v = W[j * n + l] * W[j * n + l];
    }
    kval[map_hs[hs_off + e]] = -v;
}
void launch_psd_hs(hipStream_t st, double *kval, const int64_t *map_hs, int64_t hs_off, const double *W, int n) {
    const int64_t numel = (int64_t)n * (n + 1) / 2, nent = numel * (numel + 1) / 2;
    if (nent > 0) hipLaunchKernelGGL(k_psd_hs, dim3(nblk(nent)), dim3(256), 0, st, kval, map_hs, hs_off, W, n, nent);
}
void launch_zero_words(hipStream_t st, void *p, int nwords) {
    if (nwords > 0) hipLaunchKernelGGL(k_zero_words, dim3(nblk(nwords, 64)), dim3(64), 0, st, (int *)p, nwords);
}
void launch_update_gather(hipStream_t st, const DevPlan &P, int64_t ebegin, int64_t n, int64_t hbegin, int64_t nheavy, int max_blocks) {
    if (n > 0) hipLaunchKernelGGL(k_update_gather, dim3(max_blocks > 0 ? std::min<unsigned>(nblk(n), (unsigned)max_blocks) : nblk(n)), dim3(256), 0, st, P, ebegin, n);
    if (nheavy > 0) hipLaunchKernelGGL(k_update_gather_heavy, dim3(nblk(nheavy, 4)), dim3(256), 0, st, P, hbegin, nheavy);
}
// inv_list = [n_small supernodes of width <= wmax_small | n_wide supernodes of width in (16, 64]]
void launch_invert_diag(hipStream_t st, const DevPlan &P, int n_small, int wmax_small, int n_wide) {
    const int ldl = wmax_small | 1;
    if (n_small > 0)
        hipLaunchKernelGGL(k_invert_diag, dim3(n_small), dim3(64), sizeof(double) * 2 * (size_t)wmax_small * ldl, st, P, 0, n_small);
    if (n_wide > 0) hipLaunchKernelGGL(k_invert_diag_wide, dim3(n_wide), dim3(256), 0, st, P, n_small);
}
void launch_mfma_probe(hipStream_t st, const double *A, const double *B, double *out) {
    hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, st, A, B, out);
}
void launch_permute_in(hipStream_t st, const double *b, const int *perm, double *y, int n, int *epoch, int *ticks, int nticks, int *zero, int nzero) {
    if (n > 0) hipLaunchKernelGGL(k_permute_in, dim3(nblk(n)), dim3(256), 0, st, b, perm, y, n, epoch, ticks, nticks, zero, nzero);
}
void launch_fwd_level(hipStream_t st, const DevPlan &P, int item_begin, int nitems, double *y, double *z) {
    if (nitems > 0) hipLaunchKernelGGL(k_fwd_level, dim3(nitems), dim3(256), 0, st, P, item_begin, y, z);
}
void launch_bwd_partial(hipStream_t st, const DevPlan &P, int item_begin, int nitems, const double *x) {
    if (nitems > 0) hipLaunchKernelGGL(k_bwd_partial, dim3(nitems), dim3(256), 0, st, P, item_begin, x);
}
void launch_bwd_final(hipStream_t st, const DevPlan &P, int sn_begin, int nsn, const double *z, double *x, double *xout) {
    if (nsn > 0) hipLaunchKernelGGL(k_bwd_final, dim3(nsn), dim3(256), 0, st, P, sn_begin, z, x, xout);
}
void launch_fwd_seg(hipStream_t st, const DevPlan &P, int seg, int item_begin, int nitems, int nsuper, int first, double *y, double *z) {
    if (nitems > 0) hipLaunchKernelGGL(k_fwd_seg, dim3(nitems), dim3(256), 0, st, P, seg, item_begin, nitems, nsuper, first, y, z);
}
void launch_bwd_seg(hipStream_t st, const DevPlan &P, int seg, int item_begin, int nitems, int nsuper, int first, const double *z,
                    double *x, double *xout) {
    if (nitems > 0) hipLaunchKernelGGL(k_bwd_seg, dim3(nitems), dim3(256), 0, st, P, seg, item_begin, nitems, nsuper, first, z, x, xout);
}
void launch_front_fwd(hipStream_t st, const DevPlan &P, const FrontDesc &F, double *y, double *z) {
    if (F.sb_g > 0) { launch_front_fwd_sb(st, P, F, y, z); return; }      // super-block sweep (front_sweep.hip)
    hipLaunchKernelGGL(k_front_fwd, dim3(F.nb), dim3(256), 0, st, P, F, y, z);
}
void launch_front_bwd(hipStream_t st, const DevPlan &P, const FrontDesc &F, const double *z, double *x, double *xout) {
    if (F.sb_g > 0) { launch_front_bwd_sb(st, P, F, z, x, xout); return; }
    hipLaunchKernelGGL(k_front_bwd, dim3(F.np), dim3(256), 0, st, P, F, z, x, xout);
}
void launch_spmv_residual(hipStream_t st, const DevPlan &P, const double *b, const double *xi, double *e, int n,
                          unsigned long long *slot) {
    const double *dacc = P.n_dtri_strips > 0 ? P.dense_acc : nullptr;
    if (dacc)
        hipLaunchKernelGGL(k_spmv_dense_tri, dim3(P.n_dtri_strips), dim3(512), 0, st, P.dtri_strips, P.dtri_col, P.kval, (const RefineState *)nullptr,
                           xi, xi, P.dense_acc);
    if (n > 0)
        hipLaunchKernelGGL(k_spmv_residual, dim3(nblk((int64_t)n * 4)), dim3(256), 0, st, P.sym_rowptr, P.sym_col, P.sym_q,
                           P.kval, b, xi, e, n, slot, dacc);
    if (P.n_long_rows > 0)
        hipLaunchKernelGGL(k_spmv_long, dim3(P.n_long_rows), dim3(256), 0, st, P.long_rows, P.sym_rowptr, P.sym_col, P.sym_q, P.kval, b,
                           xi, e, slot, dacc);
}
void launch_block_products(hipStream_t st, const DevPlan &P, const double *x, const double *z, double *Px, double *ATz,
                           double *Ax, int n, int m) {
    if (n + m > 0)
        hipLaunchKernelGGL(k_block_products, dim3(nblk((int64_t)(n + m) * 4)), dim3(256), 0, st, P.sym_rowptr, P.sym_col, P.sym_q,
                           P.kval, x, z, Px, ATz, Ax, n, m);
}
void launch_refine_decide(hipStream_t st, RefineState *rs, const double *scal, int phase, double reltol, double abstol, int max_iter,
                          double stop_ratio) {
    hipLaunchKernelGGL(k_refine_decide, dim3(1), dim3(64), 0, st, rs, (const unsigned long long *)scal, phase, reltol, abstol, max_iter,
                       stop_ratio);
}
void launch_refine_add(hipStream_t st, const RefineState *rs, double *x0, double *x1, const double *corr, int n) {
    if (n > 0) hipLaunchKernelGGL(k_refine_add, dim3(nblk(n)), dim3(256), 0, st, rs, x0, x1, corr, n);
}
void launch_refine_copy_out(hipStream_t st, const RefineState *rs, const double *x0, const double *x1, double *out, int nm) {
    if (nm > 0) hipLaunchKernelGGL(k_refine_copy_out, dim3(nblk(nm)), dim3(256), 0, st, rs, x0, x1, out, nm);
}
void launch_spmv_residual_cand(hipStream_t st, const DevPlan &P, const double *b, const RefineState *rs, const double *x0,
                               const double *x1, double *e, int n, unsigned long long *slot) {
    const double *dacc = P.n_dtri_strips > 0 ? P.dense_acc : nullptr;
    if (dacc)
        hipLaunchKernelGGL(k_spmv_dense_tri, dim3(P.n_dtri_strips), dim3(512), 0, st, P.dtri_strips, P.dtri_col, P.kval, rs, x0, x1, P.dense_acc);
    if (n > 0)
        hipLaunchKernelGGL(k_spmv_residual_cand, dim3(nblk((int64_t)n * 4)), dim3(256), 0, st, P.sym_rowptr, P.sym_col, P.sym_q,
                           P.kval, b, rs, x0, x1, e, n, slot, dacc);
    if (P.n_long_rows > 0)
        hipLaunchKernelGGL(k_spmv_long_cand, dim3(P.n_long_rows), dim3(256), 0, st, P.long_rows, P.sym_rowptr, P.sym_col, P.sym_q, P.kval,
                           b, rs, x0, x1, e, slot, dacc);
}
int long_row_threshold() { return kLongRow; }
int residual_blocks(int n, int m) { return (int)nblk((int64_t)(n + m) * 4); }
void launch_residuals(hipStream_t st, const DevPlan &P, const double *x, const double *z, const double *s, const double *q,
                      const double *b, double tau, double kappa, double *out, double *part, double *scal, int n, int m) {
    if (n + m <= 0) return;
    const int nb = residual_blocks(n, m);
    hipLaunchKernelGGL(k_residuals, dim3(nb), dim3(256), 0, st, P.sym_rowptr, P.sym_col, P.sym_q, P.kval, x, z, s, q, b, tau, out, part, n, m);
    hipLaunchKernelGGL(k_residuals_finish, dim3(1), dim3(256), 0, st, part, nb, tau, kappa, scal);
}
void launch_reduced(hipStream_t st, const DevPlan &P, const double *s1, const double *s2, const double *xv, const double *q,
                    const double *b, double tau, double kappa, double rhs_tau, double rhs_kappa, double *part, double *scal_out,
                    double *lhs, int n, int m) {
    if (n + m <= 0) return;
    const int nb = residual_blocks(n, m);
    hipLaunchKernelGGL(k_reduced_rows, dim3(nb), dim3(256), 0, st, P.sym_rowptr, P.sym_col, P.sym_q, P.kval, s1, s2, xv, q, b, tau, part, n, m);
    hipLaunchKernelGGL(k_reduced_finish, dim3(1), dim3(256), 0, st, part, nb, tau, kappa, rhs_tau, rhs_kappa, scal_out);
    hipLaunchKernelGGL(k_reduced_axpy, dim3(nblk(n + m)), dim3(256), 0, st, s1, s2, scal_out, lhs, n + m);
}
void launch_const_rhs(hipStream_t st, double *dst, const double *qb, int n, int nm, int N) {
    if (N > 0) hipLaunchKernelGGL(k_const_rhs, dim3(nblk(N)), dim3(256), 0, st, dst, qb, n, nm, N);
}
void launch_norm_inf(hipStream_t st, const double *v, int n, unsigned long long *slot) {
    if (n > 0) hipLaunchKernelGGL(k_norm_inf, dim3(min(nblk(n), 64u)), dim3(256), 0, st, v, n, slot);
}
void launch_add(hipStream_t st, double *dst, const double *a, int n) {
    if (n > 0) hipLaunchKernelGGL(k_add, dim3(nblk(n)), dim3(256), 0, st, dst, a, n);
}
void launch_set_rhs(hipStream_t st, double *b, const double *rhs, int nm, int n) {
    if (n > 0) hipLaunchKernelGGL(k_set_rhs, dim3(nblk(n)), dim3(256), 0, st, b, rhs, nm, n);
}
void launch_check_finite(hipStream_t st, const double *v, int n, int *flags) {
    if (n > 0) hipLaunchKernelGGL(k_check_finite, dim3(nblk(n)), dim3(256), 0, st, v, n, flags);
}

}  // namespace hipkkt
