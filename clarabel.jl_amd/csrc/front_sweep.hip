// K5 over a FRONT, super-block sweeps (round 3; VERDICT r2 task 1): the per-panel sweeps of kernels.hip (k_front_fwd / k_front_bwd)
// pay one inter-workgroup hand-off per 64-column panel -- 88 dependent hops of ~1.7 us on the 5617-column root of a random sparse
// QP, at 14 % of the HBM roofline.  Here the np panels of a front are grouped into super-blocks of kSbG = 8 consecutive panels and
// the explicit inverse of every super-block's unit-lower diagonal block (8 x 8 tiles of 64 x 64, formed after each factorisation by
// k_invert_super from the panels and the per-panel inverses) replaces the eight dependent panel solves:
//
//   forward, workgroup b (row block b, super-block B = b / g):
//       r_b = b_b - (external children) - sum_{q < gB} L[b,q] y_q          L tiles of a whole super-block per hop, prefetched
//       publish r_b, wait for r_c of the earlier panels c of the own super-block
//       y_b = sum_{gB <= c <= b} Inv[b,c] r_c                              Inv tiles prefetched like one more group of L tiles
//   backward, ticket t -> panel p = np-1-t: the mirror image with L^T (row-major copy LT) and Inv^T.
//
// Two hand-offs per super-block instead of eight: 11 x 2 hops on that root instead of 88.  The hand-off itself is unchanged
// (self-validating 16-byte slots, sweep_common.h); a second set of slots carries the partial right-hand sides r / s.  Workgroups
// have 16 wavefronts: wave w < 5 polls panel w of the awaited super-block, thread (lane = row, wave v) owns the columns v + 16 t of
// every tile (4 values per tile, two groups of 5 tiles in registers).  Same guarantees as the per-panel sweeps: tickets in arrival order
// (a workgroup only waits for lower tickets), bounded spins that fail the solve through FL_FRONTFAIL, fixed summation order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kernels.h"
#include "sweep_common.h"

#ifdef HIPKKT_SWEEP_TRACE
__device__ long long g_sweep_trace[8 * 512];
#define SWT(i) do { if (threadIdx.x == 0) g_sweep_trace[tk * 8 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int hipkkt_debug_sweep_trace(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sweep_trace), sizeof(long long) * 8 * 512); }
#else
#define SWT(i) do { } while (0)
#endif
namespace hipkkt {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int SLD = 65;            // LDS row stride of a 64 x 64 tile
constexpr int kSbThreads = 1024;   // 16 wavefronts: 4 values per tile and thread (two groups of 5 tiles = 80 VGPRs of the 128 available)
constexpr int NW = kSbThreads / 64;  // wavefronts per workgroup
constexpr int NT = 2;                // values per (half) tile and thread: 32 x 64 / 1024
static_assert(kSbMaxPanels * 16 <= 24 * 1024, "panel table of the sweeps must fit into LDS");

__device__ __forceinline__ int64_t sb_tile(const FrontDesc &F, int B, int bl, int cl) {   // symbolic.h FrontDesc::sbinv_off
    return F.sbinv_off + ((int64_t)B * (kSbG * (kSbG - 1) / 2) + bl * (bl - 1) / 2 + cl) * 8192;
}
#define SB_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ double sb_ld(const double *base, unsigned byte_off) {   // uniform base + 32-bit lane offset (saddr form)
    return *(const SB_GLOBAL double *)((const SB_GLOBAL char *)base + byte_off);
}
// wave-uniform values are forced into SGPRs (a FrontPanel record read through a plain pointer lands in VGPRs otherwise)
__device__ __forceinline__ int sb_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t sb_rfl64(int64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uint64_t)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double sum_waves(const double (*red)[64], int lane) {   // fixed tree over the NW partial sums
    double q[4];
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = (red[4 * c][lane] + red[4 * c + 1][lane]) + (red[4 * c + 2][lane] + red[4 * c + 3][lane]);
    return (q[0] + q[1]) + (q[2] + q[3]);
}
static_assert(NW == 16 && NT == 2, "the half-tile thread map assumes 16 wavefronts");

// ------------------------------------------------------------------------------------------
// Inverse of the super-blocks' diagonal blocks: Inv[b,c] = -Linv_b * sum_{c <= k < b} L[b,k] Inv[k,c]  (Inv[c,c] = Linv_c, the
// per-panel inverse of k_invert_diag / k_invert_diag_wide); the 64 x 64 x 64 products run on the matrix core.  Tiles are padded to
// 64 x 64 with zeros (panels narrower than 64 columns).
// ------------------------------------------------------------------------------------------
// The chain of a super-block's first column is 28 + 7 dependent 64 x 64 x 64 products -- 1.7 us of FP64 matrix work EACH on one
// compute unit, whatever else is done well (rounds 3 - 5: one workgroup of 16 wavefronts per (super-block, column), ~3 us per
// product, 105 us per launch on cfg 2a).  Round 6: the columns of an inverse are independent of each other (Inv[:,cl] = L^-1 applied
// to a block of unit vectors, everything is a multiplication from the LEFT), so a column block is cut into kInvSlices slices of 16
// columns, one workgroup of FOUR wavefronts each (wavefront v = the rows 16 v .. 16 v + 15 of every tile):
//   * a product is 16 matrix-core instructions per wavefront, one wavefront per SIMD;
//   * the A operand (16 rows x 64 of L[bl,kl] or Linv_bl) goes from memory straight into the wavefront's registers in operand layout,
//     unconditionally (clamped addresses; the zero padding is applied at the use), through a ring of kInvRing sets requested
//     kInvRing - 1 products ahead;
//   * the B operands Inv[kl,cl][:, slice] are the workgroup's own earlier results: they stay in LDS for the whole kernel (7 x 8 KB),
//     so there is NO barrier between the products of a row -- only two per row (the sum S is exchanged between the wavefronts before
//     the closing product with Linv_bl, and the new tile is published to the other wavefronts).
// Every element still sums the same products in the same order (k = 0 .. 63 inside a product, the products of a row in the order of
// their columns): bit-identical to the earlier forms.
constexpr int kInvThreads = 256;
constexpr int kInvSlices = 4;
constexpr int kInvRing = 4;
// row jb of operand m in the list "for jb = 1 .. : jb tiles of L, then Linv": first(jb) = (jb - 1)(jb + 2) / 2
__host__ __device__ constexpr int sb_op_row(int m) {
    return m < 2 ? 1 : m < 5 ? 2 : m < 9 ? 3 : m < 14 ? 4 : m < 20 ? 5 : m < 27 ? 6 : m < 35 ? 7 : 8;
}
static_assert(kSbG == 8 && sb_op_row(1) == 1 && sb_op_row(2) == 2 && sb_op_row(34) == 7 && sb_op_row(35) == 8, "operand list of k_invert_super");
__device__ __forceinline__ void sb_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool CW64>                     // the front's panels are 64 wide (all but its last one): FrontDesc::cw == 64
__global__ void __launch_bounds__(kInvThreads)
k_invert_super(DevPlan P, FrontDesc F) {
    __shared__ double Bs[kSbG - 1][64 * 16];      // Bs[j][k * 16 + c] = Inv[cl + j, cl](k, 16 slice + c)
    __shared__ double Ss[64 * 16];
    __shared__ int64_t s_off[kSbG];
    __shared__ int s_r[kSbG], s_w[kSbG];
    __shared__ int64_t s_diag[kSbG];
    // the long chains (cl = 0) first: blockIdx = (cl, super-block, slice)
    const int per_cl = F.nsb * kInvSlices;
    const int cl = blockIdx.x / per_cl, B = (blockIdx.x % per_cl) / kInvSlices, sl = blockIdx.x % kInvSlices;
    const int nbB = min(kSbG, F.np - kSbG * B);
    if (cl + 1 >= nbB) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = sb_rfl(tid >> 6), l15 = lane & 15, lk = lane >> 4;
    const FrontPanel *fps = P.front_panels + F.fp_off;
    if (tid < nbB) {
        const FrontPanel q = fps[kSbG * B + tid];
        s_off[tid] = q.panel_off; s_r[tid] = q.r; s_w[tid] = q.w; s_diag[tid] = q.diag_off;
    }
    __syncthreads();
    const int wcl = s_w[cl];
    {   // Inv[cl,cl] = Linv_cl: this slice's 16 columns, zero-padded lower triangle
        const double *li = P.Linv + s_diag[cl];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int idx = tid + kInvThreads * q, k = idx >> 4, c = idx & 15, col = 16 * sl + c;
            Bs[0][idx] = (k < wcl && col <= k) ? li[k + col * wcl] : 0.0;
        }
    }
    // The A operands in the order of their use: for jb = 1 .. 7: L[cl + jb, cl + jk] (jk = 0 .. jb - 1), then Linv_{cl + jb}: operand
    // m = first(jb) + jk.  This lane's part of an operand: row i = 16 wv + l15, columns k = 4 kk + lk (kk = 0 .. 15).
    const int irow = 16 * wv + l15;
    double ring[kInvRing][16];
    auto fetch_op = [&](int m) {          // m: compile-time after unrolling
        const int fjb = sb_op_row(m), fjk = m - ((fjb - 1) * (fjb + 2)) / 2;
        if (fjb < kSbG) {                 // (compile-time)
            // Past the end of a short super-block the request is repeated for its last row instead of being skipped: a load
            // under a condition makes the compiler wait for ALL loads at the next use of an earlier one (it must be right on the
            // path without the load).  cl + 1 < nbB holds for every workgroup that gets here.
            const int bl = min(cl + fjb, nbB - 1), wb = s_w[bl];
            const int ic = min(irow, wb - 1);
            if (fjk < fjb) {
                // rows past the block's width and columns past the source panel's width need no zeroing here: they meet zero rows of
                // the B operand / zero columns of Linv_bl; the addresses only have to stay inside the panel
                const int kl = min(cl + fjk, bl - 1), wk = sb_rfl(s_w[kl]), rk = sb_rfl(s_r[kl]);
                const double *base = P.Lx + sb_rfl64(s_off[kl]) + F.cw * (bl - kl);
                if (CW64) {               // uniform base per k-step + one lane offset, no vector arithmetic (kl is never the front's last panel)
                    const unsigned lo = (unsigned)(ic + lk * rk) * 8u;
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) ring[m % kInvRing][kk] = sb_ld(base + (int64_t)(4 * kk) * rk, lo);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) ring[m % kInvRing][kk] = base[ic + (int64_t)min(4 * kk + lk, wk - 1) * rk];
                }
            } else {
                const double *base = P.Linv + s_diag[bl] + ic;
#pragma unroll
                for (int kk = 0; kk < 16; kk++) ring[m % kInvRing][kk] = base[min(4 * kk + lk, ic) * wb];
            }
        }
    };
    // value of operand m at kk with its zero pattern
    auto a_val = [&](int m, int kk) -> double {
        const int fjb = sb_op_row(m), fjk = m - ((fjb - 1) * (fjb + 2)) / 2;
        if (fjk < fjb) return ring[m % kInvRing][kk];
        return (irow < s_w[cl + fjb] && 4 * kk + lk <= irow) ? ring[m % kInvRing][kk] : 0.0;   // Linv_bl: lower triangle of a w x w block
    };
#pragma unroll
    for (int u = 0; u < kInvRing - 1; u++) fetch_op(u);
    sb_bar();                             // Bs[0] is complete
    // the rows are nested (row jb + 1 inside the branch of row jb), not a sequence of seven branches: see fetch_op
    auto row = [&](auto &&self, auto jb_c) -> void {
        constexpr int jb = decltype(jb_c)::value;
        const int bl = cl + jb;
        if (bl < nbB) {                   // workgroup-uniform
            constexpr int n0 = ((jb - 1) * (jb + 2)) / 2;    // index of this row's first product
            // four partial sums (k-steps kk = c mod 4), added at the end: a matrix-core instruction that takes the result of the one
            // before it waits ~200 cycles, four independent chains issue back to back
            v4f64 pa[4];
#pragma unroll
            for (int c = 0; c < 4; c++) pa[c] = v4f64{0.0, 0.0, 0.0, 0.0};
            // the B operands of a product are read from LDS while the matrix core works on the product before it
            double bv[2][16];
#pragma unroll
            for (int kk = 0; kk < 16; kk++) bv[0][kk] = Bs[0][(4 * kk + lk) * 16 + l15];
#pragma unroll
            for (int jk = 0; jk < jb; jk++) {
                const int n = n0 + jk;
                fetch_op(n + kInvRing - 1);
                if (jk + 1 < jb) {
#pragma unroll
                    for (int kk = 0; kk < 16; kk++) bv[(jk + 1) & 1][kk] = Bs[jk + 1 < kSbG - 1 ? jk + 1 : 0][(4 * kk + lk) * 16 + l15];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 16; kk++)
                    pa[kk & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_val(n, kk), bv[jk & 1][kk], pa[kk & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            const v4f64 acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            constexpr int n = n0 + jb;
            // Inv[bl,cl] = -Linv_bl * S: S changes hands between the wavefronts (this one holds the rows 16 wv + lk + 4 reg)
            fetch_op(n + kInvRing - 1);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Ss[(16 * wv + lk + 4 * reg) * 16 + l15] = acc[reg];
            sb_bar();
            v4f64 po[4];
#pragma unroll
            for (int c = 0; c < 4; c++) po[c] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; kk++) po[kk & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_val(n, kk), Ss[(4 * kk + lk) * 16 + l15], po[kk & 3], 0, 0, 0);
            const v4f64 o = (po[0] + po[1]) + (po[2] + po[3]);
            double *t = P.SbInv + sb_tile(F, B, bl, cl);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int i = 16 * wv + lk + 4 * reg, col = 16 * sl + l15;
                const double v = -o[reg];
                if (jb < kSbG - 1) Bs[jb < kSbG - 1 ? jb : 0][i * 16 + l15] = v;
                t[i + 64 * col] = v;                                             // column-major half [i + 64 k]
                t[4096 + 64 * i + col] = v;                                      // row-major half [64 i + k]
            }
            sb_bar();                     // the new tile is visible to every wavefront; Ss may be written again
            if constexpr (jb + 1 < kSbG) self(self, std::integral_constant<int, jb + 1>{});
        }
    };
    row(row, std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------
// forward sweep.  One workgroup streams at most ~30 GB/s (measured, r03c: 160 KB per hop in 5.4 us), so every 64-row block is
// shared by TWO workgroups: ticket t -> row block b = t / 2, half h = t % 2 = the rows 32 h .. 32 h + 31 of the block.  A half-tile
// is 32 rows x 64 columns: lane -> row 32 h + (lane & 31), column phase lane >> 5; wave v -> the columns 2 v + phase + 32 t.
// Both halves publish their 32 of the panel's 64 slots; the upper half (h = 1) also needs the lower half's r (lower ticket).
// ------------------------------------------------------------------------------------------
// the two column / row phases of a wave are added inside the wave (lanes l and l + 32 hold the same row), the 16 per-wave partial
// sums of a row go through LDS and are added in a fixed tree
__device__ __forceinline__ void put_part(double (*red)[32], int wv, int lr, double v) {
    v += __shfl_xor(v, 32, 64);
    red[wv][lr] = v;                             // both half-waves store the same value
}
__device__ __forceinline__ double sum_parts(const double (*red)[32], int lr) {
    double q[4];
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = (red[4 * c][lr] + red[4 * c + 1][lr]) + (red[4 * c + 2][lr] + red[4 * c + 3][lr]);
    return (q[0] + q[1]) + (q[2] + q[3]);
}
// whole-wave wait for HALF a panel's slots (32 slots, both half-waves poll the same ones)
__device__ __forceinline__ bool half_slot_wait(const FrontSlot *p32, int lane, double &v, int *err, int *failflag, unsigned lim) {
    return front_slot_wait(p32 + (lane & 31), v, err, failflag, lim);
}

__global__ void __launch_bounds__(kSbThreads)
k_front_fwd_sb(DevPlan P, FrontDesc F, double *__restrict__ y, double *__restrict__ z) {
    __shared__ double red[NW][32];
    __shared__ double ybuf[2][kSbG][64];
    __shared__ double rbuf[kSbG][64];
    __shared__ int64_t pn_off[kSbMaxPanels];   // panel table of the front in LDS: reading a record must not wait on the tile loads in flight
    __shared__ int pn_r[kSbMaxPanels], pn_w[kSbMaxPanels];
    __shared__ int sb, okflag;
    int *sync = P.front_sync + F.sync_off;
    for (int q = threadIdx.x; q < kSbG * 64; q += kSbThreads) (&rbuf[0][0])[q] = 0.0;
    for (int q = threadIdx.x; q < F.np; q += kSbThreads) {
        const FrontPanel *fp_ = P.front_panels + F.fp_off + q;
        pn_off[q] = fp_->panel_off; pn_r[q] = fp_->r; pn_w[q] = fp_->w;
    }
    if (threadIdx.x == 0) { sb = atomicAdd(sync, 1); okflag = 1; }
    __syncthreads();
    const int tk = sb_rfl(sb);                  // wave-uniform for the compiler too: scalar branches, SGPR tile bases
    const int b = tk >> 1, h = tk & 1;
    if (b >= F.nb) return;
    SWT(0);
    // re-arm the backward sweep's block (idle during this launch), every workgroup a share
    for (int q = tk * kSbThreads + threadIdx.x; q < F.sync_blk; q += 2 * F.nb * kSbThreads) sync[F.sync_blk + q] = 0;
    FrontSlot *yslots = front_slots(sync, F.np), *rslots = yslots + (int64_t)F.np * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = sb_rfl(tid >> 6);
    const int lr = lane & 31, ph = lane >> 5, ir = 32 * h + lr, k0 = 2 * wv + ph;      // row inside the block, first column
    const FrontPanel *fps = P.front_panels + F.fp_off;
    const int g = kSbG;
    const bool own = b < F.np;
    struct { int w, f; int64_t diag_off; } me;
    {
        const FrontPanel *mp = fps + (own ? b : 0);
        me.w = sb_rfl(mp->w); me.f = sb_rfl(mp->f); me.diag_off = sb_rfl64(mp->diag_off);
    }
    const int B = own ? b / g : F.nsb, bl = own ? b - g * B : 0;
    const int i0 = own ? F.cw * b : F.W + 64 * (b - F.np);
    const int nrows = own ? me.w : min(64, F.rF - i0);
    const int i = i0 + ir;
    const bool valid = ir < nrows;
    const bool lead = wv == 0 && ph == 0;        // the 32 lanes that own this half's rows in the scalar parts
    // this row's start value: own rows  b_i - (external children), rows below the front  + (external children)
    double base = 0.0;
    if (lead && valid) {
        double G = 0.0;
        const int64_t g0 = P.front_gptr[F.gptr_off + i], g1 = P.front_gptr[F.gptr_off + i + 1];
        for (int64_t gq = g0; gq < g1; gq++) G += P.ubuf[P.front_gidx[gq]];
        base = own ? y[me.f + ir] - G : G;
    }
    const double dinv_own = (own && valid && lead) ? P.Dinv[me.f + ir] : 0.0;
    const int nQ = own ? B : F.nsb;            // super-blocks whose panels this row block accumulates
    double cur[kSbG][NT], nxt[kSbG][NT];       // the group being consumed and the one prefetched behind it
    // the L tiles of super-block G on this half row block
    auto load_L = [&](int G, double (&dst)[kSbG][NT]) {
#pragma unroll
        for (int p = 0; p < kSbG; p++) {
            const int q = g * G + p;
            const int qc = q < F.np ? q : F.np - 1;
            const int fq_r = sb_rfl(pn_r[qc]), fq_w = sb_rfl(pn_w[qc]);
            const double *pb = P.Lx + sb_rfl64(pn_off[qc]);
            const unsigned o0 = (unsigned)(i - F.cw * q + k0 * fq_r) * 8u, os = (unsigned)fq_r * (8u * 32u);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int k = k0 + 32 * t;
                dst[p][t] = (q < F.np && valid && k < fq_w && !(P.dbg & 1)) ? sb_ld(pb, o0 + (unsigned)t * os) : 0.0;
            }
        }
    };
    // The inverse tiles Inv[b, gB .. b] of the own super-block (own rows only) do not depend on the sweep: they are requested NOW, into
    // registers of their own, with unconditional loads (addresses clamped into the tiles; the zero pattern is applied where the values
    // are used).  [Round 6: they used to be the prefetch of the LAST hop -- issued by the polling wavefronts after their poll had
    // succeeded and awaited at that hop's barrier: one memory round trip on the critical path of every super-block.]
    double inv[kSbG][NT];
#pragma unroll
    for (int p = 0; p < kSbG; p++) {
        if (own && p < bl) {                     // (uniform) a full tile of the super-block inverse, column-major (rows past the panel's width are zero)
            const double *tl = P.SbInv + sb_tile(F, B, bl, p);
#pragma unroll
            for (int t = 0; t < NT; t++) inv[p][t] = sb_ld(tl, (unsigned)(ir + 64 * (k0 + 32 * t)) * 8u);
        } else if (own && p == bl) {             // the panel's own inverse (lower triangular, w x w column-major)
            const double *li = P.Linv + me.diag_off;
            const int irc = min(ir, me.w - 1);
#pragma unroll
            for (int t = 0; t < NT; t++) inv[p][t] = sb_ld(li, (unsigned)(irc + min(k0 + 32 * t, irc) * me.w) * 8u);
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) inv[p][t] = 0.0;
        }
    }
    double a0 = 0.0, a1 = 0.0;
    // wait for the y of super-block Q's panels while the next group is prefetched into `nxt`, accumulate `cur`, cur <- nxt;
    // false = give up.  The polling waves issue their share of the prefetch AFTER their poll has succeeded: a wave's loads return
    // in order, so a poll behind its tile loads could not see its slot before those have landed (they then have a whole hop to land).
    auto consume = [&](int Q, auto &&prefetch) -> bool {
        double (*yb)[64] = ybuf[Q & 1];
        if (wv < kSbG) {                         // wave w polls the 64 slots of panel g Q + w (value and validity in one load per lane)
            const int q = g * Q + wv;
            double v = 0.0;
            if (q < F.np) {
                const bool okw = front_slot_wait(yslots + q * 64 + lane, v, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
                if (!okw) { v = 0.0; if (lane == 0) okflag = 0; }
            }
            yb[wv][lane] = v;
        }
        prefetch();
        __syncthreads();
        if (!okflag) return false;
#pragma unroll
        for (int p = 0; p < kSbG; p++) {
            a0 = fma(cur[p][0], yb[p][k0], a0);
            a1 = fma(cur[p][1], yb[p][k0 + 32], a1);
        }
        return true;
    };
    bool ok = true;
    if (nQ > 0) {
        load_L(0, cur);
        for (int Q = 0; Q + 1 < nQ && ok; Q++) {
            ok = consume(Q, [&] { load_L(Q + 1, nxt); });
#pragma unroll
            for (int p = 0; p < kSbG; p++)
#pragma unroll
                for (int t = 0; t < NT; t++) cur[p][t] = nxt[p][t];
        }
        SWT(1);
        if (ok) ok = consume(nQ - 1, [] {});                       // the last hop has nothing left to request
        SWT(2);
    }
    put_part(red, wv, lr, a0 + a1);
    __syncthreads();
    SWT(3);
    if (!own) {
        if (lead && valid && ok) P.ubuf[F.ubelow_off + (i - F.W)] = base + sum_parts(red, lr);
        return;
    }
    if (wv == 0) {                               // partial right-hand side of this half panel: to the later panels of the super-block
        if (ph == 0) {
            const double r = valid ? base - sum_parts(red, lr) : 0.0;
            rbuf[bl][ir] = r;
            if (ok) front_slot_st(rslots + b * 64 + ir, r);
        }
    } else if (wv <= bl) {                       // wave w polls r of panel g B + w - 1 (both halves)
        const int c = wv - 1;
        double v = 0.0;
        const bool okw = ok && front_slot_wait(rslots + (g * B + c) * 64 + lane, v, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
        if (!okw) { v = 0.0; if (lane == 0) okflag = 0; }
        rbuf[c][lane] = v;
    } else if (wv == bl + 1 && h == 1) {         // the upper half also needs the lower half's r of this very panel (ticket tk - 1)
        double v = 0.0;
        const bool okw = ok && half_slot_wait(rslots + b * 64, lane, v, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
        if (!okw) { v = 0.0; if (lane == 0) okflag = 0; }
        rbuf[bl][lane & 31] = v;
    }
    __syncthreads();
    SWT(4);
    if (!okflag) return;
    {   // y_b = sum_c Inv[b,c] r_c  (tiles of the panels after b in the super-block are zero)
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int p = 0; p < kSbG; p++) {
            const bool m0 = valid && (p < bl || k0 <= ir), m1 = valid && (p < bl || k0 + 32 <= ir);
            s0 = fma(m0 ? inv[p][0] : 0.0, rbuf[p][k0], s0);
            s1 = fma(m1 ? inv[p][1] : 0.0, rbuf[p][k0 + 32], s1);
        }
        put_part(red, wv, lr, s0 + s1);          // (the first reduction's reads finished before the barrier above)
    }
    __syncthreads();
    SWT(5);
    if (lead) {
        const double v = valid ? sum_parts(red, lr) : 0.0;
        front_slot_st(yslots + b * 64 + ir, v);              // first: the next super-block is waiting for it
        if (valid) z[me.f + ir] = v * dinv_own;
    }
    SWT(6);
}

// ------------------------------------------------------------------------------------------
// backward sweep: ticket t -> panel p = np-1-t/2, half h = 1 - t % 2 (the UPPER column half first: the lower one needs its s).
// lane -> column 32 h + (lane & 31) of the panel, row phase lane >> 5; wave v -> the rows 2 v + phase + 32 t of every tile
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kSbThreads)
k_front_bwd_sb(DevPlan P, FrontDesc F, const double *__restrict__ z, double *__restrict__ x, double *__restrict__ xout) {
    __shared__ double red[NW][32];
    __shared__ double xbuf[2][kSbG][64];
    __shared__ double sbuf[kSbG][64];
    __shared__ int pn_w[kSbMaxPanels];          // panel widths in LDS (see k_front_fwd_sb)
    __shared__ int sb, okflag;
    int *sync = P.front_sync + F.sync_off + F.sync_blk;
    for (int q = threadIdx.x; q < kSbG * 64; q += kSbThreads) (&sbuf[0][0])[q] = 0.0;
    for (int q = threadIdx.x; q < F.np; q += kSbThreads) pn_w[q] = P.front_panels[F.fp_off + q].w;
    if (threadIdx.x == 0) { sb = atomicAdd(sync, 1); okflag = 1; }
    __syncthreads();
    const int tk = sb_rfl(sb);                  // wave-uniform for the compiler too
    if (tk >= 2 * F.np) return;
    // re-arm the forward sweep's block for the next solve (idle during this launch), every workgroup a share
    for (int q = tk * kSbThreads + threadIdx.x; q < F.sync_blk; q += 2 * F.np * kSbThreads) sync[q - F.sync_blk] = 0;
    FrontSlot *xslots = front_slots(sync, F.np), *sslots = xslots + (int64_t)F.np * 64;
    const int p = F.np - 1 - (tk >> 1), h = 1 - (tk & 1);
    const int tid = threadIdx.x, lane = tid & 63, wv = sb_rfl(tid >> 6);
    const int kc = 32 * h + (lane & 31), ph = lane >> 5, j0 = 2 * wv + ph;       // column of the panel, first tile row
    const FrontPanel *fps = P.front_panels + F.fp_off;
    struct { int w, f; int64_t diag_off, lt_off; } me;
    me.w = sb_rfl(fps[p].w); me.f = sb_rfl(fps[p].f); me.diag_off = sb_rfl64(fps[p].diag_off); me.lt_off = sb_rfl64(fps[p].lt_off);
    const int w = me.w;
    const bool cvalid = kc < w;
    const bool lead = wv == 0 && ph == 0;
    const int g = kSbG;
    const int B = p / g, pl = p - g * B, nbB = min(g, F.np - g * B);
    const double *lt = P.LT + me.lt_off;            // row-major: lt[(j - w) * w + k], j = local panel row
    const int perm_own = (cvalid && lead) ? P.perm[me.f + kc] : 0;
    const double zin = (cvalid && lead) ? z[me.f + kc] : 0.0;
    double a0 = 0.0, a1 = 0.0;
    // (A) rows below the front: x is final
    {
        const int *rows = P.sn_rows + F.rows_off;
        for (int ib = F.W; ib < F.rF; ib += 64) {
            double xv[NT], l[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int i = ib + j0 + 32 * t;
                const bool in = i < F.rF;
                xv[t] = in ? x[rows[in ? i : F.W]] : 0.0;
                l[t] = (in && cvalid) ? lt[(int64_t)(i - F.cw * p - w) * w + kc] : 0.0;
            }
            a0 = fma(l[0], xv[0], a0);
            a1 = fma(l[1], xv[1], a1);
        }
    }
    // (B) later super-blocks in descending order (groups 0 .. nG-1), then (group nG) the inverse tiles Inv[p .. , p]^T
    const int nG = F.nsb - 1 - B;
    double cur[kSbG][NT], nxt[kSbG][NT];
    auto load_L = [&](int G, double (&dst)[kSbG][NT]) {      // L[q, p]^T tiles of super-block Q = nsb - 1 - G, from the row-major copy
        const int Q = F.nsb - 1 - G;
#pragma unroll
        for (int pp = 0; pp < kSbG; pp++) {
            const int q = g * Q + pp;
            const int fq_w = sb_rfl(pn_w[q < F.np ? q : F.np - 1]);
            const unsigned o0 = (unsigned)((F.cw * (q - p) + j0 - w) * w + kc) * 8u, os = (unsigned)w * (8u * 32u);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int jr = j0 + 32 * t;
                dst[pp][t] = (q < F.np && cvalid && jr < fq_w && !(P.dbg & 1)) ? sb_ld(lt, o0 + (unsigned)t * os) : 0.0;
            }
        }
    };
    // the inverse tiles Inv[p .. , p]^T of the own super-block: requested now, unconditionally, into registers of their own (see
    // k_front_fwd_sb)
    double inv[kSbG][NT];
#pragma unroll
    for (int cl = 0; cl < kSbG; cl++) {
        if (cl > pl && cl < nbB) {               // (uniform) a full tile of the super-block inverse, row-major half (columns past the width are zero)
            const double *tl = P.SbInv + sb_tile(F, B, cl, pl) + 4096;
#pragma unroll
            for (int t = 0; t < NT; t++) inv[cl][t] = sb_ld(tl, (unsigned)((j0 + 32 * t) * 64 + kc) * 8u);
        } else if (cl == pl) {                   // the panel's own inverse, transposed copy: Linv[i][k] at [k + i w]
            const double *lit = P.LinvT + me.diag_off;
            const int kcc = min(kc, w - 1);
#pragma unroll
            for (int t = 0; t < NT; t++) inv[cl][t] = sb_ld(lit, (unsigned)(kcc + min(max(j0 + 32 * t, kcc), w - 1) * w) * 8u);
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) inv[cl][t] = 0.0;
        }
    }
    auto consume = [&](int G, auto &&prefetch) -> bool {        // see k_front_fwd_sb
        const int Q = F.nsb - 1 - G;
        double (*xb)[64] = xbuf[G & 1];
        if (wv < kSbG) {
            const int q = g * Q + wv;
            double v = 0.0;
            if (q < F.np) {
                const bool okw = front_slot_wait(xslots + q * 64 + lane, v, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
                if (!okw) { v = 0.0; if (lane == 0) okflag = 0; }
            }
            xb[wv][lane] = v;
        }
        prefetch();
        __syncthreads();
        if (!okflag) return false;
#pragma unroll
        for (int pp = 0; pp < kSbG; pp++) {
            a0 = fma(cur[pp][0], xb[pp][j0], a0);
            a1 = fma(cur[pp][1], xb[pp][j0 + 32], a1);
        }
        return true;
    };
    bool ok = true;
    if (nG > 0) {
        load_L(0, cur);
        for (int G = 0; G + 1 < nG && ok; G++) {
            ok = consume(G, [&] { load_L(G + 1, nxt); });
#pragma unroll
            for (int pp = 0; pp < kSbG; pp++)
#pragma unroll
                for (int t = 0; t < NT; t++) cur[pp][t] = nxt[pp][t];
        }
        if (ok) ok = consume(nG - 1, [] {});
    }
    put_part(red, wv, lane & 31, a0 + a1);
    __syncthreads();
    if (wv == 0) {                               // partial right-hand side of this half panel: to the EARLIER panels of the super-block
        if (ph == 0) {
            const double s = cvalid ? zin - sum_parts(red, lane & 31) : 0.0;
            sbuf[pl][kc] = s;
            if (ok) front_slot_st(sslots + p * 64 + kc, s);
        }
    } else if (pl + wv < nbB) {                  // wave v polls s of panel p + v (both halves)
        double v = 0.0;
        const bool okw = ok && front_slot_wait(sslots + (p + wv) * 64 + lane, v, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
        if (!okw) { v = 0.0; if (lane == 0) okflag = 0; }
        sbuf[pl + wv][lane] = v;
    } else if (pl + wv == nbB && h == 0) {       // the lower half also needs the upper half's s of this very panel (ticket tk - 1)
        double v = 0.0;
        const bool okw = ok && half_slot_wait(sslots + p * 64 + 32, lane, v, sync + 1, P.flags + FL_FRONTFAIL, P.spin_limit);
        if (!okw) { v = 0.0; if (lane == 0) okflag = 0; }
        sbuf[pl][32 + (lane & 31)] = v;
    }
    __syncthreads();
    if (!okflag) return;
    {   // x_p = sum_{c >= p} Inv[c,p]^T s_c
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int cl = 0; cl < kSbG; cl++) {
            const bool m0 = cvalid && (cl > pl || (j0 >= kc && j0 < w)), m1 = cvalid && (cl > pl || (j0 + 32 >= kc && j0 + 32 < w));
            s0 = fma(m0 ? inv[cl][0] : 0.0, sbuf[cl][j0], s0);
            s1 = fma(m1 ? inv[cl][1] : 0.0, sbuf[cl][j0 + 32], s1);
        }
        put_part(red, wv, lane & 31, s0 + s1);
    }
    __syncthreads();
    if (lead) {
        const double v = cvalid ? sum_parts(red, lane & 31) : 0.0;
        front_slot_st(xslots + p * 64 + kc, v);              // first: the previous super-block is waiting for it
        if (cvalid) {
            x[me.f + kc] = v;
            xout[perm_own] = v;
        }
    }
}

void launch_front_fwd_sb(hipStream_t st, const DevPlan &P, const FrontDesc &F, double *y, double *z) {
    hipLaunchKernelGGL(k_front_fwd_sb, dim3(2 * F.nb), dim3(kSbThreads), 0, st, P, F, y, z);
}
void launch_front_bwd_sb(hipStream_t st, const DevPlan &P, const FrontDesc &F, const double *z, double *x, double *xout) {
    hipLaunchKernelGGL(k_front_bwd_sb, dim3(2 * F.np), dim3(kSbThreads), 0, st, P, F, z, x, xout);
}
void launch_invert_super(hipStream_t st, const DevPlan &P, const FrontDesc &F) {
    if (F.sb_g <= 0 || F.nsb <= 0) return;
    if (F.cw == 64) hipLaunchKernelGGL(k_invert_super<true>, dim3(F.nsb * (kSbG - 1) * kInvSlices), dim3(kInvThreads), 0, st, P, F);
    else hipLaunchKernelGGL(k_invert_super<false>, dim3(F.nsb * (kSbG - 1) * kInvSlices), dim3(kInvThreads), 0, st, P, F);
}

}  // namespace hipkkt
