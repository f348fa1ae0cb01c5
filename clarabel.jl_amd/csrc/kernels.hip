// HIP kernels of the KKT path for gfx950 (MI355X / CDNA4).  See DESIGN.md §5 for the roofline
// that bounds each one.  Everything is Float64; indices are int32 except panel offsets (int64).
//
//   value updates     k_scatter_values / k_scale_values / k_set_hs          (HBM/latency, K1-K2)
//   regulariser       k_maxabs_diag + k_init_panels                          (K3)
//   numeric LDL^T     k_factor_level  (LDS-resident diagonal block + TRSM)   (K4, latency/LDS)
//                     k_update_stage  (gather - MFMA f64 16x16x4 - scatter)  (K4, MFMA bound)
//   triangular solves k_fwd_level / k_bwd_partial / k_bwd_final              (K5, latency/HBM)
//   refinement        k_spmv_residual, k_norm_inf, k_axpy ...                (K6-K8)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_plan.h"

namespace hipkkt {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_max_abs(unsigned long long *slot, double v) {
    // |v| as an ordered integer; NaN (all-ones exponent, non-zero mantissa) compares largest,
    // so a NaN anywhere surfaces as a non-finite maximum
    unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(v));
    atomicMax(slot, bits);
}

__device__ __forceinline__ double wave_max(double v) {
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = (o > v || o != o) ? o : v;
    }
    return v;
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
    a = wave_max(a);
    if ((threadIdx.x & 63) == 0) atomic_max_abs(slot, a);
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
        pan[(lo + row) + (int64_t)k * r] = T[k * LDT + row] / dd[k];
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

__global__ void __launch_bounds__(256)
k_update_stage(DevPlan P, int group_begin) {
    __shared__ double Ct[kMaxSnWidth * LDC];
    const UpdGroup G = P.upd_groups[group_begin + blockIdx.x];
    const int t = G.tgt;
    const int ft = P.sn_first[t];
    const int wt = P.sn_first[t + 1] - ft;
    const int rt = (int)(P.sn_rowptr[t + 1] - P.sn_rowptr[t]);
    double *tp = P.Lx + P.sn_panel[t];
    const int tid = threadIdx.x;
    const int nrt = min(kUpdRows, rt - G.row_base);
    for (int idx = tid; idx < nrt * wt; idx += 256) {
        int row = idx % nrt, col = idx / nrt;
        Ct[col * LDC + row] = tp[(G.row_base + row) + (int64_t)col * rt];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    for (int q = G.task_begin; q < G.task_end; q++) {
        const UpdTask T = P.upd_tasks[q];
        const int s = T.src;
        const int fs = P.sn_first[s];
        const int K = P.sn_first[s + 1] - fs;
        const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
        const double *sp = P.Lx + P.sn_panel[s];
        const double *dv = P.D + fs;
        const int *srows = P.sn_rows + P.sn_rowptr[s];
        const int *rel = P.rel + T.rel_off;
        const int mt = (T.nrows + 15) >> 4, nt = (T.ncols + 15) >> 4;
        for (int tile = wave; tile < mt * nt; tile += 4) {
            const int tm = tile % mt, tn = tile / mt;
            const int ai = tm * 16 + l15, bj = tn * 16 + l15;
            const bool av = ai < T.nrows, bv = bj < T.ncols;
            const double *ap = sp + (T.row_lo + ai);
            const double *bp = sp + (T.col_lo + bj);
            v4f64 acc = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < K; k0 += 4) {
                const int kk = k0 + lk;
                const bool kv = kk < K;
                double a = 0.0, b = 0.0;
                if (av && kv) a = ap[(int64_t)kk * r];
                if (bv && kv) b = bp[(int64_t)kk * r] * dv[kk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            // C/D layout of the f64 16x16x4 form: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int ii = tm * 16 + lk + 4 * reg;
                const int jj = tn * 16 + l15;
                if (ii < T.nrows && jj < T.ncols) {
                    const int rp = rel[(T.row_lo + ii) - T.col_lo] - G.row_base;
                    const int cp = srows[T.col_lo + jj] - ft;
                    Ct[cp * LDC + rp] -= acc[reg];
                }
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < nrt * wt; idx += 256) {
        int row = idx % nrt, col = idx / nrt;
        tp[(G.row_base + row) + (int64_t)col * rt] = Ct[col * LDC + row];
    }
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
k_permute_in(const double *__restrict__ b, const int *__restrict__ perm, double *__restrict__ y, int n) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) y[k] = b[perm[k]];
}

__global__ void __launch_bounds__(256)
k_fwd_level(DevPlan P, int item_begin, double *__restrict__ y, double *__restrict__ z) {
    __shared__ double yv[kMaxSnWidth];
    __shared__ double part[4][kMaxSnWidth];
    __shared__ double Ld[kMaxSnWidth * kMaxSnWidth];
    const FacItem it = P.slv_items[item_begin + blockIdx.x];
    const int s = it.sn;
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
    const double *pan = P.Lx + P.sn_panel[s];
    const double *ld = P.Ldiag + P.sn_diag[s];
    const int tid = threadIdx.x;
    {
        const int c = tid & 63, pq = tid >> 6;
        double acc = 0.0;
        if (c < w) {
            const int64_t g0 = P.g_ptr[f + c], g1 = P.g_ptr[f + c + 1];
            for (int64_t g = g0 + pq; g < g1; g += 4) acc += P.ubuf[P.g_idx[g]];
        }
        part[pq][c] = acc;
    }
    for (int idx = tid; idx < w * w; idx += 256) Ld[idx] = ld[idx];
    __syncthreads();
    if (tid < 64) {
        double x = 0.0;
        if (tid < w) x = y[f + tid] - (((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]);
        for (int k = 0; k < w; k++) {
            const double xk = __shfl(x, k, 64);
            if (tid > k && tid < w) x -= Ld[tid + k * w] * xk;
        }
        if (tid < w) yv[tid] = x;
    }
    __syncthreads();
    if (it.blk == 0 && tid < w) {
        y[f + tid] = yv[tid];
        z[f + tid] = yv[tid] * P.Dinv[f + tid];
    }
    const int row = w + it.blk * 256 + tid;
    if (row < r) {
        double a = 0.0;
        for (int k = 0; k < w; k++) a += pan[row + (int64_t)k * r] * yv[k];
        P.ubuf[P.u_off[s] + (row - w)] = a;
    }
}

__global__ void __launch_bounds__(256)
k_bwd_partial(DevPlan P, int item_begin, const double *__restrict__ x) {
    __shared__ double red[4][kMaxSnWidth];
    const FacItem it = P.slv_items[item_begin + blockIdx.x];
    const int s = it.sn;
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
    const double *pan = P.Lx + P.sn_panel[s];
    const int *rows = P.sn_rows + P.sn_rowptr[s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = w + it.blk * 256 + tid;
    const double xr = row < r ? x[rows[row]] : 0.0;
    // every thread owns one row; reduce each column over the 256 rows: wave shuffle then LDS
    for (int k = 0; k < w; k++) {
        double v = row < r ? pan[row + (int64_t)k * r] * xr : 0.0;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < w) P.pbuf[P.p_off[s] + (int64_t)it.blk * w + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

__global__ void __launch_bounds__(64)
k_bwd_final(DevPlan P, int sn_begin, const double *__restrict__ z, double *__restrict__ x,
            double *__restrict__ xout) {
    __shared__ double Ld[kMaxSnWidth * kMaxSnWidth];
    const int s = P.lvl_sn[sn_begin + blockIdx.x];
    const int f = P.sn_first[s];
    const int w = P.sn_first[s + 1] - f;
    const int r = (int)(P.sn_rowptr[s + 1] - P.sn_rowptr[s]);
    const double *ld = P.Ldiag + P.sn_diag[s];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < w * w; idx += 64) Ld[idx] = ld[idx];
    const int nblk = (r - w + 255) / 256;
    double v = 0.0;
    if (tid < w) {
        double acc = 0.0;
        const double *pb = P.pbuf + P.p_off[s] + tid;
        for (int b = 0; b < nblk; b++) acc += pb[(int64_t)b * w];
        v = z[f + tid] - acc;
    }
    __syncthreads();
    // unit-upper solve with L11^T: x_k = v_k - sum_{i>k} L11[i][k] x_i
    for (int k = w - 1; k >= 0; k--) {
        const double xk = __shfl(v, k, 64);
        if (tid < k) v -= Ld[k + tid * w] * xk;
    }
    if (tid < w) {
        x[f + tid] = v;
        xout[P.perm[f + tid]] = v;
    }
}

// ------------------------------------------------------------------------------------------
// K6-K8: iterative refinement pieces (ref: kktsolver_directldl.jl:389-466)
// e = b - K*xi with the symmetric CSR view of the unregularised K; 8 lanes per row
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_spmv_residual(const int64_t *__restrict__ rowptr, const int *__restrict__ col, const int64_t *__restrict__ qidx,
                const double *__restrict__ kval, const double *__restrict__ b, const double *__restrict__ xi,
                double *__restrict__ e, int n, unsigned long long *__restrict__ norm_slot) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = gid >> 3, sub = gid & 7;
    double acc = 0.0;
    if (row < n) {
        const int64_t p0 = rowptr[row], p1 = rowptr[row + 1];
        for (int64_t p = p0 + sub; p < p1; p += 8) acc += kval[qidx[p]] * xi[col[p]];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    double a = 0.0;
    if (row < n && sub == 0) {
        const double ev = b[row] - acc;
        e[row] = ev;
        a = fabs(ev);
    }
    a = wave_max(a);
    if ((threadIdx.x & 63) == 0) atomic_max_abs(norm_slot, a);
}

__global__ void k_norm_inf(const double *__restrict__ v, int n, unsigned long long *__restrict__ slot) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double a = i < n ? fabs(v[i]) : 0.0;
    a = wave_max(a);
    if ((threadIdx.x & 63) == 0) atomic_max_abs(slot, a);
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
// host-callable launchers (used by hipkkt.cpp; all launches go to the handle's stream)
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
void launch_update_stage(hipStream_t st, const DevPlan &P, int group_begin, int ngroups) {
    if (ngroups > 0) hipLaunchKernelGGL(k_update_stage, dim3(ngroups), dim3(256), 0, st, P, group_begin);
}
void launch_mfma_probe(hipStream_t st, const double *A, const double *B, double *out) {
    hipLaunchKernelGGL(k_mfma_probe, dim3(1), dim3(64), 0, st, A, B, out);
}
void launch_permute_in(hipStream_t st, const double *b, const int *perm, double *y, int n) {
    if (n > 0) hipLaunchKernelGGL(k_permute_in, dim3(nblk(n)), dim3(256), 0, st, b, perm, y, n);
}
void launch_fwd_level(hipStream_t st, const DevPlan &P, int item_begin, int nitems, double *y, double *z) {
    if (nitems > 0) hipLaunchKernelGGL(k_fwd_level, dim3(nitems), dim3(256), 0, st, P, item_begin, y, z);
}
void launch_bwd_partial(hipStream_t st, const DevPlan &P, int item_begin, int nitems, const double *x) {
    if (nitems > 0) hipLaunchKernelGGL(k_bwd_partial, dim3(nitems), dim3(256), 0, st, P, item_begin, x);
}
void launch_bwd_final(hipStream_t st, const DevPlan &P, int sn_begin, int nsn, const double *z, double *x, double *xout) {
    if (nsn > 0) hipLaunchKernelGGL(k_bwd_final, dim3(nsn), dim3(64), 0, st, P, sn_begin, z, x, xout);
}
void launch_spmv_residual(hipStream_t st, const DevPlan &P, const double *b, const double *xi, double *e, int n,
                          unsigned long long *slot) {
    if (n > 0)
        hipLaunchKernelGGL(k_spmv_residual, dim3(nblk((int64_t)n * 8)), dim3(256), 0, st, P.sym_rowptr, P.sym_col, P.sym_q,
                           P.kval, b, xi, e, n, slot);
}
void launch_norm_inf(hipStream_t st, const double *v, int n, unsigned long long *slot) {
    if (n > 0) hipLaunchKernelGGL(k_norm_inf, dim3(nblk(n)), dim3(256), 0, st, v, n, slot);
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
