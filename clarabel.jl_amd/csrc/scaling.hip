// update_scaling! + get_Hs! of the symmetric cones on the device, written straight into the resident KKT
// values (kktsolver_directldl.jl:197-245 then consumes them unchanged).  Given (s, z) in cone order:
//   Zero          Hs = 0                                                       (coneops_zerocone.jl:91)
//   Nonnegative   lambda = sqrt(s z), w = sqrt(s / z), Hs = w^2                (coneops_nncone.jl:77-101)          bit-exact
//   SecondOrder   eta, w, lambda, sparse (d, u, v) or the dense <= 4 block     (coneops_socone.jl:75-192)          sums are tree sums
//   PSDTriangle   W = R R^T from the caller's R (the Cholesky / SVD of :78-143 stay with the caller), then the skron block of
//                 k_psd_hs (kernels.hip)                                       (coneops_psdtrianglecone.jl:145-161)
// Expressions keep the reference's association; this file is compiled with -ffp-contract=off so that a*x + b*y is two rounded
// products and a sum like Julia's, not an FMA.  Everything is HBM / latency bound: 2 m doubles in, (w, lambda) out, O(m) K entries.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace hipkkt {

// one entry per row of a Zero / Nonnegative cone: kind 0 zero, 1 nonnegative, other rows are skipped
__global__ void __launch_bounds__(256)
k_scaling_diag(const signed char *__restrict__ row_kind, const int64_t *__restrict__ row_hs, const int64_t *__restrict__ map_hs,
               const double *__restrict__ s, const double *__restrict__ z, double *__restrict__ w, double *__restrict__ lam,
               double *__restrict__ kval, int64_t m) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int kind = row_kind[i];
    if (kind == 0) {
        w[i] = 0.0; lam[i] = 0.0;
        kval[map_hs[row_hs[i]]] = -0.0;
    } else if (kind == 1) {
        const double l = sqrt(s[i] * z[i]), ww = sqrt(s[i] / z[i]);
        lam[i] = l; w[i] = ww;
        kval[map_hs[row_hs[i]]] = -(ww * ww);
    }
}

__device__ __forceinline__ double block_sum(double v, double *red) {      // 256 threads; every thread gets the total
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    __syncthreads();                                                    // red[] may still be read from the previous sum
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double sqrt_soc_residual(double z0, double z1norm) {   // coneops_socone.jl:395-407
    const double r = (z0 - z1norm) * (z0 + z1norm);
    return r > 0.0 ? sqrt(r) : 0.0;
}

// one workgroup per second-order cone.  desc[c] = {first row, dim, first Hs entry, offset into the concatenated (u, v) of the sparse
// cones or -1 for a dense (dim <= 4) block, ordinal among the sparse cones or -1}
__global__ void __launch_bounds__(256)
k_scaling_soc(const int64_t *__restrict__ desc, const int64_t *__restrict__ map_hs, const double *__restrict__ s_all,
              const double *__restrict__ z_all, double *__restrict__ w_all, double *__restrict__ lam_all, double *__restrict__ eta_out,
              double *__restrict__ soc_u, double *__restrict__ soc_v, double *__restrict__ soc_eta2, double *__restrict__ kval,
              int *__restrict__ fail) {
    __shared__ double red[4];
    const int c = blockIdx.x, t = threadIdx.x;
    const int64_t row0 = desc[5 * c], dim = desc[5 * c + 1], hs0 = desc[5 * c + 2], uv0 = desc[5 * c + 3], ord = desc[5 * c + 4];
    const double *s = s_all + row0, *z = z_all + row0;
    double *w = w_all + row0, *lam = lam_all + row0;
    double a = 0.0, b = 0.0;
    for (int64_t i = 1 + t; i < dim; i += 256) { a += z[i] * z[i]; b += s[i] * s[i]; }
    const double z1 = sqrt(block_sum(a, red)), s1 = sqrt(block_sum(b, red));
    const double z0 = z[0], s0 = s[0];
    const double zscale = sqrt_soc_residual(z0, z1), sscale = sqrt_soc_residual(s0, s1);
    if (zscale == 0.0 || sscale == 0.0) { if (t == 0) atomicOr(fail, 1); return; }     // :88-90, uniform over the workgroup
    const double eta = sqrt(sscale / zscale);
    // w = s / sscale +- z / zscale, then normalised (:96-113)
    a = 0.0;
    for (int64_t i = t; i < dim; i += 256) {
        double v = s[i] / sscale;
        if (i == 0) v += z0 / zscale; else { v -= z[i] / zscale; a += v * v; }
        w[i] = v;
    }
    const double w1 = sqrt(block_sum(a, red));        // block_sum's barriers also order the w[] stores before the reads below
    const double wscale = sqrt_soc_residual(s0 / sscale + z0 / zscale, w1);
    if (wscale == 0.0) { if (t == 0) atomicOr(fail, 1); return; }
    a = 0.0;
    for (int64_t i = 1 + t; i < dim; i += 256) { const double v = w[i] / wscale; w[i] = v; a += v * v; }
    const double w1sq = block_sum(a, red);
    const double w0 = sqrt(1.0 + w1sq);
    // lambda = W z (:115-123)
    const double gamma = 0.5 * wscale;
    const double cs = (gamma + z0 / zscale) / sscale, cz = (gamma + s0 / sscale) / zscale;
    const double cinv = 1.0 / (s0 / sscale + z0 / zscale + 2.0 * gamma), root = sqrt(sscale * zscale);
    for (int64_t i = 1 + t; i < dim; i += 256) {
        double l = cs * s[i] + cz * z[i];
        l *= cinv;
        lam[i] = l * root;
    }
    const double eta2 = eta * eta;
    if (t == 0) { w[0] = w0; lam[0] = gamma * root; eta_out[c] = eta; }
    if (uv0 >= 0) {
        // sparse expansion terms (:125-153) and the diagonal block eta^2 * [d, 1, ..., 1] (:166-171)
        const double alpha = 2.0 * w0, wsq = w0 * w0 + w1sq, wsqinv = 1.0 / wsq, d = wsqinv / 2.0;
        const double u0 = sqrt(wsq - d), u1 = alpha / u0, v1 = sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv));
        double *u = soc_u + uv0, *v = soc_v + uv0;
        for (int64_t i = t; i < dim; i += 256) {
            const double wi = i == 0 ? w0 : w[i];
            u[i] = i == 0 ? u0 : u1 * wi;
            v[i] = i == 0 ? 0.0 : v1 * wi;
            kval[map_hs[hs0 + i]] = -(i == 0 ? eta2 * d : eta2);
        }
        if (t == 0) soc_eta2[ord] = eta2;
    } else if (t == 0) {
        // dense packed upper triangle of eta^2 (2 w w' - J), dim <= 4 (:173-189)
        const double r2 = sqrt(2.0);
        double wl[4];
        wl[0] = w0;
        for (int i = 1; i < (int)dim; i++) wl[i] = w[i];
        int64_t h = 0;
        kval[map_hs[hs0 + h++]] = -(((r2 * wl[0] - 1.0) * (r2 * wl[0] + 1.0)) * eta2);
        for (int col = 1; col < (int)dim; col++)
            for (int row = 0; row <= col; row++) {
                double e = 2.0 * wl[row] * wl[col];
                if (row == col) e += 1.0;
                kval[map_hs[hs0 + h++]] = -(e * eta2);
            }
    }
}

// W = R R^T of one PSD cone (n x n, column-major), both triangles with the same summation order so that W is exactly symmetric
__global__ void __launch_bounds__(256)
k_psd_rrt(const double *__restrict__ R, double *__restrict__ W, int n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    const int i = (int)(e % n), j = (int)(e / n);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    double acc = 0.0;
    for (int k = 0; k < n; k++) acc += R[lo + (int64_t)k * n] * R[hi + (int64_t)k * n];
    W[e] = acc;
}

void launch_scaling_diag(hipStream_t st, const signed char *row_kind, const int64_t *row_hs, const int64_t *map_hs, const double *s,
                         const double *z, double *w, double *lam, double *kval, int64_t m) {
    if (m > 0)
        hipLaunchKernelGGL(k_scaling_diag, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, row_kind, row_hs, map_hs, s, z, w, lam,
                           kval, m);
}
void launch_scaling_soc(hipStream_t st, int nsoc, const int64_t *desc, const int64_t *map_hs, const double *s, const double *z,
                        double *w, double *lam, double *eta_out, double *soc_u, double *soc_v, double *soc_eta2, double *kval,
                        int *fail) {
    if (nsoc > 0)
        hipLaunchKernelGGL(k_scaling_soc, dim3(nsoc), dim3(256), 0, st, desc, map_hs, s, z, w, lam, eta_out, soc_u, soc_v, soc_eta2,
                           kval, fail);
}
void launch_psd_rrt(hipStream_t st, const double *R, double *W, int n) {
    const int64_t nn = (int64_t)n * n;
    if (nn > 0) hipLaunchKernelGGL(k_psd_rrt, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, R, W, n);
}

}  // namespace hipkkt
