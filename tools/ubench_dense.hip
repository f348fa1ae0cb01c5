// Microbenchmark of the inner loop of k_update_dense<4> (one wavefront per 64x64 tile, K = 4 x 64) with the
// pieces switched on one at a time, to see which of them keeps the kernel below the MFMA rate:
//   V0 MFMAs only   V1 + operand preparation (masks, scaling by -d)   V2 + operand loads (double buffered)
//   V3 + per-task restart of the load pipeline (4 tasks of K = 64)    V4 + tile load / store
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_dense.hip -o /tmp/ubench_dense
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define HK_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ double ld_off(const double *base, unsigned byte_off) {
    return *(const HK_GLOBAL double *)((const HK_GLOBAL char *)base + byte_off);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
struct Raw { double a[4], b[4], d; };

template <int V>
__device__ __forceinline__ void load(Raw &f, const double *sp, const double *dv, const unsigned (&coff)[4], const unsigned (&roff)[4],
                                     unsigned r8, int K, int k0, int lk) {
    int kk = k0 + lk;
    kk = kk < K ? kk : K - 1;
    const unsigned ko = (unsigned)kk * r8;
    if (V >= 2) {
        f.d = ld_off(dv, (unsigned)kk * 8u);
#pragma unroll
        for (int t = 0; t < 4; t++) f.a[t] = ld_off(sp, coff[t] + ko);
#pragma unroll
        for (int t = 0; t < 4; t++) f.b[t] = ld_off(sp, roff[t] + ko);
    } else {
        f.d = 1.0 + ko * 1e-9;
#pragma unroll
        for (int t = 0; t < 4; t++) f.a[t] = 1e-3 * (coff[t] + 1);
#pragma unroll
        for (int t = 0; t < 4; t++) f.b[t] = 1e-3 * (roff[t] + 1);
    }
}
template <int V>
__device__ __forceinline__ void mma(const Raw &f, v4f64 (&acc)[4][4], unsigned mbits, int K, int k0, int lk) {
    double a[4], b[4];
    if (V >= 1) {
        const double dk = (k0 + lk < K) ? -f.d : 0.0;
#pragma unroll
        for (int t = 0; t < 4; t++) a[t] = ((mbits >> t) & 1u) ? f.a[t] * dk : 0.0;
#pragma unroll
        for (int t = 0; t < 4; t++) b[t] = ((mbits >> (4 + t)) & 1u) ? f.b[t] : 0.0;
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++) { a[t] = f.a[t]; b[t] = f.b[t]; }
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int ti = 0; ti < 4; ti++) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], b[ti], acc[tj][ti], 0, 0, 0);
}

// src: per tile a private 4 x (64 cols x R rows) region; tiles: 64x64 each
template <int V>
__global__ void __launch_bounds__(256, 2) k_dense(const double *src, const double *dvec, double *tiles, int R, int ntask, int K) {
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int g = rfl(blockIdx.x * 4 + wave);
    const int l15 = lane & 15, lk = lane >> 4;
    double *tp = tiles + (size_t)g * 4096;
    v4f64 acc[4][4];
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) {
                const int ii = ti * 16 + l15, jj = tj * 16 + lk + 4 * reg;
                acc[tj][ti][reg] = V >= 4 ? tp[ii + jj * 64] : 1.0 + ii * 1e-6;
            }
    for (int q = 0; q < ntask; q++) {
        // every tile reads rows [rowA, rowA+64) and [rowB, rowB+64) of panel q (shared by all tiles, like a front)
        const double *sp = src + (size_t)q * 64 * R;
        const double *dv = dvec + q * 64;
        const unsigned r8 = (unsigned)R * 8u;
        const int rowA = rfl((g * 64) % (R - 64)), rowB = rfl((g * 192 + 64) % (R - 64));
        unsigned roff[4], coff[4], mbits = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int ii = x * 16 + l15;
            const bool ok = ii < 64 - (g & 1);
            roff[x] = (unsigned)(rowA + (ok ? ii : 0)) * 8u;
            mbits |= ok ? 16u << x : 0u;
            coff[x] = (unsigned)(rowB + (ok ? ii : 0)) * 8u;
            mbits |= ok ? 1u << x : 0u;
        }
        const int Kt = V >= 3 ? K : K * ntask;
        Raw fa, fb;
        load<V>(fa, sp, dv, coff, roff, r8, Kt, 0, lk);
        for (int k0 = 0; k0 < Kt; k0 += 8) {
            load<V>(fb, sp, dv, coff, roff, r8, Kt, k0 + 4, lk);
            __builtin_amdgcn_sched_barrier(0);
            mma<V>(fa, acc, mbits, Kt, k0, lk);
            __builtin_amdgcn_sched_barrier(0);
            load<V>(fa, sp, dv, coff, roff, r8, Kt, k0 + 8, lk);
            __builtin_amdgcn_sched_barrier(0);
            mma<V>(fb, acc, mbits, Kt, k0 + 4, lk);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V < 3) break;
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) {
                const int ii = ti * 16 + l15, jj = tj * 16 + lk + 4 * reg;
                if (V >= 4 || acc[tj][ti][reg] == 12345.678) tp[ii + jj * 64] = acc[tj][ti][reg];
            }
}

// V5: persistent, one wavefront per SIMD, grid-stride over tiles; the NEXT tile travels global -> LDS
// (global_load_lds_dwordx4, no registers) during the current tile's MFMAs; accumulators start from LDS.
__global__ void __launch_bounds__(256, 1) k_dense_persist(const double *src, const double *dvec, double *tiles, int R, int ntask, int K, int ntiles) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int l15 = lane & 15, lk = lane >> 4;
    double *tb = lds + wave * 4096;
    int g = rfl(blockIdx.x * 4 + wave);
    const int stride = gridDim.x * 4;
    if (g >= ntiles) return;
    auto prefetch = [&](int gt) {
        const double *tpn = tiles + (size_t)gt * 4096;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            const int col = 2 * i + (lane >> 5), row = 2 * (lane & 31);
            __builtin_amdgcn_global_load_lds((const HK_GLOBAL void *)(tpn + row + col * 64),
                                             (__attribute__((address_space(3))) void *)(tb + i * 128), 16, 0, 0);
        }
    };
    prefetch(g);
    for (;;) {
        double *tp = tiles + (size_t)g * 4096;
        v4f64 acc[4][4];
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): the tile is in LDS  (gfx9 encoding: vmcnt low bits 3:0 + 15:14, lgkm 11:8, exp 6:4)
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++)
#pragma unroll
                for (int ti = 0; ti < 4; ti++) {
                    const int ii = ti * 16 + l15, jj = tj * 16 + lk + 4 * reg;
                    acc[tj][ti][reg] = tb[ii + jj * 64];
                }
        const int gn = g + stride;
        for (int q = 0; q < ntask; q++) {
            const double *sp = src + (size_t)q * 64 * R;
            const double *dv = dvec + q * 64;
            const unsigned r8 = (unsigned)R * 8u;
            const int rowA = rfl((g * 64) % (R - 64)), rowB = rfl((g * 192 + 64) % (R - 64));
            unsigned roff[4], coff[4], mbits = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int ii = x * 16 + l15;
                const bool ok = ii < 64 - (g & 1);
                roff[x] = (unsigned)(rowA + (ok ? ii : 0)) * 8u;
                mbits |= ok ? 16u << x : 0u;
                coff[x] = (unsigned)(rowB + (ok ? ii : 0)) * 8u;
                mbits |= ok ? 1u << x : 0u;
            }
            Raw fa, fb;
            load<3>(fa, sp, dv, coff, roff, r8, K, 0, lk);
            for (int k0 = 0; k0 < K; k0 += 8) {
                load<3>(fb, sp, dv, coff, roff, r8, K, k0 + 4, lk);
                __builtin_amdgcn_sched_barrier(0);
                mma<3>(fa, acc, mbits, K, k0, lk);
                __builtin_amdgcn_sched_barrier(0);
                load<3>(fa, sp, dv, coff, roff, r8, K, k0 + 8, lk);
                __builtin_amdgcn_sched_barrier(0);
                mma<3>(fb, acc, mbits, K, k0 + 4, lk);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q == 0 && gn < ntiles) { __builtin_amdgcn_sched_barrier(0); prefetch(gn); __builtin_amdgcn_sched_barrier(0); }
        }
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++)
#pragma unroll
                for (int ti = 0; ti < 4; ti++) {
                    const int ii = ti * 16 + l15, jj = tj * 16 + lk + 4 * reg;
                    tp[ii + jj * 64] = acc[tj][ti][reg];
                }
        if (gn >= ntiles) break;
        g = gn;
    }
}

template <int V>
int run(const double *src, const double *dv, double *tiles, int R, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : {64, 256, 512, 768, 1024}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_dense<V>, dim3(wgs), dim3(256), 0, st, src, dv, tiles, R, 4, 64);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double flops = (double)wgs * 4 * 2.0 * 64 * 64 * 256;
        printf("V%d  %4d WGs (%4d tiles): %7.1f us  %5.1f TFLOP/s\n", V, wgs, wgs * 4, best * 1e3, flops / best / 1e9);
    }
    return 0;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int R = 5617;
    double *src, *dv, *tiles;
    CK(hipMalloc(&src, (size_t)4 * 64 * R * 8)); CK(hipMalloc(&dv, 256 * 8)); CK(hipMalloc(&tiles, (size_t)4096 * 4096 * 8));
    CK(hipMemset(src, 0, (size_t)4 * 64 * R * 8)); CK(hipMemset(dv, 0, 256 * 8)); CK(hipMemset(tiles, 0, (size_t)4096 * 4096 * 8));
    if (run<0>(src, dv, tiles, R, st)) return 1;
    if (run<1>(src, dv, tiles, R, st)) return 1;
    if (run<2>(src, dv, tiles, R, st)) return 1;
    if (run<3>(src, dv, tiles, R, st)) return 1;
    if (run<4>(src, dv, tiles, R, st)) return 1;
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipFuncSetAttribute((const void *)k_dense_persist, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        for (int ntiles : {1024, 2048, 3072, 3656, 4096}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k_dense_persist, dim3(256), dim3(256), 131072, st, src, dv, tiles, R, 4, 64, ntiles);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double flops = (double)ntiles * 2.0 * 64 * 64 * 256;
            printf("V5 persistent 256 WGs, %4d tiles: %7.1f us  %5.1f TFLOP/s\n", ntiles, best * 1e3, flops / best / 1e9);
        }
    }
    return 0;
}
