// What keeps a wavefront's FP64 matrix-core stream below its rate?  One wavefront per SIMD (256 workgroups x 4), 80 steps of 16
// v_mfma_f64_16x16x4_f64 on 16 accumulators, with per step: MODE 0 nothing else; 1 + 8 LDS reads (operands) ; 2 + LDS-only barrier;
// 3 + 4 LDS writes; 4 + 5 global loads (prefetched 4 steps ahead); 5 = 4 with the barrier in the middle of the 16 instructions
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_mix.hip -o tools/bin/ubench_mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int MODE, int OCC>
__global__ void __launch_bounds__(256, OCC) k_mix(const double *src, double *out, int steps) {
    __shared__ double sb[2][1280];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lk = lane >> 4;
    v4f64 acc[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < 2560; i += 256) (&sb[0][0])[i] = 1e-3 * (i & 255);
    __syncthreads();
    double a[4], b[4], g[4][5];
    for (int t = 0; t < 4; t++) { a[t] = 1.0 + 1e-6 * (lane + t); b[t] = 1.0 - 1e-6 * (lane + t); }
    const double *p = src + (size_t)blockIdx.x * 64 + lane;
    if (MODE >= 4) for (int u = 0; u < 4; u++) for (int q = 0; q < 5; q++) g[u][q] = p[(size_t)(u * 5 + q) * 4096];
    for (int s = 0; s < steps; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double *B = sb[u & 1];
            if (MODE >= 3) for (int q = 0; q < 4; q++) B[q * 320 + wave * 80 + lane] = MODE >= 4 ? g[u][q] : a[q];
            if (MODE >= 4) for (int q = 0; q < 5; q++) g[u][q] = p[(size_t)(((s + u + 4) % 64) * 5 + q) * 4096];
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 5) {
                for (int tj = 0; tj < 2; tj++) for (int ti = 0; ti < 4; ti++) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], b[ti], acc[tj][ti], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE >= 2) bar();
            double na[4], nb[4];
            if (MODE >= 1) for (int t = 0; t < 4; t++) { na[t] = B[640 + lk * 80 + 16 * t + l15]; nb[t] = B[lk * 80 + 16 * t + l15]; }
            __builtin_amdgcn_sched_barrier(0);
            for (int tj = MODE == 5 ? 2 : 0; tj < 4; tj++) for (int ti = 0; ti < 4; ti++) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], b[ti], acc[tj][ti], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 1) for (int t = 0; t < 4; t++) { a[t] = na[t] * 1.0000001; b[t] = nb[t]; }
        }
    }
    double sacc = 0;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int r = 0; r < 4; r++) sacc += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = sacc;
}
int main() {
    double *src, *out; hipMalloc(&src, (size_t)64 * 5 * 4096 * 8 + 1024 * 64 * 8); hipMalloc(&out, 1024 * 256 * 8);
    hipMemset(src, 0, (size_t)64 * 5 * 4096 * 8 + 1024 * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 80;
    auto run = [&](const char *nm, auto k, int wgs) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) { hipEventRecord(e0, 0); hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, src, out, steps); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-58s %7.1f us  = %5.1f cycles at 2.4 GHz per instruction and SIMD, %5.1f TFLOP/s\n", nm, best * 1e3, best * 1e-3 * 2.4e9 / (steps * 16 * (wgs < 256 ? 1.0 : wgs / 256.0)), 1024.0 * steps * 16 * 2.0 * wgs * 4 / (best * 1e-3) / 1e12);
    };
    run("0 matrix core only, 1 wavefront per SIMD", k_mix<0, 1>, 256); run("1 + 8 LDS reads per step", k_mix<1, 1>, 256); run("2 + LDS barrier per step", k_mix<2, 1>, 256);
    run("3 + 4 LDS writes per step", k_mix<3, 1>, 256); run("4 + 5 global loads per step, 4 steps ahead", k_mix<4, 1>, 256); run("5 = 4, barrier in the middle", k_mix<5, 1>, 256);
    run("0 matrix core only, 2 wavefronts per SIMD (512 workgroups)", k_mix<0, 2>, 512); run("4 with 2 wavefronts per SIMD", k_mix<4, 2>, 512);
    // (a quarter of the chip: the same time per wavefront as line 0 -- the 100 cycles are no power or clock effect; the cycles column of
    // this line counts the busy SIMDs only)
    run("0 matrix core only, 1 wavefront per SIMD, 64 workgroups only", k_mix<0, 1>, 64);
    return 0;
}
