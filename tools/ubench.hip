// Microbenchmarks used to calibrate DESIGN.md's kernel models on the MI355X box:
//   f64 MFMA peak (v_mfma_f64_16x16x4_f64), shader clock, dependent-FMA / LDS / L2 latencies,
//   kernel-boundary cost inside a hipGraph.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_mfma(double *out, int iters) {
    v4f64 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ void k_mfma_dep(double *out, int iters, long long *cyc) {
    v4f64 a0 = {0, 0, 0, 0};
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    long long t1 = clock64();
    out[threadIdx.x] = a0[0];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_chain(double *out, int iters, long long *cyc) {
    double a = 1.0 + threadIdx.x, b = 1.0000001, c = 1e-9;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    for (int i = 0; i < iters; i++) a = fma(a, b, c);
    long long t1 = clock64();
    long long w1 = wall_clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
__global__ void k_div_chain(double *out, int iters, long long *cyc) {
    double a = 1.0 + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) a = 1.0 / (a + 1.5);
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chase(const int *next, int iters, int *out, long long *cyc) {
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) p = next[p];
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_chase(int iters, int *out, long long *cyc) {
    __shared__ int nx[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) nx[i] = (i * 17 + 1) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) p = nx[p];
    long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_barrier_loop(int iters, int *out, long long *cyc) {
    __shared__ double s[256];
    s[threadIdx.x] = threadIdx.x;
    long long t0 = clock64();
    double a = 0;
    for (int i = 0; i < iters; i++) { __syncthreads(); a += s[(threadIdx.x + i) & 255]; s[threadIdx.x] = a; }
    long long t1 = clock64();
    out[threadIdx.x] = (int)a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_shfl_chain(int iters, double *out, long long *cyc) {
    double a = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) a += __shfl(a, i & 63, 64);
    long long t1 = clock64();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_empty(int *p) { if (p && threadIdx.x == 1000) p[0] = 1; }

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double *dout; CK(hipMalloc(&dout, 256 * 1024 * 64 * sizeof(double)));
    long long *dcyc, hc[4]; CK(hipMalloc(&dcyc, 4 * sizeof(long long)));
    int *dint; CK(hipMalloc(&dint, (1 << 23) + 4096));
    float ms;
    // 1. MFMA f64 peak: 256 CUs x 4 SIMD x WPS waves
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps, iters = 20000;
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, st, dout, 100);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, st, dout, iters);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4 peak: %d wave/SIMD  %.2f ms  %.1f TFLOP/s\n", wps, ms, flops / ms / 1e9);
    }
    // 2. dependent MFMA latency + clock
    hipLaunchKernelGGL(k_mfma_dep, dim3(1), dim3(64), 0, st, dout, 10000, dcyc);
    CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
    printf("dependent mfma_f64: %.1f cycles each\n", hc[0] / 10000.0);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_fma_chain, dim3(1), dim3(64), 0, st, dout, 1000000, dcyc);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
    printf("fma_f64 chain: %.2f cycles/fma, shader clock %.0f MHz (1 wave), wall_clock64 ticks %lld in %.3f ms -> %.1f MHz\n",
           hc[0] / 1e6, hc[0] / ms / 1e3, hc[1], ms, hc[1] / ms / 1e3);
    hipLaunchKernelGGL(k_div_chain, dim3(1), dim3(64), 0, st, dout, 100000, dcyc);
    CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
    printf("1/(x+c) f64 chain: %.1f cycles each\n", hc[0] / 1e5);
    // 3. pointer chase: L2-resident (256 KB) and HBM-sized (64 MB... limited to 4MB ints here)
    for (int n : {1 << 10, 1 << 16, 1 << 20}) {
        std::vector<int> h(n);
        for (int i = 0; i < n; i++) h[i] = (int)(((long long)i * 1000003LL + 12345) % n);
        CK(hipMemcpy(dint, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, dint, 2000, dint + (1 << 20), dcyc);
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, st, dint, 20000, dint + (1 << 20), dcyc);
        CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
        printf("global pointer chase over %d ints: %.0f cycles/load\n", n, hc[0] / 20000.0);
    }
    hipLaunchKernelGGL(k_lds_chase, dim3(1), dim3(64), 0, st, 100000, dint, dcyc);
    CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
    printf("LDS chase: %.1f cycles/read\n", hc[0] / 1e5);
    hipLaunchKernelGGL(k_barrier_loop, dim3(1), dim3(256), 0, st, 100000, dint, dcyc);
    CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
    printf("barrier + lds read + lds write loop (256 thr): %.1f cycles/iter\n", hc[0] / 1e5);
    hipLaunchKernelGGL(k_shfl_chain, dim3(1), dim3(64), 0, st, 100000, dout, dcyc);
    CK(hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost));
    printf("__shfl(variable lane) + add chain: %.1f cycles/iter\n", hc[0] / 1e5);
    // 4. kernel boundary: eager and graph, 1000 dependent tiny launches
    for (int mode = 0; mode < 2; mode++) {
        const int n = 1000;
        hipGraphExec_t ge = nullptr;
        if (mode == 1) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, dint);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        }
        CK(hipEventRecord(e0, st));
        if (mode == 0) for (int i = 0; i < n; i++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, dint);
        else CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %d empty 1-WG kernels back to back: %.2f us each\n", mode ? "graph" : "eager", n, ms * 1e3 / n);
    }
    return 0;
}
