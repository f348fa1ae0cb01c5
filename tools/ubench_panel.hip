// Microbenchmark of the pivot loop of k_factor_panel (clarabel.jl_amd/csrc/kernels.hip): where do the
// cycles of one pivot step go?  Variants: 0 = full step, 1 = no division (multiply by d), 2 = no barrier
// (wrong results, timing only), 3 = no trailing FMAs.   Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_panel.hip -o tools/ubench_panel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double rl(double x, int l) {
    const unsigned long long u = __double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
template <int VAR>
__global__ void __launch_bounds__(256) k_loop(const double *A, double *out, long long *stamps, double eps, double delta) {
    __shared__ double colD[2][64];
    __shared__ double colO[2][64];
    __shared__ double Yt[2][64 * 65];
    const int tid = threadIdx.x, lane = tid & 63, v = tid >> 6;
    const int w = 64;
    double aD[16], aO[16];
#pragma unroll
    for (int c = 0; c < 16; c++) { aD[c] = A[lane + (4 * c + v) * 128]; aO[c] = A[64 + lane + (4 * c + v) * 128]; }
    double *yD = &Yt[0][lane * 65], *yO = &Yt[1][lane * 65];
    long long t0 = clock64();
    long long tb = 0, td = 0, tf = 0;
#pragma unroll
    for (int k = 0; k < 64; k++) {
        if (k < w) {
            const int ck = k >> 2, vk = k & 3, pb = k & 1;
            long long s0 = clock64();
            if (v == vk) { colD[pb][lane] = aD[ck]; colO[pb][lane] = aO[ck]; }
            if (VAR != 2) __syncthreads();
            const double cvD = colD[pb][lane], cvO = colO[pb][lane];
            double d = VAR == 4 ? rl(cvD, k) : colD[pb][k];
            long long s1 = clock64();
            if (d < eps) d = delta;
            const double dinv = VAR == 1 ? d * 0.5 : 1.0 / d;
            const double liD = cvD * dinv;
            const double liO = cvO * dinv;
            if (v == vk) { yD[k] = liD; yO[k] = liO; }
            long long s2 = clock64();
            if (VAR != 3) {
                if (v > vk) { const double cj = VAR == 4 ? rl(cvD, 4 * ck + v) : colD[pb][4 * ck + v]; aD[ck] = fma(-liD, cj, aD[ck]); aO[ck] = fma(-liO, cj, aO[ck]); }
#pragma unroll
                for (int c = ck + 1; c < 16; c++) { const double cj = VAR == 4 ? rl(cvD, 4 * c + v) : colD[pb][4 * c + v]; aD[c] = fma(-liD, cj, aD[c]); aO[c] = fma(-liO, cj, aO[c]); }
            }
            long long s3 = clock64();
            tb += s1 - s0; td += s2 - s1; tf += s3 - s2;
        }
    }
    long long t1 = clock64();
    double acc = 0;
#pragma unroll
    for (int c = 0; c < 16; c++) acc += aD[c] + aO[c];
    out[tid] = acc + Yt[0][tid] + Yt[1][tid];
    if (tid == 0) { stamps[0] = t1 - t0; stamps[1] = tb; stamps[2] = td; stamps[3] = tf; }
}
int main() {
    std::vector<double> h(128 * 64);
    for (int i = 0; i < 128; i++) for (int j = 0; j < 64; j++) h[i + j * 128] = (i == j ? 70.0 : 0.0) + 0.01 * ((i * 31 + j * 17) % 13);
    double *dA, *dout; long long *ds, hs[4];
    (void)0;
    hipMalloc(&dA, h.size() * 8); hipMalloc(&dout, 256 * 8); hipMalloc(&ds, 32);
    hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const char *names[5] = {"full", "no-division", "no-barrier", "no-fma", "readlane-cj"};
    for (int rep = 0; rep < 2; rep++)
    for (int var = 0; var < 5; var++) {
        if (var == 0) hipLaunchKernelGGL(k_loop<0>, dim3(1), dim3(256), 0, 0, dA, dout, ds, 1e-13, 2e-7);
        if (var == 1) hipLaunchKernelGGL(k_loop<1>, dim3(1), dim3(256), 0, 0, dA, dout, ds, 1e-13, 2e-7);
        if (var == 2) hipLaunchKernelGGL(k_loop<2>, dim3(1), dim3(256), 0, 0, dA, dout, ds, 1e-13, 2e-7);
        if (var == 4) hipLaunchKernelGGL(k_loop<4>, dim3(1), dim3(256), 0, 0, dA, dout, ds, 1e-13, 2e-7);
        if (var == 3) hipLaunchKernelGGL(k_loop<3>, dim3(1), dim3(256), 0, 0, dA, dout, ds, 1e-13, 2e-7);
        hipMemcpy(hs, ds, 32, hipMemcpyDeviceToHost);
        if (rep) printf("%-12s total %6lld cycles = %5.0f/step ; barrier+read-d %5.0f  div+li %5.0f  fma %5.0f (per step, wave 0)\n", names[var], hs[0], hs[0] / 64.0, hs[1] / 64.0, hs[2] / 64.0, hs[3] / 64.0);
    }
    return 0;
}
