// Does hipExtStreamCreateWithCUMask partition the MI355X's CUs between two streams, and how does a latency-bound
// kernel fare next to an MFMA-bound one with and without the partition?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_cumask.hip -o tools/bin/ubench_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(256, 2) k_mfma(double *out, int iters) {
    v4f64 a[16];
    for (int i = 0; i < 16; i++) a[i] = {0, 0, 0, 0};
    double x = 1.0 + threadIdx.x * 1e-9, y = 0.5;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) a[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[j], 0, 0, 0);
    double s = 0;
    for (int j = 0; j < 16; j++) s += a[j][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// latency-bound: dependent FMA chain + LDS + barrier, like a pivot loop
__global__ void __launch_bounds__(512) k_chain(double *out, int iters) {
    __shared__ double s[512];
    double a = threadIdx.x;
    s[threadIdx.x] = a;
    for (int i = 0; i < iters; i++) {
        __syncthreads();
        double b = s[(threadIdx.x + i) & 511];
        for (int j = 0; j < 8; j++) a = fma(a, 1.0000001, b);
        s[threadIdx.x] = a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
int main() {
    int ncu = 0; CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d\n", ncu);
    double *o1, *o2; CK(hipMalloc(&o1, 1 << 24)); CK(hipMalloc(&o2, 1 << 24));
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    for (int mode = 0; mode < 3; mode++) {
        hipStream_t sa, sb;
        if (mode == 0) { CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb)); }
        else {
            // mode 1: bits interleaved over XCDs assumed (first 168 bits / last 88 bits); mode 2: per-XCD contiguous (21 of each 32)
            std::vector<uint32_t> ma(8, 0), mb(8, 0);
            for (int b = 0; b < 256; b++) {
                bool far = mode == 1 ? (b < 168) : ((b % 32) < 21);
                (far ? ma : mb)[b / 32] |= 1u << (b % 32);
            }
            CK(hipExtStreamCreateWithCUMask(&sa, 8, ma.data()));
            CK(hipExtStreamCreateWithCUMask(&sb, 8, mb.data()));
        }
        float ms;
        for (int wgs : {168, 256, 512}) {
            hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, sa, o1, 10);
            CK(hipStreamSynchronize(sa));
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, sa, o1, 200);
            CK(hipEventRecord(e1, sa)); CK(hipStreamSynchronize(sa)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("mode %d: mfma alone on A, %d WGs: %.1f us\n", mode, wgs, ms * 1e3);
        }
        hipLaunchKernelGGL(k_chain, dim3(87), dim3(512), 0, sb, o2, 10);
        CK(hipStreamSynchronize(sb));
        CK(hipEventRecord(f0, sb));
        hipLaunchKernelGGL(k_chain, dim3(87), dim3(512), 0, sb, o2, 300);
        CK(hipEventRecord(f1, sb)); CK(hipStreamSynchronize(sb)); CK(hipEventElapsedTime(&ms, f0, f1));
        printf("mode %d: chain alone on B (87 WGs): %.1f us\n", mode, ms * 1e3);
        for (int wgs : {168, 256}) {
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, sa, o1, 400);
            CK(hipEventRecord(e1, sa));
            CK(hipEventRecord(f0, sb));
            for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k_chain, dim3(87), dim3(512), 0, sb, o2, 300);
            CK(hipEventRecord(f1, sb));
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            float ma_, mb_; CK(hipEventElapsedTime(&ma_, e0, e1)); CK(hipEventElapsedTime(&mb_, f0, f1));
            printf("mode %d: concurrent: mfma(%d WGs, 2x iters) %.1f us, 4 x chain %.1f us (%.1f each)\n", mode, wgs, ma_ * 1e3, mb_ * 1e3, mb_ * 250);
        }
    }
    return 0;
}
