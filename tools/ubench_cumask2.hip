// Maps hipExtStreamCreateWithCUMask bit patterns to the MFMA throughput they leave (effective CU count) on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
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
int main() {
    double *o1; CK(hipMalloc(&o1, 1 << 26));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Pat { const char *name; int words; std::function<bool(int)> f; };
    std::vector<Pat> pats = {
        {"all256", 8, [](int b) { return true; }},
        {"low128", 8, [](int b) { return b < 128; }},
        {"high128", 8, [](int b) { return b >= 128; }},
        {"low64", 8, [](int b) { return b < 64; }},
        {"low168", 8, [](int b) { return b < 168; }},
        {"low192", 8, [](int b) { return b < 192; }},
        {"low224", 8, [](int b) { return b < 224; }},
        {"high88", 8, [](int b) { return b >= 168; }},
        {"even", 8, [](int b) { return (b & 1) == 0; }},
        {"mod32<21", 8, [](int b) { return (b % 32) < 21; }},
        {"mod8<5", 8, [](int b) { return (b % 8) < 5; }},
        {"mod16<11", 8, [](int b) { return (b % 16) < 11; }},
        {"words4_all", 4, [](int b) { return true; }},
        {"words2_all", 2, [](int b) { return true; }},
        {"words1_all", 1, [](int b) { return true; }},
        {"words1_low21", 1, [](int b) { return b < 21; }},
    };
    float base = 0;
    for (auto &p : pats) {
        std::vector<uint32_t> m(p.words, 0);
        int pc = 0;
        for (int b = 0; b < p.words * 32; b++) if (p.f(b)) { m[b / 32] |= 1u << (b % 32); pc++; }
        hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, p.words, m.data()));
        hipLaunchKernelGGL(k_mfma, dim3(64), dim3(256), 0, s, o1, 10);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_mfma, dim3(4096), dim3(256), 0, s, o1, 100);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (base == 0) base = ms;
        printf("%-14s bits set %3d: 4096 WGs %.1f us -> effective CUs %.0f\n", p.name, pc, ms * 1e3, 256.0 * base / ms);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
