// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the KKT kernels use (task 3a):
// every kernel below reads `bytes` once and writes `bytes` once, past the 256 MiB Infinity Cache, so
//     counter / known bytes  =  the factor to apply to that counter for that access shape.
//   k_cal_stream16   16 B per lane, contiguous (the guide's calibrated case: FETCH_SIZE reports 1/2)
//   k_cal_stream8     8 B per lane, contiguous (512 B per wave instruction)
//   k_cal_seg8        8 B per lane, four 128-B row segments of a column-major matrix per wave instruction, column stride 64 KB
//                     (the operand / tile loads and the tile stores of k_update_dense, kernels.hip)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_traffic.hip -o tools/bin/ubench_traffic
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -o p -- tools/bin/ubench_traffic     (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) k_cal_stream16(const double2 *__restrict__ in, double2 *__restrict__ out, int64_t n2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        double2 v = in[i];
        v.x += 1.0;
        out[i] = v;
    }
}
__global__ void __launch_bounds__(256) k_cal_stream8(const double *__restrict__ in, double *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[i] + 1.0;
}
// column-major r x nc matrix; one wave = 16 rows x 64 columns, 4 columns per instruction
__global__ void __launch_bounds__(256) k_cal_seg8(const double *__restrict__ in, double *__restrict__ out, int r, int nc) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lk = lane >> 4;
    const int64_t nwr = r / 16, nw = nwr * (nc / 64);
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += (int64_t)gridDim.x * 4) {
        const int64_t row = 16 * (w % nwr) + l15, c0 = 64 * (w / nwr);
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = in[row + (c0 + 4 * q + lk) * (int64_t)r];
#pragma unroll
        for (int q = 0; q < 16; q++) out[row + (c0 + 4 * q + lk) * (int64_t)r] = v[q] + 1.0;
    }
}

int main() {
    const int r = 8192, nc = 16384;
    const int64_t n = (int64_t)r * nc;            // 2^27 doubles = 1 GiB
    double *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(a, 0, n * 8);
    (void)hipMemset(b, 0, n * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("known bytes per launch: read %lld  write %lld\n", (long long)n * 8, (long long)n * 8);
    for (int rep = 0; rep < 3; rep++) {
        float ms[3];
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_cal_stream16, dim3(8192), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n / 2);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[0], e0, e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_cal_stream8, dim3(8192), dim3(256), 0, 0, a, b, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[1], e0, e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_cal_seg8, dim3(8192), dim3(256), 0, 0, a, b, r, nc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[2], e0, e1);
        printf("rep %d: stream16 %.3f ms (%.2f TB/s r+w)  stream8 %.3f ms (%.2f)  seg8 %.3f ms (%.2f)\n", rep, ms[0],
               2.0 * n * 8 / ms[0] / 1e9, ms[1], 2.0 * n * 8 / ms[1] / 1e9, ms[2], 2.0 * n * 8 / ms[2] / 1e9);
    }
    (void)hipFree(a); (void)hipFree(b);
    return 0;
}
