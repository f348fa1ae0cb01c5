// Box probe (diagnostic, hipkkt_box_probe): what the latency-bound kernels of the factorisation depend on and the matrix-core /
// memory-bound ones do not -- the shader clock a single busy wavefront really gets, and the round trip of a flag between two
// workgroups on the same XCD and on different XCDs (relaxed agent-scope atomics, what the hand-offs of front_block.hip /
// front_sweep.hip are made of).  bench.py puts the result into its JSON line so that a slow line names its cause.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hipkkt {

__device__ __forceinline__ int pb_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pb_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int pb_xcc() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

// one wavefront, a dependent FP64 fma chain: shader cycles (s_memtime) against the 100 MHz constant clock
__global__ void k_probe_clock(long long *out, double *sink, int n) {
    double x = 1.0 + threadIdx.x * 1e-9;
    const double y = 1.0000001;
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int q = 0; q < 16; q++) x = fma(x, y, 1e-9);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    sink[threadIdx.x] = x;
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

// ctl: [0] arrivals, [1] partner on the same XCD, [2] partner on another XCD, [3] go, [8 .. 8 + n) XCC id + 1 of every workgroup;
// ping / pong words on lines of their own.  Every workgroup is one wavefront: all 256 are resident at once.
__global__ void k_probe_handoff(int *ctl, long long *out, int rounds) {
    const int b = blockIdx.x, n = gridDim.x;
    const int xid = pb_xcc();
    __shared__ int role;
    if (threadIdx.x == 0) {
        pb_st(ctl + 8 + b, xid + 1);
        atomicAdd(ctl, 1);
        role = 0;
        if (b == 0) {
            unsigned spins = 0;
            while (pb_ld(ctl) < n && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
            int same = -1, other = -1, nx = 0;
            unsigned seen = 0;
            for (int q = 0; q < n; q++) {
                const int x = pb_ld(ctl + 8 + q) - 1;
                if (x < 0) continue;
                if (!(seen & (1u << x))) { seen |= 1u << x; nx++; }
                if (q == 0) continue;
                if (x == xid && same < 0) same = q;
                if (x != xid && other < 0) other = q;
            }
            pb_st(ctl + 1, same); pb_st(ctl + 2, other); pb_st(ctl + 4, nx);
            pb_st(ctl + 3, 1);
            role = 1;
        } else {
            unsigned spins = 0;
            while (pb_ld(ctl + 3) == 0 && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
            if (pb_ld(ctl + 1) == b) role = 2;
            if (pb_ld(ctl + 2) == b) role = 3;
        }
    }
    __syncthreads();
    if (role == 0 || threadIdx.x != 0) return;
    int *ping = ctl + 512, *pong = ctl + 640;
    const unsigned lim = 1u << 20;                  // bounded: a lost partner ends the probe instead of hanging the device
    if (role == 1) {
        for (int phase = 0; phase < 2; phase++) {
            out[phase] = -1;
            if (pb_ld(ctl + 1 + phase) < 0) continue;
            const int base = phase * (rounds + 10);
            const long long t0 = wall_clock64();
            bool ok = true;
            for (int r = 1; r <= rounds && ok; r++) {
                pb_st(ping, base + r);
                unsigned s = 0;
                while (pb_ld(pong) != base + r) if (++s > lim) { ok = false; break; }
            }
            if (ok) out[phase] = wall_clock64() - t0;
        }
        out[2] = pb_ld(ctl + 4);
    } else {
        const int phase = role - 2;
        const int base = phase * (rounds + 10);
        for (int r = 1; r <= rounds; r++) {
            unsigned s = 0;
            bool ok = true;
            while (pb_ld(ping) != base + r) if (++s > lim) { ok = false; break; }
            if (!ok) return;
            pb_st(pong, base + r);
        }
    }
}

// out: [0] effective shader GHz of one busy wavefront, [1] flag round trip same XCD (ns), [2] other XCD (ns), [3] XCDs seen,
//      [4] hipDeviceProp clockRate (MHz), [5] memoryClockRate (MHz), [6] compute units, [7] constant-clock ticks per microsecond assumed (100)
int box_probe(int device, double *out) {
    // the caller's current device is restored, the work runs on a private stream (no device-wide synchronisation: other
    // streams of the process keep running), the buffers are released on every path
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return 1;
    int *ctl = nullptr; long long *o = nullptr; double *sink = nullptr;
    hipStream_t st = nullptr;
    int rc = 1;
    hipDeviceProp_t pr;
    double ghz = 0.0, same = -1.0, other = -1.0, nx = 0.0;
    long long h[4] = {0, 0, 0, 0};
    do {
        if (hipGetDeviceProperties(&pr, device) != hipSuccess) break;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; break; }
        if (hipMalloc(&ctl, 4096 * sizeof(int)) != hipSuccess || hipMalloc(&o, 64) != hipSuccess || hipMalloc(&sink, 64 * sizeof(double)) != hipSuccess) break;
        bool ok = true;
        for (int rep = 0; rep < 3 && ok; rep++) {        // the last repetition counts (clocks ramp up)
            hipLaunchKernelGGL(k_probe_clock, dim3(1), dim3(64), 0, st, o, sink, 20000);
            ok = hipMemcpyAsync(h, o, 16, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            if (ok && h[1] > 0) ghz = (double)h[0] / ((double)h[1] * 10.0);
        }
        const int rounds = 1000;
        for (int rep = 0; rep < 2 && ok; rep++) {
            ok = hipMemsetAsync(ctl, 0, 4096 * sizeof(int), st) == hipSuccess;
            hipLaunchKernelGGL(k_probe_handoff, dim3(256), dim3(64), 0, st, ctl, o, rounds);
            ok = ok && hipMemcpyAsync(h, o, 24, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
            if (!ok) break;
            same = h[0] > 0 ? h[0] * 10.0 / rounds : -1.0;
            other = h[1] > 0 ? h[1] * 10.0 / rounds : -1.0;
            nx = (double)h[2];
        }
        if (!ok) break;
        out[0] = ghz; out[1] = same; out[2] = other; out[3] = nx;
        out[4] = pr.clockRate / 1000.0; out[5] = pr.memoryClockRate / 1000.0; out[6] = pr.multiProcessorCount; out[7] = 100.0;
        rc = 0;
    } while (0);
    if (ctl) (void)hipFree(ctl);
    if (o) (void)hipFree(o);
    if (sink) (void)hipFree(sink);
    if (st) (void)hipStreamDestroy(st);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return rc;
}

}  // namespace hipkkt
