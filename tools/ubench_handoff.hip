// Flag hand-off latency between two workgroups on MI355X: same XCD vs different XCDs, agent-scope relaxed
// atomics (what the front sweeps use).  One round trip = A stores seq, B sees it and stores seq back, A sees it.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_handoff.hip -o tools/bin/ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ int ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}
// ctl: [0] ready count, [1] partner same xcd, [2] partner other xcd, [3] go, [8..8+n) xcc table; ping/pong words at [512], [640] (separate lines)
__global__ void k_handoff(int *ctl, long long *out, int rounds) {
    const int b = blockIdx.x, n = gridDim.x;
    const int xid = xcc_id();
    if (threadIdx.x == 0) {
        st(ctl + 8 + b, xid + 1);
        atomicAdd(ctl, 1);
    }
    __shared__ int role;
    if (threadIdx.x == 0) {
        role = 0;
        if (b == 0) {
            while (ld(ctl) < n) __builtin_amdgcn_s_sleep(2);
            int same = -1, other = -1;
            for (int q = 1; q < n; q++) {
                const int x = ld(ctl + 8 + q) - 1;
                if (x == xid && same < 0) same = q;
                if (x != xid && other < 0) other = q;
            }
            st(ctl + 1, same); st(ctl + 2, other);
            st(ctl + 3, 1);
            role = 1;
        } else {
            while (ld(ctl + 3) == 0) __builtin_amdgcn_s_sleep(2);
            if (ld(ctl + 1) == b) role = 2;
            if (ld(ctl + 2) == b) role = 3;
        }
    }
    __syncthreads();
    if (role == 0 || threadIdx.x != 0) return;
    int *ping = ctl + 512, *pong = ctl + 640;
    if (role == 1) {
        for (int phase = 0; phase < 2; phase++) {
            const int base = phase * (rounds + 10);
            const long long t0 = wall_clock64();
            for (int r = 1; r <= rounds; r++) {
                st(ping, base + r);
                while (ld(pong) != base + r) {}
            }
            const long long t1 = wall_clock64();
            out[phase] = t1 - t0;
        }
        out[2] = ld(ctl + 1); out[3] = ld(ctl + 2);
    } else {
        const int phase = role - 2;
        const int base = phase * (rounds + 10);
        for (int r = 1; r <= rounds; r++) {
            while (ld(ping) != base + r) {}
            st(pong, base + r);
        }
    }
}
int main() {
    int *ctl; long long *out, h[4];
    CK(hipMalloc(&ctl, 4096 * 4)); CK(hipMalloc(&out, 64));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(ctl, 0, 4096 * 4));
        hipLaunchKernelGGL(k_handoff, dim3(256), dim3(64), 0, 0, ctl, out, 2000);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
        // wall_clock64 ticks at 100 MHz
        printf("round trip: same XCD (WG %lld) %.0f ns, other XCD (WG %lld) %.0f ns\n", h[2], h[0] * 10.0 / 2000, h[3], h[1] * 10.0 / 2000);
    }
    return 0;
}
