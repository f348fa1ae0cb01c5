// Hop latency of a chain of workgroups that pass a 64-double payload down the line (the pattern of the front sweeps):
//   A: payload stores, s_waitcnt vmcnt(0), flag store; consumer polls the flag, then loads the payload
//   B: one 16-byte store per lane (value, value-bits ^ KEY); consumer polls the payload itself (self-validating tag)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_chain.hip -o tools/bin/ubench_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;
struct __attribute__((aligned(16))) Slot { double v; u64 h; };
constexpr u64 KEY = 0x5bd1e995a5a5a5a5ull;
__device__ __forceinline__ int ldi(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ldd(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ Slot ld_slot(const Slot *p) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    Slot s;
    s.v = __longlong_as_double((long long)(((u64)r[1] << 32) | r[0]));
    s.h = ((u64)r[3] << 32) | r[2];
    return s;
}
__device__ __forceinline__ void st_slot(Slot *p, double v) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const u64 b = (u64)__double_as_longlong(v), h = b ^ KEY;
    v4u r = {(unsigned)b, (unsigned)(b >> 32), (unsigned)h, (unsigned)(h >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(r) : "memory");
}
template <int MODE>
__global__ void k_chain(int *ticket, int *flags, double *pay, Slot *slots, long long *tstamp, int n) {
    __shared__ int sb;
    __shared__ double buf[64];
    if (threadIdx.x == 0) sb = atomicAdd(ticket, 1);
    __syncthreads();
    const int b = sb, lane = threadIdx.x;
    double x = 1.0;
    if (b > 0) {
        if (MODE == 0) {
            while (ldi(flags + b - 1) == 0) {}
            x = ldd(pay + (size_t)(b - 1) * 64 + lane);
        } else {
            for (;;) {
                const Slot s = ld_slot(slots + (size_t)(b - 1) * 64 + lane);
                const bool okl = ((u64)__double_as_longlong(s.v) ^ s.h) == KEY;
                if (__ballot(okl) == ~0ull) { x = s.v; break; }
            }
        }
    }
    // stand-in for the per-hop work: one LDS exchange + a few FMAs
    buf[lane] = x;
    __syncthreads();
    double a = 0;
    for (int k = 0; k < 16; k++) a = fma(buf[(lane + k) & 63], 1e-3, a);
    a += x + 1.0;
    if (MODE == 0) {
        __hip_atomic_store(pay + (size_t)b * 64 + lane, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        st_slot(slots + (size_t)b * 64 + lane, a);
    }
    if (lane == 0) tstamp[b] = wall_clock64();
}
// C: like the front sweeps -- workgroup b consumes EVERY earlier slot in order (one poll + a little work each) before
//    it publishes its own, so slot q is polled by all later workgroups at once.  NAP > 0: s_sleep(NAP) between polls.
template <int NAP>
__global__ void k_chain_all(int *ticket, Slot *slots, long long *tstamp, int n) {
    __shared__ int sb;
    __shared__ double buf[64];
    if (threadIdx.x == 0) sb = atomicAdd(ticket, 1);
    __syncthreads();
    const int b = sb, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 1.0;
    for (int q = 0; q < b; q++) {
        if (wv == 0) {
            double x;
            for (;;) {
                const Slot s = ld_slot(slots + (size_t)q * 64 + lane);
                const bool okl = ((u64)__double_as_longlong(s.v) ^ s.h) == KEY;
                if (__ballot(okl) == ~0ull) { x = s.v; break; }
                if (NAP > 0) __builtin_amdgcn_s_sleep(NAP);
            }
            buf[lane] = x;
        }
        __syncthreads();
        for (int k = 0; k < 16; k++) acc = fma(buf[(lane + k + wv) & 63], 1e-3, acc);
        __syncthreads();
    }
    if (wv == 0) {
        st_slot(slots + (size_t)b * 64 + lane, acc * 1e-3 + 1.0);
        if (lane == 0) tstamp[b] = wall_clock64();
    }
}
// NOTE on D: issue and wait are separate asm statements here, which is NOT safe in product code (the compiler may copy
// an output register between them, i.e. before the data has landed: DESIGN.md section 9); a mis-read can only make a
// poll round look invalid and repeat it, so the timing conclusion (more polls in flight are slower) stands.
// D: as C, but the polling wave keeps FOUR polls in flight (a new one every s_sleep(GAP)) instead of one round trip
//    at a time: the detection delay after the value becomes visible drops from up to a full round trip to ~GAP.
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void issue_slot(v4u_t &r, const Slot *p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
}
__device__ __forceinline__ bool slot_ok(const v4u_t &r, double &v) {
    const u64 b = ((u64)r[1] << 32) | r[0], h = ((u64)r[3] << 32) | r[2];
    v = __longlong_as_double((long long)b);
    return __ballot((b ^ h) == KEY) == ~0ull;
}
template <int GAP>
__global__ void k_chain_pipe(int *ticket, Slot *slots, long long *tstamp, int n) {
    __shared__ int sb;
    __shared__ double buf[64];
    if (threadIdx.x == 0) sb = atomicAdd(ticket, 1);
    __syncthreads();
    const int b = sb, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 1.0;
    for (int q = 0; q < b; q++) {
        if (wv == 0) {
            const Slot *p = slots + (size_t)q * 64 + lane;
            v4u_t r0, r1, r2, r3;
            double x = 0;
            issue_slot(r0, p); __builtin_amdgcn_s_sleep(GAP);
            issue_slot(r1, p); __builtin_amdgcn_s_sleep(GAP);
            issue_slot(r2, p); __builtin_amdgcn_s_sleep(GAP);
            issue_slot(r3, p);
            for (;;) {
                asm volatile("s_waitcnt vmcnt(3)" : "+v"(r0) : : "memory");
                if (slot_ok(r0, x)) break;
                issue_slot(r0, p);
                asm volatile("s_waitcnt vmcnt(3)" : "+v"(r1) : : "memory");
                if (slot_ok(r1, x)) break;
                issue_slot(r1, p);
                asm volatile("s_waitcnt vmcnt(3)" : "+v"(r2) : : "memory");
                if (slot_ok(r2, x)) break;
                issue_slot(r2, p);
                asm volatile("s_waitcnt vmcnt(3)" : "+v"(r3) : : "memory");
                if (slot_ok(r3, x)) break;
                issue_slot(r3, p);
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
            buf[lane] = x;
        }
        __syncthreads();
        for (int k = 0; k < 16; k++) acc = fma(buf[(lane + k + wv) & 63], 1e-3, acc);
        __syncthreads();
    }
    if (wv == 0) {
        st_slot(slots + (size_t)b * 64 + lane, acc * 1e-3 + 1.0);
        if (lane == 0) tstamp[b] = wall_clock64();
    }
}
template <int GAP>
int run_pipe(const char *name) {
    const int n = 96;
    int *ticket; Slot *slots; long long *ts;
    CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&slots, (size_t)n * 64 * 16)); CK(hipMalloc(&ts, n * 8));
    long long h[96];
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(ticket, 0, 4)); CK(hipMemset(slots, 0, (size_t)n * 64 * 16));
        hipLaunchKernelGGL(k_chain_pipe<GAP>, dim3(n), dim3(256), 0, 0, ticket, slots, ts, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, ts, n * 8, hipMemcpyDeviceToHost));
        printf("%s: %.0f ns per hop (hops 8..%d)\n", name, (h[n - 1] - h[8]) * 10.0 / (n - 9), n - 1);
    }
    return 0;
}

template <int NAP>
int run_all(const char *name) {
    const int n = 96;
    int *ticket; Slot *slots; long long *ts;
    CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&slots, (size_t)n * 64 * 16)); CK(hipMalloc(&ts, n * 8));
    long long h[96];
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(ticket, 0, 4)); CK(hipMemset(slots, 0, (size_t)n * 64 * 16));
        hipLaunchKernelGGL(k_chain_all<NAP>, dim3(n), dim3(256), 0, 0, ticket, slots, ts, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, ts, n * 8, hipMemcpyDeviceToHost));
        printf("%s: %d workgroups, every one consumes all earlier slots: %.0f ns per hop (hops 8..%d)\n", name, n,
               (h[n - 1] - h[8]) * 10.0 / (n - 9), n - 1);
    }
    return 0;
}

template <int MODE>
int run(const char *name) {
    const int n = 2048;
    int *ticket, *flags; double *pay; Slot *slots; long long *ts;
    CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&flags, n * 4)); CK(hipMalloc(&pay, (size_t)n * 64 * 8));
    CK(hipMalloc(&slots, (size_t)n * 64 * 16)); CK(hipMalloc(&ts, n * 8));
    static long long h[2048];
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(ticket, 0, 4)); CK(hipMemset(flags, 0, n * 4)); CK(hipMemset(slots, 0, (size_t)n * 64 * 16));
        hipLaunchKernelGGL(k_chain<MODE>, dim3(n), dim3(64), 0, 0, ticket, flags, pay, slots, ts, n);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, ts, n * 8, hipMemcpyDeviceToHost));
        printf("%s: %d hops, %.0f ns per hop (hops 100..%d)\n", name, n, (h[n - 1] - h[100]) * 10.0 / (n - 101), n - 1);
    }
    return 0;
}
int main() {
    if (run<0>("A flag + payload")) return 1;
    if (run<1>("B tagged 16-byte")) return 1;
    if (run_all<0>("C all-consume, eager polls")) return 1;
    if (run_pipe<1>("D all-consume, 4 polls in flight, gap s_sleep(1)")) return 1;
    if (run_pipe<3>("D all-consume, 4 polls in flight, gap s_sleep(3)")) return 1;
    if (run_pipe<6>("D all-consume, 4 polls in flight, gap s_sleep(6)")) return 1;
    return 0;
}
