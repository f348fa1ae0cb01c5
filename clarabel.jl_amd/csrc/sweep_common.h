// Hand-off primitives of the persistent front sweeps, shared by kernels.hip (one hop per panel) and front_sweep.hip (super-block
// sweeps).  See the comment block above k_front_fwd in kernels.hip for the protocol.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_plan.h"

namespace hipkkt {

__device__ __forceinline__ int front_ld_flag(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double front_ld(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void front_st(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Hand-off slots: the 64 values of a panel travel as 64 x 16 bytes {value, bits(value) ^ KEY}, written with ONE
// agent-scope 16-byte store per lane and polled with one 16-byte load per lane.  A slot is valid when its tag
// checks against its value, so no separate flag is needed (no "payload complete" wait before the flag, no second
// round trip for the payload after it: 0.7 us per hop instead of 1.3 us in tools/ubench_chain.hip), a torn or
// stale read can only fail the check, and a zeroed slot is invalid.
constexpr unsigned long long kSlotKey = 0x5bd1e995a5a5a5a5ull;
__device__ __forceinline__ FrontSlot front_slot_ld(const FrontSlot *p) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    FrontSlot s;
    s.v = __longlong_as_double((long long)(((unsigned long long)r[1] << 32) | r[0]));
    s.h = ((unsigned long long)r[3] << 32) | r[2];
    return s;
}
__device__ __forceinline__ void front_slot_st(FrontSlot *p, double v) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), h = b ^ kSlotKey;
    v4u r = {(unsigned)b, (unsigned)(b >> 32), (unsigned)h, (unsigned)(h >> 32)};
    // s_nop: the data VGPRs of a store wider than 64 bits must not be overwritten in the next wait states (the
    // compiler's hazard recogniser does not look inside inline asm)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(r) : "memory");
}
// whole-wave wait for the 64 slots of one panel (lane = slot); false = timed out / another workgroup failed
__device__ __forceinline__ bool front_slot_wait(const FrontSlot *p, double &v, int *err, int *failflag, unsigned lim) {
    for (unsigned spins = 0;; spins++) {
        const FrontSlot s = front_slot_ld(p);
        const bool okl = ((unsigned long long)__double_as_longlong(s.v) ^ s.h) == kSlotKey;
        if (__ballot(okl) == ~0ull) { v = s.v; return true; }
        if ((spins & 127u) == 127u || lim < 128u) {
            if (spins > lim) {
                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(failflag, 1);
                return false;
            }
            if (front_ld_flag(err) != 0) return false;
        }
    }
}
__device__ __forceinline__ FrontSlot *front_slots(int *sync_block, int np) {
    return (FrontSlot *)(sync_block + ((2 + np + 31) & ~31));   // header = whole 128-byte lines (symbolic.cpp sync_blk)
}

}  // namespace hipkkt
