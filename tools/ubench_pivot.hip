// Microbenchmark of the pivot chain of k_front_block's wave 0 (clarabel.jl_amd/csrc/front_block.hip): one wavefront, 8 blocks of 8
// pivots of a 64 x 64 tile held lane = row, no barriers / matrix-core work around it -- what does the dependent chain cost?
//   lat    dependent-issue latency of v_fma_f64, v_mul_f64, v_rcp_f64, v_readlane pairs (cycles per op)
//   cur    the round-3 recurrence: readlane d -> pivot rule -> rcp + 2 Newton steps -> l = a * dinv -> fma into the next column
//   fast   the same elimination with the pivot-to-pivot chain shortened to  d_k = fma(-c^2, dinv_{k-1}, a)  -> rcp -> 3 fma:
//          the sub-diagonal entry c and the diagonal entry a of column k are fetched one pivot EARLIER (they do not depend on
//          dinv_{k-1} before the last update), the pivot rule is evaluated next to the reciprocal
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_pivot.hip -o tools/bin/ubench_pivot
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double rl(double x, int l) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double rcp2(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
// 3 dependent operations after the hardware reciprocal (2^-27 or better): r (1 + e + e^2), e = 1 - d r
__device__ __forceinline__ double rcp3(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    const double e2 = fma(e, e, e);
    return fma(r, e2, r);
}

__global__ void k_lat(double *out, long long *st, double seed) {
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 256; i++) x = fma(x, y, 1e-9);
    long long t1 = clock64();
#pragma unroll
    for (int i = 0; i < 256; i++) x = x * y;
    long long t2 = clock64();
#pragma unroll
    for (int i = 0; i < 64; i++) x = __builtin_amdgcn_rcp(x) + 0.0 * y;
    long long t3 = clock64();
#pragma unroll
    for (int i = 0; i < 64; i++) x = rl(x, i & 63) * y;
    long long t4 = clock64();
    // throughput: 8 independent fma chains
    double z[8];
#pragma unroll
    for (int q = 0; q < 8; q++) z[q] = x + q;
    long long t5 = clock64();
#pragma unroll
    for (int i = 0; i < 64; i++)
#pragma unroll
        for (int q = 0; q < 8; q++) z[q] = fma(z[q], y, 1e-9);
    long long t6 = clock64();
    for (int q = 0; q < 8; q++) x += z[q];
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { st[0] = t1 - t0; st[1] = t2 - t1; st[2] = t3 - t2; st[3] = t4 - t3; st[4] = t6 - t5; }
}

template <int VAR>
__global__ void __launch_bounds__(64) k_piv(const double *A, double *Lout, double *Dout, long long *st, double eps, double delta, int cold) {
    const int lane = threadIdx.x;
    // cold: drop the instruction cache first -- the real kernel runs its fully unrolled pivot code ONCE per workgroup, i.e. every
    // instruction of it comes from L2 (round 5: is that what 290 cycles per pivot in the kernel against 101 here are made of?)
    if (cold) asm volatile("s_icache_inv\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    double a[64];
#pragma unroll
    for (int c = 0; c < 64; c++) a[c] = A[lane + 64 * c];
    int nreg = 0;
    long long tp = 0;
#pragma unroll
    for (int Bk = 0; Bk < 8; Bk++) {
        const long long t0 = clock64();
        double pcol[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pcol[q] = a[8 * Bk + q];
        double lcol[8], dk[8];
        if (VAR == 0) {
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int k = 8 * Bk + kk;
                const double reg = pcol[kk];
                double d = rl(reg, k);
                if (d < eps) { d = delta; nreg++; }
                const double dinv = rcp2(d);
                const double li = reg * dinv;
                lcol[kk] = li;
                dk[kk] = d;
#pragma unroll
                for (int jj = kk + 1; jj < 8; jj++) pcol[jj] = fma(-li, rl(reg, 8 * Bk + jj), pcol[jj]);
            }
        } else {
            // chain values of pivot kk: d = fma(-csq, dinv_prev, akk); for kk = 0: d = akk (csq = 0)
            double akk = rl(pcol[0], 8 * Bk), csq = 0.0, dinv_prev = 0.0;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                const int k = 8 * Bk + kk;
                double d = fma(-csq, dinv_prev, akk);
                const bool bad = d < eps;                                    // next to the reciprocal, not in front of it
                double dinv = rcp3(d);
                if (bad) { d = delta; dinv = 1.0 / delta; nreg++; }
                dk[kk] = d;
                // fetch the next pivot's a and c BEFORE this pivot's update of them is applied on the vector side: what they need from
                // this pivot is applied on the chain (the fma above) -- only valid for the sub-diagonal neighbour, so fetch AFTER the
                // vector update of pivot kk - 1 (done below in the previous iteration) and BEFORE that of pivot kk
                if (kk < 7) {
                    akk = rl(pcol[kk + 1], k + 1);
                    const double cn = rl(pcol[kk], k + 1);
                    csq = cn * cn;
                }
                dinv_prev = dinv;
                const double reg = pcol[kk];
                const double li = reg * dinv;
                lcol[kk] = li;
#pragma unroll
                for (int jj = kk + 1; jj < 8; jj++) pcol[jj] = fma(-li, rl(reg, 8 * Bk + jj), pcol[jj]);
            }
        }
        tp += clock64() - t0;
        // rank-8 update of the columns right of the block (VALU here; the kernel does it on the matrix core after a barrier)
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            a[8 * Bk + kk] = lcol[kk];
            if (lane == 8 * Bk + kk) Dout[8 * Bk + kk] = dk[kk];
#pragma unroll
            for (int c = 8 * Bk + 8; c < 64; c++) a[c] = fma(-lcol[kk], rl(lcol[kk], c) * dk[kk], a[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < 64; c++) Lout[lane + 64 * c] = a[c];
    if (lane == 0) { st[0] = tp; st[1] = nreg; }
}

// the same chain ("fast" arithmetic) as a ROLLED loop over the 8 blocks with the tile in LDS (lane = row, stride 65): 1/8 of the code,
// seven of eight iterations run from a warm instruction cache whatever the state at entry
__global__ void __launch_bounds__(64) k_piv_rolled(const double *A, double *Lout, double *Dout, long long *st, double eps, double delta, int cold) {
    __shared__ double T[64 * 65];
    const int lane = threadIdx.x;
    for (int c = 0; c < 64; c++) T[lane * 65 + c] = A[lane + 64 * c];
    __syncthreads();
    if (cold) asm volatile("s_icache_inv\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    int nreg = 0;
    long long tp = 0;
#pragma unroll 1
    for (int Bk = 0; Bk < 8; Bk++) {
        const long long t0 = clock64();
        double pcol[8], lcol[8], dk[8];
#pragma unroll
        for (int q = 0; q < 8; q++) pcol[q] = T[lane * 65 + 8 * Bk + q];
        double akk = rl(pcol[0], 8 * Bk), csq = 0.0, dinv_prev = 0.0;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const int k = 8 * Bk + kk;
            double d = fma(-csq, dinv_prev, akk);
            const bool bad = d < eps;
            double dinv = rcp3(d);
            if (bad) { d = delta; dinv = 1.0 / delta; nreg++; }
            dk[kk] = d;
            if (kk < 7) {
                akk = rl(pcol[kk + 1], k + 1);
                const double cn = rl(pcol[kk], k + 1);
                csq = cn * cn;
            }
            dinv_prev = dinv;
            const double reg = pcol[kk];
            const double li = reg * dinv;
            lcol[kk] = li;
#pragma unroll
            for (int jj = kk + 1; jj < 8; jj++) pcol[jj] = fma(-li, rl(reg, 8 * Bk + jj), pcol[jj]);
        }
        tp += clock64() - t0;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            T[lane * 65 + 8 * Bk + kk] = lcol[kk];
            if (lane == 8 * Bk + kk) Dout[8 * Bk + kk] = dk[kk];
        }
        for (int c = 8 * Bk + 8; c < 64; c++) {
            double a = T[lane * 65 + c];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) a = fma(-lcol[kk], rl(lcol[kk], c) * dk[kk], a);
            T[lane * 65 + c] = a;
        }
    }
    for (int c = 0; c < 64; c++) Lout[lane + 64 * c] = T[lane * 65 + c];
    if (lane == 0) { st[0] = tp; st[1] = nreg; }
}

// accuracy of the reciprocal refinements against the correctly rounded quotient 1.0 / d: the raw v_rcp_f64, one Newton step
// r (1 + e), the three-term form r (1 + e + e^2) used on the pivot chain, two Newton steps
__global__ void k_rcpacc(double *out, unsigned long long seed) {
    unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    double m0 = 0, m1 = 0, m3 = 0, m2 = 0;
    for (int it = 0; it < 4096; it++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const double mant = 1.0 + (double)(x >> 12) * (1.0 / 4503599627370496.0);
        const int ex = (int)((x >> 3) & 127) - 64;
        const double d = ldexp(mant, ex) * ((x & 1) ? -1.0 : 1.0);
        const double q = 1.0 / d;
        const double r = __builtin_amdgcn_rcp(d);
        const double e = fma(-d, r, 1.0);
        const double r1 = fma(r, e, r);
        const double r3 = fma(r, fma(e, e, e), r);
        double r2 = fma(fma(-d, r, 1.0), r, r); r2 = fma(fma(-d, r2, 1.0), r2, r2);
        const double u = fabs(q) * 1.1102230246251565e-16;   // half an ulp-ish unit: 2^-53 |q|
        m0 = fmax(m0, fabs(r - q) / u); m1 = fmax(m1, fabs(r1 - q) / u); m3 = fmax(m3, fabs(r3 - q) / u); m2 = fmax(m2, fabs(r2 - q) / u);
    }
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    out[4 * t] = m0; out[4 * t + 1] = m1; out[4 * t + 2] = m3; out[4 * t + 3] = m2;
}

int main() {
    std::vector<double> h(64 * 64);
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) h[i + j * 64] = (i == j ? 70.0 : 0.0) + 0.01 * (((i + 1) * 31 + (j + 1) * 17 + (i ^ j)) % 13);
    for (int i = 0; i < 64; i++) for (int j = 0; j < i; j++) h[j + i * 64] = h[i + j * 64];
    double *dA, *dL, *dD, *dout; long long *ds, hs[8];
    hipMalloc(&dA, h.size() * 8); hipMalloc(&dL, h.size() * 8); hipMalloc(&dD, 64 * 8); hipMalloc(&dout, 64 * 8); hipMalloc(&ds, 64);
    hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, dout, ds, 1.5);
        hipMemcpy(hs, ds, 64, hipMemcpyDeviceToHost);
        if (rep) printf("latency (cycles per dependent op): fma %.1f  mul %.1f  rcp+add %.1f  readlane+mul %.1f ; 8 independent fma chains: %.1f cycles per fma\n", hs[0] / 256.0, hs[1] / 256.0,
                        hs[2] / 64.0, hs[3] / 64.0, hs[4] / 512.0);
    }
    std::vector<double> L0(64 * 64), L1(64 * 64), D0(64), D1(64);
    for (int rep = 0; rep < 2; rep++)
        for (int var = 0; var < 2; var++) {
            if (var == 0) hipLaunchKernelGGL(k_piv<0>, dim3(1), dim3(64), 0, 0, dA, dL, dD, ds, 1e-13, 2e-7, 0);
            else hipLaunchKernelGGL(k_piv<1>, dim3(1), dim3(64), 0, 0, dA, dL, dD, ds, 1e-13, 2e-7, 0);
            hipMemcpy(hs, ds, 64, hipMemcpyDeviceToHost);
            hipMemcpy(var ? L1.data() : L0.data(), dL, 64 * 64 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(var ? D1.data() : D0.data(), dD, 64 * 8, hipMemcpyDeviceToHost);
            if (rep) printf("%s: %lld cycles for 64 pivots (8 blocks, without the rank-8 updates) = %.0f per pivot, nreg %lld\n", var ? "fast" : "cur ", hs[0], hs[0] / 64.0, hs[1]);
        }
    // instruction cache dropped at entry (s_icache_inv): unrolled against rolled
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_piv<1>, dim3(1), dim3(64), 0, 0, dA, dL, dD, ds, 1e-13, 2e-7, 1);
        hipMemcpy(hs, ds, 64, hipMemcpyDeviceToHost);
        printf("fast, unrolled, instruction cache dropped at entry: %lld cycles = %.0f per pivot\n", hs[0], hs[0] / 64.0);
    }
    for (int cold = 0; cold < 2; cold++)
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_piv_rolled, dim3(1), dim3(64), 0, 0, dA, dL, dD, ds, 1e-13, 2e-7, cold);
            hipMemcpy(hs, ds, 64, hipMemcpyDeviceToHost);
            printf("fast, ROLLED over the blocks (tile in LDS), %s: %lld cycles = %.0f per pivot\n", cold ? "cache dropped at entry" : "warm", hs[0], hs[0] / 64.0);
        }
    {
        std::vector<double> L2(64 * 64), D2(64);
        hipMemcpy(L2.data(), dL, 64 * 64 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(D2.data(), dD, 64 * 8, hipMemcpyDeviceToHost);
        double e = 0;
        for (int i = 0; i < 64; i++) e = std::fmax(e, std::fabs(D2[i] - D0[i]) / std::fabs(D0[i]));
        printf("rolled vs cur: max rel diff D %.2e\n", e);
    }
    {
        double *da; const int nt = 64 * 256;
        hipMalloc(&da, nt * 4 * 8);
        hipLaunchKernelGGL(k_rcpacc, dim3(64), dim3(256), 0, 0, da, 12345ull);
        std::vector<double> ha(nt * 4);
        hipMemcpy(ha.data(), da, nt * 4 * 8, hipMemcpyDeviceToHost);
        double m[4] = {0, 0, 0, 0};
        for (int t = 0; t < nt; t++) for (int q = 0; q < 4; q++) m[q] = std::fmax(m[q], ha[4 * t + q]);
        printf("reciprocal error in units of 2^-53 |1/d| over 6.7e7 values: raw v_rcp_f64 %.3g, one Newton step %.3g, r(1+e+e^2) %.3g, two Newton steps %.3g\n", m[0], m[1], m[2], m[3]);
    }
    double eL = 0, eD = 0;
    for (int i = 0; i < 64; i++) for (int j = 0; j < i; j++) eL = std::fmax(eL, std::fabs(L1[i + 64 * j] - L0[i + 64 * j]) / std::fmax(1e-300, std::fabs(L0[i + 64 * j])));
    for (int i = 0; i < 64; i++) eD = std::fmax(eD, std::fabs(D1[i] - D0[i]) / std::fabs(D0[i]));
    printf("fast vs cur: max rel diff L %.2e  D %.2e\n", eL, eD);
    return 0;
}
