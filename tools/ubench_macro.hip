// Microbenchmark for the Schur-update tile kernels on front-like data (K = ntask x 64 columns of R-row panels):
//   base   one wavefront per 64 x 64 tile, operands straight from the panels (dense_tile.h dense_tile_core_full), 2 wavefronts per SIMD
//   macro  one workgroup per 2 x 2 tiles: the four operand row ranges of a k-step are fetched ONCE per workgroup (4 values per thread,
//          P k-steps ahead), laid down in LDS and read back by the four wavefronts (each still owns one tile with 16 accumulators:
//          the same products in the same order as `base`)
// Prints microseconds and TFLOP/s for T tiles, and the largest difference between the two results (must be 0).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I clarabel.jl_amd/csrc -I tools tools/ubench_macro.hip -o tools/bin/ubench_macro
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "experiments/dense_macro.h"
using namespace hipkkt;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256, 2) k_base(const double *Lx, const double *D, const MacroGroup *mg, const MacroTask *mt, int nmacro) {
    const int lane = threadIdx.x & 63, wave = rfl(threadIdx.x >> 6);
    const int m = blockIdx.x;
    if (m >= nmacro) return;
    const MacroGroup G = mg[m];
    const int wi = wave >> 1, wj = wave & 1, l15 = lane & 15, lk = lane >> 4;
    double *tp = rfl_ptr(const_cast<double *>(Lx) + G.tile_off[wave]);
    const int rt = rfl(G.rt[wj]);
    v4f64 acc[4][4];
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) acc[tj][ti][reg] = ld_off(tp, (unsigned)(ti * 16 + l15 + (tj * 16 + lk + 4 * reg) * rt) * 8u);
    for (int q = rfl(G.task_begin); q < rfl(G.task_end); q++) {
        const MacroTask T = mt[q];
        const double *sp = rfl_ptr(Lx + T.panel_off), *dv = rfl_ptr(D + T.dfirst);
        const unsigned r8 = (unsigned)rfl(T.r8);
        const int K = rfl(T.K), ro = rfl(T.row_off[wi]), co = rfl(T.row_off[2 + wj]);
        double fa[2][4], fb[2][4], fd[2];
        auto load = [&](int s, int k0) {
            const unsigned ko = (unsigned)(k0 + lk) * r8;
            fd[s] = ld_off(dv, (unsigned)(k0 + lk) * 8u);
#pragma unroll
            for (int t = 0; t < 4; t++) { fa[s][t] = ld_off(sp, (unsigned)(co + t * 16 + l15) * 8u + ko); fb[s][t] = ld_off(sp, (unsigned)(ro + t * 16 + l15) * 8u + ko); }
        };
        auto mma = [&](int s) {
            const double nd = -fd[s];
            double a[4];
#pragma unroll
            for (int t = 0; t < 4; t++) a[t] = fa[s][t] * nd;
#pragma unroll
            for (int tj = 0; tj < 4; tj++)
#pragma unroll
                for (int ti = 0; ti < 4; ti++) acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tj], fb[s][ti], acc[tj][ti], 0, 0, 0);
        };
        load(0, 0);
        for (int k0 = 0; k0 < K; k0 += 8) {
            load(1, k0 + 4);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            __builtin_amdgcn_sched_barrier(0);
            if (k0 + 8 < K) load(0, k0 + 8);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) st_off(tp, (unsigned)(ti * 16 + l15 + (tj * 16 + lk + 4 * reg) * rt) * 8u, acc[tj][ti][reg]);
}

// the product's full-tile core (dense_tile.h), one wavefront per tile
__global__ void __launch_bounds__(256, 2) k_prod(DevPlan P, const DenseGroup *dg, int ntiles) {
    const int t = blockIdx.x * 4 + rfl(threadIdx.x >> 6);
    if (t >= ntiles) return;
    const DenseGroup G = dg[t];
    dense_tile_core_full(P, rfl_ptr(P.Lx + G.tile_off), rfl(G.rt), rfl(G.task_begin), rfl(G.task_end), threadIdx.x & 63);
}

template <int P, int OCC>
__global__ void __launch_bounds__(256, OCC) k_macro(DevPlanLite Pl, const MacroGroup *mg, const MacroTask *mt, int nmacro) {
    if ((int)blockIdx.x < nmacro) dense_macro_tile<P>(Pl.Lx, Pl.D, mg + blockIdx.x, mt);
}

int main(int argc, char **argv) {
    const int R = 5632, ntask = argc > 1 ? atoi(argv[1]) : 5, nmacro = argc > 2 ? atoi(argv[2]) : 680;
    const int nsrc = ntask * 64;
    // source panels: ntask panels of R rows x 64 columns; targets: a separate region of nmacro x 4 tiles (rt = 64)
    const size_t src_d = (size_t)R * nsrc, tgt_d = (size_t)nmacro * 4 * 4096;
    std::vector<double> h(src_d + tgt_d), hd(nsrc);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5; };
    for (auto &v : h) v = rnd();
    for (auto &v : hd) v = 1.0 + 0.5 * rnd();
    std::vector<MacroGroup> mg(nmacro);
    std::vector<MacroTask> mt((size_t)nmacro * ntask);
    for (int m = 0; m < nmacro; m++) {
        const int nb = R / 64 - 1, J0 = (2 * (m % 37)) % (nb - 4), I0 = J0 + 2 + 2 * ((m / 37) % ((nb - J0 - 3) / 2));
        for (int w = 0; w < 4; w++) mg[m].tile_off[w] = (int64_t)src_d + ((int64_t)m * 4 + w) * 4096;
        mg[m].rt[0] = mg[m].rt[1] = 64;
        mg[m].task_begin = m * ntask; mg[m].task_end = (m + 1) * ntask; mg[m].nsteps = ntask * 16; mg[m].pad = 0;
        for (int q = 0; q < ntask; q++) {
            MacroTask &T = mt[(size_t)m * ntask + q];
            T.panel_off = (int64_t)q * 64 * R; T.r8 = R * 8; T.K = 64; T.dfirst = q * 64;
            T.row_off[0] = 64 * I0; T.row_off[1] = 64 * (I0 + 1); T.row_off[2] = 64 * J0; T.row_off[3] = 64 * (J0 + 1);
        }
    }
    double *dL, *dD; MacroGroup *dg; MacroTask *dt;
    CK(hipMalloc(&dL, h.size() * 8)); CK(hipMalloc(&dD, hd.size() * 8)); CK(hipMalloc(&dg, mg.size() * sizeof(MacroGroup))); CK(hipMalloc(&dt, mt.size() * sizeof(MacroTask)));
    CK(hipMemcpy(dD, hd.data(), hd.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, mg.data(), mg.size() * sizeof(MacroGroup), hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, mt.data(), mt.size() * sizeof(MacroTask), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = (double)nmacro * 4 * 2.0 * 64 * 64 * nsrc;
    std::vector<double> ref(tgt_d), out(tgt_d);
    DevPlanLite Pl{dL, dD};
    auto run = [&](const char *name, auto launch, std::vector<double> &res) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipMemcpy(dL, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            (void)hipEventRecord(e0, 0);
            launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::fmin(best, ms);
        }
        (void)hipMemcpy(res.data(), dL + src_d, tgt_d * 8, hipMemcpyDeviceToHost);
        printf("%-28s %8.1f us  %6.2f TFLOP/s  (%d tiles, K = %d)\n", name, best * 1e3, flops / (best * 1e-3) / 1e12, nmacro * 4, nsrc);
    };
    {   // the same work as records of the product kernel
        std::vector<DenseGroup> pg((size_t)nmacro * 4);
        std::vector<DenseTask> pt((size_t)nmacro * 4 * ntask);
        for (int m = 0; m < nmacro; m++)
            for (int w = 0; w < 4; w++) {
                DenseGroup &G = pg[(size_t)m * 4 + w];
                G.tile_off = mg[m].tile_off[w]; G.rt = 64; G.nrt = 64; G.wt = 64; G.pad = 1;
                G.task_begin = (int)((size_t)(m * 4 + w) * ntask); G.task_end = G.task_begin + ntask;
                for (int q = 0; q < ntask; q++) {
                    const MacroTask &T = mt[(size_t)m * ntask + q];
                    DenseTask &X = pt[(size_t)G.task_begin + q];
                    X = DenseTask{};
                    X.panel_off = T.panel_off; X.r8 = T.r8; X.K = T.K; X.dfirst = T.dfirst;
                    X.row_lo = T.row_off[w >> 1]; X.nrows = 64; X.col_lo = T.row_off[2 + (w & 1)]; X.ncols = 64;
                }
            }
        DenseGroup *dpg; DenseTask *dpt;
        CK(hipMalloc(&dpg, pg.size() * sizeof(DenseGroup))); CK(hipMalloc(&dpt, pt.size() * sizeof(DenseTask)));
        CK(hipMemcpy(dpg, pg.data(), pg.size() * sizeof(DenseGroup), hipMemcpyHostToDevice));
        CK(hipMemcpy(dpt, pt.data(), pt.size() * sizeof(DenseTask), hipMemcpyHostToDevice));
        DevPlan DP{};
        DP.Lx = dL; DP.D = dD; DP.dtasks = dpt;
        run("product core (wave per tile)", [&] { hipLaunchKernelGGL(k_prod, dim3(nmacro), dim3(256), 0, 0, DP, dpg, nmacro * 4); }, ref);
    }
    std::vector<double> prod = ref;
    run("base (wave per tile, occ 2)", [&] { hipLaunchKernelGGL(k_base, dim3(nmacro), dim3(256), 0, 0, dL, dD, dg, dt, nmacro); }, ref);
    { double e = 0; for (size_t i = 0; i < tgt_d; i++) e = std::fmax(e, std::fabs(prod[i] - ref[i])); printf("    product vs base: max |diff| = %.3e\n", e); }
    auto cmp = [&](const char *name) {
        double e = 0; for (size_t i = 0; i < tgt_d; i++) e = std::fmax(e, std::fabs(out[i] - ref[i]));
        printf("    %s vs base: max |diff| = %.3e\n", name, e);
    };
    run("macro P=2 occ 2", [&] { hipLaunchKernelGGL((k_macro<2, 2>), dim3(nmacro), dim3(256), 0, 0, Pl, dg, dt, nmacro); }, out); cmp("macro P=2");
    run("macro P=4 occ 2", [&] { hipLaunchKernelGGL((k_macro<4, 2>), dim3(nmacro), dim3(256), 0, 0, Pl, dg, dt, nmacro); }, out); cmp("macro P=4");
    run("macro P=8 occ 2", [&] { hipLaunchKernelGGL((k_macro<8, 2>), dim3(nmacro), dim3(256), 0, 0, Pl, dg, dt, nmacro); }, out); cmp("macro P=8");
    run("macro P=4 occ 1", [&] { hipLaunchKernelGGL((k_macro<4, 1>), dim3(nmacro), dim3(256), 0, 0, Pl, dg, dt, nmacro); }, out); cmp("macro P=4 occ1");
    run("macro P=8 occ 1", [&] { hipLaunchKernelGGL((k_macro<8, 1>), dim3(nmacro), dim3(256), 0, 0, Pl, dg, dt, nmacro); }, out); cmp("macro P=8 occ1");
    return 0;
}
