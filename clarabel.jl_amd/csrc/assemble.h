// Host-side KKT assembly for the L1 seam (DirectLDLKKTSolver constructor):
// builds the :triu CSC image of K = [P A' . ; A -Hs -B ; . -B' D] together with the LDLDataMap
// index vectors and the expected pivot signs.  ref: src/kktsolvers/direct-ldl/
// directldl_kkt_assembly.jl:15-175, directldl_datamaps.jl, src/utils/csc_assembly.jl,
// src/kktsolvers/kktsolver_directldl.jl:112-126.  0-based throughout.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace hipkkt {

struct SparseMap {
    int kind = 0;                   // 1 SOC, 2 GenPow
    int pdim = 0;
    std::vector<int64_t> vec[3];    // SOC: u, v ; GenPow: q, r, p
    int64_t D[3] = {0, 0, 0};
};

struct KKTImage {
    int64_t n = 0, m = 0, p = 0, N = 0, nnzP = 0, nnzA = 0, nHs = 0;
    std::vector<int64_t> colptr, rowval;
    std::vector<double> nzval;
    std::vector<int64_t> mapP, mapA, mapHs, diagP, diag_full, dsigns;
    std::vector<SparseMap> smaps;
};

// index arrays are 0-based here (the C ABI converts).  Returns "" on success.
std::string assemble_kkt(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Px,
                         const int64_t *Ap, const int64_t *Ai, const double *Ax, int64_t ncones,
                         const int64_t *numel, const int32_t *hs_dense, const int32_t *sparse_kind,
                         const int64_t *dim1, KKTImage &K);

// the same image built by count -> scan -> fill kernels on the current device (assemble_dev.hip); `stream` is a hipStream_t
std::string assemble_kkt_device(void *stream, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Px,
                                const int64_t *Ap, const int64_t *Ai, const double *Ax, int64_t ncones, const int64_t *numel,
                                const int32_t *hs_dense, const int32_t *sparse_kind, const int64_t *dim1, KKTImage &K);

}  // namespace hipkkt
