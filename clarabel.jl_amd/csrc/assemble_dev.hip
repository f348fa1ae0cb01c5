// KKT assembly ON THE DEVICE for the L1 seam (SURVEY.md section 7 step 6 / kernel K9; north_star: "KKT assembly ... writes
// a coalesced CSC image in HBM").  Count -> scan -> fill kernels that reproduce, entry for entry, the :triu image and the
// LDLDataMap index vectors the reference builds on the host:
//   src/kktsolvers/direct-ldl/directldl_kkt_assembly.jl:15-175   (_assemble_kkt_matrix: colcount pass, fill pass)
//   src/utils/csc_assembly.jl:19-260                             (block / diagonal / column-vector count + fill primitives)
//   src/kktsolvers/direct-ldl/directldl_datamaps.jl:8-167        (SOC / GenPow expansion maps)
// Entry order inside a column (what makes the maps bit-identical): P's entries in P's order (+ a structural zero on the
// diagonal where P has none) | row i of A by ascending column | the Hs entries of that column (diagonal cone: the
// diagonal; dense cone: rows row0..row0+local) | expansion columns: the cone's rows, then the diagonal.
// The only data-dependent step is the transposition of A: entries are scattered into their row's slot range with an
// atomic cursor and then RANKED by column inside the row, so the final order does not depend on the order of the atomics.
// 0-based; the host twin (assemble.cpp) stays as the reference the GPU tests compare this against.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "assemble.h"
#include "runtime_pool.h"

namespace hipkkt {

namespace {

struct AsmDev {
    int64_t n, m, N, ncones, next;
    const int64_t *Pp, *Pi, *Ap, *Ai;
    const double *Px, *Ax;
    const int64_t *cone_row0;     // [ncones+1] first row of every cone (cone-local numbering, 0..m)
    const int64_t *cone_hsoff;    // [ncones]   offset of the cone's block in the Hs vector
    const int32_t *cone_dense;    // [ncones]
    const int64_t *ext_row0, *ext_len, *ext_off;   // [next] expansion column e: first KKT row, length, offset in extidx
    unsigned *rowcnt, *cursor;    // [m]
    int64_t *colptr, *rowval, *tcol, *tq, *mapP, *mapA, *mapHs, *extidx, *extD;
    double *nzval;
};

__device__ __forceinline__ int64_t find_cone(const AsmDev &D, int64_t i) {   // largest k with cone_row0[k] <= i (empty cones skipped)
    int64_t lo = 0, hi = D.ncones;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (D.cone_row0[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void k_asm_rowcount(AsmDev D, int64_t nnzA) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnzA) atomicAdd(D.rowcnt + D.Ai[q], 1u);
}

// colptr[c + 1] = number of entries of column c  (ref: the colcount pass, directldl_kkt_assembly.jl:43-100)
__global__ void k_asm_colcount(AsmDev D) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D.N) return;
    int64_t cnt;
    if (c < D.n) {
        const int64_t p0 = D.Pp[c], p1 = D.Pp[c + 1];
        cnt = p1 - p0 + ((p0 == p1 || D.Pi[p1 - 1] != c) ? 1 : 0);            // csc_assembly.jl:207-220 missing diagonal
    } else if (c < D.n + D.m) {
        const int64_t i = c - D.n, k = find_cone(D, i), local = i - D.cone_row0[k];
        cnt = (int64_t)D.rowcnt[i] + (D.cone_dense[k] ? local + 1 : 1);
    } else {
        cnt = D.ext_len[c - D.n - D.m] + 1;
    }
    D.colptr[c + 1] = cnt;
    if (c == 0) D.colptr[0] = 0;
}

// in-place inclusive scan of colptr[1..N] by one workgroup (set-up time; N <= a few 10^6)
__global__ void __launch_bounds__(1024) k_asm_scan(int64_t *v, int64_t n) {
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        part[threadIdx.x] = i < n ? v[i] : 0;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int64_t add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n) v[i] = part[threadIdx.x] + carry;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

__global__ void k_asm_fill_P(AsmDev D) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D.n) return;
    const int64_t p0 = D.Pp[j], p1 = D.Pp[j + 1];
    int64_t d = D.colptr[j];
    for (int64_t q = p0; q < p1; q++, d++) {
        D.rowval[d] = D.Pi[q];
        D.nzval[d] = D.Px[q];
        D.mapP[q] = d;
    }
    if (p0 == p1 || D.Pi[p1 - 1] != j) { D.rowval[d] = j; D.nzval[d] = 0.0; }
}

// A' : entry q of column j goes to SOME slot of its row's range; k_asm_place_A then orders the range by column
__global__ void k_asm_scatter_A(AsmDev D) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D.n) return;
    for (int64_t q = D.Ap[j]; q < D.Ap[j + 1]; q++) {
        const int64_t i = D.Ai[q];
        const int64_t d = D.colptr[D.n + i] + atomicAdd(D.cursor + i, 1u);
        D.tcol[d] = j;
        D.tq[d] = q;
    }
}

// one wavefront per row of A: rank of every entry = number of entries of the row in smaller columns.  Rows with more than
// kAsmLongRow entries (a budget row 1'x = 1 has n of them: O(k^2 / 64) per lane was 29 ms on cfg 3) are left to
// k_asm_place_A_long, one workgroup each with the row's columns staged through LDS in chunks.
constexpr int kAsmLongRow = 512;
__global__ void __launch_bounds__(256) k_asm_place_A(AsmDev D) {
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= D.m) return;
    const int lane = threadIdx.x & 63;
    const int64_t base = D.colptr[D.n + i], k = D.rowcnt[i];
    if (k > kAsmLongRow) return;
    for (int64_t a = lane; a < k; a += 64) {
        const int64_t c = D.tcol[base + a], q = D.tq[base + a];
        int64_t rank = 0;
        for (int64_t b = 0; b < k; b++) rank += D.tcol[base + b] < c ? 1 : 0;
        const int64_t d = base + rank;
        D.rowval[d] = c;
        D.nzval[d] = D.Ax[q];
        D.mapA[q] = d;
    }
}

__global__ void __launch_bounds__(1024) k_asm_place_A_long(AsmDev D, const int64_t *__restrict__ long_rows) {
    __shared__ int64_t chunk[2048];
    const int64_t i = long_rows[blockIdx.x];
    const int64_t base = D.colptr[D.n + i], k = D.rowcnt[i];
    // every thread ranks the entries a = tid, tid + 1024, ... against the whole row, read chunk by chunk through LDS
    for (int64_t a0 = 0; a0 < k; a0 += 1024) {
        const int64_t a = a0 + threadIdx.x;
        const int64_t c = a < k ? D.tcol[base + a] : 0;
        int64_t rank = 0;
        for (int64_t b0 = 0; b0 < k; b0 += 2048) {
            __syncthreads();
            for (int64_t b = threadIdx.x; b < 2048 && b0 + b < k; b += 1024) chunk[b] = D.tcol[base + b0 + b];
            __syncthreads();
            const int64_t nb_ = k - b0 < 2048 ? k - b0 : 2048;
            if (a < k)
                for (int64_t b = 0; b < nb_; b++) rank += chunk[b] < c ? 1 : 0;
        }
        if (a < k) {
            const int64_t q = D.tq[base + a], d = base + rank;
            D.rowval[d] = c;
            D.nzval[d] = D.Ax[q];
            D.mapA[q] = d;
        }
    }
}

// Hs entries of column n + i (ref: directldl_kkt_assembly.jl:131-147 with csc_assembly.jl diag / dense-triu fills)
__global__ void k_asm_fill_Hs(AsmDev D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D.m) return;
    const int64_t k = find_cone(D, i), row0 = D.cone_row0[k], local = i - row0;
    const int64_t base = D.colptr[D.n + i] + D.rowcnt[i];
    if (!D.cone_dense[k]) {
        D.rowval[base] = D.n + i;
        D.nzval[base] = 0.0;
        D.mapHs[D.cone_hsoff[k] + local] = base;
    } else {
        const int64_t h = D.cone_hsoff[k] + local * (local + 1) / 2;
        for (int64_t rr = 0; rr <= local; rr++) {
            D.rowval[base + rr] = D.n + row0 + rr;
            D.nzval[base + rr] = 0.0;
            D.mapHs[h + rr] = base + rr;
        }
    }
}

// expansion columns of the sparse cones (ref: directldl_datamaps.jl:42-59, 116-144): the vector, then the diagonal
__global__ void k_asm_fill_ext(AsmDev D) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= D.next) return;
    const int64_t c = D.n + D.m + e, base = D.colptr[c], len = D.ext_len[e], r0 = D.ext_row0[e], off = D.ext_off[e];
    for (int64_t t = 0; t < len; t++) {
        D.rowval[base + t] = r0 + t;
        D.nzval[base + t] = 0.0;
        D.extidx[off + t] = base + t;
    }
    D.rowval[base + len] = c;
    D.nzval[base + len] = 0.0;
    D.extD[e] = base + len;
}

// temporaries of one assembly: blocks of the process-wide cache (runtime_pool.h), handed back once the stream is idle
struct DevBuf {
    std::vector<std::pair<void *, size_t>> ptrs;
    hipStream_t st;
    int device = 0;
    bool ok = true;
    DevBuf() { (void)hipGetDevice(&device); }
    template <class T>
    T *alloc(size_t n) {
        size_t cap = 0;
        void *p = hipkkt::RuntimePool::get().dev_alloc(device, std::max<size_t>(n, 1) * sizeof(T), &cap);
        if (!p) { ok = false; return nullptr; }
        ptrs.push_back({p, cap});
        return (T *)p;
    }
    template <class T>
    T *up(const T *h, size_t n) {
        T *p = alloc<T>(n);
        if (p && n && hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, st) != hipSuccess) ok = false;
        return p;
    }
    ~DevBuf() {
        if (!ptrs.empty()) (void)hipStreamSynchronize(st);      // hipFree used to wait implicitly
        for (auto &a : ptrs) hipkkt::RuntimePool::get().dev_free(device, a.first, a.second);
    }
};

inline unsigned nb(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

// Same contract as assemble_kkt (assemble.h); the image is built on the device of the current context, on `stream`,
// and copied into K.  Structure checks (triangularity, ranges, cone sizes) are the host twin's.
std::string assemble_kkt_device(void *stream_, int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Px,
                                const int64_t *Ap, const int64_t *Ai, const double *Ax, int64_t ncones, const int64_t *numel,
                                const int32_t *hs_dense, const int32_t *sparse_kind, const int64_t *dim1, KKTImage &K) {
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();   // the sticky per-thread error may hold a failure that is not ours (an embedding host's, or an allocation the
                               // runtime pool recovered from): the check at the end of this function must only see this function's launches
    K = KKTImage();
    K.n = n; K.m = m; K.nnzP = Pp[n]; K.nnzA = Ap[n];
    // ---- host: cone descriptors (structure only, O(#cones)) and the checks of the host twin
    std::vector<int64_t> row0(ncones + 1, 0), hsoff(ncones, 0), e_row0, e_len, e_off;
    std::vector<int32_t> dense(ncones, 0);
    int64_t rows = 0, extn = 0;
    for (int64_t c = 0; c < ncones; c++) {
        if (numel[c] < 0) return "negative cone dimension";
        row0[c] = rows;
        hsoff[c] = K.nHs;
        dense[c] = hs_dense[c] ? 1 : 0;
        K.nHs += hs_dense[c] ? numel[c] * (numel[c] + 1) / 2 : numel[c];
        if (sparse_kind[c] == 1 || sparse_kind[c] == 2) {
            SparseMap sm;
            sm.kind = sparse_kind[c];
            sm.pdim = sm.kind == 1 ? 2 : 3;
            if (sm.kind == 1) {   // columns: v, then u (directldl_datamaps.jl:42-59)
                sm.vec[0].resize(numel[c]); sm.vec[1].resize(numel[c]);
                for (int t = 0; t < 2; t++) { e_row0.push_back(n + rows); e_len.push_back(numel[c]); e_off.push_back(extn); extn += numel[c]; }
            } else {
                if (dim1[c] < 0 || dim1[c] > numel[c]) return "bad GenPow dim1";
                const int64_t d1 = dim1[c], d2 = numel[c] - d1;
                sm.vec[0].resize(d1); sm.vec[1].resize(d2); sm.vec[2].resize(numel[c]);
                e_row0.push_back(n + rows); e_len.push_back(d1); e_off.push_back(extn); extn += d1;            // q
                e_row0.push_back(n + rows + d1); e_len.push_back(d2); e_off.push_back(extn); extn += d2;       // r
                e_row0.push_back(n + rows); e_len.push_back(numel[c]); e_off.push_back(extn); extn += numel[c]; // p
            }
            K.p += sm.pdim;
            K.smaps.push_back(std::move(sm));
        } else if (sparse_kind[c] != 0) return "unknown sparse_kind";
        rows += numel[c];
    }
    row0[ncones] = rows;
    if (rows != m) return "cone dimensions do not sum to the number of rows of A";
    for (int64_t j = 0; j < n; j++)
        for (int64_t q = Pp[j]; q < Pp[j + 1]; q++)
            if (Pi[q] > j || Pi[q] < 0) return "P must be upper triangular";
    for (int64_t q = 0; q < K.nnzA; q++)
        if (Ai[q] < 0 || Ai[q] >= m) return "A row index out of range";
    // empty cones would break the binary search over row starts: drop them from the lookup tables
    std::vector<int64_t> lrow0, lhsoff;
    std::vector<int32_t> ldense;
    for (int64_t c = 0; c < ncones; c++)
        if (numel[c] > 0) { lrow0.push_back(row0[c]); lhsoff.push_back(hsoff[c]); ldense.push_back(dense[c]); }
    const int64_t lcones = (int64_t)lrow0.size();
    lrow0.push_back(rows);
    K.N = n + m + K.p;
    const int64_t N = K.N, next = K.p;

    // ---- device
    DevBuf B;
    B.st = st;
    AsmDev D{};
    D.n = n; D.m = m; D.N = N; D.ncones = std::max<int64_t>(lcones, 1); D.next = next;
    D.Pp = B.up(Pp, n + 1); D.Pi = B.up(Pi, K.nnzP); D.Px = B.up(Px, K.nnzP);
    D.Ap = B.up(Ap, n + 1); D.Ai = B.up(Ai, K.nnzA); D.Ax = B.up(Ax, K.nnzA);
    D.cone_row0 = B.up(lrow0.data(), lrow0.size());
    D.cone_hsoff = B.up(lhsoff.data(), lhsoff.size());
    D.cone_dense = B.up(ldense.data(), ldense.size());
    D.ext_row0 = B.up(e_row0.data(), e_row0.size());
    D.ext_len = B.up(e_len.data(), e_len.size());
    D.ext_off = B.up(e_off.data(), e_off.size());
    D.rowcnt = B.alloc<unsigned>(m);
    D.cursor = B.alloc<unsigned>(m);
    D.colptr = B.alloc<int64_t>(N + 1);
    if (!B.ok) return "device allocation failed during the KKT assembly";
    (void)hipMemsetAsync(D.rowcnt, 0, std::max<int64_t>(m, 1) * sizeof(unsigned), st);
    (void)hipMemsetAsync(D.cursor, 0, std::max<int64_t>(m, 1) * sizeof(unsigned), st);
    if (K.nnzA) hipLaunchKernelGGL(k_asm_rowcount, dim3(nb(K.nnzA)), dim3(256), 0, st, D, K.nnzA);
    if (N) hipLaunchKernelGGL(k_asm_colcount, dim3(nb(N)), dim3(256), 0, st, D);
    if (N) hipLaunchKernelGGL(k_asm_scan, dim3(1), dim3(1024), 0, st, D.colptr + 1, N);
    K.colptr.assign(N + 1, 0);
    if (hipMemcpyAsync(K.colptr.data(), D.colptr, (N + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return "device error during the KKT assembly (count / scan)";
    if (N == 0) K.colptr[0] = 0;
    const int64_t nnz = K.colptr[N];
    D.rowval = B.alloc<int64_t>(nnz); D.nzval = B.alloc<double>(nnz);
    D.tcol = B.alloc<int64_t>(nnz); D.tq = B.alloc<int64_t>(nnz);
    D.mapP = B.alloc<int64_t>(K.nnzP); D.mapA = B.alloc<int64_t>(K.nnzA); D.mapHs = B.alloc<int64_t>(K.nHs);
    D.extidx = B.alloc<int64_t>(extn); D.extD = B.alloc<int64_t>(next);
    if (!B.ok) return "device allocation failed during the KKT assembly";
    if (n) hipLaunchKernelGGL(k_asm_fill_P, dim3(nb(n)), dim3(256), 0, st, D);
    if (n) hipLaunchKernelGGL(k_asm_scatter_A, dim3(nb(n)), dim3(256), 0, st, D);
    if (m) hipLaunchKernelGGL(k_asm_place_A, dim3(nb(m, 4)), dim3(256), 0, st, D);
    {   // long rows of A (from A's row histogram, computed here on the host: O(nnz))
        std::vector<int64_t> cnt(m, 0), longs;
        for (int64_t q = 0; q < K.nnzA; q++) cnt[Ai[q]]++;
        for (int64_t i = 0; i < m; i++)
            if (cnt[i] > kAsmLongRow) longs.push_back(i);
        if (!longs.empty()) {
            const int64_t *dl = B.up(longs.data(), longs.size());
            if (!B.ok) return "device allocation failed during the KKT assembly";
            hipLaunchKernelGGL(k_asm_place_A_long, dim3((unsigned)longs.size()), dim3(1024), 0, st, D, dl);
            if (hipStreamSynchronize(st) != hipSuccess) return "device error during the KKT assembly (long rows)";   // `longs` is a local
        }
    }
    if (m) hipLaunchKernelGGL(k_asm_fill_Hs, dim3(nb(m)), dim3(256), 0, st, D);
    if (next) hipLaunchKernelGGL(k_asm_fill_ext, dim3(nb(next)), dim3(256), 0, st, D);
    K.rowval.assign(nnz, 0); K.nzval.assign(nnz, 0.0);
    K.mapP.assign(K.nnzP, 0); K.mapA.assign(K.nnzA, 0); K.mapHs.assign(K.nHs, 0);
    std::vector<int64_t> extidx(extn, 0), extD(next, 0);
    auto down = [&](void *h, const void *d, size_t bytes) { return bytes == 0 || hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st) == hipSuccess; };
    bool ok = down(K.rowval.data(), D.rowval, nnz * sizeof(int64_t)) && down(K.nzval.data(), D.nzval, nnz * sizeof(double)) &&
              down(K.mapP.data(), D.mapP, K.nnzP * sizeof(int64_t)) && down(K.mapA.data(), D.mapA, K.nnzA * sizeof(int64_t)) &&
              down(K.mapHs.data(), D.mapHs, K.nHs * sizeof(int64_t)) && down(extidx.data(), D.extidx, extn * sizeof(int64_t)) &&
              down(extD.data(), D.extD, next * sizeof(int64_t));
    if (!ok || hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return "device error during the KKT assembly (fill)";
    // ---- host: hand the expansion indices to the sparse maps, diagonal maps and pivot signs (closed form)
    {
        int64_t e = 0;
        for (SparseMap &sm : K.smaps) {
            if (sm.kind == 1) {   // column e = v, e + 1 = u
                std::copy(extidx.begin() + e_off[e], extidx.begin() + e_off[e] + e_len[e], sm.vec[1].begin());
                std::copy(extidx.begin() + e_off[e + 1], extidx.begin() + e_off[e + 1] + e_len[e + 1], sm.vec[0].begin());
            } else {
                for (int t = 0; t < 3; t++) std::copy(extidx.begin() + e_off[e + t], extidx.begin() + e_off[e + t] + e_len[e + t], sm.vec[t].begin());
            }
            for (int t = 0; t < sm.pdim; t++) sm.D[t] = extD[e + t];
            e += sm.pdim;
        }
    }
    K.diag_full.resize(N);
    K.diagP.resize(n);
    for (int64_t j = 0; j < N; j++) K.diag_full[j] = K.colptr[j + 1] - 1;
    for (int64_t j = 0; j < n; j++) K.diagP[j] = K.colptr[j + 1] - 1;
    K.dsigns.assign(N, 1);
    for (int64_t j = n; j < n + m; j++) K.dsigns[j] = -1;
    int64_t pp = n + m;
    for (const SparseMap &sm : K.smaps) {
        if (sm.kind == 1) { K.dsigns[pp] = -1; K.dsigns[pp + 1] = 1; }
        else { K.dsigns[pp] = -1; K.dsigns[pp + 1] = -1; K.dsigns[pp + 2] = 1; }
        pp += sm.pdim;
    }
    return "";
}

}  // namespace hipkkt
