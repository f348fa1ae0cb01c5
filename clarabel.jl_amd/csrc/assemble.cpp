#include "assemble.h"

namespace hipkkt {

namespace {
// column-oriented builder: pass 1 counts, pass 2 places entries in arrival order
struct ColumnBuilder {
    std::vector<int64_t> &colptr, &rowval;
    std::vector<double> &nzval;
    std::vector<int64_t> cursor;
    bool counting = true;
    ColumnBuilder(KKTImage &K) : colptr(K.colptr), rowval(K.rowval), nzval(K.nzval) {}
    // returns the nz index the entry landed on (pass 2) or -1 (pass 1)
    int64_t put(int64_t col, int64_t row, double v) {
        if (counting) { colptr[col + 1]++; return -1; }
        int64_t d = cursor[col]++;
        rowval[d] = row;
        nzval[d] = v;
        return d;
    }
    void finish_counting() {
        for (size_t j = 0; j + 1 < colptr.size(); j++) colptr[j + 1] += colptr[j];
        rowval.assign(colptr.back(), 0);
        nzval.assign(colptr.back(), 0.0);
        cursor.assign(colptr.begin(), colptr.end() - 1);
        counting = false;
    }
};
}  // namespace

std::string assemble_kkt(int64_t n, int64_t m, const int64_t *Pp, const int64_t *Pi, const double *Px,
                         const int64_t *Ap, const int64_t *Ai, const double *Ax, int64_t ncones,
                         const int64_t *numel, const int32_t *hs_dense, const int32_t *sparse_kind,
                         const int64_t *dim1, KKTImage &K) {
    K = KKTImage();
    K.n = n; K.m = m; K.nnzP = Pp[n]; K.nnzA = Ap[n];
    int64_t rows = 0;
    for (int64_t c = 0; c < ncones; c++) {
        if (numel[c] < 0) return "negative cone dimension";
        rows += numel[c];
        K.nHs += hs_dense[c] ? numel[c] * (numel[c] + 1) / 2 : numel[c];
        if (sparse_kind[c] == 1 || sparse_kind[c] == 2) {
            SparseMap sm;
            sm.kind = sparse_kind[c];
            sm.pdim = sm.kind == 1 ? 2 : 3;
            if (sm.kind == 1) { sm.vec[0].resize(numel[c]); sm.vec[1].resize(numel[c]); }
            else {
                if (dim1[c] < 0 || dim1[c] > numel[c]) return "bad GenPow dim1";
                sm.vec[0].resize(dim1[c]); sm.vec[1].resize(numel[c] - dim1[c]); sm.vec[2].resize(numel[c]);
            }
            K.p += sm.pdim;
            K.smaps.push_back(std::move(sm));
        } else if (sparse_kind[c] != 0) return "unknown sparse_kind";
    }
    if (rows != m) return "cone dimensions do not sum to the number of rows of A";
    for (int64_t j = 0; j < n; j++)
        for (int64_t q = Pp[j]; q < Pp[j + 1]; q++)
            if (Pi[q] > j || Pi[q] < 0) return "P must be upper triangular";
    for (int64_t q = 0; q < K.nnzA; q++)
        if (Ai[q] < 0 || Ai[q] >= m) return "A row index out of range";
    K.N = n + m + K.p;
    K.colptr.assign(K.N + 1, 0);
    K.mapP.assign(K.nnzP, 0); K.mapA.assign(K.nnzA, 0); K.mapHs.assign(K.nHs, 0);
    ColumnBuilder B(K);
    auto no_diag = [&](int64_t j) { return Pp[j] == Pp[j + 1] || Pi[Pp[j + 1] - 1] != j; };

    for (int pass = 0; pass < 2; pass++) {
        // upper-left block: P, then a structural zero where P lacks its diagonal (kept LAST in the column)
        for (int64_t j = 0; j < n; j++)
            for (int64_t q = Pp[j]; q < Pp[j + 1]; q++) {
                int64_t d = B.put(j, Pi[q], Px[q]);
                if (pass) K.mapP[q] = d;
            }
        for (int64_t j = 0; j < n; j++)
            if (no_diag(j)) B.put(j, j, 0.0);
        // upper-right block: A' (A visited column by column)
        for (int64_t j = 0; j < n; j++)
            for (int64_t q = Ap[j]; q < Ap[j + 1]; q++) {
                int64_t d = B.put(n + Ai[q], j, Ax[q]);
                if (pass) K.mapA[q] = d;
            }
        // lower-right: Hs blocks, then the expansion columns of sparse cones
        int64_t row = n, pcol = n + m, h = 0;
        size_t si = 0;
        for (int64_t c = 0; c < ncones; c++) {
            int64_t d = numel[c];
            if (!hs_dense[c]) {
                for (int64_t i = 0; i < d; i++) {
                    int64_t z = B.put(row + i, row + i, 0.0);
                    if (pass) K.mapHs[h + i] = z;
                }
                h += d;
            } else {
                for (int64_t cc = 0; cc < d; cc++)
                    for (int64_t rr = 0; rr <= cc; rr++) {
                        int64_t z = B.put(row + cc, row + rr, 0.0);
                        if (pass) K.mapHs[h] = z;
                        h++;
                    }
            }
            if (sparse_kind[c] == 1) {
                SparseMap &sm = K.smaps[si++];
                for (int64_t i = 0; i < d; i++) { int64_t z = B.put(pcol, row + i, 0.0); if (pass) sm.vec[1][i] = z; }      // v
                for (int64_t i = 0; i < d; i++) { int64_t z = B.put(pcol + 1, row + i, 0.0); if (pass) sm.vec[0][i] = z; }  // u
                for (int t = 0; t < 2; t++) { int64_t z = B.put(pcol + t, pcol + t, 0.0); if (pass) sm.D[t] = z; }
                pcol += 2;
            } else if (sparse_kind[c] == 2) {
                SparseMap &sm = K.smaps[si++];
                int64_t d1 = dim1[c], d2 = d - d1;
                for (int64_t i = 0; i < d1; i++) { int64_t z = B.put(pcol, row + i, 0.0); if (pass) sm.vec[0][i] = z; }           // q
                for (int64_t i = 0; i < d2; i++) { int64_t z = B.put(pcol + 1, row + d1 + i, 0.0); if (pass) sm.vec[1][i] = z; }  // r
                for (int64_t i = 0; i < d; i++) { int64_t z = B.put(pcol + 2, row + i, 0.0); if (pass) sm.vec[2][i] = z; }        // p
                for (int t = 0; t < 3; t++) { int64_t z = B.put(pcol + t, pcol + t, 0.0); if (pass) sm.D[t] = z; }
                pcol += 3;
            }
            row += d;
        }
        if (pass == 0) B.finish_counting();
    }
    K.diag_full.resize(K.N);
    K.diagP.resize(n);
    for (int64_t j = 0; j < K.N; j++) K.diag_full[j] = K.colptr[j + 1] - 1;
    for (int64_t j = 0; j < n; j++) K.diagP[j] = K.colptr[j + 1] - 1;
    K.dsigns.assign(K.N, 1);
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
