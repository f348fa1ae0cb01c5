"""CPU model of the register / memory layouts of clarabel.jl_amd/csrc/front_block2.hip (round 5): the index algebra the kernel relies
on, executed with a numpy model of one wavefront's v_mfma_f64_16x16x4_f64 and checked against dense products.

Lane l = 16 lk + l15.  One instruction:  D[m][n] += sum_k A[m][k] B[k][n]  with  A[l15][lk] = a[l],  B[lk][l15] = b[l],
D[lk + 4 r][l15] = c[l][r]  (the maps hipkkt_selftest_mfma checks on the device).  A 64 x 64 tile lives TRANSPOSED in the
accumulators: wave w, acc[sub][reg][l] = tile[16 w + l15][16 sub + lk + 4 reg].  Then
  * a product C^T = S^T R^T takes its B operand straight from the accumulators: k-step kk <-> acc[kk // 4][kk % 4];
  * the shared operand S comes in "operand order": value (mu, kappa) at (16 (mu // 16) + kappa // 4) * 64 + 16 (kappa % 4) + mu % 16;
  * a stream record (block of 8 pivots) is consumed as l = p T, T = L_bb^-T D_b^-1, and the rank-8 update of the columns right of it.
What the GPU tests prove on the device, this file pins as a specification that runs anywhere."""
import numpy as np

LANES = np.arange(64)
L15, LK = LANES & 15, LANES >> 4


def mfma(a, b, c):
    """one v_mfma_f64_16x16x4_f64 of a wavefront: a, b [64], c [64, 4] -> c"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[L15, LK] = a
    B[LK, L15] = b
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[:, r] += D[LK + 4 * r, L15]
    return out


def to_acc(tile, w):
    """rows 16 w .. 16 w + 15 of a 64 x 64 tile in the transposed accumulator layout: [sub][lane, reg]"""
    acc = np.zeros((4, 64, 4))
    for sub in range(4):
        for reg in range(4):
            acc[sub][:, reg] = tile[16 * w + L15, 16 * sub + LK + 4 * reg]
    return acc


def from_acc(acc, w, out):
    for sub in range(4):
        for reg in range(4):
            out[16 * w + L15, 16 * sub + LK + 4 * reg] = acc[sub][:, reg]


def operand_order(M):
    """M(mu, kappa) -> flat [4096] in operand order"""
    flat = np.zeros(4096)
    mu, ka = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
    flat[(16 * (mu // 16) + ka // 4) * 64 + 16 * (ka % 4) + mu % 16] = M[mu, ka]
    return flat


def test_regular_step_and_updates_in_the_transposed_layout():
    rng = np.random.default_rng(0)
    A_ij = rng.standard_normal((64, 64))              # this workgroup's rows of panel j
    A_ik = rng.standard_normal((64, 64))              # ... of panel k > j
    Minv = np.triu(rng.standard_normal((64, 64)))     # L_jj^-T D_j^-1, upper triangular
    Lkj_D = rng.standard_normal((64, 64))             # rows of the diagonal workgroup k: L(k,j) D_j
    OT_minv = operand_order(Minv.T)                   # value (c, k) = Minv[k][c]
    OT_l = operand_order(Lkj_D)                       # value (row c of block k, column m of panel j)
    X = np.zeros((64, 64)); Aik_new = np.zeros((64, 64))
    for w in range(4):
        acc_j, acc_k = to_acc(A_ij, w), to_acc(A_ik, w)
        x = np.zeros((4, 64, 4))
        for so in range(4):
            for kk in range(4 * so + 4):              # k <= c: the k-steps below a strip's diagonal are skipped
                x[so] = mfma(OT_minv[(so * 16 + kk) * 64 + LANES], acc_j[kk >> 2][:, kk & 3], x[so])
        from_acc(x, w, X)
        for so in range(4):                           # A_ik^T -= (L(k,j) D) X^T, B operand = -X from the registers
            for kk in range(16):
                acc_k[so] = mfma(OT_l[(so * 16 + kk) * 64 + LANES], -x[kk >> 2][:, kk & 3], acc_k[so])
        from_acc(acc_k, w, Aik_new)
    assert np.allclose(X, A_ij @ Minv, rtol=0, atol=1e-12)
    assert np.allclose(Aik_new, A_ik - X @ Lkj_D.T, rtol=0, atol=1e-11)


def test_stream_records_are_consumed_as_l_equals_p_times_T():
    rng = np.random.default_rng(1)
    # the producer's tile: symmetric, diagonally dominant; LDL^T by blocks of 8 gives the records
    G = rng.standard_normal((64, 64))
    Tile = G @ G.T + 64 * np.eye(64)
    P_rows = rng.standard_normal((64, 64))            # the consumer's rows of the same panel
    Lref = np.linalg.cholesky(Tile)
    dref = np.diag(Lref) ** 2
    Lunit = Lref / np.diag(Lref)
    Xref = np.linalg.solve(Lunit * dref, P_rows.T).T  # rows l of the consumer: l D L^T = p  ->  l = p L^-T D^-1
    work = Tile.copy()
    xr = [to_acc(P_rows, w) for w in range(4)]
    out = np.zeros((64, 64))
    for Bk in range(8):
        o, sb, par = 8 * Bk, Bk >> 1, Bk & 1
        # producer: eliminate the 8 pivots of the block (raw columns cr = d l of ALL rows of the tile, zero on the rows up to the block's own)
        blk = work[o:o + 8, o:o + 8]
        Lb = np.linalg.cholesky(blk); db = np.diag(Lb) ** 2; Lbu = Lb / np.diag(Lb)
        Tm = np.linalg.inv(Lbu).T / db                # T = L_bb^-T D_b^-1  (8 x 8 upper)
        lcols = work[:, o:o + 8] @ Tm                 # l of every row of the tile for these 8 pivots
        raw = lcols * db
        raw[: o + 8, :] = 0.0
        work[o + 8:, o + 8:] -= lcols[o + 8:, :] @ raw[o + 8:, :].T
        # the record in its chunk layout: chunks 2 q + e: raw[16 q + l15][4 e + lk]; 8 + e: T[4 e + lk][l15] (l15 < 8)
        rec = np.zeros((12, 64))
        for q in range(4):
            for e in range(2):
                rec[2 * q + e] = raw[16 * q + L15, 4 * e + LK]
        for e in range(2):
            rec[8 + e] = np.where(L15 < 8, Tm[4 * e + LK, np.minimum(L15, 7)], 0.0)
        # consumer, wave by wave
        for w in range(4):
            lT = np.zeros((64, 4))
            lT = mfma(rec[8], xr[w][sb][:, 2 * par], lT)
            lT = mfma(rec[9], xr[w][sb][:, 2 * par + 1], lT)
            assert np.allclose(lT[:, 2:], 0.0)        # rows 8 .. 15 of the padded product
            xr[w][sb][:, 2 * par] = lT[:, 0]          # the finished columns stay in the tile's registers
            xr[w][sb][:, 2 * par + 1] = lT[:, 1]
            for q in range(sb + par, 4):              # rank-8 update of the columns right of the block
                xr[w][q] = mfma(rec[2 * q], -lT[:, 0], xr[w][q])
                xr[w][q] = mfma(rec[2 * q + 1], -lT[:, 1], xr[w][q])
    for w in range(4):
        from_acc(xr[w], w, out)
    assert np.allclose(out, Xref, rtol=0, atol=1e-10)
