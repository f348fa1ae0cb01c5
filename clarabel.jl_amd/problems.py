"""Seeded synthetic problem generators for the benchmark configs.

All problems are feasible by construction: x0 ~ N(0,1), s0 strictly inside the cone,
b = A x0 + s0, q ~ N(0,1).  ``np.random.default_rng(seed)``; Float64."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .cone_api import (ExponentialConeT, GenPowerConeT, NonnegativeConeT, PowerConeT, PSDTriangleConeT, SecondOrderConeT, ZeroConeT,
                       triangular_number)


def _sparse_rows(rng, nrows, ncols, k, window=None, centers=None):
    """nrows x ncols CSC with k nonzeros per row, N(0,1) values; uniform columns or a +-window band."""
    rows = np.repeat(np.arange(nrows), k)
    if window is None:
        cols = rng.integers(0, ncols, size=nrows * k)
    else:
        c = np.repeat(centers, k)
        cols = np.clip(c + rng.integers(-window, window + 1, size=nrows * k), 0, ncols - 1)
    vals = rng.standard_normal(nrows * k)
    M = sp.coo_matrix((vals, (rows, cols)), shape=(nrows, ncols)).tocsc()
    M.sum_duplicates()
    return M


def _psd_P(rng, n, k, window=None):
    """P = S + S' + diag(rowsum|.| + 0.1): sparse, symmetric, diagonally dominant (PSD)."""
    S = _sparse_rows(rng, n, n, k, window, np.arange(n) if window is not None else None)
    P = S + S.T
    P = P + sp.diags(np.asarray(abs(P).sum(axis=1)).ravel() + 0.1)
    return sp.csc_matrix(P)


def random_sparse_qp(n=10000, m=20000, seed=2, kA=3, kP=1, window=None):
    """cfg 1 (n=1000,m=2000,seed 1,kA=4,kP=2), cfg 2a (defaults, uniform columns) and
    cfg 2b (window=50, kA=4, kP=2): NN(m) cone only."""
    rng = np.random.default_rng(seed)
    centers = (np.arange(m) * n) // m if window is not None else None
    A = _sparse_rows(rng, m, n, kA, window, centers)
    P = _psd_P(rng, n, kP, window)
    x0 = rng.standard_normal(n)
    s0 = rng.uniform(0.1, 1.0, m)
    b = A @ x0 + s0
    q = rng.standard_normal(n)
    return P, q, A, b, [NonnegativeConeT(m)]


def portfolio_socp(n=5000, nsoc=50, socdim=101, seed=3):
    """cfg 3: Zero(1) [1'x = 1], NN(n) [x >= 0], nsoc x SOC(socdim) [||G_k x|| <= t-style rows]."""
    rng = np.random.default_rng(seed)
    P = _psd_P(rng, n, 2)
    blocks = [sp.csc_matrix(np.ones((1, n))), -sp.identity(n, format="csc")]
    for _ in range(nsoc):
        blocks.append(_sparse_rows(rng, socdim, n, 5))
    A = sp.vstack(blocks).tocsc()
    m = A.shape[0]
    x0 = np.abs(rng.standard_normal(n))
    x0 /= x0.sum()
    s0 = np.zeros(m)
    s0[1:1 + n] = x0 + 0.0  # slack of -x + s = 0 -> s = x >= 0
    b = np.zeros(m)
    b[0] = 1.0
    off = 1 + n
    Ax = A @ x0
    for _ in range(nsoc):
        t = rng.standard_normal(socdim)
        t[0] = np.linalg.norm(t[1:]) + rng.uniform(0.1, 1.0)
        b[off:off + socdim] = Ax[off:off + socdim] + t
        off += socdim
    q = rng.standard_normal(n)
    cones = [ZeroConeT(1), NonnegativeConeT(n)] + [SecondOrderConeT(socdim)] * nsoc
    return P, q, A, b, cones


def sdp_blocks(n=1000, ncones=20, dim=50, seed=5):
    """cfg 5: ncones x PSDTriangle(dim), rows with 3 nnz, b_k = A_k x0 + svec(random PD), P = 0.01 I."""
    rng = np.random.default_rng(seed)
    ne = triangular_number(dim)
    A = _sparse_rows(rng, ncones * ne, n, 3)
    x0 = rng.standard_normal(n)
    b = A @ x0
    il = np.tril_indices(dim)
    r, c = il[1], il[0]
    isd = r == c
    for k in range(ncones):
        M = rng.standard_normal((dim, dim))
        S = M @ M.T / dim + np.eye(dim)
        sv = np.where(isd, S[r, c], S[r, c] * np.sqrt(2.0))
        b[k * ne:(k + 1) * ne] += sv
    P = sp.identity(n, format="csc") * 0.01
    q = rng.standard_normal(n)
    return P, q, A, b, [PSDTriangleConeT(dim)] * ncones


def batch_problem(seed):
    """cfg 4: one of 256 Maros-Meszaros-like QPs (the real set is not available offline): n in
    [50,2000], m in [n,3n], mixed Zero/NN rows, uniform or windowed pattern, column scaling 10^U(-2,2)."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 2001))
    m = int(rng.integers(n, 3 * n + 1))
    window = None if rng.random() < 0.5 else int(rng.integers(5, 60))
    centers = (np.arange(m) * n) // m if window is not None else None
    A = _sparse_rows(rng, m, n, int(rng.integers(2, 6)), window, centers)
    scale = 10.0 ** rng.uniform(-2, 2, n)
    A = sp.csc_matrix(A @ sp.diags(scale))
    P = _psd_P(rng, n, int(rng.integers(1, 3)), window)
    nzero = int(rng.integers(0, max(1, min(n // 4, m // 4))))
    x0 = rng.standard_normal(n)
    s0 = rng.uniform(0.1, 1.0, m)
    s0[:nzero] = 0.0
    b = A @ x0 + s0
    q = rng.standard_normal(n)
    cones = ([ZeroConeT(nzero)] if nzero else []) + [NonnegativeConeT(m - nzero)]
    return P, q, A, b, cones


def nonsymmetric_mix(n=300, nexp=60, npow=40, ngenpow=10, nn=100, nzero=10, socdim=8, seed=7, kA=4):
    """The non-symmetric cones of the reference next to the symmetric ones (no benchmark config uses them; this is the parity case
    for the 3 x 3 dense Hs blocks of the Exponential / Power cone and the rank-3 expansion of the Generalized Power cone):
    Zero(nzero), NN(nn), nexp x Exp, npow x Pow(alpha ~ U(0.1, 0.9)), ngenpow x GenPow(len(alpha) in 2..4, dim2 in 1..3), SOC(socdim).
    Strictly feasible by construction (s0 on a positive multiple of each cone's central ray), P positive definite."""
    rng = np.random.default_rng(seed)
    cones, s0 = [], []
    if nzero:
        cones.append(ZeroConeT(nzero))
        s0.append(np.zeros(nzero))
    if nn:
        cones.append(NonnegativeConeT(nn))
        s0.append(rng.uniform(0.1, 1.0, nn))
    for _ in range(nexp):          # (-1.0514, 0.5564, 1.2590): the point the reference starts this cone from (coneops_expcone.jl:42-44)
        cones.append(ExponentialConeT())
        s0.append(rng.uniform(0.5, 2.0) * np.array([-1.051383945322714, 0.556409619469370, 1.258967884768947]))
    for _ in range(npow):
        a = float(rng.uniform(0.1, 0.9))
        cones.append(PowerConeT(a))
        s0.append(rng.uniform(0.5, 2.0) * np.array([np.sqrt(1.0 + a), np.sqrt(2.0 - a), 0.0]))
    for _ in range(ngenpow):
        d1, d2 = int(rng.integers(2, 5)), int(rng.integers(1, 4))
        a = rng.uniform(0.2, 1.0, d1)
        a = a / a.sum()
        a[-1] = 1.0 - a[:-1].sum()
        cones.append(GenPowerConeT(tuple(float(v) for v in a), d2))
        s0.append(rng.uniform(0.5, 2.0) * np.concatenate([np.sqrt(1.0 + a), np.zeros(d2)]))
    if socdim:
        cones.append(SecondOrderConeT(socdim))
        v = rng.standard_normal(socdim - 1)
        s0.append(np.concatenate([[1.0 + np.linalg.norm(v)], v]))
    s0 = np.concatenate(s0)
    m = s0.size
    A = _sparse_rows(rng, m, n, kA)
    P = _psd_P(rng, n, 2)
    x0 = rng.standard_normal(n)
    b = A @ x0 + s0
    q = rng.standard_normal(n)
    return P, q, A, b, cones
