"""CPU check of the index arithmetic of k_psd_hs (clarabel.jl_amd/csrc/kernels.hip): entry e of the packed upper
triangle of W (x)_s W  <->  (a <= b)  <->  ((i <= j), (k <= l)), evaluated with the kernel's four cases and order of
operations, must reproduce a literal restatement of the reference's skron! loop + pack_triu
(coneops_psdtrianglecone.jl:153-161, 502-540) bit for bit, and so must the vectorised host stand-in.  The GPU side of the same statement is
tests/test_gpu_kkt.py::test_assembly_bit_exact_and_factor_solve_parity[sdp_*]."""
import math

import numpy as np
import pytest

import clarabel_jl_amd  # noqa: F401
from julia_standin.cones import PSDTriangleCone


def _tri_root(e):
    t = int((math.sqrt(8.0 * e + 1.0) - 1.0) * 0.5)
    while t * (t + 1) // 2 > e:
        t -= 1
    while (t + 1) * (t + 2) // 2 <= e:
        t += 1
    return t


def _reference_skron_loop(A):
    """literal restatement of skron!(out, A), coneops_psdtrianglecone.jl:502-540 (1-based loops kept)"""
    n = A.shape[0]
    numel = n * (n + 1) // 2
    out = np.zeros((numel, numel))
    sqrt2 = math.sqrt(2.0)
    col = 1
    for l in range(1, n + 1):
        for k in range(1, l + 1):
            row = 1
            kl_eq = k == l
            for j in range(1, n + 1):
                Ajl, Ajk = A[j - 1, l - 1], A[j - 1, k - 1]
                for i in range(1, j + 1):
                    if row > col:
                        break
                    ij_eq = i == j
                    if not ij_eq and not kl_eq:
                        out[row - 1, col - 1] = A[i - 1, k - 1] * Ajl + A[i - 1, l - 1] * Ajk
                    elif ij_eq and not kl_eq:
                        out[row - 1, col - 1] = sqrt2 * Ajl * Ajk
                    elif not ij_eq and kl_eq:
                        out[row - 1, col - 1] = sqrt2 * A[i - 1, l - 1] * Ajk
                    else:
                        out[row - 1, col - 1] = Ajl * Ajl
                    row += 1
            col += 1
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8])
def test_kernel_formula_matches_reference_skron(n):
    rng = np.random.default_rng(n)
    c = PSDTriangleCone(n)
    for _ in range(2):
        M = rng.standard_normal((n, n))
        S = M @ M.T + n * np.eye(n)
        M = rng.standard_normal((n, n))
        Z = M @ M.T + n * np.eye(n)
        assert c.update_scaling(c.mat_to_svec(S), c.mat_to_svec(Z), 1.0)
        nent = c.numel * (c.numel + 1) // 2
        host = np.zeros(nent)
        c.get_Hs(host)
        W = c.RRt
        ref = _reference_skron_loop(W)[c._hs_r, c._hs_c]          # pack_triu (mathutils.jl:402-412)
        assert np.array_equal(host, ref)                          # vectorised stand-in == the reference's loop
        s2 = math.sqrt(2.0)
        dev = np.zeros(nent)
        for e in range(nent):                                     # k_psd_hs: same index arithmetic and case split
            b = _tri_root(e)
            a = e - b * (b + 1) // 2
            j = _tri_root(a)
            i = a - j * (j + 1) // 2
            l = _tri_root(b)
            k = b - l * (l + 1) // 2
            if i != j and k != l:
                dev[e] = W[i, k] * W[j, l] + W[i, l] * W[j, k]
            elif i == j and k != l:
                dev[e] = (s2 * W[j, l]) * W[j, k]
            elif i != j:
                dev[e] = (s2 * W[i, l]) * W[j, k]
            else:
                dev[e] = W[j, l] * W[j, l]
        assert np.array_equal(dev, ref)


def test_identity_scaling_block_is_exact_identity():
    c = PSDTriangleCone(4)
    c.set_identity_scaling()
    blk = np.zeros(c.numel * (c.numel + 1) // 2)
    c.get_Hs(blk)
    full = np.zeros((c.numel, c.numel))
    full[c._hs_r, c._hs_c] = blk
    assert np.array_equal(full, np.eye(c.numel))
