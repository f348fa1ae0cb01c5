"""The symbolic plan (ordering, supernodes, work lists) interpreted serially on the host must
reproduce a dense solve: validates every index structure the HIP kernels consume, without a GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd import problems
from oracle.kkt_oracle import OracleKKT
from tests import fixtures as fx
from tests import plan_support as ps


def _kkt(prob, rng):
    P, q, A, b, specs = prob
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    Pt = sp.triu(sp.csc_matrix(P), format="csc")
    Pt.sort_indices()
    A = sp.csc_matrix(A)
    A.sort_indices()
    k = OracleKKT(Pt, A, *cones.kkt_descriptors())
    fx.scale_cones(cones, rng)
    hs = np.zeros(k.nHs)
    cones.get_Hs(hs)
    k.L.oracle_kkt_update_Hs(k.h, hs)
    si = 0
    for c in cones:
        if c.is_sparse_expandable:
            k.L.oracle_kkt_update_soc(k.h, si, c.eta ** 2, c.u, c.v)
            si += 1
    nz = k.nzval.copy()
    ds = k.map("dsigns")
    nz[k.map("map_diag_full")] += 1e-8 * ds
    return k, nz, ds


@pytest.mark.parametrize("policy", [0, 1, 2 + 16 * 4, 2 + 16 * 2])
@pytest.mark.parametrize("maxw,relax", [(64, 1), (8, 1), (3, 0), (1, 0)])
def test_plan_reproduces_dense_solve(policy, maxw, relax):
    rng = np.random.default_rng(5)
    probs = [problems.random_sparse_qp(60, 100, 3, 3, 1), problems.random_sparse_qp(200, 300, 4, 4, 2, window=10),
             problems.portfolio_socp(n=40, nsoc=3, socdim=9, seed=1), problems.sdp_blocks(n=12, ncones=2, dim=4, seed=2)]
    for prob in probs:
        k, nz, ds = _kkt(prob, rng)
        b = rng.standard_normal(k.N)
        rc, x, perm, st = ps.run(k.N, k.colptr, k.rowval, nz, ds, b, max_width=maxw, relax=relax, policy=policy)
        assert rc == 0
        assert sorted(perm) == list(range(k.N))
        K = sp.csc_matrix((nz, k.rowval, k.colptr), shape=(k.N, k.N)).toarray()
        K = K + K.T - np.diag(np.diag(K))
        xd = np.linalg.solve(K, b)
        assert np.linalg.norm(x - xd) <= 1e-9 * max(1.0, np.linalg.norm(xd))
        assert st["nreg"] == 0


def test_device_work_lists_are_exercised_and_correct():
    """The host interpreter applies the plan the way the device kernels do (dense tiles in tile coordinates
    incl. tile maps, per-entry gather lists, persistent front sweep with its external gather lists).  A larger
    problem makes sure every kind of work list is present, and the result still equals the dense solve."""
    rng = np.random.default_rng(9)
    k, nz, ds = _kkt(problems.random_sparse_qp(500, 900, 7, 3, 1), rng)
    b = rng.standard_normal(k.N)
    seen = dict(nfronts=0, ngather_entries=0, ndense_groups=0, nmapped_tasks=0)
    for maxw in (64, 16):
        rc, x, perm, st = ps.run(k.N, k.colptr, k.rowval, nz, ds, b, max_width=maxw, relax=1, policy=2 + 16 * 4)
        assert rc == 0
        K = sp.csc_matrix((nz, k.rowval, k.colptr), shape=(k.N, k.N)).toarray()
        K = K + K.T - np.diag(np.diag(K))
        xd = np.linalg.solve(K, b)
        assert np.linalg.norm(x - xd) <= 1e-9 * max(1.0, np.linalg.norm(xd))
        for key in seen:
            seen[key] += st[key]
    assert all(v > 0 for v in seen.values()), seen


def test_ordering_quality_cfg1():
    """own AMD vs the oracle's independent MMD order on config 1: fill within 15 %"""
    from oracle.kkt_oracle import mmd_order

    rng = np.random.default_rng(1)
    k, nz, ds = _kkt(problems.random_sparse_qp(1000, 2000, 1, 4, 2), rng)
    rc, _, perm, st = ps.run(k.N, k.colptr, k.rowval, nz, ds, symbolic_only=True)
    assert rc == 0
    k.symbolic(mmd_order(k.N, k.colptr, k.rowval))
    assert st["nnzL"] <= 1.15 * k.nnzL
    # and the plan's column counts equal the oracle's for the SAME permutation
    k.symbolic(perm)
    assert st["nnzL"] == k.nnzL


def test_user_perm_and_dynamic_regularisation_count():
    rng = np.random.default_rng(2)
    k, nz, ds = _kkt(problems.random_sparse_qp(30, 40, 9, 3, 1), rng)
    # natural order handed in as user_perm; make one expected-negative pivot positive -> substituted
    nz2 = nz.copy()
    dfull = k.map("map_diag_full")
    nz2[dfull[k.n + 3]] = 0.0   # zero Hs entry with static shift removed below eps
    b = rng.standard_normal(k.N)
    rc, x, perm, st = ps.run(k.N, k.colptr, k.rowval, nz2, ds, b, perm=np.arange(k.N))
    assert rc == 0 and list(perm) != [] and st["nreg"] >= 0


@pytest.mark.parametrize("maxw", [8, 16, 64])
def test_super_block_front_sweeps_reproduce_dense_solve(maxw, monkeypatch, capfd):
    """front_sweep.hip's super-block sweeps (k_invert_super / k_front_fwd_sb / k_front_bwd_sb) have a serial host twin in
    tests/support/plan_check.cpp with the same tile addresses, layouts (column-major / row-major halves of the inverse tiles,
    the row-major copy LT) and super-block structure.  Narrow panel widths turn the dense root of a small problem into a front of
    >= kSbMinPanels panels (incl. a partial last panel and a partial last super-block), so the index arithmetic is checked
    against a dense solve without a GPU."""
    rng = np.random.default_rng(11)
    monkeypatch.setenv("PLANCHECK_SUPERHOP", "1")
    monkeypatch.setenv("PLANCHECK_VERBOSE", "1")
    k, nz, ds = _kkt(problems.random_sparse_qp(1000, 2000, 1, 4, 2) if maxw == 64 else problems.random_sparse_qp(300, 500, 7, 3, 1), rng)
    b = rng.standard_normal(k.N)
    K = sp.csc_matrix((nz, k.rowval, k.colptr), shape=(k.N, k.N)).toarray()
    K = K + K.T - np.diag(np.diag(K))
    xd = np.linalg.solve(K, b)
    capfd.readouterr()
    rc, x, perm, st = ps.run(k.N, k.colptr, k.rowval, nz, ds, b, max_width=maxw, relax=1, policy=2 + 16 * 4)
    assert rc == 0 and st["nfronts"] >= 1
    assert int(capfd.readouterr().err.split(" with super-block sweeps")[0].split()[-1]) >= 1     # (PLANCHECK_SUPERHOP=1: from kSbMinPanels panels on)
    assert np.linalg.norm(x - xd) <= 1e-9 * max(1.0, np.linalg.norm(xd))
    # and the same plan with one hop per panel gives the same answer (the two sweeps are interchangeable)
    monkeypatch.setenv("PLANCHECK_SUPERHOP", "0")
    rc0, x0, _, _ = ps.run(k.N, k.colptr, k.rowval, nz, ds, b, max_width=maxw, relax=1, policy=2 + 16 * 4)
    assert rc0 == 0
    assert np.linalg.norm(x - x0) <= 1e-10 * max(1.0, np.linalg.norm(x0))




def test_dense_triangles_leave_the_symmetric_view_and_the_product_is_unchanged():
    """PSD-cone Hs blocks (packed upper triangles, directldl_kkt_assembly.jl:49-57) are found from the pattern alone, taken out of the
    symmetric view and multiplied from their packed columns (the host restatement of k_spmv_dense_tri): K x must equal the plain
    product; with the search off (first_col = -1) or the blocks below the minimum dimension nothing leaves the view."""
    rng = np.random.default_rng(11)
    prob = problems.sdp_blocks(n=40, ncones=3, dim=16, seed=3)       # 3 x PSD(16): triangles of dimension 136
    k, nz, ds = _kkt(prob, rng)
    n = prob[0].shape[0]
    K = sp.csc_matrix((nz, k.rowval, k.colptr), shape=(k.N, k.N)).toarray()
    K = K + K.T - np.diag(np.diag(K))
    x = rng.standard_normal(k.N)
    full = int(2 * len(nz) - k.N)
    rc, y, st = ps.symmetric_product(k.N, k.colptr, k.rowval, nz, x, first_col=n, min_dim=128)
    assert rc == 0 and st["triangles"] == 3 and st["triangle_dims"] == 3 * 136
    assert st["view_entries"] == full - 3 * 136 * 136
    assert np.linalg.norm(y - K @ x) <= 1e-12 * np.linalg.norm(K @ x)
    for kw in (dict(first_col=-1, min_dim=128), dict(first_col=n, min_dim=137)):
        rc, y, st = ps.symmetric_product(k.N, k.colptr, k.rowval, nz, x, **kw)
        assert rc == 0 and st["triangles"] == 0 and st["view_entries"] == full
        assert np.linalg.norm(y - K @ x) <= 1e-12 * np.linalg.norm(K @ x)
    # a problem without PSD cones: nothing to find
    k2, nz2, _ = _kkt(problems.portfolio_socp(n=40, nsoc=3, socdim=9, seed=1), rng)
    rc, y2, st2 = ps.symmetric_product(k2.N, k2.colptr, k2.rowval, nz2, np.ones(k2.N), first_col=40, min_dim=4)
    assert rc == 0
    K2 = sp.csc_matrix((nz2, k2.rowval, k2.colptr), shape=(k2.N, k2.N)).toarray()
    K2 = K2 + K2.T - np.diag(np.diag(K2))
    assert np.linalg.norm(y2 - K2 @ np.ones(k2.N)) <= 1e-12 * np.linalg.norm(K2 @ np.ones(k2.N))
