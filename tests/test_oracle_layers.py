"""Layer-level pins of the oracle below the end-to-end answers (the reference has none there,
SURVEY.md §4): KKT layout vs the hand-derived Appendix B example, assembly vs an independent scipy
construction, LDL factor/solve vs dense numpy, dynamic regularisation, IR stopping rules."""
import numpy as np
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from oracle.kkt_oracle import OracleKKT, OracleKKTSolver, mmd_order
from tests import fixtures as fx


def _kkt_for(P, A, specs):
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    P = sp.triu(sp.csc_matrix(P), format="csc")
    P.sort_indices()
    A = sp.csc_matrix(A)
    A.sort_indices()
    return OracleKKT(P, A, *cones.kkt_descriptors()), cones, P, A


def test_appendix_b_layout():
    """SURVEY.md Appendix B (derived from directldl_kkt_assembly.jl + csc_assembly.jl), 1-based there."""
    P, c, A, b, specs = fx.basic_qp()
    k, cones, _, _ = _kkt_for(P, A, specs)
    assert (k.N, k.n, k.m, k.p, k.nnzK) == (8, 2, 6, 0, 17)
    assert list(k.colptr + 1) == [1, 2, 4, 7, 9, 11, 14, 16, 18]
    assert list(k.map("map_P") + 1) == [1, 2, 3]
    assert list(k.map("map_A") + 1) == [4, 7, 11, 14, 5, 9, 12, 16]
    assert list(k.map("map_Hs") + 1) == [6, 8, 10, 13, 15, 17]
    assert list(k.map("map_diagP") + 1) == [1, 3]
    assert list(k.map("map_diag_full") + 1) == [1, 3, 6, 8, 10, 13, 15, 17]
    assert list(k.map("dsigns")) == [1, 1, -1, -1, -1, -1, -1, -1]
    rows = k.rowval + 1
    assert list(rows) == [1, 1, 2, 1, 2, 3, 1, 4, 2, 5, 1, 2, 6, 1, 7, 2, 8]


def test_genpow_expansion_layout_hand_derived():
    """GenPowExpansionMap (directldl_datamaps.jl:81-144) on a 2-variable problem with cones [NN(1), GenPow(dim1=2,
    dim2=1)]: the :triu image gets three extra columns -- q over the cone's first dim1 rows, r over its last dim2
    rows, p over all of them -- each closed by its diagonal entry D, with Dsigns (-1,-1,+1) (:98).  Layout derived
    by hand from _csc_colcount_sparsecone / _csc_fill_sparsecone (:101-144); 0-based here."""
    P = sp.csc_matrix(np.diag([1.0, 2.0]))
    A = sp.csc_matrix(np.array([[1.0, 2.0], [3.0, 0.0], [0.0, 4.0], [5.0, 6.0]]))
    numel = np.array([1, 3])
    k = OracleKKT(P, A, numel, np.array([0, 0], dtype=np.int32), np.array([0, 2], dtype=np.int32), np.array([0, 2]))
    assert (k.N, k.n, k.m, k.p, k.nnzK, k.nsparse) == (9, 2, 4, 3, 21, 1)
    assert list(k.colptr) == [0, 1, 2, 5, 7, 9, 12, 15, 17, 21]
    assert list(k.rowval) == [0, 1, 0, 1, 2, 0, 3, 1, 4, 0, 1, 5, 3, 4, 6, 5, 7, 3, 4, 5, 8]
    assert list(k.map("map_Hs")) == [4, 6, 8, 11]
    assert list(k.sparse_map(0, 0)) == [12, 13]          # q
    assert list(k.sparse_map(0, 1)) == [15]              # r
    assert list(k.sparse_map(0, 2)) == [17, 18, 19]      # p
    assert list(k.sparse_map(0, 3)) == [14, 16, 20]      # D
    assert list(k.map("dsigns")) == [1, 1, -1, -1, -1, -1, -1, -1, 1]
    # _csc_update_sparsecone(::GenPowerCone), :146-167: K[q] = q*(-sqrt(mu)) etc., D = (-1,-1,+1)
    pv, qv, rv = np.array([0.1, 0.2, 0.3]), np.array([1.5, -2.5]), np.array([4.0])
    k.L.oracle_kkt_update_genpow(k.h, 0, 3.0, pv, qv, rv)
    nz = k.nzval
    assert list(nz[[12, 13]]) == [-4.5, 7.5] and list(nz[[15]]) == [-12.0]
    assert np.array_equal(nz[[17, 18, 19]], pv * -3.0) and list(nz[[14, 16, 20]]) == [-1.0, -1.0, 1.0]


def _random_problem(rng, n, m_nn, soc_dims=(), psd_dims=(), zero=0, pdiag=True):
    specs = []
    if zero:
        specs.append(cl.ZeroConeT(zero))
    if m_nn:
        specs.append(cl.NonnegativeConeT(m_nn))
    for d in soc_dims:
        specs.append(cl.SecondOrderConeT(d))
    for d in psd_dims:
        specs.append(cl.PSDTriangleConeT(d))
    m = sum(cl.cones.nvars(s) for s in specs)
    A = sp.random(m, n, density=min(1.0, 4.0 / n), random_state=np.random.RandomState(rng.integers(1 << 30)),
                  format="csc")
    S = sp.random(n, n, density=min(1.0, 2.0 / n), random_state=np.random.RandomState(rng.integers(1 << 30)),
                  format="csc")
    P = S + S.T
    if pdiag:
        P = P + sp.diags(np.asarray(abs(P).sum(axis=1)).ravel() + 0.1)
    return sp.csc_matrix(P), sp.csc_matrix(A), specs


def _dense_kkt(k):
    K = sp.csc_matrix((k.nzval, k.rowval, k.colptr), shape=(k.N, k.N)).toarray()
    return K + K.T - np.diag(np.diag(K))


def test_assembly_matches_independent_construction():
    """K = [P A' ; A -Hs] (+ SOC expansion columns) rebuilt with scipy, after an Hs/SOC update."""
    rng = np.random.default_rng(7)
    P, A, specs = _random_problem(rng, 12, 9, soc_dims=(3, 7), psd_dims=(3,), zero=2, pdiag=False)
    k, cones, Pt, At = _kkt_for(P, A, specs)
    n, m = 12, A.shape[0]
    assert k.p == 2 and k.N == n + m + 2
    # every column has ascending rows and ends on the diagonal
    cp, rv = k.colptr, k.rowval
    for j in range(k.N):
        r = rv[cp[j]:cp[j + 1]]
        assert np.all(np.diff(r) > 0) and r[-1] == j
    k.symbolic(None)
    # random interior scaling on every cone
    s = np.zeros(m)
    z = np.zeros(m)
    for c, r in zip(cones.cones, cones.rng_cones):
        if isinstance(c, cl.cones.PSDTriangleCone):
            for v in (s, z):
                M = rng.standard_normal((c.n, c.n))
                v[r] = c.mat_to_svec(M @ M.T + c.n * np.eye(c.n))
        elif isinstance(c, cl.cones.SecondOrderCone):
            for v in (s, z):
                t = rng.standard_normal(c.dim)
                t[0] = np.linalg.norm(t[1:]) + 1.0 + rng.random()
                v[r] = t
        else:
            s[r] = rng.random(c.numel) + 0.5
            z[r] = rng.random(c.numel) + 0.5
    assert cones.update_scaling(s, z, 1.0)
    hs = np.zeros(cones.nnz_Hs)
    cones.get_Hs(hs)
    k.L.oracle_kkt_update_Hs(k.h, hs)
    soc7 = [c for c in cones.cones if getattr(c, "is_sparse_expandable", False)][0]
    k.L.oracle_kkt_update_soc(k.h, 0, soc7.eta ** 2, soc7.u, soc7.v)
    K = _dense_kkt(k)
    Pd = Pt.toarray()
    Pd = Pd + Pd.T - np.diag(np.diag(Pd))
    assert np.allclose(K[:n, :n], Pd)
    assert np.allclose(K[n:n + m, :n], At.toarray())
    # -Hs blocks: compare the action with mul_Hs on the non-expanded cones
    H = -K[n:n + m, n:n + m]
    for c, r in zip(cones.cones, cones.rng_cones):
        x = rng.standard_normal(c.numel)
        y = np.zeros(c.numel)
        c.mul_Hs(y, x, np.zeros(c.numel))
        if getattr(c, "is_sparse_expandable", False):
            # W'W = eta^2 (D + u u' - v v')  (test_coneops_secondordercone.jl:60-66)
            eta2 = c.eta ** 2
            D = np.ones(c.dim)
            D[0] = c.d
            assert np.allclose(eta2 * (D * x + c.u * (c.u @ x) - c.v * (c.v @ x)), y, rtol=1e-10, atol=1e-12)
            assert np.allclose(np.diag(H[r, r]), eta2 * D)
            vcol, ucol = K[n + r.start:n + r.stop, n + m], K[n + r.start:n + r.stop, n + m + 1]
            assert np.allclose(vcol, -eta2 * c.v) and np.allclose(ucol, -eta2 * c.u)
            assert np.isclose(K[n + m, n + m], -eta2) and np.isclose(K[n + m + 1, n + m + 1], eta2)
        else:
            assert np.allclose(H[r, r] @ x, y, rtol=1e-9, atol=1e-11)


def test_factor_solve_vs_dense_numpy():
    rng = np.random.default_rng(11)
    for trial in range(4):
        P, A, specs = _random_problem(rng, 40 + 10 * trial, 60, soc_dims=(6,), zero=3)
        k, cones, Pt, At = _kkt_for(P, A, specs)
        perm = mmd_order(k.N, k.colptr, k.rowval) if trial % 2 == 0 else None
        k.symbolic(perm)
        hs = rng.random(cones.nnz_Hs) + 0.1
        k.L.oracle_kkt_update_Hs(k.h, hs)
        soc = [c for c in cones.cones if getattr(c, "is_sparse_expandable", False)][0]
        soc.set_identity_scaling()
        k.L.oracle_kkt_update_soc(k.h, 0, 1.0, soc.u, soc.v)
        eps = __import__("ctypes").c_double(0)
        assert k.L.oracle_kkt_regularize_and_refactor(k.h, 1, 1e-8, 2.0 ** -104, eps)
        K = _dense_kkt(k)
        dsg = k.map("dsigns")
        Kreg = K + eps.value * np.diag(dsg.astype(float))
        b = rng.standard_normal(k.N)
        x = k.ldl_solve(b)
        xd = np.linalg.solve(Kreg, b)
        assert np.linalg.norm(x - xd) <= 1e-9 * max(1.0, np.linalg.norm(xd))
        assert np.allclose(k.symv(b), K @ b, rtol=1e-13, atol=1e-13)


def test_dynamic_regularization_and_ir_rules():
    """zero cone => -Hs block is 0, quasidefiniteness comes from static eps only; with static
    regularisation off the pivots hit the dynamic rule D = delta*sign."""
    P, c, A, b, specs = fx.eq_constrained(1)
    st = cl.Settings(static_regularization_enable=False)
    cones = cl.CompositeCone(specs)
    Pt = sp.triu(P, format="csc")
    ks = OracleKKTSolver(Pt, sp.csc_matrix(A), cones, 2, 3, st, ordering="natural")
    cones.set_identity_scaling()
    assert ks.kktsolver_update(cones)
    assert ks.k.L.oracle_kkt_nreg(ks.k.h) == 0  # A' D^-1 A pivots are negative as expected
    # duplicate constraint rows -> exactly singular (2,2) block -> one pivot is substituted
    A2 = sp.vstack([A, A]).tocsc()
    cones2 = cl.CompositeCone([cl.ZeroConeT(4)])
    ks2 = OracleKKTSolver(Pt, A2, cones2, 4, 3, st, ordering="natural")
    assert ks2.kktsolver_update(cones2)
    assert ks2.k.L.oracle_kkt_nreg(ks2.k.h) == 2
    # IR: with default settings the refined solution satisfies the reference's stopping rule
    st = cl.Settings()
    ks3 = OracleKKTSolver(Pt, A2, cones2, 4, 3, st, ordering="natural")
    assert ks3.kktsolver_update(cones2)
    rx, rz = np.array([1.0, -2.0, 0.5]), np.array([2.0, 0.0, 2.0, 0.0])
    ks3.kktsolver_setrhs(rx, rz)
    x, z = np.zeros(3), np.zeros(4)
    assert ks3.kktsolver_solve(x, z)
    assert ks3.last_ir_steps <= st.iterative_refinement_max_iter
    K = _dense_kkt(ks3.k)
    res = np.concatenate([rx, rz]) - K @ np.concatenate([x, z])
    assert np.max(np.abs(res)) < 1e-6
