"""Pin the ORACLE chain (stand-in IPM caller + CPU restatement of the :qdldl KKT path) on every
known answer the reference's own tests hold for this path.  tol = the reference's
own 1e-3 (1e-7 for data updating); our results are asserted tighter where the literal allows."""
import numpy as np
import pytest
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from tests import fixtures as fx

TOL = 1e-3


def solve(prob, fac, **kw):
    P, q, A, b, cones = prob
    s = cl.Solver(P, q, A, b, cones, cl.Settings(**kw), kktsolver_factory=fac)
    return s, s.solve()


def test_qp_feasible(oracle_factory):  # basic_qp.jl:62-73, linear_solvers.jl:17-28
    _, sol = solve(fx.basic_qp(), oracle_factory)
    assert sol.status == "SOLVED"
    assert np.linalg.norm(sol.x - [0.3, 0.7]) < 1e-6
    assert abs(sol.obj_val - 1.8800000298331538) < 1e-6
    assert abs(sol.obj_val_dual - 1.8800000298331538) < 1e-6


def test_qp_primal_infeasible(oracle_factory):  # basic_qp.jl:75-88
    P, c, A, b, cones = fx.basic_qp()
    b[0] = -1.0
    b[3] = -1.0
    _, sol = solve((P, c, A, b, cones), oracle_factory)
    assert sol.status == "PRIMAL_INFEASIBLE"
    assert np.isnan(sol.obj_val) and np.isnan(sol.obj_val_dual)


def test_qp_dual_infeasible(oracle_factory):  # basic_qp.jl:90-99
    _, sol = solve(fx.basic_qp_dualinf(), oracle_factory)
    assert sol.status == "DUAL_INFEASIBLE"
    assert np.isnan(sol.obj_val)


def test_qp_dual_infeasible_nonqsd(oracle_factory):  # basic_qp.jl:101-115
    P, c, A, b, _ = fx.basic_qp_dualinf()
    _, sol = solve((P, c, A[0:1, :], b[0:1], [cl.NonnegativeConeT(1)]), oracle_factory)
    assert sol.status == "DUAL_INFEASIBLE"


def test_qp_univariate(oracle_factory):  # basic_qp.jl:44-60
    _, sol = solve(fx.univariate_qp(), oracle_factory)
    assert abs(sol.x[0]) < TOL and abs(sol.obj_val) < TOL and abs(sol.obj_val_dual) < TOL


def test_lp_feasible(oracle_factory):  # basic_lp.jl:26-37 (two-solve LP initialisation)
    _, sol = solve(fx.basic_lp(), oracle_factory)
    assert sol.status == "SOLVED"
    assert np.linalg.norm(sol.x - [-0.5, 0.5, -0.5]) < 1e-6
    assert abs(sol.obj_val + 3.0) < 1e-6 and abs(sol.obj_val_dual + 3.0) < 1e-6


def test_lp_primal_infeasible(oracle_factory):  # basic_lp.jl:39-52
    P, c, A, b, cones = fx.basic_lp()
    b[0] = -1
    b[3] = -1
    _, sol = solve((P, c, A, b, cones), oracle_factory)
    assert sol.status == "PRIMAL_INFEASIBLE"


def test_lp_dual_infeasible(oracle_factory):  # basic_lp.jl:54-67
    P, c, A, b, cones = fx.basic_lp()
    A = A.tolil()
    A[3, 0] = 1.0
    c = np.array([1.0, 0.0, 0.0])
    _, sol = solve((P, c, sp.csc_matrix(A), b, cones), oracle_factory)
    assert sol.status == "DUAL_INFEASIBLE"


@pytest.mark.parametrize("variant,xref", [(1, [0.0, 1.0, 1.0]), (2, [10 / 6, 1 / 6, 1 / 6])])
def test_eq_constrained(oracle_factory, variant, xref):  # basic_eq_constrained.jl:16-42
    _, sol = solve(fx.eq_constrained(variant), oracle_factory)
    assert sol.status == "SOLVED"
    assert np.linalg.norm(sol.x - xref) < 1e-6


def test_eq_constrained_redundant_rows(oracle_factory):  # basic_eq_constrained.jl:44-60
    P, c, A, b, cones = fx.eq_constrained(1)
    _, sol = solve((P, c, sp.vstack([A, A]).tocsc(), np.concatenate([b, b]), cones + cones), oracle_factory)
    assert sol.status == "SOLVED"
    assert np.linalg.norm(sol.x - [0.0, 1.0, 1.0]) < TOL


def test_unconstrained(oracle_factory):  # basic_unconstrained.jl:16-26
    _, sol = solve(fx.unconstrained(), oracle_factory)
    assert sol.status == "SOLVED"
    assert np.linalg.norm(sol.x - [-1.0, -2.0, 3.0]) < 1e-6


def test_socp_dense_soc(oracle_factory):  # basic_socp.jl:41-53, linear_solvers.jl:30-45 (SOC dim 3: dense Hs path)
    _, sol = solve(fx.basic_socp(), oracle_factory)
    assert sol.status == "SOLVED"
    assert np.linalg.norm(sol.x - [-0.5, 0.435603, -0.245459]) < 5e-5  # literal has 6 digits
    assert abs(sol.obj_val + 8.4590e-01) < 1e-4 and abs(sol.obj_val_dual + 8.4590e-01) < 1e-4


def test_socp_infeasible(oracle_factory):  # basic_socp.jl:71-84
    P, c, A, b, cones = fx.basic_socp()
    b[6] = -10.0
    _, sol = solve((P, c, A, b, cones), oracle_factory)
    assert sol.status == "PRIMAL_INFEASIBLE"


def test_socp_lasso_sparse_expansion(oracle_factory):  # socp-lasso.jl:56-65 (status only)
    s, sol = solve(fx.lasso_socp(), oracle_factory)
    assert s.kktsystem.kktsolver.p == 2  # the rank-2 expansion columns are present
    assert sol.status == "SOLVED"


def test_sdp(oracle_factory):  # basic_sdp.jl:31-48, linear_solvers.jl:47-67
    _, sol = solve(fx.basic_sdp(), oracle_factory)
    refsol = [-3.0729833267361095, 0.3696004167288786, -0.022226685581313674,
              0.31441213129613066, -0.026739700851545107, -0.016084530571308823]
    assert sol.status == "SOLVED"
    assert np.linalg.norm(sol.x - refsol) < 1e-6
    assert abs(sol.obj_val - 4.840076866013861) < 1e-6


def test_data_updating_P_and_A(oracle_factory):  # data_updating.jl:32-169, tol 1e-7
    P, q, A, b, cones = fx.updating_data()
    s1, _ = solve((P, q, A, b, cones), oracle_factory)
    P2 = P.tolil()
    P2[0, 0] = 100.0
    P2 = sp.csc_matrix(P2)
    s1.update_P(sp.triu(P2, format="csc").data)
    sol1 = s1.solve()
    _, sol2 = solve((P2, q, A, b, cones), oracle_factory)
    assert np.linalg.norm(sol1.x - sol2.x) < 1e-7
    A2 = A.copy()
    A2.data[1] = -1000.0  # data_updating.jl:78-97 changes A[2,2]; any in-pattern change will do
    s1.update_A(A2.data)
    sol1 = s1.solve()
    _, sol3 = solve((P2, q, A2, b, cones), oracle_factory)
    assert np.linalg.norm(sol1.x - sol3.x) < 1e-7
    q2 = np.array([1000.0, 1.0])
    b2 = np.array([0.0, 1.0, 1.0, 1.0])
    s1.update_q(q2)
    s1.update_b(b2)
    sol1 = s1.solve()
    _, sol4 = solve((P2, q2, A2, b2, cones), oracle_factory)
    assert np.linalg.norm(sol1.x - sol4.x) < 1e-7
