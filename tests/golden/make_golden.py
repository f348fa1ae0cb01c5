#!/usr/bin/env python
"""Writes tests/golden/reference_known_answers.json.

The reference is Julia and cannot be executed in this image (no `julia`, QDLDL.jl/AMD.jl not vendored),
so the golden vectors are of two kinds, kept apart in the file:

  "reference": literal problem data and literal expected answers transcribed from the reference's own
               test files (file:line cited per entry) -- these pin the ORACLE and the HIP path end-to-end
               at the reference's tolerance;
  "layout"   : the KKT image hand-derived from directldl_kkt_assembly.jl + csc_assembly.jl for the QP
               fixture -- pins assembly order and every LDLDataMap index;
  "oracle"   : outputs of THIS repository's oracle on the same fixtures (iterations, x, objective) --
               regression vectors that travel to the GPU box; they are NOT reference outputs.

Run:  python tests/golden/make_golden.py [out.json]     (from the repo root; needs oracle/liboracle_kkt.so;
      default output = tests/golden/reference_known_answers.json; tests/test_golden_file.py regenerates the file into a
      temporary directory and compares it with the committed one)
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import clarabel_jl_amd  # noqa: E402,F401  (registers the dotted package directory)
import julia_standin as cl  # noqa: E402  (the numpy stand-in of the Julia caller: Solver, Settings)
from oracle.kkt_oracle import OracleKKTSolver  # noqa: E402
from tests import fixtures as fx  # noqa: E402

REFERENCE = [
    # name, fixture, status, x, obj, tol, citation
    ("qp", fx.basic_qp, "SOLVED", [0.3, 0.7], 1.8800000298331538, 1e-3, "test/OptTests/basic_qp.jl:6-19,62-73; linear_solvers.jl:17-28"),
    ("qp_univariate", fx.univariate_qp, "SOLVED", [0.0], 0.0, 1e-3, "test/OptTests/basic_qp.jl:44-60"),
    ("qp_dual_infeasible", fx.basic_qp_dualinf, "DUAL_INFEASIBLE", None, None, 1e-3, "test/OptTests/basic_qp.jl:90-99"),
    ("lp", fx.basic_lp, "SOLVED", [-0.5, 0.5, -0.5], -3.0, 1e-3, "test/OptTests/basic_lp.jl:6-16,26-37"),
    ("eq_constrained_1", lambda: fx.eq_constrained(1), "SOLVED", [0.0, 1.0, 1.0], None, 1e-3, "test/OptTests/basic_eq_constrained.jl:16-29"),
    ("eq_constrained_2", lambda: fx.eq_constrained(2), "SOLVED", [10 / 6, 1 / 6, 1 / 6], None, 1e-3, "test/OptTests/basic_eq_constrained.jl:31-42"),
    ("unconstrained", fx.unconstrained, "SOLVED", [-1.0, -2.0, 3.0], None, 1e-3, "test/OptTests/basic_unconstrained.jl:16-26"),
    ("socp", fx.basic_socp, "SOLVED", [-0.5, 0.435603, -0.245459], -8.4590e-01, 1e-3, "test/OptTests/basic_socp.jl:6-30,41-53; linear_solvers.jl:30-45"),
    ("sdp", fx.basic_sdp, "SOLVED", [-3.0729833267361095, 0.3696004167288786, -0.022226685581313674,
                                     0.31441213129613066, -0.026739700851545107, -0.016084530571308823],
     4.840076866013861, 1e-3, "test/OptTests/basic_sdp.jl:6-20,31-48; linear_solvers.jl:47-67"),
    # the non-symmetric cones: 3 x 3 dense Hs blocks (Exponential, Power) and the rank-3 expansion (Generalized Power)
    ("exp", fx.basic_exp, "SOLVED", [-9.425995201329599, 4.828561507482018, 14.59743362204262, 1.0000012112102774,
                                     7.65314081561849, -29.99999978458479, -0.0],
     -54.41243965302268, 1e-3, "test/OptTests/basic_exp.jl:6-37,54-71"),
    ("pow", fx.basic_pow, "SOLVED", None, -1.8458, 1e-3, "test/OptTests/basic_pow.jl:6-39,56-62"),
    ("genpow", fx.basic_genpow, "SOLVED", None, -1.8458, 1e-3, "test/OptTests/basic_genpow.jl:7-33,50-56"),
]

# SURVEY.md Appendix B (1-based, exactly as Julia would hold them)
LAYOUT_QP = dict(
    citation="src/kktsolvers/direct-ldl/directldl_kkt_assembly.jl:15-175, src/utils/csc_assembly.jl:19-260",
    N=8, nnzKKT=17,
    colptr=[1, 2, 4, 7, 9, 11, 14, 16, 18],
    rowval=[1, 1, 2, 1, 2, 3, 1, 4, 2, 5, 1, 2, 6, 1, 7, 2, 8],
    map_P=[1, 2, 3], map_A=[4, 7, 11, 14, 5, 9, 12, 16], map_Hsblocks=[6, 8, 10, 13, 15, 17],
    map_diagP=[1, 3], map_diag_full=[1, 3, 6, 8, 10, 13, 15, 17], Dsigns=[1, 1, -1, -1, -1, -1, -1, -1])


def dump_problem(prob):
    P, q, A, b, cones = prob
    P = sp.csc_matrix(P)
    A = sp.csc_matrix(A)
    return dict(n=int(P.shape[0]), m=int(A.shape[0]),
                P=dict(colptr=P.indptr.tolist(), rowval=P.indices.tolist(), nzval=P.data.tolist()),
                A=dict(colptr=A.indptr.tolist(), rowval=A.indices.tolist(), nzval=A.data.tolist()),
                q=np.asarray(q, float).tolist(), b=np.asarray(b, float).tolist(),
                cones=[dump_cone(c) for c in cones])


def dump_cone(c):
    """[type name, field] with the field the reference's JSON form carries (json.jl:142-158): dim; alpha; (); [alpha, dim2]"""
    name = type(c).__name__
    if name == "ExponentialConeT":
        return [name, []]
    if name == "PowerConeT":
        return [name, float(c.alpha)]
    if name == "GenPowerConeT":
        return [name, [list(c.alpha), int(c.dim2)]]
    return [name, int(c.dim)]


def build():
    out = dict(reference=[], layout=dict(qp=LAYOUT_QP), oracle=[])
    for name, mk, status, x, obj, tol, cite in REFERENCE:
        out["reference"].append(dict(name=name, citation=cite, problem=dump_problem(mk()), status=status, x=x, obj=obj, tol=tol))
        P, q, A, b, cones = mk()
        s = cl.Solver(P, q, A, b, cones, cl.Settings(),
                      kktsolver_factory=lambda *a: OracleKKTSolver(*a, ordering="natural"))
        sol = s.solve()
        # cross_order_tol: how far a run on ANOTHER elimination order may sit from this one.  1e-8 = the IPM's stopping tolerance for the
        # symmetric fixtures; the non-symmetric ones move by up to 3e-9 (objective, relative) between orders in the oracle itself
        # (tests/test_nonsymmetric_cones.py::test_spread_of_the_reference_arithmetic_between_elimination_orders holds it under 2.5e-8)
        out["oracle"].append(dict(name=name, ordering="natural", status=sol.status, iterations=int(sol.iterations),
                                  x=np.asarray(sol.x).tolist(),
                                  obj=None if np.isnan(sol.obj_val) else float(sol.obj_val),
                                  r_prim=float(sol.r_prim), r_dual=float(sol.r_dual),
                                  cross_order_tol=2.5e-8 if name in ("exp", "pow", "genpow") else 1e-8))
    return out


def main(path=None):
    out = build()
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_known_answers.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
