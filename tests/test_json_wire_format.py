"""the reference's JSON problem files (src/json.jl).  Mirrors test/UnitTests/test_json.jl
(write, reload, solve both, compare x to 1e-10 and the status; reload with max_iter = 1 -> MAX_ITERATIONS) on the CPU
oracle chain, plus the schema itself (0-based CSC dictionaries, one-key cone dictionaries, +-Inf as floatmax)."""
import json
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd import jsonio


def _problem():  # test_json.jl:4-12
    P = sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 2.0]]))
    c = np.array([1.0, 1.0])
    A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]]))
    b = np.array([1.0, 1.0, 1.0])
    cones = [cl.NonnegativeConeT(1), cl.ZeroConeT(1), cl.NonnegativeConeT(1)]
    return P, c, A, b, cones


def test_write_reload_solve(tmp_path, oracle_factory):  # test_json.jl:14-27
    P, c, A, b, cones = _problem()
    f = str(tmp_path / "problem.json")
    jsonio.save_to_file(f, P, c, A, b, cones, cl.Settings())
    P2, c2, A2, b2, cones2, st2 = jsonio.load_from_file(f)
    s1 = cl.Solver(P, c, A, b, cones, cl.Settings(), kktsolver_factory=oracle_factory).solve()
    s2 = cl.Solver(P2, c2, A2, b2, cones2, st2, kktsolver_factory=oracle_factory).solve()
    assert np.allclose(s1.x, s2.x, atol=1e-10, rtol=0)
    assert s1.status == s2.status == "SOLVED"


def test_reload_with_custom_settings(tmp_path, oracle_factory):  # test_json.jl:30-35
    P, c, A, b, cones = _problem()
    f = str(tmp_path / "problem.json")
    jsonio.save_to_file(f, P, c, A, b, cones)
    st = cl.Settings()
    st.max_iter = 1
    P2, c2, A2, b2, cones2, st2 = jsonio.load_from_file(f, st)
    assert st2 is st
    s3 = cl.Solver(P2, c2, A2, b2, cones2, st2, kktsolver_factory=oracle_factory).solve()
    assert s3.status == "MAX_ITERATIONS"


def test_schema(tmp_path):  # json.jl:118-158
    P, c, A, b, _ = _problem()
    cones = [cl.ZeroConeT(1), cl.NonnegativeConeT(2), cl.SecondOrderConeT(3), cl.PSDTriangleConeT(2)]
    st = cl.Settings()
    assert st.time_limit == float("inf")
    f = str(tmp_path / "p.json")
    jsonio.save_to_file(f, P, c, A, b, cones, st)
    d = json.load(open(f))
    assert list(d) == ["settings", "P", "q", "A", "b", "cones"]
    assert d["A"] == {"m": 3, "n": 2, "colptr": [0, 2, 4], "rowval": [0, 1, 0, 2], "nzval": [1.0, 1.0, 1.0, 1.0]}
    assert d["cones"] == [{"ZeroConeT": 1}, {"NonnegativeConeT": 2}, {"SecondOrderConeT": 3}, {"PSDTriangleConeT": 2}]
    assert d["settings"]["time_limit"] == sys.float_info.max            # sanitize_settings!, json.jl:87-97
    P2, c2, A2, b2, cones2, st2 = jsonio.load_from_file(f)
    assert (P2 != sp.csc_matrix(P)).nnz == 0 and (A2 != A).nnz == 0
    assert np.array_equal(c2, c) and np.array_equal(b2, b) and cones2 == cones
    assert st2.time_limit == float("inf") and st2.max_iter == st.max_iter  # desanitize_settings!, :100-110


def test_foreign_file(tmp_path):
    """a file as the reference writes it: settings fields this mirror does not carry are kept, non-symmetric cones are
    refused with a clear message"""
    d = {"settings": {"max_iter": 50, "direct_kkt_solver": True, "time_limit": sys.float_info.max},
         "P": {"m": 1, "n": 1, "colptr": [0, 1], "rowval": [0], "nzval": [2.0]}, "q": [1.0],
         "A": {"m": 1, "n": 1, "colptr": [0, 1], "rowval": [0], "nzval": [1.0]}, "b": [1.0],
         "cones": [{"NonnegativeConeT": 1}]}
    f = str(tmp_path / "ref.json")
    json.dump(d, open(f, "w"))
    P, q, A, b, cones, st = jsonio.load_from_file(f)
    assert st.max_iter == 50 and st.extra == {"direct_kkt_solver": True} and st.time_limit == float("inf")
    assert cones == [cl.NonnegativeConeT(1)] and P[0, 0] == 2.0
    d["cones"] = [{"ExponentialConeT": []}]                      # json.jl:201-202 (non-symmetric cones: tests/test_nonsymmetric_cones.py)
    json.dump(d, open(f, "w"))
    assert jsonio.load_from_file(f)[4] == [cl.ExponentialConeT()]
    d["cones"] = [{"HyperbolicConeT": 3}]
    json.dump(d, open(f, "w"))
    with pytest.raises(ValueError):
        jsonio.load_from_file(f)
