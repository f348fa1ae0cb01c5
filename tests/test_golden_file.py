"""Committed golden vectors (tests/golden/reference_known_answers.json, made by tests/golden/make_golden.py):
  * "reference" = literal data + literal answers of the reference's own tests (citations inside the file)
  * "layout"    = hand-derived KKT image of the QP fixture, 1-based like Julia
  * "oracle"    = this repo's oracle outputs (regression only)
CPU tests check the oracle against all three; the gpu test checks the HIP path against the same file."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_known_answers.json")) as f:
    GOLD = json.load(f)

CONE_TYPES = {"ZeroConeT": cl.ZeroConeT, "NonnegativeConeT": cl.NonnegativeConeT,
              "SecondOrderConeT": cl.SecondOrderConeT, "PSDTriangleConeT": cl.PSDTriangleConeT,
              "ExponentialConeT": lambda _: cl.ExponentialConeT(), "PowerConeT": cl.PowerConeT,
              "GenPowerConeT": lambda v: cl.GenPowerConeT(v[0], v[1])}


def load_problem(p):
    n, m = p["n"], p["m"]
    P = sp.csc_matrix((p["P"]["nzval"], p["P"]["rowval"], p["P"]["colptr"]), shape=(n, n))
    A = sp.csc_matrix((p["A"]["nzval"], p["A"]["rowval"], p["A"]["colptr"]), shape=(m, n))
    cones = [CONE_TYPES[t](d) for t, d in p["cones"]]
    return P, np.array(p["q"]), A, np.array(p["b"]), cones


def check_reference_entry(e, sol):
    assert sol.status == e["status"], e["name"]
    if e["x"] is not None:
        assert np.linalg.norm(sol.x - np.array(e["x"])) < e["tol"], e["name"]
    if e["obj"] is not None:
        assert abs(sol.obj_val - e["obj"]) < e["tol"], e["name"]


@pytest.mark.parametrize("e", GOLD["reference"], ids=lambda e: e["name"])
def test_oracle_meets_reference_known_answers(e, oracle_factory):
    P, q, A, b, cones = load_problem(e["problem"])
    sol = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=oracle_factory).solve()
    check_reference_entry(e, sol)


def test_oracle_layout_matches_hand_derived_image(oracle_factory):
    """bit-exact: assembly order + every LDLDataMap index (the file is 1-based, the oracle 0-based)"""
    L = GOLD["layout"]["qp"]
    P, q, A, b, specs = load_problem(GOLD["reference"][0]["problem"])
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    Pt = sp.triu(P, format="csc")
    o = oracle_factory(Pt, sp.csc_matrix(A), cones, A.shape[0], P.shape[0], cl.Settings(), ordering="natural").k
    assert (o.N, o.nnzK) == (L["N"], L["nnzKKT"])
    assert np.array_equal(o.colptr + 1, L["colptr"]) and np.array_equal(o.rowval + 1, L["rowval"])
    for key, name in [("map_P", "map_P"), ("map_A", "map_A"), ("map_Hsblocks", "map_Hs"),
                      ("map_diagP", "map_diagP"), ("map_diag_full", "map_diag_full")]:
        assert np.array_equal(o.map(name) + 1, L[key]), key
    assert np.array_equal(o.map("dsigns"), L["Dsigns"])


@pytest.mark.parametrize("e", GOLD["oracle"], ids=lambda e: e["name"])
def test_oracle_regression_vectors(e, oracle_factory):
    ref = next(r for r in GOLD["reference"] if r["name"] == e["name"])
    P, q, A, b, cones = load_problem(ref["problem"])
    sol = cl.Solver(P, q, A, b, cones, cl.Settings(),
                    kktsolver_factory=lambda *a: oracle_factory(*a, ordering=e["ordering"])).solve()
    assert sol.status == e["status"] and sol.iterations == e["iterations"]
    if e["obj"] is not None:
        assert abs(sol.obj_val - e["obj"]) <= 1e-10 * max(1.0, abs(e["obj"]))
        assert np.max(np.abs(sol.x - np.array(e["x"]))) <= 1e-8 * max(1.0, np.max(np.abs(e["x"])))


@pytest.mark.gpu
@pytest.mark.parametrize("e", GOLD["reference"], ids=lambda e: e["name"])
def test_hip_path_meets_golden_file(e):
    """the HIP KKT path against the committed vectors only (nothing from oracle/ is executed here)"""
    from clarabel_jl_amd.kktsolver import HipKKTSolver

    P, q, A, b, cones = load_problem(e["problem"])
    s = cl.Solver(P, q, A, b, cones, cl.Settings())
    assert isinstance(s.kktsystem.kktsolver, HipKKTSolver)
    sol = s.solve()
    check_reference_entry(e, sol)
    # regression against the oracle's stored run.  The stored run used the NATURAL elimination order, the
    # product uses its own fill-reducing order, so the two IPM runs differ by rounding amplified through the
    # iterations: they agree to the IPM's own stopping tolerance (1e-8), not to the same-order parity bound
    # (1e-10, enforced in tests/test_gpu_kkt.py where the oracle is handed the product's permutation).
    o = next(r for r in GOLD["oracle"] if r["name"] == e["name"])
    assert sol.iterations == o["iterations"]
    tol = o.get("cross_order_tol", 1e-8)          # (the non-symmetric fixtures: the oracle's own measured spread between orders, see make_golden.py)
    if o["obj"] is not None:
        assert abs(sol.obj_val - o["obj"]) <= tol * max(1.0, abs(o["obj"]))
        assert abs(sol.r_prim - o["r_prim"]) <= tol and abs(sol.r_dual - o["r_dual"]) <= tol


@pytest.mark.gpu
def test_hip_layout_matches_hand_derived_image():
    from clarabel_jl_amd.kktsolver import HipKKTSolver

    L = GOLD["layout"]["qp"]
    P, q, A, b, specs = load_problem(GOLD["reference"][0]["problem"])
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    hk = HipKKTSolver(sp.triu(P, format="csc"), sp.csc_matrix(A), cones, A.shape[0], P.shape[0], cl.Settings())
    colptr, rowval, _ = hk.h.kkt()
    assert np.array_equal(colptr + 1, L["colptr"]) and np.array_equal(rowval + 1, L["rowval"])
    for w, key in enumerate(["map_P", "map_A", "map_Hsblocks", "map_diagP", "map_diag_full"]):
        assert np.array_equal(hk.h.map(w) + 1, L[key]), key
    assert np.array_equal(hk.h.dsigns(), L["Dsigns"])


def test_golden_file_regenerates(tmp_path):
    """The committed recipe (tests/golden/make_golden.py) still runs and reproduces the committed file: the "reference" and
    "layout" sections exactly (they are literals of the reference's tests / SURVEY App. B), the "oracle" section up to
    rounding (status and iteration counts exactly, x and objective to 1e-12: BLAS / libm builds may differ in the last bits)."""
    import subprocess
    import sys

    out = tmp_path / "regen.json"
    root = os.path.dirname(HERE)
    subprocess.check_call([sys.executable, os.path.join(HERE, "golden", "make_golden.py"), str(out)], cwd=root)
    with open(out) as f:
        new = json.load(f)
    assert new["reference"] == GOLD["reference"]
    assert new["layout"] == GOLD["layout"]
    assert [e["name"] for e in new["oracle"]] == [e["name"] for e in GOLD["oracle"]]
    for a, c in zip(new["oracle"], GOLD["oracle"]):
        assert a["status"] == c["status"] and a["iterations"] == c["iterations"] and a["ordering"] == c["ordering"], a["name"]
        assert np.max(np.abs(np.array(a["x"]) - np.array(c["x"]))) <= 1e-12 * max(1.0, np.max(np.abs(c["x"]))), a["name"]
        if c["obj"] is None:
            assert a["obj"] is None
        else:
            assert abs(a["obj"] - c["obj"]) <= 1e-12 * max(1.0, abs(c["obj"])), a["name"]
