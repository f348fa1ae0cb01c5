"""BASELINE.json's configs at FULL size on the GPU: against the CPU oracle on the same permutation
(test_full_size_matches_oracle) and through size-independent properties:
  * the refined solve satisfies the reference's own stopping rule against the TRUE (unregularised) K,
    recomputed on the host with scipy from the library's resident image;
  * linearity of the factorisation's solve operator;
  * determinism: refactor + solve twice -> bit-identical results;
  * the end-to-end IPM ends SOLVED with consistent primal/dual objectives.
Plus the batch config (cfg 4): a sample of the 256 seeded problems against the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver
from tests.fixtures import scale_cones, scale_cones_late

pytestmark = pytest.mark.gpu

FULL = {
    "cfg2a": lambda: problems.random_sparse_qp(10000, 20000, 2, 3, 1),
    "cfg2b": lambda: problems.random_sparse_qp(10000, 20000, 2, 4, 2, window=50),
    "cfg3": lambda: problems.portfolio_socp(seed=3),
    "cfg5": lambda: problems.sdp_blocks(seed=5),
}


def _prep(prob):
    P, q, A, b, specs = prob
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    Pt = sp.triu(sp.csc_matrix(P), format="csc")
    Pt.sort_indices()
    A = sp.csc_matrix(A)
    A.sort_indices()
    return Pt, A, cones


def _sym_K(h):
    colptr, rowval, nzval = h.kkt()
    U = sp.csc_matrix((nzval, rowval, colptr), shape=(h.N, h.N))
    return U + U.T - sp.diags(U.diagonal())


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_solve_properties(name):
    rng = np.random.default_rng(11)
    Pt, A, cones = _prep(FULL[name]())
    m, n = A.shape
    st = cl.Settings()
    hk = HipKKTSolver(Pt, A, cones, m, n, st)
    scale_cones(cones, rng)
    assert hk.kktsolver_update(cones)
    K = _sym_K(hk.h)
    N, p = hk.h.N, hk.h.p
    # refined solve vs the true K (reference stopping rule: abstol 1e-12 + reltol 1e-13 * |b|, or no further
    # improvement by the stop ratio; 1e-9 relative is the bound asserted at every size)
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    lx, lz = np.zeros(n), np.zeros(m)
    hk.kktsolver_setrhs(rx, rz)
    assert hk.kktsolver_solve(lx, lz)
    if p == 0:
        b = np.concatenate([rx, rz])
        res = b - K @ np.concatenate([lx, lz])
        assert np.max(np.abs(res)) <= 1e-9 * max(1.0, np.max(np.abs(b)))
    # linearity of x = K_fact^-1 b
    b1, b2 = rng.standard_normal(N), rng.standard_normal(N)
    x1, x2, x12 = hk.h.ldl_solve(b1), hk.h.ldl_solve(b2), hk.h.ldl_solve(2.0 * b1 - 3.0 * b2)
    scale = max(1.0, np.max(np.abs(x1)), np.max(np.abs(x2)))
    assert np.max(np.abs(x12 - (2.0 * x1 - 3.0 * x2))) <= 1e-8 * scale
    # K_fact x = b up to the static regulariser: residual against K + eps*diag(signs)
    eps = hk.diagonal_regularizer
    Kf = K + sp.diags(eps * hk.h.dsigns().astype(float))
    assert np.max(np.abs(Kf @ x1 - b1)) <= 1e-7 * max(1.0, np.max(np.abs(x1)) * abs(K).sum(axis=1).max())
    # determinism
    assert hk.kktsolver_update(cones)
    assert np.array_equal(hk.h.ldl_solve(b1), x1)


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_matches_oracle(name, oracle_factory):
    """BASELINE.json's configs at FULL size against the CPU oracle on the same permutation (one scalar QDLDL
    factorisation each: ~0.1 s cfg 2b, ~7 s cfg 3, ~20 s cfg 5, ~25 s cfg 2a on one host core): resident K bit for bit,
    regulariser and the number of dynamically regularised pivots equal, unrefined LDL solve to 1e-9, refined solve
    (kktsolver_solve!) to 1e-10."""
    rng = np.random.default_rng(23)
    Pt, A, cones = _prep(FULL[name]())
    m, n = A.shape
    st = cl.Settings()
    hk = HipKKTSolver(Pt, A, cones, m, n, st)
    ok_ = oracle_factory(Pt, A, cones, m, n, st, ordering=hk.h.perm())
    o = ok_.k
    assert hk.h.nnzL == o.nnzL
    scale_cones(cones, rng)
    assert hk.kktsolver_update(cones) and ok_.kktsolver_update(cones)
    assert np.array_equal(hk.h.kkt()[2], o.nzval)
    assert abs(hk.diagonal_regularizer - ok_.diagonal_regularizer) <= 1e-16 * max(1.0, ok_.diagonal_regularizer)
    in_twin = hk.h.counters()["in_twin"]      # cfg 5: a factorisation that broke down in the cheap order was repeated in the robust one
    b = rng.standard_normal(o.N)
    if not in_twin:
        assert hk.last_nreg == o.L.oracle_kkt_nreg(o.h)
        xg, xc = hk.h.ldl_solve(b), o.ldl_solve(b)
        assert np.max(np.abs(xg - xc)) <= 1e-9 * max(1.0, np.max(np.abs(xc)))
    else:
        # The oracle cannot follow into the robust order (minimum degree on K: 5e12 scalar flops for cfg 5, hours on a host core),
        # so the twin's UNREFINED factorisation is held to the size-independent property instead: its solve is the exact solve of
        # the regularised matrix up to a backward error at rounding level,  |(K + eps S) x - b| <= 1e-9 (|K| |x| + |b|)  row by
        # row -- the same bound the same-order comparison above implies.
        K = _sym_K(hk.h)
        Kf = K + sp.diags(hk.diagonal_regularizer * hk.h.dsigns().astype(float))
        xg = hk.h.ldl_solve(b)
        bound = abs(Kf) @ np.abs(xg) + np.abs(b)
        assert np.all(np.abs(Kf @ xg - b) <= 1e-9 * bound)
    for rep in range(2):
        rx, rz = rng.standard_normal(n), rng.standard_normal(m)
        lx_g, lz_g, lx_c, lz_c = np.zeros(n), np.zeros(m), np.zeros(n), np.zeros(m)
        hk.kktsolver_setrhs(rx, rz)
        ok_.kktsolver_setrhs(rx, rz)
        assert hk.kktsolver_solve(lx_g, lz_g) and ok_.kktsolver_solve(lx_c, lz_c)
        scale = max(1.0, np.max(np.abs(lx_c)), np.max(np.abs(lz_c)))
        assert np.max(np.abs(lx_g - lx_c)) <= 1e-10 * scale and np.max(np.abs(lz_g - lz_c)) <= 1e-10 * scale


@pytest.mark.parametrize("trigger", ["forced", "late-iterate"])
def test_twin_factorisation_matches_oracle_in_the_robust_order(trigger, oracle_factory, monkeypatch):
    """hipkkt_refactor's fallback: a factorisation that breaks down in the cheap "cone rows first" order is
    repeated on a twin handle in the minimum-degree order on K.  At cfg 5's size the oracle cannot follow into that order (5e12 scalar
    flops), so the twin is compared HERE, on the same problem family reduced to 8 x PSD(24), n = 300: the cheap order is chosen
    (3.2e8 against 3.8e9 flops) and the oracle factors the twin's permutation in seconds.  Same assertions as
    test_full_size_matches_oracle: resident K bit for bit, regulariser and dynamic-regularisation count equal, unrefined solve 1e-9,
    refined solve 1e-10.  "forced": HIPKKT_FORCE_TWIN=1 declares the cheap order broken down on a benign iterate; "late-iterate": a
    nearly complementary scaling spread over 12 decades, the kind of iterate on which it really does (skipped if it survives)."""
    if trigger == "forced":
        monkeypatch.setenv("HIPKKT_FORCE_TWIN", "1")
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    rng = np.random.default_rng(31)
    Pt, A, cones = _prep(problems.sdp_blocks(n=300, ncones=8, dim=24, seed=5))
    m, n = A.shape
    st = cl.Settings()
    hk = HipKKTSolver(Pt, A, cones, m, n, st)
    assert hk.h.counters()["ordering"] == 1, "the cheap order was not chosen: the twin path is not exercised"
    perm_cheap = hk.h.perm()
    if trigger == "forced":
        scale_cones(cones, rng)
    else:
        scale_cones_late(cones, rng, mu=1e-10, span=6.0)
    ok = hk.kktsolver_update(cones)
    c = hk.h.counters()
    if trigger != "forced" and not c["in_twin"]:
        pytest.skip(f"this iterate did not break the cheap order down (factorisation ok = {ok})")
    assert ok and c["in_twin"] and c["twin_exists"] and c["twin_refactors"] >= 1
    perm_twin = hk.h.perm()                       # the order of the factorisation in use: the twin's
    assert sorted(perm_twin) == list(range(len(perm_twin))) and not np.array_equal(perm_twin, perm_cheap)
    ok_ = oracle_factory(Pt, A, cones, m, n, st, ordering=perm_twin)
    assert ok_.kktsolver_update(cones)
    o = ok_.k
    assert np.array_equal(hk.h.kkt()[2], o.nzval)
    assert abs(hk.diagonal_regularizer - ok_.diagonal_regularizer) <= 1e-16 * max(1.0, ok_.diagonal_regularizer)
    assert hk.last_nreg == o.L.oracle_kkt_nreg(o.h)
    b = rng.standard_normal(o.N)
    xg, xc = hk.h.ldl_solve(b), o.ldl_solve(b)
    # UNREFINED solves of two factorisations of the same permuted matrix differ by rounding x condition number; this K (dense PSD
    # blocks, 1e-8 regulariser) is worse conditioned than the full-size configs, so the bound is MEASURED instead of assumed: x* = the
    # solution of the regularised system refined with extended-precision residuals; the HIP solve must be as close to it as the
    # oracle's own unrefined solve is (within a factor 10), and within 1e-9 wherever the oracle is
    K = _sym_K(hk.h)
    Kf = np.asarray((K + sp.diags(hk.diagonal_regularizer * hk.h.dsigns().astype(float))).todense(), dtype=np.longdouble)
    xs = xc.astype(np.longdouble)
    for _ in range(4):
        xs = xs + o.ldl_solve(np.asarray(b - Kf @ xs, dtype=np.float64))
    err_c = float(np.max(np.abs(xc - xs)) / max(1.0, float(np.max(np.abs(xs)))))
    err_g = float(np.max(np.abs(xg - xs)) / max(1.0, float(np.max(np.abs(xs)))))
    print(f"\n[twin-parity {trigger}] unrefined forward error vs the extended-precision solution: oracle {err_c:.2e}, hip twin {err_g:.2e}; "
          f"hip vs oracle {np.max(np.abs(xg - xc)) / max(1.0, np.max(np.abs(xc))):.2e}")
    assert err_g <= max(1e-9, 10.0 * err_c)
    # REFINED solves (kktsolver_solve!): 1e-10 against the oracle where the oracle itself is that close to the exact solution of
    # K x = b; on the ill-conditioned late iterate the refinement stops on the RESIDUAL (abstol 1e-12 + reltol 1e-13 |b|, or no
    # further gain), which leaves the solution itself undetermined by cond(K) x that residual -- there the HIP solution must be as
    # close to the extended-precision solution as the oracle's own is (factor 10), and meet the same residual bound against the true K
    Kt = np.asarray(K.todense(), dtype=np.longdouble)
    for rep in range(2):
        rx, rz = rng.standard_normal(n), rng.standard_normal(m)
        lx_g, lz_g, lx_c, lz_c = np.zeros(n), np.zeros(m), np.zeros(n), np.zeros(m)
        hk.kktsolver_setrhs(rx, rz)
        ok_.kktsolver_setrhs(rx, rz)
        assert hk.kktsolver_solve(lx_g, lz_g) and ok_.kktsolver_solve(lx_c, lz_c)
        bb = np.concatenate([rx, rz])
        xg_, xc_ = np.concatenate([lx_g, lz_g]), np.concatenate([lx_c, lz_c])
        xs = xc_.astype(np.longdouble)
        for _ in range(6):
            xs = xs + o.ldl_solve(np.asarray(bb - Kt @ xs, dtype=np.float64))
        scale = max(1.0, float(np.max(np.abs(xs))))
        eg, ec = float(np.max(np.abs(xg_ - xs))) / scale, float(np.max(np.abs(xc_ - xs))) / scale
        rg, rc_ = float(np.max(np.abs(bb - Kt @ xg_))), float(np.max(np.abs(bb - Kt @ xc_)))
        print(f"[twin-parity {trigger}] refined solve {rep}: forward error oracle {ec:.2e}, hip twin {eg:.2e}; residual vs the true K oracle {rc_:.2e}, hip {rg:.2e}; "
              f"hip vs oracle {np.max(np.abs(xg_ - xc_)) / scale:.2e}")
        assert eg <= max(1e-10, 10.0 * ec)
        assert rg <= max(1e-9 * max(1.0, float(np.max(np.abs(bb)))), 10.0 * rc_)


@pytest.mark.parametrize("name", ["cfg2a", "cfg3", "cfg5"])
def test_full_size_ipm_end_to_end(name):
    P, q, A, b, cones = FULL[name]()
    sol = cl.Solver(P, q, A, b, cones, cl.Settings()).solve()
    assert sol.status == "SOLVED"
    assert abs(sol.obj_val - sol.obj_val_dual) <= 1e-6 * max(1.0, abs(sol.obj_val))
    assert sol.r_prim < 1e-8 and sol.r_dual < 1e-8


def _traces_agree(tg, tc, upto, tol=1e-10):
    """per-iteration records of two IPM runs (julia_standin/ipm.py Solver.trace) up to iteration `upto`: objective relative,
    residuals absolute (info.jl:50-51), step length and centring parameter"""
    worst = {}
    for it in range(upto + 1):
        a, b = _trace_at(tg, it), _trace_at(tc, it)
        assert a is not None and b is not None, it
        for key, rel in (("cost_primal", True), ("cost_dual", True), ("res_primal", False), ("res_dual", False)):
            d = abs(a[key] - b[key]) / (max(1.0, abs(b[key])) if rel else 1.0)
            worst[key] = max(worst.get(key, 0.0), d)
            assert d <= tol, (it, key, a[key], b[key])
    return worst


@pytest.mark.slow
@pytest.mark.parametrize("name,max_iter", [("cfg3", None), ("cfg2a", 5), ("cfg5", None)])
def test_full_size_ipm_matches_oracle(name, max_iter, oracle_factory, capsys, monkeypatch):
    """The whole IPM at FULL size against the oracle on the same elimination order, iterate by iterate: objective (relative) and
    residuals (absolute, info.jl:50-51) to 1e-10 at EVERY iteration, status and iteration count equal.  Oracle cost on one host core:
    cfg 3 ~5.4 s per iteration (~80 s), cfg 2a ~25 s per iteration (the first 5 iterations: max_iter = 5 on both sides), cfg 5
    ~18 s per iteration.  cfg 5's last iteration breaks down in the cheap order ON BOTH PATHS: the HIP path repeats it on its
    robust-order twin and ends SOLVED; the oracle, held to the cheap order (it cannot follow into the twin's at this size: that is
    test_twin_factorisation_matches_oracle_in_the_robust_order's job on the reduced config), must stop on the factorisation failure
    at exactly that iteration (NUMERICAL_ERROR, or ALMOST_SOLVED when its last good iterate meets the reduced tolerances), and every iteration before it agrees to 1e-10."""
    P, q, A, b, cones = FULL[name]()
    stg = cl.Settings() if max_iter is None else cl.Settings(max_iter=max_iter)
    sg = cl.Solver(P, q, A, b, cones, stg)
    sg.trace = []
    solg = sg.solve()
    hg = sg.kktsystem.kktsolver.h
    twin_its = hg.counters()["twin_refactors"]
    # the cheap order's permutation (the handle reports the twin's while the last factorisation lives there)
    perm = hg.perm() if not hg.counters()["in_twin"] else None
    if perm is None:
        monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
        Pt, At, cn = _prep((P, q, A, b, cones))
        perm = HipKKTSolver(Pt, At, cn, At.shape[0], At.shape[1], cl.Settings()).h.perm()
    stc = cl.Settings() if max_iter is None else cl.Settings(max_iter=max_iter)
    sc = cl.Solver(P, q, A, b, cones, stc, kktsolver_factory=lambda *a: oracle_factory(*a, ordering=perm))
    sc.trace = []
    solc = sc.solve()
    with capsys.disabled():
        print(f"\n[full-size-ipm {name}] hip {solg.status} in {solg.iterations} iterations (factorisations repeated on the twin: {twin_its}); "
              f"oracle {solc.status} in {solc.iterations}")
    if twin_its == 0:
        assert solg.status == solc.status and solg.iterations == solc.iterations
        worst = _traces_agree(sg.trace, sc.trace, solc.iterations if max_iter is None else min(max_iter, solc.iterations))
        assert abs(solg.obj_val - solc.obj_val) <= 1e-10 * max(1.0, abs(solc.obj_val))
        assert abs(solg.r_prim - solc.r_prim) <= 1e-10 and abs(solg.r_dual - solc.r_dual) <= 1e-10
    else:
        # the oracle stops where the cheap order breaks down; the HIP path went on from there on its twin
        # (a failed factorisation ends the reference's loop with NUMERICAL_ERROR, which info_post_process! turns into ALMOST_SOLVED when
        # the last good iterate meets the reduced tolerances: solver.jl:368, info.jl:198-212 -- cfg 5's does)
        assert solg.status == "SOLVED" and solc.status in ("NUMERICAL_ERROR", "ALMOST_SOLVED")
        last_common = len(sc.trace) - 1
        assert last_common >= 5 and solg.iterations >= last_common
        worst = _traces_agree(sg.trace, sc.trace, last_common)
    with capsys.disabled():
        print(f"[full-size-ipm {name}] worst differences over the compared iterations: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))


BATCH_SAMPLE = (100, 113, 126, 137, 150, 168, 187, 201, 222, 240, 255, 271, 300, 318, 339, 355)   # 126 ends ALMOST_SOLVED on both paths


def _trace_at(trace, it):
    for rec in trace:
        if rec["iter"] == it:
            return rec
    return None


@pytest.mark.parametrize("seed", [pytest.param(s_, marks=() if s_ in BATCH_SAMPLE else pytest.mark.slow) for s_ in range(100, 356)])
def test_batch_config_matches_oracle(seed, oracle_factory, capsys):
    """cfg 4: seeds 100..355 are the 256 problems of the batch; EVERY one against the oracle on the same elimination order
    (the 240 outside the original sample carry the `slow` marker: -m "gpu and not slow" skips them).
    Gate: status equal, iterations equal or +-1, objective and residuals to 1e-10.  Where the plain 1e-10 is not met
    the cause is measured, not assumed: the ORACLE is run on a second elimination order (SuperLU MMD) and what IT moves by between
    the two orders (CPU vs CPU, the reference's own arithmetic) is added four-fold to the gate and logged; there is no third tier -- these problems
    carry column scalings of 10^U(-2,2), and the last IPM iterations of some are decided by digits no LDL^T reproduces across
    orderings.  When the iteration counts differ by one, the two runs are compared at their last COMMON iterate instead."""
    P, q, A, b, cones = problems.batch_problem(seed)
    sg = cl.Solver(P, q, A, b, cones, cl.Settings())
    sg.trace = []
    solg = sg.solve()
    perm = sg.kktsystem.kktsolver.h.perm()
    sc = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: oracle_factory(*a, ordering=perm))
    sc.trace = []
    solc = sc.solve()
    assert solg.status == solc.status
    assert abs(solg.iterations - solc.iterations) <= 1
    if solg.status not in ("SOLVED", "ALMOST_SOLVED"):
        return
    itc = min(solg.iterations, solc.iterations)
    tg, tc = _trace_at(sg.trace, itc), _trace_at(sc.trace, itc)
    assert tg is not None and tc is not None
    dobj = abs(tg["cost_primal"] - tc["cost_primal"]) / max(1.0, abs(tc["cost_primal"]))
    dres = max(abs(tg["res_primal"] - tc["res_primal"]), abs(tg["res_dual"] - tc["res_dual"]))
    if solg.iterations == solc.iterations:      # the final answers themselves
        dobj = max(dobj, abs(solg.obj_val - solc.obj_val) / max(1.0, abs(solc.obj_val)))
        dres = max(dres, abs(solg.r_prim - solc.r_prim), abs(solg.r_dual - solc.r_dual))
    if dobj <= 1e-10 and dres <= 1e-10 and solg.iterations == solc.iterations:
        return
    # not within the plain gate: measure what the reference arithmetic itself does between two orderings on this problem
    s2 = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: oracle_factory(*a, ordering="mmd"))
    s2.trace = []
    sol2 = s2.solve()
    it2 = min(itc, sol2.iterations)
    t2, tc2 = _trace_at(s2.trace, it2), _trace_at(sc.trace, it2)
    spread_obj = abs(t2["cost_primal"] - tc2["cost_primal"]) / max(1.0, abs(tc2["cost_primal"]))
    spread_res = max(abs(t2["res_primal"] - tc2["res_primal"]), abs(t2["res_dual"] - tc2["res_dual"]))
    if it2 == itc and sol2.iterations == solc.iterations:
        spread_obj = max(spread_obj, abs(sol2.obj_val - solc.obj_val) / max(1.0, abs(solc.obj_val)))
        spread_res = max(spread_res, abs(sol2.r_prim - solc.r_prim), abs(sol2.r_dual - solc.r_dual))
    with capsys.disabled():
        print(f"\n[batch-parity seed {seed}] {solg.status}: iterations hip/oracle/oracle(mmd) = {solg.iterations}/{solc.iterations}/"
              f"{sol2.iterations}; at iterate {itc}: |dobj| {dobj:.2e}, |dres| {dres:.2e}; cause = the oracle's own spread between "
              f"two elimination orders: obj {spread_obj:.2e}, res {spread_res:.2e}")
    if dobj <= 1e-10 + 4.0 * spread_obj and dres <= 1e-10 + 4.0 * spread_res:
        if solg.iterations != solc.iterations:  # a termination / step-length test decided below 1e-10: the oracle must show it too
            assert abs(sol2.iterations - solc.iterations) <= 1
        return
    # Nothing is waved through beyond that.  (Rounds 3 - 4 had a third tier here, |dobj| <= 1e-4, for seed 324: a stop-ratio branch of the
    # iterative refinement taken differently because the HIP path's first, unrefined solve was 1.6x less accurate than the oracle's.  Cause
    # found in round 5 -- the solve kernels multiply by explicit inverses of the diagonal blocks, tools/seed324_inverse_vs_substitution.py
    # -- and removed: wide blocks whose inverse has large entries take one refinement step against the factored block, kernels.hip
    # k_invert_diag_wide.  Seed 324 now meets the plain 1e-10 with equal iteration counts.)
    raise AssertionError(f"seed {seed}: |dobj| {dobj:.3e}, |dres| {dres:.3e}, iterations hip/oracle/oracle(mmd) {solg.iterations}/{solc.iterations}/{sol2.iterations} "
                         f"-- outside 1e-10 and outside the oracle's own spread between two elimination orders (obj {spread_obj:.3e}, res {spread_res:.3e})")


@pytest.mark.parametrize("refined", [True, False])
def test_refined_block_solves_keep_the_refinement_branches_of_the_reference(refined, monkeypatch, capsys):
    """Batch seed 324 with the ORACLE driving the IPM and the HIP solver shadowing it on identical inputs (same elimination order): with
    the refined block solves (kernels.hip k_invert_diag_wide: wide diagonal blocks whose explicit inverse has large entries take one
    refinement step in the solve kernels) EVERY solve of the run takes the same number of iterative-refinement steps on both paths --
    the branch of kktsolver_directldl.jl:437-444 that rounds 3 - 4 saw flip at iteration 19 does not.  With them switched off
    (HIPKKT_ACCURATE=0) the flip is back: that is the cause, shown on the device (the CPU side of the evidence is
    tests/test_block_refinement.py)."""
    from clarabel_jl_amd.kktsolver import HipKKTSolver
    from oracle.kkt_oracle import OracleKKTSolver
    from tests.fixtures import ShadowKKT

    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    if not refined:
        monkeypatch.setenv("HIPKKT_ACCURATE", "0")
    P, q, A, b, cones = problems.batch_problem(324)
    sh = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: ShadowKKT(HipKKTSolver, OracleKKTSolver, *a))
    sol = sh.solve()
    kk = sh.kktsystem.kktsolver
    log = kk.log
    first = next((k for k, r in enumerate(log) if r[2] != r[3]), None)
    marked = kk.g.h.counters()["accurate_factorisations"]
    with capsys.disabled():
        print(f"\n[refined-block-solves seed 324, refined={refined}] oracle-driven run {sol.status} in {sol.iterations} iterations, {len(log)} solves; factorisations with "
              f"marked blocks: {marked}; first solve with different refinement step counts: {None if first is None else log[first][:4]}")
    assert sol.status == "SOLVED" and sol.iterations == 23
    if refined:
        assert marked > 0, "no block was marked: the mechanism under test did not run"
        assert first is None, log[first]
        assert max(r[1] for r in log) <= 1e-6          # identical inputs, |K| up to 1e15: the refined solutions agree this far at every solve
    else:
        assert marked == 0
        assert first is not None and log[first][0] >= 15, "the flip of rounds 3 - 4 did not reproduce with the refinement off"


@pytest.mark.parametrize("name", ["cfg2a", "cfg3", "cfg5"])
def test_streamed_pivot_chain_equals_whole_tile_handoff(name, monkeypatch):
    """front_block.hip, round 4: the diagonal workgroups of a front batch hand their tile over block by block (8 pivots at a time, the
    next diagonal workgroup eliminates its rows of that panel by substitution behind the producer) instead of as one explicit inverse
    after all 64 pivots (HIPKKT_FB_STREAM=0, the round-3 chain).  Different rounding of L(i, i-1) (substitution vs product with the
    inverse), same factorisation: D, the dynamic-regularisation count and the unrefined LDL solve agree to rounding, the refined solve
    meets the same stopping rule, and the streamed run is deterministic (bit-identical when repeated)."""
    rng = np.random.default_rng(21)
    Pt, A, cones = _prep(FULL[name]())
    m, n = A.shape
    scale_cones(cones, rng)
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    monkeypatch.setenv("HIPKKT_FB_STREAM", "0")
    hk0 = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    assert hk0.kktsolver_update(cones)
    assert not hk0.h.counters()["streamed_chain"]
    monkeypatch.setenv("HIPKKT_FB_STREAM", "1")
    hk1 = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    assert hk1.kktsolver_update(cones)
    c = hk1.h.counters()
    if c["front_batches"] == 0:
        pytest.skip("no front of this problem qualifies for the front-batch kernel (cfg 5: its fronts share their levels)")
    assert c["streamed_chain"] and c["front_block"] and c["sweep_timeouts"] == 0
    if c["in_twin"] or hk0.h.counters()["in_twin"]:
        pytest.skip("this iterate broke down in the cheap order on one of the two chains: the twins are compared by the oracle tests")
    assert hk0.last_nreg == hk1.last_nreg
    d0, d1 = hk0.h.debug_dump(5), hk1.h.debug_dump(5)
    assert np.all(np.sign(d0) == np.sign(d1))
    assert np.max(np.abs(d1 - d0) / np.maximum(np.abs(d0), 1e-300)) <= 1e-6      # pivots of an ill-conditioned K: rounding, amplified
    b = rng.standard_normal(hk0.h.N)
    x0, x1 = hk0.h.ldl_solve(b), hk1.h.ldl_solve(b)
    assert np.max(np.abs(x1 - x0)) <= 1e-9 * max(1.0, np.max(np.abs(x0)))
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    sols = []
    for hk in (hk0, hk1):
        lx, lz = np.zeros(n), np.zeros(m)
        hk.kktsolver_setrhs(rx, rz)
        assert hk.kktsolver_solve(lx, lz)
        sols.append(np.concatenate([lx, lz]))
    assert np.max(np.abs(sols[1] - sols[0])) <= 1e-9 * max(1.0, np.max(np.abs(sols[0])))
    assert hk1.kktsolver_update(cones)
    assert np.array_equal(hk1.h.ldl_solve(b), x1)
    assert np.array_equal(hk1.h.debug_dump(5), d1)


@pytest.mark.parametrize("name", ["cfg2a", "cfg3"])
def test_front_block_second_form_equals_first_form(name, monkeypatch):
    """front_block2.hip (round 5: every tile transposed in the matrix-core accumulators, operands of the steps left of the diagonal
    straight from registers / global memory, the streamed block of 8 pivots consumed as l = p T with T = L_bb^-T D_b^-1) against
    front_block.hip (HIPKKT_FB_V2=0, round 4's streamed chain by substitution).  Same factorisation, another rounding of L(i, i-1):
    D, the dynamic-regularisation count and the unrefined LDL solve agree to rounding, the refined solves agree, and the second form
    is deterministic (bit-identical when repeated)."""
    rng = np.random.default_rng(22)
    Pt, A, cones = _prep(FULL[name]())
    m, n = A.shape
    scale_cones(cones, rng)
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    monkeypatch.setenv("HIPKKT_FB_V2", "0")
    hk0 = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    assert hk0.kktsolver_update(cones)
    monkeypatch.setenv("HIPKKT_FB_V2", "1")
    hk1 = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    assert hk1.kktsolver_update(cones)
    c = hk1.h.counters()
    assert c["front_batches"] > 0 and c["front_block"] and c["sweep_timeouts"] == 0
    assert hk0.h.counters()["front_block"]
    assert hk0.last_nreg == hk1.last_nreg
    d0, d1 = hk0.h.debug_dump(5), hk1.h.debug_dump(5)
    assert np.all(np.sign(d0) == np.sign(d1))
    assert np.max(np.abs(d1 - d0) / np.maximum(np.abs(d0), 1e-300)) <= 1e-6      # pivots of an ill-conditioned K: rounding, amplified
    b = rng.standard_normal(hk0.h.N)
    x0, x1 = hk0.h.ldl_solve(b), hk1.h.ldl_solve(b)
    assert np.max(np.abs(x1 - x0)) <= 1e-9 * max(1.0, np.max(np.abs(x0)))
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    sols = []
    for hk in (hk0, hk1):
        lx, lz = np.zeros(n), np.zeros(m)
        hk.kktsolver_setrhs(rx, rz)
        assert hk.kktsolver_solve(lx, lz)
        sols.append(np.concatenate([lx, lz]))
    assert np.max(np.abs(sols[1] - sols[0])) <= 1e-9 * max(1.0, np.max(np.abs(sols[0])))
    assert hk1.kktsolver_update(cones)
    assert np.array_equal(hk1.h.ldl_solve(b), x1)
    assert np.array_equal(hk1.h.debug_dump(5), d1)


def test_split_k_of_long_tiles_equals_unsplit_updates(monkeypatch):
    """hipkkt_setup.cpp plan_split_k / kernels.hip k_split_reduce (round 4): in cfg 5 the tiles of the variables' block receive 80+
    contributions per update batch next to thousands of tiles with 4-10; they are cut into chunks accumulated by separate wavefronts
    and added in a fixed order.  Same contributions, another association order: D, the dynamic-regularisation count and the unrefined
    solve agree with the unsplit factorisation (HIPKKT_SPLIT_K=0) to rounding, and the split run is deterministic."""
    rng = np.random.default_rng(23)
    Pt, A, cones = _prep(FULL["cfg5"]())
    m, n = A.shape
    scale_cones(cones, rng)
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    monkeypatch.setenv("HIPKKT_SPLIT_K", "0")
    hk0 = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    assert hk0.kktsolver_update(cones)
    monkeypatch.setenv("HIPKKT_SPLIT_K", "1")
    hk1 = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    assert hk1.kktsolver_update(cones)
    assert not hk0.h.counters()["in_twin"] and not hk1.h.counters()["in_twin"]
    assert hk0.last_nreg == hk1.last_nreg
    d0, d1 = hk0.h.debug_dump(5), hk1.h.debug_dump(5)
    assert np.all(np.sign(d0) == np.sign(d1)) and not np.array_equal(d0, d1)      # (the split is really on: another association order)
    assert np.max(np.abs(d1 - d0) / np.abs(d0)) <= 1e-8
    b = rng.standard_normal(hk0.h.N)
    x0, x1 = hk0.h.ldl_solve(b), hk1.h.ldl_solve(b)
    assert np.max(np.abs(x1 - x0)) <= 1e-9 * max(1.0, np.max(np.abs(x0)))
    assert hk1.kktsolver_update(cones)
    assert np.array_equal(hk1.h.ldl_solve(b), x1) and np.array_equal(hk1.h.debug_dump(5), d1)


def test_dense_triangle_spmv_matches_host_product(monkeypatch):
    """kernels.hip k_spmv_dense_tri (round 4): the packed upper triangles of PSD-cone Hs blocks leave the symmetric CSR view of the
    refinement's SpMV and are multiplied from their values alone.  The residual e = b - K x the device computes (debug_dump 18, right
    after an unrefined solve: max_iter = 0) against the host product with the K the handle holds, entry by entry at the rounding level
    of the row's terms -- with the triangles in their own kernel and (HIPKKT_DENSE_TRI=0) inside the view; the refined solves of the
    two modes agree and take the same number of steps."""
    rng = np.random.default_rng(29)
    Pt, A, cones = _prep(problems.sdp_blocks(n=300, ncones=8, dim=24, seed=7))     # 8 triangles of dimension 300
    m, n = A.shape
    scale_cones(cones, rng)
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("HIPKKT_DENSE_TRI", mode)
        hk = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
        assert hk.kktsolver_update(cones)
        assert int(hk.h.debug_dump(19)[0]) == (8 if mode == "1" else 0)
        hk.h.setrhs(rx, rz)
        x, z = np.zeros(n), np.zeros(m)
        ok, steps = hk.h.solve(x, z, True, max_iter=0)
        assert ok and steps == 0
        e = hk.h.debug_dump(18)
        colptr, rowval, nz = hk.h.kkt()
        N = hk.h.N
        K = sp.csc_matrix((nz, rowval, colptr), shape=(N, N))
        K = K + sp.triu(K, 1).T
        xs = np.concatenate([x, z, np.zeros(N - n - m)])
        bs = np.concatenate([rx, rz, np.zeros(N - n - m)])
        scale = abs(K) @ np.abs(xs) + np.abs(bs)
        assert np.all(np.abs(e - (bs - K @ xs)) <= 1e-13 * scale), mode
        x2, z2 = np.zeros(n), np.zeros(m)
        ok, steps2 = hk.h.solve(x2, z2, True)
        assert ok
        out[mode] = (x, z, x2, z2, steps2)
    assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1])      # (the unrefined solve never sees the SpMV)
    assert out["1"][4] == out["0"][4]
    for a_, b_ in ((out["1"][2], out["0"][2]), (out["1"][3], out["0"][3])):
        assert np.max(np.abs(a_ - b_)) <= 1e-12 * max(1.0, np.max(np.abs(b_)))


@pytest.mark.parametrize("name", ["cfg1", "cfg1w16", "cfg2a"])
def test_super_block_sweeps_equal_panel_sweeps(name, monkeypatch):
    """front_sweep.hip (two hand-offs per super-block of 8 panels, explicit inverse of the super-block's diagonal block) against the
    panel-by-panel sweeps of round 2 (HIPKKT_SUPERHOP=0) on the same factorisation: cfg 1's 12-panel root (below the default threshold
    of 16 panels, forced on with HIPKKT_SUPERHOP=1: partial second super-block), the same root cut into panels of 16 columns
    (supernode_max_width = 16: k_invert_super's path for panels narrower than its 64 x 64 tiles) and cfg 2a's 88-panel root (default)."""
    rng = np.random.default_rng(31)
    prob = problems.random_sparse_qp(1000, 2000, 1, 4, 2) if name.startswith("cfg1") else FULL[name]()
    Pt, A, cones = _prep(prob)
    m, n = A.shape
    scale_cones(cones, rng)
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    kw = dict(supernode_max_width=16) if name == "cfg1w16" else {}
    sols = []
    for sh in ("1", "0"):
        monkeypatch.setenv("HIPKKT_SUPERHOP", sh)
        hk = HipKKTSolver(Pt, A, cones, m, n, cl.Settings(), **kw)
        assert hk.kktsolver_update(cones)
        b = np.random.default_rng(32).standard_normal(hk.h.N)
        x = hk.h.ldl_solve(b)
        assert np.array_equal(hk.h.ldl_solve(b), x)                      # deterministic
        sols.append(x)
        assert hk.h.counters()["sweep_timeouts"] == 0
    assert np.max(np.abs(sols[0] - sols[1])) <= 1e-10 * max(1.0, np.max(np.abs(sols[1])))


@pytest.mark.parametrize("name", ["cfg2a", "cfg3"])
def test_extra_tiles_in_front_block_launches_are_bit_identical(name, monkeypatch):
    """hipkkt_factor.cpp fb_extra_tiles_of_stage: the partial last round of a front batch's far updates rides in the NEXT k_front_block
    launch as extra workgroups (compute units the panel chain leaves idle) instead of in the stage's own launch.  Same tiles, same tile
    code, same order of accumulation per tile, and no tile the next panel kernel needs: the factorisation and the solves are
    bit-identical to HIPKKT_FB_EXTRA=0, the profile says how many tiles moved, and nothing is applied twice or dropped (refined
    solve against the true K)."""
    rng = np.random.default_rng(41)
    Pt, A, cones = _prep(FULL[name]())
    m, n = A.shape
    scale_cones(cones, rng)
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("HIPKKT_FB_EXTRA", flag)
        hk = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
        assert hk.kktsolver_update(cones)
        b = np.random.default_rng(42).standard_normal(hk.h.N)
        x = hk.h.ldl_solve(b)
        hk.h.set_profiling(True)
        assert hk.kktsolver_update(cones)
        prof = hk.h.profile()
        assert np.array_equal(hk.h.ldl_solve(b), x)          # the eager profiled path applies the same split
        hk.h.set_profiling(2)                                # ... and mode 2 keeps every tile in its stage's own launch
        assert hk.kktsolver_update(cones)
        assert hk.h.profile()["front_block_extra_tiles"] == 0 and np.array_equal(hk.h.ldl_solve(b), x)
        hk.h.set_profiling(False)
        out.append((x, hk.h.debug_dump(5), prof))
    (x1, d1, p1), (x0, d0, p0) = out
    # ... and so is the mask-free core of the full tiles (dense_tile.h dense_tile_core_full) against the general tile code
    monkeypatch.setenv("HIPKKT_FB_EXTRA", "1")
    monkeypatch.setenv("HIPKKT_FULL_TILES", "0")
    hk = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    assert hk.kktsolver_update(cones)
    assert np.array_equal(hk.h.ldl_solve(np.random.default_rng(42).standard_normal(hk.h.N)), x1) and np.array_equal(hk.h.debug_dump(5), d1)
    assert p1["front_block_extra_tiles"] > 0 and p0["front_block_extra_tiles"] == 0
    assert abs((p1["dense4_flops"] + p1["front_block_extra_flops"]) - p0["dense4_flops"]) <= 1e-9 * p0["dense4_flops"] or p1["dense4_launches"] != p0["dense4_launches"]
    assert np.array_equal(d1, d0) and np.array_equal(x1, x0)


@pytest.mark.parametrize("name", ["cfg2a", "cfg3"])
def test_deferred_gather_on_the_side_stream_is_bit_identical(name, monkeypatch):
    """hipkkt_setup.cpp split_gather_stages / hipkkt_factor.cpp enqueue_gather (round 6): the entries of the bottom batch's big
    per-entry gather that land beyond the next update batch run on the side stream next to that batch's levels and are joined
    before its far stage.  Every entry keeps its pair list: the factor (D) and the solve are bit-identical to the whole gather in
    line (switch GATHER_OVERLAP=0), in the graph path and in the eager profiled path, and the split really happened on cfg 2a."""
    rng = np.random.default_rng(43)
    Pt, A, cones = _prep(FULL[name]())
    m, n = A.shape
    scale_cones(cones, rng)
    monkeypatch.setenv("HIPKKT_PLAN_CACHE", "0")
    out = []
    for flag in ("1", "0"):
        monkeypatch.setenv("HIPKKT_GATHER_OVERLAP", flag)
        hk = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
        for _ in range(2):                                   # graph capture, then its replay
            assert hk.kktsolver_update(cones)
        b = np.random.default_rng(44).standard_normal(hk.h.N)
        x = hk.h.ldl_solve(b)
        d = hk.h.debug_dump(5)
        hk.h.set_profiling(True)
        assert hk.kktsolver_update(cones)
        assert np.array_equal(hk.h.ldl_solve(b), x) and np.array_equal(hk.h.debug_dump(5), d)
        hk.h.set_profiling(False)
        out.append((x, d, hk.h.counters()))
    (x1, d1, c1), (x0, d0, c0) = out
    assert np.array_equal(d1, d0) and np.array_equal(x1, x0)
    if name == "cfg2a":
        assert c1["deferred_gather_entries"] > 3_000_000 and c0["deferred_gather_entries"] == 0
