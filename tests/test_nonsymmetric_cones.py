"""The non-symmetric cones of the stand-in caller (julia_standin/cones_nonsym.py: Exponential, Power, Generalized Power) -- the
producers of the 3 x 3 dense Hs blocks and of the rank-3 expansion data the KKT path takes (directldl_datamaps.jl:81-167).

The restated formulas are held against what they must satisfy MATHEMATICALLY (finite differences of the barrier the reference
states, conjugacy of the primal and the dual barrier, the secant equations of the primal-dual scaling), not against a second copy of
themselves; the reference's known answers for such problems (test/OptTests/basic_exp.jl, basic_pow.jl, basic_genpow.jl) are in
tests/golden/reference_known_answers.json and checked by tests/test_golden_file.py.  CPU only."""
import math

import numpy as np
import pytest
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd import problems
from julia_standin.cones_nonsym import ExponentialCone, GenPowerCone, PowerCone, _wright_omega
from tests import fixtures as fx


def _fd_grad(f, x, h=1e-6):
    g = np.zeros(x.size)
    for i in range(x.size):
        e = np.zeros(x.size)
        e[i] = h
        g[i] = (f(x + e) - f(x - e)) / (2 * h)
    return g


def _fd_hess(grad, x, h=1e-6):
    H = np.zeros((x.size, x.size))
    for i in range(x.size):
        e = np.zeros(x.size)
        e[i] = h
        H[:, i] = (grad(x + e) - grad(x - e)) / (2 * h)
    return 0.5 * (H + H.T)


def _dual_interior(K, rng):
    z, s = np.zeros(K.numel), np.zeros(K.numel)
    K.unit_initialization(z, s)
    for _ in range(100):
        zt = z * rng.uniform(0.5, 2.0) + 0.15 * rng.standard_normal(K.numel)
        if K.is_dual_feasible(zt):
            return zt
    raise AssertionError("no interior point found")


def _primal_interior(K, rng):
    z, s = np.zeros(K.numel), np.zeros(K.numel)
    K.unit_initialization(z, s)
    for _ in range(100):
        st = s * rng.uniform(0.5, 2.0) + 0.15 * rng.standard_normal(K.numel)
        if K.is_primal_feasible(st):
            return st
    raise AssertionError("no interior point found")


CONES = {"exp": lambda: ExponentialCone(), "pow_0.6": lambda: PowerCone(0.6), "pow_0.1": lambda: PowerCone(0.1), "pow_0.5": lambda: PowerCone(0.5),
         "genpow_2_1": lambda: GenPowerCone([0.6, 0.4], 1), "genpow_4_3": lambda: GenPowerCone([0.1, 0.2, 0.3, 0.4], 3)}


def test_wright_omega_solves_its_equation():  # coneops_expcone.jl:403-467: w + log(w) = z
    for z in [0.0, 0.3, 1.0, 2.5, 1.0 + math.pi - 1e-9, 1.0 + math.pi + 1e-9, 7.0, 40.0, 1e3, 1e6]:
        w = _wright_omega(z)
        assert abs(w + math.log(w) - z) <= 1e-12 * max(1.0, z), z
    with pytest.raises(ValueError):
        _wright_omega(-0.1)


@pytest.mark.parametrize("name", list(CONES))
def test_dual_gradient_and_hessian_are_those_of_the_dual_barrier(name):
    """update_dual_grad_H against central differences of barrier_dual (the barrier written in the header of each reference file).
    For the Generalized Power cone the Hessian is never stored: it is  D + p p' - q q' - r r'  (coneops_genpowcone.jl:111-134)."""
    rng = np.random.default_rng(5)
    K = CONES[name]()
    for _ in range(5):
        z = _dual_interior(K, rng)
        K.update_dual_grad_H(z)
        g = K.grad.copy()
        assert np.allclose(g, _fd_grad(K.barrier_dual, z), rtol=2e-6, atol=1e-7)

        def grad_at(x):
            K.update_dual_grad_H(x)
            return K.grad.copy()

        Hfd = _fd_hess(grad_at, z)
        K.update_dual_grad_H(z)
        if isinstance(K, GenPowerCone):
            d = np.concatenate([K.d1, np.full(K.dim2, K.d2)])
            q = np.concatenate([K.q, np.zeros(K.dim2)])
            r = np.concatenate([np.zeros(K.dim1), K.r])
            H = np.diag(d) + np.outer(K.p, K.p) - np.outer(q, q) - np.outer(r, r)
            K.mu = 1.7
            y = np.zeros(K.numel)
            x = rng.standard_normal(K.numel)
            K.mul_Hs(y, x, None)
            assert np.allclose(y, 1.7 * (H @ x), rtol=1e-12, atol=1e-12)
            blk = np.zeros(K.numel)
            K.get_Hs(blk)
            assert np.array_equal(blk, 1.7 * d)                      # the diagonal part only goes into the Hs block
        else:
            H = K.H_dual
        assert np.allclose(H, Hfd, rtol=5e-6, atol=1e-6 * np.max(np.abs(Hfd)))
        assert np.all(np.linalg.eigvalsh(H) > 0)
        assert abs(float(g @ z) + K.degree) <= 1e-10 * K.degree          # <z, grad f*(z)> = -nu (logarithmic homogeneity)


@pytest.mark.parametrize("name", list(CONES))
def test_primal_gradient_is_the_conjugate_of_the_dual_barrier(name):
    """gradient_primal (Wright omega / one-sided Newton) returns g(s) with  -g(s) in the dual cone  and  grad f*(-g(s)) = -s ,
    the defining relation of the conjugate barrier; <s, g(s)> = -nu.
    Power cone with alpha != 1/2: as restated from coneops_powcone.jl:449-478 the one-sided Newton iteration starts from a closed-form
    x0 that lies to the RIGHT of the root of its (decreasing) function, so its first correction is negative and the iteration halts at
    once (coneops_nonsymmetric_common.jl:183-188): the primal gradient is the closed-form approximation, good to 1 % (alpha 0.6) ..
    25 % (alpha 0.1).  It only feeds the primal-dual scaling's second secant pair and the centrality check; the relation
    <s, g(s)> = -nu holds exactly by construction of g1, g2 from g3.  Kept as the reference has it."""
    rng = np.random.default_rng(6)
    K = CONES[name]()
    approx = isinstance(K, PowerCone) and K.alpha != 0.5
    for _ in range(5):
        s = _primal_interior(K, rng)
        if isinstance(K, GenPowerCone):
            g = np.zeros(K.numel)
            K.gradient_primal(g, s)
        else:
            g = K.gradient_primal(s)
        assert K.is_dual_feasible(-g)
        K.update_dual_grad_H(-g)
        if approx:
            assert np.max(np.abs(K.grad + s)) <= 0.3 * np.max(np.abs(s))
        else:
            assert np.allclose(K.grad, -s, rtol=1e-7, atol=1e-9)
        assert abs(float(g @ s) + K.degree) <= 1e-7 * K.degree
        # barrier_primal(s) = <s, g> - f*(-g) = -nu - f*(-g)
        assert abs(K.barrier_primal(s) - (-K.degree - K.barrier_dual(-g))) <= 1e-9 * max(1.0, abs(K.barrier_primal(s)))


@pytest.mark.parametrize("name", ["exp", "pow_0.6", "pow_0.1", "pow_0.5"])
def test_primal_dual_scaling_meets_its_secant_equations(name):
    """coneops_nonsymmetric_common.jl:82-165: Hs z = s and Hs zt = st with zt = f'(s), st = f*'(z) (the BFGS-type scaling);
    Hs symmetric positive definite; packed into the block column by column (mathutils.jl:402-412); the dual scaling is mu H*(z)."""
    rng = np.random.default_rng(7)
    K = CONES[name]()
    for _ in range(5):
        z, s = _dual_interior(K, rng), _primal_interior(K, rng)
        if float(z @ s) <= 0:
            continue
        assert K.update_scaling(s, z, float(z @ s) / 3, "primal_dual")
        Hs = K.Hs.copy()
        zt, st = K.gradient_primal(s), K.grad.copy()
        assert np.allclose(Hs, Hs.T)
        assert np.all(np.linalg.eigvalsh(Hs) > 0)
        if not np.allclose(Hs, (float(z @ s) / 3) * K.H_dual):      # (falls back to the dual scaling near the central path)
            assert np.allclose(Hs @ z, s, rtol=1e-9, atol=1e-10)
            assert np.allclose(Hs @ zt, st, rtol=1e-7, atol=1e-9)
        blk = np.zeros(6)
        K.get_Hs(blk)
        assert np.array_equal(blk, [Hs[0, 0], Hs[0, 1], Hs[1, 1], Hs[0, 2], Hs[1, 2], Hs[2, 2]])
        assert K.update_scaling(s, z, 0.37, "dual")
        assert np.array_equal(K.Hs, 0.37 * K.H_dual)


@pytest.mark.parametrize("name", ["exp", "pow_0.6", "pow_0.1"])
def test_higher_order_correction_is_half_the_third_derivative(name):
    """higher_correction! (coneops_expcone.jl:300-367, coneops_powcone.jl:319-405): eta = 1/2 f*'''(z)[u, v] with u = H*(z)^-1 ds
    (combined_ds_shift! then SUBTRACTS it: shift = sigma mu grad f*(z) - eta) -- against the central difference of the dual Hessian
    along u."""
    rng = np.random.default_rng(8)
    K = CONES[name]()
    for _ in range(5):
        z = _dual_interior(K, rng)
        ds, v = rng.standard_normal(3), rng.standard_normal(3)
        K.update_dual_grad_H(z)
        K.z[:] = z
        H = K.H_dual.copy()
        eta = K.higher_correction(ds, v)
        u = np.linalg.solve(H, ds)
        h = 1e-5
        K.update_dual_grad_H(z + h * u)
        Hp = K.H_dual.copy()
        K.update_dual_grad_H(z - h * u)
        Hm = K.H_dual.copy()
        ref = 0.5 * ((Hp - Hm) / (2 * h)) @ v
        assert np.allclose(eta, ref, rtol=2e-5, atol=1e-6 * np.max(np.abs(ref)))


def test_composite_cone_with_non_symmetric_members():
    """compositecone_type.jl:28-66 / coneops_compositecone.jl: degree, block ranges, KKT descriptors, symmetric cones stepped first"""
    specs = [cl.ZeroConeT(1), cl.NonnegativeConeT(2), cl.NonnegativeConeT(1), cl.ExponentialConeT(), cl.PowerConeT(0.3),
             cl.GenPowerConeT([0.25, 0.75], 2), cl.SecondOrderConeT(6)]
    K = cl.CompositeCone(cl.cones_new_collapsed(specs))
    assert not K.is_symmetric() and not K.allows_primal_dual_scaling()
    assert cl.CompositeCone([cl.ExponentialConeT(), cl.PowerConeT(0.5)]).allows_primal_dual_scaling()
    assert K.degree == 0 + 3 + 3 + 3 + 3 + 1 and K.numel == 1 + 3 + 3 + 3 + 4 + 6
    assert [r.stop - r.start for r in K.rng_blocks] == [1, 3, 6, 6, 4, 6]
    numel, hs_dense, sparse_kind, dim1 = K.kkt_descriptors()
    assert list(numel) == [1, 3, 3, 3, 4, 6] and list(hs_dense) == [0, 0, 1, 1, 0, 0]
    assert list(sparse_kind) == [0, 0, 0, 0, 2, 1] and list(dim1) == [0, 0, 0, 0, 2, 0]
    assert list(K.kkt_cone_kinds()) == [0, 1, -1, -1, -1, 2]           # no on-device scaling for such a cone set
    z, s = np.zeros(K.numel), np.zeros(K.numel)
    K.unit_initialization(z, s)
    assert np.array_equal(z[:4], [0, 1, 1, 1]) and z[4] == -1.051383945322714 and np.array_equal(z, s)
    assert np.allclose(z[7:10], [math.sqrt(1.3), math.sqrt(1.7), 0.0]) and np.allclose(z[10:14], [math.sqrt(1.25), math.sqrt(1.75), 0, 0])
    K.use_settings(cl.Settings())
    # a step that leaves the exponential cone at full length is cut back by factors of 0.8 from 1 - sqrt(eps)
    dz, ds = np.zeros(K.numel), np.zeros(K.numel)
    ds[4:7] = [3.0, 0.0, 0.0]
    a, _ = K.step_length(dz, ds, z, s, 1.0)
    assert 0 < a < 1.0 and K.cones[2].is_primal_feasible(s[4:7] + a * ds[4:7])
    k = round(math.log(a / (1.0 - math.sqrt(np.finfo(float).eps))) / math.log(0.8))
    assert abs(a - (1.0 - math.sqrt(np.finfo(float).eps)) * 0.8 ** k) <= 1e-15
    assert np.isfinite(K.compute_barrier(z, s, dz, ds, a))


@pytest.mark.parametrize("name,mk", [("exp", fx.basic_exp), ("pow", fx.basic_pow), ("genpow", fx.basic_genpow)])
def test_spread_of_the_reference_arithmetic_between_elimination_orders(name, mk, oracle_factory):
    """What the oracle itself moves by between elimination orders on the reference's three non-symmetric fixtures (CPU vs CPU): the
    bound tests/test_golden_file.py::test_hip_path_meets_golden_file applies to such problems (`cross_order_tol` of the golden
    file) is this measured spread, not a guess.  Iteration counts do not move."""
    runs = []
    for order in ["natural", "mmd", 1, 2]:
        P, q, A, b, cones = mk()

        def fac(*a, order=order):
            if isinstance(order, str):
                return oracle_factory(*a, ordering=order)
            N = oracle_factory(*a, ordering="natural").k.N
            return oracle_factory(*a, ordering=np.random.default_rng(order).permutation(N))

        runs.append(cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=fac).solve())
    r0 = runs[0]
    assert all(r.status == "SOLVED" and r.iterations == r0.iterations for r in runs)
    spread = max(abs(r.obj_val - r0.obj_val) / max(1.0, abs(r0.obj_val)) for r in runs)
    spread_res = max(max(abs(r.r_prim - r0.r_prim), abs(r.r_dual - r0.r_dual)) for r in runs)
    assert spread <= 2.5e-8 and spread_res <= 2.5e-8          # (measured: exp 7e-10, pow 3e-9, genpow 2e-15; residuals <= 1e-10)
    assert spread > 1e-12 or name == "genpow"                  # ... and it is NOT at the 1e-10 level for the 3 x 3 blocks


def test_mixed_problem_solves_and_hands_over_what_the_kkt_path_expects(oracle_factory):
    """problems.nonsymmetric_mix through the stand-in IPM + oracle: SOLVED; at the final scaling the Exponential / Power blocks are
    symmetric positive definite 3 x 3 matrices and the Generalized Power cones' expansion keeps K quasi-definite
    (D + p p' - q q' - r r' positive definite)."""
    P, q, A, b, cones = problems.nonsymmetric_mix(n=60, nexp=8, npow=6, ngenpow=3, nn=20, nzero=3, socdim=5, seed=3)
    S = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=oracle_factory)
    sol = S.solve()
    assert sol.status == "SOLVED"
    assert np.linalg.norm(A @ sol.x + sol.s - b) <= 1e-6 * max(1.0, np.linalg.norm(b))
    seen = set()
    for c, r in zip(S.cones.cones, S.cones.rng_cones):
        s, z = sol.s[r], sol.z[r]
        if isinstance(c, (ExponentialCone, PowerCone)):
            ev = np.linalg.eigvalsh(c.Hs)                       # (at the last iterate the block's condition number exceeds 1e16)
            assert ev[-1] > 0 and ev[0] > -1e-12 * ev[-1]
            seen.add(type(c).__name__)
        elif isinstance(c, GenPowerCone):
            d = np.concatenate([c.d1, np.full(c.dim2, c.d2)])
            qq = np.concatenate([c.q, np.zeros(c.dim2)])
            rr = np.concatenate([np.zeros(c.dim1), c.r])
            ev = np.linalg.eigvalsh(np.diag(d) + np.outer(c.p, c.p) - np.outer(qq, qq) - np.outer(rr, rr))
            assert ev[-1] > 0 and ev[0] > -1e-12 * ev[-1]
            seen.add("GenPowerCone")
        if not getattr(c, "is_symmetric", True):
            assert float(s @ z) >= -1e-7
    assert seen == {"ExponentialCone", "PowerCone", "GenPowerCone"}


def test_json_wire_format_carries_the_non_symmetric_cones(tmp_path):
    """json.jl:142-158 / :190-213: PowerConeT -> alpha, ExponentialConeT -> (), GenPowerConeT -> [alpha, dim2]"""
    import json

    from clarabel_jl_amd import jsonio

    P, q, A, b, cones = fx.basic_genpow()
    cones = cones + [cl.ExponentialConeT(), cl.PowerConeT(0.25)]
    A = sp.vstack([A, sp.csc_matrix((6, 6))]).tocsc()
    b = np.concatenate([b, np.zeros(6)])
    f = str(tmp_path / "p.json")
    jsonio.save_to_file(f, P, q, A, b, cones)
    d = json.load(open(f))
    assert d["cones"] == [{"GenPowerConeT": [[0.6, 0.4], 1]}, {"GenPowerConeT": [[0.1, 0.9], 1]}, {"ZeroConeT": 2},
                          {"ExponentialConeT": []}, {"PowerConeT": 0.25}]
    P2, q2, A2, b2, cones2, _ = jsonio.load_from_file(f)
    assert cones2 == cones and (A2 != A).nnz == 0 and np.array_equal(b2, b)
