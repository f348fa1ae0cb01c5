"""Problem fixtures taken from the reference's own tests (data only, /root/reference/test/OptTests).
Each returns (P, q, A, b, cone_specs) in scipy/numpy form."""
import numpy as np
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl


def basic_qp():  # basic_qp.jl:6-19
    P = sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 2.0]]))
    c = np.array([1.0, 1.0])
    A0 = np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
    A = sp.csc_matrix(np.vstack([-A0, A0]))
    b = np.array([-1.0, 0.0, 0.0, 1.0, 0.7, 0.7])
    return P, c, A, b, [cl.NonnegativeConeT(3), cl.NonnegativeConeT(3)]


def basic_qp_dualinf():  # basic_qp.jl:20-30
    P = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 1.0]]))
    c = np.array([1.0, -1.0])
    A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 0.0]]))
    b = np.array([1.0, 1.0])
    return P, c, A, b, [cl.NonnegativeConeT(2)]


def univariate_qp():  # basic_qp.jl:44-60
    return (sp.identity(1, format="csc"), np.zeros(1), sp.identity(1, format="csc"), np.ones(1),
            [cl.NonnegativeConeT(1)])


def basic_lp():  # basic_lp.jl:6-16
    P = sp.csc_matrix((3, 3))
    A = sp.vstack([sp.identity(3), -sp.identity(3)]).tocsc() * 2.0
    c = np.array([3.0, -2.0, 1.0])
    b = np.ones(6)
    return P, c, A, b, [cl.NonnegativeConeT(3), cl.NonnegativeConeT(3)]


def eq_constrained(variant=1):  # basic_eq_constrained.jl:16-42
    P = sp.identity(3, format="csc")
    if variant == 1:
        c = np.zeros(3)
        A = sp.csc_matrix(np.array([[0.0, 1.0, 1.0], [0.0, 1.0, -1.0]]))
    else:
        c = np.array([1.0, 2.0, 3.0])
        A = sp.csc_matrix(np.array([[1.0, 1.0, 1.0], [0.0, 1.0, -1.0]]))
    b = np.array([2.0, 0.0])
    return P, c, A, b, [cl.ZeroConeT(2)]


def unconstrained():  # basic_unconstrained.jl:16-26
    return sp.identity(3, format="csc"), np.array([1.0, 2.0, -3.0]), sp.csc_matrix((0, 3)), np.zeros(0), []


def basic_socp():  # basic_socp.jl:6-30
    P = np.array([[1.4652521089139698, 0.6137176286085666, -1.1527861771130112],
                  [0.6137176286085666, 2.219109946678485, -1.4400420548730628],
                  [-1.1527861771130112, -1.4400420548730628, 1.6014483534926371]])
    I3 = sp.identity(3)
    A = sp.vstack([I3 * 2.0, -I3 * 2.0, I3]).tocsc()
    c = np.array([0.1, -2.0, 1.0])
    b = np.concatenate([np.ones(6), np.zeros(3)])
    return sp.csc_matrix(P), c, A, b, [cl.NonnegativeConeT(3), cl.NonnegativeConeT(3), cl.SecondOrderConeT(3)]


def basic_sdp():  # basic_sdp.jl:6-20
    P = sp.identity(6, format="csc")
    c = np.zeros(6)
    A = sp.identity(6, format="csc")
    b = np.array([-3.0, 1.0, 4.0, 1.0, 2.0, 5.0])
    return P, c, A, b, [cl.PSDTriangleConeT(3)]


def basic_exp():  # basic_exp.jl:6-37
    A1 = np.hstack([np.ones((1, 3)), np.zeros((1, 4))])                      # ZeroCone
    A2 = np.hstack([np.zeros((3, 2)), -np.eye(3), np.zeros((3, 2))])         # NNCone
    A3 = np.zeros((3, 7))                                                    # expcone
    A3[0, 0] = -1.0
    A3[1, 2] = -1.0
    A3[2, 4] = -1.0
    c = np.array([1.0, 0.5, -2.0, -0.1, 1.0, 3.0, 0.0])
    P = sp.identity(7, format="csc") * 1e-1
    A = sp.csc_matrix(np.vstack([A1, A2, A3]))
    b = np.concatenate([[10.0], np.zeros(3), np.zeros(3)])
    return P, c, A, b, [cl.ZeroConeT(1), cl.NonnegativeConeT(3), cl.ExponentialConeT()]


def basic_pow():  # basic_pow.jl:6-39: x = (x1, y, z1, x2, y2, z2), (x1, y, z1) in K_pow(0.6), (x2, y2, z2) in K_pow(0.1)
    P = sp.csc_matrix((6, 6))
    q = np.zeros(6)
    q[2] = q[5] = -1.0
    A1 = np.eye(6)
    A2 = np.array([[1.0, 2.0, 0.0, 3.0, 0.0, 0.0]])      # x1 + 2 y + 3 x2 == 3
    A3 = np.array([[0.0, 0.0, 0.0, 0.0, 1.0, 0.0]])      # y2 == 1
    A = -sp.csc_matrix(np.vstack([A1, A2, A3]))
    b = np.concatenate([np.zeros(6), [-3.0], [-1.0]])
    return P, q, A, b, [cl.PowerConeT(0.6), cl.PowerConeT(0.1), cl.ZeroConeT(1), cl.ZeroConeT(1)]


def basic_genpow():  # basic_genpow.jl:7-33: the same problem with the two power cones written as generalized power cones
    P = sp.csc_matrix((6, 6))
    q = np.zeros(6)
    q[2] = q[5] = -1.0
    A = sp.csc_matrix(np.vstack([-np.eye(6), [[1.0, 2.0, 0.0, 3.0, 0.0, 0.0]], [[0.0, 0.0, 0.0, 0.0, 1.0, 0.0]]]))
    b = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.0, 1.0])
    return P, q, A, b, [cl.GenPowerConeT([0.6, 0.4], 1), cl.GenPowerConeT([0.1, 0.9], 1), cl.ZeroConeT(2)]


def updating_data():  # data_updating.jl:6-21
    P = sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 2.0]]))
    q = np.array([1.0, 1.0])
    I2 = sp.identity(2)
    A = sp.vstack([-I2, I2]).tocsc()
    b = np.ones(4)
    return P, q, A, b, [cl.NonnegativeConeT(2), cl.NonnegativeConeT(2)]


def lasso_socp(seed=12345, n=8):
    """Shape of socp-lasso.jl:6-54 (SOC of dim 50n+2 -> sparse expansion path).  The Julia
    MersenneTwister stream is not reproducible here, so the random data are numpy-seeded."""
    rng = np.random.default_rng(seed)
    m = 50 * n
    F = rng.random((m, n))
    vtrue = np.where(rng.random(n) < 0.3, rng.random(n), 0.0)
    bb = F @ vtrue + 0.1 * rng.random(m)
    mu = 0.1 * np.max(np.abs(F.T @ bb))
    Z = np.zeros
    I = np.eye
    A1 = -np.block([[np.ones((1, 1)), Z((1, 2 * n + 1)), np.ones((1, 1)), Z((1, m))],
                    [-np.ones((1, 1)), Z((1, 2 * n)), np.ones((1, 1)), Z((1, m + 1))],
                    [Z((m, 1)), -2 * F, Z((m, n + 2)), I(m)]])
    A2 = -np.block([[Z((n, 1)), I(n), -I(n), Z((n, m + 2))],
                    [Z((n, 1)), -I(n), -I(n), Z((n, m + 2))]])
    A3 = -np.block([[Z((1, 2 * n + 1)), -np.ones((1, 1)), Z((1, m + 1))],
                    [Z((1, 2 * n + 2)), -np.ones((1, 1)), Z((1, m))],
                    [Z((m, 2 * n + 3)), -I(m)]])
    b1 = np.concatenate([[1.0, 1.0], -2 * bb])
    b2 = np.zeros(2 * n)
    b3 = np.zeros(m + 2)
    c = np.concatenate([[1.0], np.zeros(n), mu * np.ones(n), np.zeros(m + 2)])
    P = sp.identity(len(c), format="csc")
    A = sp.csc_matrix(np.vstack([A1, A2, A3]))
    b = np.concatenate([b1, b2, b3])
    return P, c, A, b, [cl.NonnegativeConeT(len(b1)), cl.NonnegativeConeT(len(b2)), cl.SecondOrderConeT(len(b3))]


def scale_cones(cones, rng):
    """put every cone at a random strictly interior (s, z) and update its scaling"""
    m = cones.numel
    s, z = np.zeros(m), np.zeros(m)
    for c, r in zip(cones.cones, cones.rng_cones):
        if isinstance(c, cl.cones.PSDTriangleCone):
            for v in (s, z):
                M = rng.standard_normal((c.n, c.n))
                v[r] = c.mat_to_svec(M @ M.T + c.n * np.eye(c.n))
        elif isinstance(c, cl.cones.SecondOrderCone):
            for v in (s, z):
                t = rng.standard_normal(c.dim)
                t[0] = np.linalg.norm(t[1:]) + 0.5 + rng.random()
                v[r] = t
        else:
            s[r] = rng.random(c.numel) + 0.2
            z[r] = rng.random(c.numel) + 0.2
    assert cones.update_scaling(s, z, 1.0)
    return s, z


def scale_cones_nonsymmetric(cones, rng, strategy="primal_dual"):
    """scale_cones for a cone set with Exponential / Power / Generalized Power members: (s, z) = a positive multiple of each such
    cone's central ray plus a perturbation that keeps both strictly inside (checked with the cone's own feasibility tests), the
    symmetric members as in scale_cones; update_scaling! with mu = <s, z> / (degree + 1) and the given scaling strategy."""
    m = cones.numel
    s, z = np.zeros(m), np.zeros(m)
    for c, r in zip(cones.cones, cones.rng_cones):
        if getattr(c, "is_symmetric", True):
            if isinstance(c, cl.cones.SecondOrderCone):
                for v in (s, z):
                    t = rng.standard_normal(c.dim)
                    t[0] = np.linalg.norm(t[1:]) + 0.5 + rng.random()
                    v[r] = t
            elif isinstance(c, cl.cones.ZeroCone):
                pass
            else:
                assert isinstance(c, cl.cones.NonnegativeCone)
                s[r] = rng.random(c.numel) + 0.2
                z[r] = rng.random(c.numel) + 0.2
            continue
        z0, s0 = np.zeros(c.numel), np.zeros(c.numel)
        c.unit_initialization(z0, s0)
        for v, inside in ((s, c.is_primal_feasible), (z, c.is_dual_feasible)):
            for _ in range(200):
                t = s0 * rng.uniform(0.5, 2.0) + 0.2 * rng.standard_normal(c.numel)
                if inside(t) and inside(s0 + 0.5 * (t - s0)):
                    break
            else:
                raise AssertionError("no interior point")
            v[r] = t
    mu = float(s @ z) / (cones.degree + 1)
    assert mu > 0 and cones.update_scaling(s, z, mu, strategy)
    return s, z, mu


def scale_cones_late(cones, rng, mu=1e-9, span=6.0):
    """a late-IPM-like iterate: s and z nearly complementary (s o z ~ mu e) with eigenvalues / entries spread over 10^span --
    the kind of scaling on which an elimination order that takes the cone blocks first can break down"""
    m = cones.numel
    s, z = np.zeros(m), np.zeros(m)
    for c, r in zip(cones.cones, cones.rng_cones):
        if isinstance(c, cl.cones.PSDTriangleCone):
            Q, _ = np.linalg.qr(rng.standard_normal((c.n, c.n)))
            ls = 10.0 ** rng.uniform(-span, 0.0, c.n)
            s[r] = c.mat_to_svec((Q * ls) @ Q.T)
            z[r] = c.mat_to_svec((Q * (mu / ls)) @ Q.T)
        elif isinstance(c, cl.cones.SecondOrderCone):
            for v in (s, z):
                t = rng.standard_normal(c.dim)
                t[0] = np.linalg.norm(t[1:]) * (1.0 + 10.0 ** rng.uniform(-span, -1.0))
                v[r] = t
        else:
            ls = 10.0 ** rng.uniform(-span, 0.0, c.numel)
            s[r] = ls
            z[r] = mu / ls
    assert cones.update_scaling(s, z, mu)
    return s, z


class ShadowKKT:
    """Test infrastructure: the ORACLE drives an IPM run while the HIP solver is handed the very same inputs at every KKT call
    (same elimination order); per solve the two solutions and refinement-step counts are recorded in `log` as
    (iteration, rel_dx, ir_hip, ir_oracle, HIP's residual norms, the oracle's, max|D| / min|D| of the factorisation).  Used to establish the CAUSE when two IPM trajectories part."""
    batch_constant_rhs = False

    def __init__(self, hip_cls, oracle_cls, *a):
        import numpy as np

        self.np = np
        self.g = hip_cls(*a)
        self.c = oracle_cls(*a, ordering=self.g.h.perm())
        self.settings = self.c.settings
        self.it = 0
        self.log = []

    def kktsolver_update(self, cones_):
        self.it += 1
        okc = self.c.kktsolver_update(cones_)
        self.g.kktsolver_update(cones_)
        try:
            d = self.np.abs(self.g.h.debug_dump(5))                 # the pivots of this factorisation: max / min = a lower estimate of cond(K)
            self.kappa = float(d.max() / max(d.min(), 1e-300))
        except Exception:
            self.kappa = float("nan")
        return okc

    def kktsolver_setrhs(self, rx, rz):
        self.c.kktsolver_setrhs(rx, rz)
        self.g.kktsolver_setrhs(rx, rz)

    def kktsolver_solve(self, lx, lz):
        np = self.np
        n, m = self.g.n, self.g.m
        gx, gz = np.zeros(n), np.zeros(m)
        self.g.kktsolver_solve(gx, gz)
        cx, cz = (lx if lx is not None else np.zeros(n)), (lz if lz is not None else np.zeros(m))
        okc = self.c.kktsolver_solve(cx, cz)
        xc, xg = np.concatenate([cx, cz]), np.concatenate([gx, gz])
        try:
            gn = tuple(float(v) for v in self.g.h.debug_dump(17)[:3])       # HIP: ||e|| before the last step, ||b||, ||e|| after it
        except Exception:
            gn = ()
        self.log.append((self.it, float(np.max(np.abs(xg - xc)) / max(1.0, np.max(np.abs(xc)))), int(self.g.last_ir_steps), int(self.c.last_ir_steps),
                         gn, tuple(float(v) for v in getattr(self.c, "last_norms", ())), getattr(self, "kappa", float("nan"))))
        return okc

    def __getattr__(self, k):
        return getattr(self.c, k)
