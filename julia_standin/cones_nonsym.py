"""The non-symmetric cones of the reference (Exponential, Power, Generalized Power) for the stand-in IPM caller.

Like julia_standin/cones.py this is host-side CALLER code that the reference keeps in Julia:
test infrastructure, not the accelerated path.  It exists so that the KKT path is driven end to end by what these cones hand to
it -- 3 x 3 dense Hs blocks (``get_Hs`` of the Exponential / Power cone, coneops_expcone.jl:92-100, coneops_powcone.jl:94-102)
and the rank-3 expansion data of the Generalized Power cone (coneops_genpowcone.jl:91-108 + directldl_datamaps.jl:146-167) --
and by the reference's known answers for such problems (test/OptTests/basic_exp.jl, basic_pow.jl, basic_genpow.jl).

Restated from
  src/cones/coneops_expcone.jl, coneops_powcone.jl, coneops_genpowcone.jl, coneops_nonsymmetric_common.jl,
  src/cones/cone_types.jl:201-320, src/utils/mathutils.jl:12-18 (logsafe), :427-466 (3 x 3 Cholesky)
with the same arithmetic (order of operations of every formula kept; numpy float64, 0-based indices)."""
from __future__ import annotations

import math

import numpy as np

_EPS = float(np.finfo(np.float64).eps)
_FLOATMAX = float(np.finfo(np.float64).max)


def logsafe(v):  # mathutils.jl:12-18
    if v < 0:
        return -_FLOATMAX
    if v == 0:
        return -math.inf
    return math.log(v)


def _chol3_factor(A):  # mathutils.jl:427-451; None = not positive definite
    L = np.zeros((3, 3))
    t = A[0, 0]
    if t <= 0:
        return None
    L[0, 0] = math.sqrt(t)
    L[1, 0] = A[1, 0] / L[0, 0]
    t = A[1, 1] - L[1, 0] * L[1, 0]
    if t <= 0:
        return None
    L[1, 1] = math.sqrt(t)
    L[2, 0] = A[2, 0] / L[0, 0]
    L[2, 1] = (A[2, 1] - L[1, 0] * L[2, 0]) / L[1, 1]
    t = A[2, 2] - L[2, 0] * L[2, 0] - L[2, 1] * L[2, 1]
    if t <= 0:
        return None
    L[2, 2] = math.sqrt(t)
    return L


def _chol3_solve(L, b):  # mathutils.jl:455-466 (the unrolled substitution, same expressions)
    l11, l21, l22, l31, l32, l33 = L[0, 0], L[1, 0], L[1, 1], L[2, 0], L[2, 1], L[2, 2]
    c1 = b[0] / l11
    c2 = (b[1] * l11 - b[0] * l21) / (l11 * l22)
    c3 = (b[2] * l11 * l22 - b[1] * l11 * l32 + b[0] * l21 * l32 - b[0] * l22 * l31) / (l11 * l22 * l33)
    x1 = (c1 * l22 * l33 - c2 * l21 * l33 + c3 * l21 * l32 - c3 * l22 * l31) / (l11 * l22 * l33)
    x2 = (c2 * l33 - c3 * l32) / (l22 * l33)
    x3 = c3 / l33
    return np.array([x1, x2, x3])


def backtrack_search(dq, q, alpha_init, alpha_min, step, is_in_cone):  # coneops_nonsymmetric_common.jl:5-33
    alpha = alpha_init
    while True:
        wq = q + alpha * dq
        if is_in_cone(wq):
            break
        alpha *= step
        if alpha < alpha_min:
            alpha = 0.0
            break
    return alpha


def _newton_raphson_onesided(x0, f0, f1):  # coneops_nonsymmetric_common.jl:170-192
    x = x0
    it = 0
    while it < 100:
        it += 1
        dfdx = f1(x)
        dx = -f0(x) / dfdx
        if dx < _EPS or abs(dx / x) < math.sqrt(_EPS) or abs(dfdx) < _EPS:
            break
        x += dx
    return x


def _wright_omega(z):  # coneops_expcone.jl:412-467 (Algorithm 4, section 8.4 of S. Akle Serrano's thesis, as the reference codes it)
    if z < 0:
        raise ValueError(f"argument not in supported range : {z}")
    if z < 1.0 + math.pi:
        zm1 = z - 1.0
        p = zm1
        w = 1 + 0.5 * p
        p *= zm1
        w += (1 / 16.0) * p
        p *= zm1
        w -= (1 / 192.0) * p
        p *= zm1
        w -= (1 / 3072.0) * p
        p *= zm1
        w += (13 / 61440.0) * p
    else:
        logz = logsafe(z)
        zinv = 1.0 / z
        w = z - logz
        q = logz * zinv
        w += q
        q *= zinv
        w += q * (logz / 2 - 1.0)
        # (:451 of the reference multiplies q by zinv WITHOUT storing the product: the last term uses log(z)/z^2 as well)
        w += q * (logz * logz / 3.0 - (3 / 2.0) * logz + 1.0)
    r = z - w - logsafe(w)
    for _ in range(2):
        wp1 = w + 1.0
        t = wp1 * (wp1 + (2.0 * r) / 3.0)
        w *= 1 + (r / wp1) * (t - 0.5 * r) / (t - r)
        r = (2 * w * w - 8 * w - 1) / (72.0 * (wp1 * wp1 * wp1 * wp1 * wp1 * wp1)) * r * r * r * r
    return w


class _Cone3:
    """what the Exponential and the Power cone share: three rows, a dense 3 x 3 Hs, dual or primal-dual scaling
    (coneops_nonsymmetric_common.jl:50-165)"""

    is_sparse_expandable = False
    hs_is_diagonal = False
    is_symmetric = False
    allows_primal_dual_scaling = True
    kind_code = -1          # no on-device scaling for these (include/hipkkt.h hipkkt_set_cone_types knows 0..3)
    dim = 3
    numel = 3
    degree = 3

    def __init__(self):
        self.H_dual = np.zeros((3, 3))
        self.Hs = np.zeros((3, 3))
        self.grad = np.zeros(3)
        self.z = np.zeros(3)

    def rectify_equilibration(self, delta, e):  # coneops_defaults.jl:32-44
        delta[:] = e.mean() / e
        return True

    def margins(self, z, pd):
        raise RuntimeError("This function should never be reached.")

    def scaled_unit_shift(self, z, alpha, pd):
        raise RuntimeError("This function should never be reached.")

    def set_identity_scaling(self):
        raise RuntimeError("This function should never be reached.")

    def update_scaling(self, s, z, mu, strategy="primal_dual"):  # coneops_expcone.jl:63-83, coneops_powcone.jl:65-85
        self.update_dual_grad_H(z)
        if strategy == "dual":
            self._use_dual_scaling(mu)
        else:
            self._use_primal_dual_scaling(s, z)
        self.z[:] = z
        return True

    def _use_dual_scaling(self, mu):  # :71-79
        self.Hs[:] = mu * self.H_dual

    def _use_primal_dual_scaling(self, s, z):  # :82-165
        Hs, H_dual = self.Hs, self.H_dual
        st = self.grad
        zt = self.gradient_primal(s)
        dot_sz = float(z[0] * s[0] + z[1] * s[1] + z[2] * s[2])
        mu = dot_sz / 3
        mut = float(zt[0] * st[0] + zt[1] * st[1] + zt[2] * st[2]) / 3
        ds = s + mu * st
        dz = z + mu * zt
        dot_dsz = float(ds[0] * dz[0] + ds[1] * dz[1] + ds[2] * dz[2])
        de1 = mu * mut - 1
        de2 = float(zt @ (H_dual @ zt)) - 3 * mut * mut
        if abs(de1) > math.sqrt(_EPS) and abs(de2) > _EPS and dot_sz > 0 and dot_dsz > 0:
            tmp = np.array([mut * st[i] - H_dual[i, 0] * zt[0] - H_dual[i, 1] * zt[1] - H_dual[i, 2] * zt[2] for i in range(3)])
            Hs[:] = H_dual
            for i in range(3):
                for j in range(3):
                    Hs[i, j] -= st[i] * st[j] / 3 + tmp[i] * tmp[j] / de2
            t = mu * float(np.linalg.norm(Hs))          # Frobenius norm
            assert t > 0
            axis_z = np.array([z[1] * zt[2] - z[2] * zt[1], z[2] * zt[0] - z[0] * zt[2], z[0] * zt[1] - z[1] * zt[0]])
            axis_z /= float(np.linalg.norm(axis_z))
            for i in range(3):
                for j in range(i, 3):
                    Hs[i, j] = s[i] * s[j] / dot_sz + ds[i] * ds[j] / dot_dsz + t * axis_z[i] * axis_z[j]
            Hs[1, 0] = Hs[0, 1]
            Hs[2, 0] = Hs[0, 2]
            Hs[2, 1] = Hs[1, 2]
        else:
            self._use_dual_scaling(mu)

    def get_Hs(self, block):  # pack_triu, mathutils.jl:402-412: column by column, rows 0..col
        Hs = self.Hs
        block[0] = Hs[0, 0]
        block[1] = Hs[0, 1]
        block[2] = Hs[1, 1]
        block[3] = Hs[0, 2]
        block[4] = Hs[1, 2]
        block[5] = Hs[2, 2]

    def mul_Hs(self, y, x, work):
        Hs = self.Hs
        x0, x1, x2 = x[0], x[1], x[2]
        for i in range(3):
            y[i] = Hs[i, 0] * x0 + Hs[i, 1] * x1 + Hs[i, 2] * x2

    def affine_ds(self, ds, s):
        ds[:] = s

    def combined_ds_shift(self, shift, step_z, step_s, sigma_mu):  # coneops_expcone.jl:130-148
        eta = self.higher_correction(step_s, step_z)
        shift[:] = self.grad * sigma_mu - eta

    def ds_from_dz_offset(self, out, ds, work, z):
        out[:] = ds

    def step_length(self, dz, ds, z, s, alpha_max, settings):  # coneops_expcone.jl:166-187
        step = settings.linesearch_backtrack_step
        amin = settings.min_terminate_step_length
        az = backtrack_search(dz, z, alpha_max, amin, step, self.is_dual_feasible)
        as_ = backtrack_search(ds, s, alpha_max, amin, step, self.is_primal_feasible)
        return az, as_

    def compute_barrier(self, z, s, dz, ds, alpha):  # coneops_expcone.jl:189-211
        return self.barrier_dual(z + alpha * dz) + self.barrier_primal(s + alpha * ds)


class ExponentialCone(_Cone3):
    """coneops_expcone.jl.  Primal: s3 >= s2 exp(s1 / s2), s2, s3 > 0; dual: z3 >= -z1 exp(z2 / z1 - 1), z3 > 0, z1 < 0;
    dual barrier f*(z) = -log(z2 - z1 - z1 log(z3 / -z1)) - log(-z1) - log(z3)."""

    def unit_initialization(self, z, s):  # :36-52
        s[0] = -1.051383945322714
        s[1] = 0.556409619469370
        s[2] = 1.258967884768947
        z[:] = s

    def barrier_dual(self, z):  # :223-232
        lg = logsafe(-z[2] / z[0])
        return -logsafe(-z[2] * z[0]) - logsafe(z[1] - z[0] - z[0] * lg)

    def barrier_primal(self, s):  # :234-248
        om = _wright_omega(1 - s[0] / s[1] - logsafe(s[1] / s[2]))
        om = (om - 1) * (om - 1) / om
        return -logsafe(om) - 2 * logsafe(s[1]) - logsafe(s[2]) - 3

    def is_primal_feasible(self, s):  # :253-266
        if s[2] > 0 and s[1] > 0:
            return s[1] * logsafe(s[2] / s[1]) - s[0] > 0
        return False

    def is_dual_feasible(self, z):  # :269-281
        if z[2] > 0 and z[0] < 0:
            return z[1] - z[0] - z[0] * logsafe(-z[2] / z[0]) > 0
        return False

    def gradient_primal(self, s):  # :284-297
        om = _wright_omega(1 - s[0] / s[1] - logsafe(s[1] / s[2]))
        g1 = 1.0 / ((om - 1.0) * s[1])
        g2 = g1 + g1 * logsafe(om * s[1] / s[2]) - 1.0 / s[1]
        g3 = om / ((1.0 - om) * s[2])
        return np.array([g1, g2, g3])

    def higher_correction(self, ds, v):  # :319-367 (third-order correction at the scaling point z)
        eta = np.zeros(3)
        L = _chol3_factor(self.H_dual)
        if L is None:
            return eta
        u = _chol3_solve(L, ds)
        z = self.z
        eta[1] = 1.0
        eta[2] = -z[0] / z[2]
        eta[0] = logsafe(eta[2])
        psi = z[0] * eta[0] - z[0] + z[1]
        dpu = float(eta[0] * u[0] + eta[1] * u[1] + eta[2] * u[2])
        dpv = float(eta[0] * v[0] + eta[1] * v[1] + eta[2] * v[2])
        coef = ((u[0] * (v[0] / z[0] - v[2] / z[2]) + u[2] * (z[0] * v[2] / z[2] - v[0]) / z[2]) * psi - 2 * dpu * dpv) / (psi * psi * psi)
        eta *= coef
        inv_psi2 = 1.0 / psi / psi
        eta[0] += ((1 / psi - 2 / z[0]) * u[0] * v[0] / (z[0] * z[0]) - u[2] * v[2] / (z[2] * z[2]) / psi
                   + dpu * inv_psi2 * (v[0] / z[0] - v[2] / z[2]) + dpv * inv_psi2 * (u[0] / z[0] - u[2] / z[2]))
        eta[2] += (2 * (z[0] / psi - 1) * u[2] * v[2] / (z[2] * z[2] * z[2]) - (u[2] * v[0] + u[0] * v[2]) / (z[2] * z[2]) / psi
                   + dpu * inv_psi2 * (z[0] * v[2] / (z[2] * z[2]) - v[0] / z[2]) + dpv * inv_psi2 * (z[0] * u[2] / (z[2] * z[2]) - u[0] / z[2]))
        eta /= 2
        return eta

    def update_dual_grad_H(self, z):  # :370-400
        g, H = self.grad, self.H_dual
        lg = logsafe(-z[2] / z[0])
        r = -z[0] * lg - z[0] + z[1]
        c2 = 1.0 / r
        g[0] = c2 * lg - 1 / z[0]
        g[1] = -c2
        g[2] = (c2 * z[0] - 1) / z[2]
        H[0, 0] = (r * r - z[0] * r + lg * lg * z[0] * z[0]) / (r * z[0] * z[0] * r)
        H[0, 1] = -lg / (r * r)
        H[1, 0] = H[0, 1]
        H[1, 1] = 1 / (r * r)
        H[0, 2] = (z[1] - z[0]) / (r * r * z[2])
        H[2, 0] = H[0, 2]
        H[1, 2] = -z[0] / (r * r * z[2])
        H[2, 1] = H[1, 2]
        H[2, 2] = (r * r - z[0] * r + z[0] * z[0]) / (r * r * z[2] * z[2])


class PowerCone(_Cone3):
    """coneops_powcone.jl.  Primal: s1^a s2^(1-a) >= |s3|, s1, s2 >= 0; dual: (z1/a)^a (z2/(1-a))^(1-a) >= |z3|;
    dual barrier f*(z) = -log((z1/a)^2a (z2/(1-a))^(2-2a) - z3^2) - (1-a) log z1 - a log z2."""

    def __init__(self, alpha):
        super().__init__()
        self.alpha = float(alpha)

    def unit_initialization(self, z, s):  # :36-54
        a = self.alpha
        s[0] = math.sqrt(1.0 + a)
        s[1] = math.sqrt(1.0 + (1.0 - a))
        s[2] = 0.0
        z[:] = s

    def barrier_dual(self, z):  # :228-237
        a = self.alpha
        return -logsafe((z[0] / a) ** (2 * a) * (z[1] / (1 - a)) ** (2 - 2 * a) - z[2] * z[2]) - (1 - a) * logsafe(z[0]) - a * logsafe(z[1])

    def barrier_primal(self, s):  # :239-251
        a = self.alpha
        g = self.gradient_primal(s)
        return logsafe((-g[0] / a) ** (2 * a) * (-g[1] / (1 - a)) ** (2 - 2 * a) - g[2] * g[2]) + (1 - a) * logsafe(-g[0]) + a * logsafe(-g[1]) - 3

    def is_primal_feasible(self, s):  # :256-269
        a = self.alpha
        if s[0] > 0 and s[1] > 0:
            return math.exp(2 * a * logsafe(s[0]) + 2 * (1 - a) * logsafe(s[1])) - s[2] * s[2] > 0
        return False

    def is_dual_feasible(self, z):  # :272-285
        a = self.alpha
        if z[0] > 0 and z[1] > 0:
            return math.exp(2 * a * logsafe(z[0] / a) + 2 * (1 - a) * logsafe(z[1] / (1 - a))) - z[2] * z[2] > 0
        return False

    def gradient_primal(self, s):  # :288-317
        a = self.alpha
        phi = s[0] ** (2 * a) * s[1] ** (2 - 2 * a)
        g = np.zeros(3)
        abs_s = abs(s[2])
        if abs_s > _EPS:
            g[2] = self._newton_raphson(abs_s, phi, a)
            if s[2] < 0:
                g[2] = -g[2]
            g[0] = -(a * g[2] * s[2] + 1 + a) / s[0]
            g[1] = -((1 - a) * g[2] * s[2] + 2 - a) / s[1]
        else:
            g[2] = 0.0
            g[0] = -(1 + a) / s[0]
            g[1] = -(2 - a) / s[1]
        return g

    @staticmethod
    def _newton_raphson(s3, phi, a):  # :449-478
        x0 = -1.0 / s3 + (2 * s3 + math.sqrt(phi * phi / s3 / s3 + 3 * phi)) / (phi - s3 * s3)
        t0 = -2 * a * logsafe(a) - 2 * (1 - a) * logsafe(1 - a)

        def f0(x):
            t1 = x * x
            t2 = 2 * x / s3
            return (2 * a * logsafe(2 * a * t1 + (1 + a) * t2) + 2 * (1 - a) * logsafe(2 * (1 - a) * t1 + (2 - a) * t2)
                    - logsafe(phi) - logsafe(t1 + t2) - 2 * logsafe(t2) + t0)

        def f1(x):
            t1 = x * x
            t2 = x * 2 / s3
            return 2 * a * a / (a * x + (1 + a) / s3) + 2 * (1 - a) * (1 - a) / ((1 - a) * x + (2 - a) / s3) - 2 * (x + 1 / s3) / (t1 + t2)

        return _newton_raphson_onesided(x0, f0, f1)

    def higher_correction(self, ds, v):  # :329-405
        eta = np.zeros(3)
        L = _chol3_factor(self.H_dual)
        if L is None:
            return eta
        u = _chol3_solve(L, ds)
        z, a = self.z, self.alpha
        phi = (z[0] / a) ** (2 * a) * (z[1] / (1 - a)) ** (2 - 2 * a)
        psi = phi - z[2] * z[2]
        eta[0] = 2 * a * phi / z[0]
        eta[1] = 2 * (1 - a) * phi / z[1]
        eta[2] = -2 * z[2]
        h11 = 2 * a * (2 * a - 1) * phi / (z[0] * z[0])
        h12 = 4 * a * (1 - a) * phi / (z[0] * z[1])
        h22 = 2 * (1 - a) * (1 - 2 * a) * phi / (z[1] * z[1])
        dpu = float(eta[0] * u[0] + eta[1] * u[1] + eta[2] * u[2])
        dpv = float(eta[0] * v[0] + eta[1] * v[1] + eta[2] * v[2])
        Hv = np.array([h11 * v[0] + h12 * v[1], h12 * v[0] + h22 * v[1], -2 * v[2]])
        coef = (float(u[0] * Hv[0] + u[1] * Hv[1] + u[2] * Hv[2]) * psi - 2 * dpu * dpv) / (psi * psi * psi)
        coef2 = 4 * a * (2 * a - 1) * (1 - a) * phi * (u[0] / z[0] - u[1] / z[1]) * (v[0] / z[0] - v[1] / z[1]) / psi
        inv_psi2 = 1 / psi / psi
        eta[0] = coef * eta[0] - 2 * (1 - a) * u[0] * v[0] / (z[0] * z[0] * z[0]) + coef2 / z[0] + Hv[0] * dpu * inv_psi2
        eta[1] = coef * eta[1] - 2 * a * u[1] * v[1] / (z[1] * z[1] * z[1]) - coef2 / z[1] + Hv[1] * dpu * inv_psi2
        eta[2] = coef * eta[2] + Hv[2] * dpu * inv_psi2
        Hu = np.array([h11 * u[0] + h12 * u[1], h12 * u[0] + h22 * u[1], -2 * u[2]])
        for i in range(3):
            eta[i] = (eta[i] + Hu[i] * dpv * inv_psi2) / 2
        return eta

    def update_dual_grad_H(self, z):  # :408-442
        H, a, g = self.H_dual, self.alpha, self.grad
        phi = (z[0] / a) ** (2 * a) * (z[1] / (1 - a)) ** (2 - 2 * a)
        psi = phi - z[2] * z[2]
        gp0 = 2 * a * phi / (z[0] * psi)
        gp1 = 2 * (1 - a) * phi / (z[1] * psi)
        gp2 = -2 * z[2] / psi
        H[0, 0] = gp0 * gp0 - 2 * a * (2 * a - 1) * phi / (z[0] * z[0] * psi) + (1 - a) / (z[0] * z[0])
        H[0, 1] = gp0 * gp1 - 4 * a * (1 - a) * phi / (z[0] * z[1] * psi)
        H[1, 0] = H[0, 1]
        H[1, 1] = gp1 * gp1 - 2 * (1 - a) * (1 - 2 * a) * phi / (z[1] * z[1] * psi) + a / (z[1] * z[1])
        H[0, 2] = gp0 * gp2
        H[2, 0] = H[0, 2]
        H[1, 2] = gp1 * gp2
        H[2, 1] = H[1, 2]
        H[2, 2] = gp2 * gp2 + 2 / psi
        g[0] = -2 * a * phi / (z[0] * psi) - (1 - a) / z[0]
        g[1] = -2 * (1 - a) * phi / (z[1] * psi) - a / z[1]
        g[2] = 2 * z[2] / psi


class GenPowerCone:
    """coneops_genpowcone.jl: prod_i s_i^(a_i) >= ||s[d1:]||, s[:d1] >= 0.  Hs = mu (D + p p' - q q' - r r') is never formed:
    the KKT matrix carries D on the diagonal and q, r, p as three extra columns (directldl_datamaps.jl:81-167)."""

    is_sparse_expandable = True          # :14-18
    hs_is_diagonal = True                # :82-86
    is_symmetric = False
    allows_primal_dual_scaling = False   # :21
    kind_code = -1
    sparse_kind = 2                      # include/hipkkt.h HIPKKT_SPARSE_GENPOW

    def __init__(self, alpha, dim2):
        self.alpha = np.asarray(alpha, dtype=np.float64).copy()
        self.dim1 = len(self.alpha)
        self.dim2 = int(dim2)
        self.dim = self.numel = self.dim1 + self.dim2
        self.degree = self.dim1 + 1
        d = self.dim
        # GenPowerConeData, cone_types.jl:262-303
        self.grad = np.zeros(d)
        self.z = np.zeros(d)
        self.mu = 1.0
        self.p = np.zeros(d)
        self.q = np.zeros(self.dim1)
        self.r = np.zeros(self.dim2)
        self.d1 = np.zeros(self.dim1)
        self.d2 = 0.0
        self.psi = 1.0 / float(np.dot(self.alpha, self.alpha))

    def rectify_equilibration(self, delta, e):
        delta[:] = e.mean() / e
        return True

    def margins(self, z, pd):
        raise RuntimeError("This function should never be reached.")

    def scaled_unit_shift(self, z, alpha, pd):
        raise RuntimeError("This function should never be reached.")

    def set_identity_scaling(self):
        raise RuntimeError("This function should never be reached.")

    def unit_initialization(self, z, s):  # :35-54
        s[: self.dim1] = np.sqrt(1.0 + self.alpha)
        s[self.dim1:] = 0.0
        z[:] = s

    def update_scaling(self, s, z, mu, strategy="dual"):  # :64-80
        self.update_dual_grad_H(z)
        self.mu = mu
        self.z[:] = z
        return True

    def get_Hs(self, block):  # :91-108: the diagonal D only; the three extra entries belong to the expansion
        block[: self.dim1] = self.mu * self.d1
        block[self.dim1:] = self.mu * self.d2

    def mul_Hs(self, y, x, work):  # :111-134
        d1 = self.dim1
        coef_p = float(np.dot(self.p, x))
        coef_q = float(np.dot(self.q, x[:d1]))
        coef_r = float(np.dot(self.r, x[d1:]))
        y[:d1] = self.d1 * x[:d1] - coef_q * self.q
        y[d1:] = self.d2 * x[d1:] - coef_r * self.r
        y += coef_p * self.p
        y *= self.mu

    def affine_ds(self, ds, s):
        ds[:] = s

    def combined_ds_shift(self, shift, step_z, step_s, sigma_mu):  # :149-168: no third-order correction for this cone
        shift[:] = self.grad * sigma_mu

    def ds_from_dz_offset(self, out, ds, work, z):
        out[:] = ds

    def step_length(self, dz, ds, z, s, alpha_max, settings):  # :185-207
        step = settings.linesearch_backtrack_step
        amin = settings.min_terminate_step_length
        az = backtrack_search(dz, z, alpha_max, amin, step, self.is_dual_feasible)
        as_ = backtrack_search(ds, s, alpha_max, amin, step, self.is_primal_feasible)
        return az, as_

    def compute_barrier(self, z, s, dz, ds, alpha):  # :209-236
        return self.barrier_primal(s + alpha * ds) + self.barrier_dual(z + alpha * dz)

    def is_primal_feasible(self, s):  # :250-271
        d1, a = self.dim1, self.alpha
        if np.all(s[:d1] > 0):
            res = 0.0
            for i in range(d1):
                res += 2 * a[i] * logsafe(s[i])
            return math.exp(res) - float(np.sum(s[d1:] * s[d1:])) > 0
        return False

    def is_dual_feasible(self, z):  # :274-295
        d1, a = self.dim1, self.alpha
        if np.all(z[:d1] > 0):
            res = 0.0
            for i in range(d1):
                res += 2 * a[i] * logsafe(z[i] / a[i])
            return math.exp(res) - float(np.sum(z[d1:] * z[d1:])) > 0
        return False

    def barrier_primal(self, s):  # :297-315
        g = np.zeros(self.dim)
        self.gradient_primal(g, s)
        return -self.barrier_dual(-g) - self.degree

    def barrier_dual(self, z):  # :318-339
        d1, a = self.dim1, self.alpha
        res = 0.0
        for i in range(d1):
            res += 2 * a[i] * logsafe(z[i] / a[i])
        res = math.exp(res) - float(np.sum(z[d1:] * z[d1:]))
        barrier = -logsafe(res)
        for i in range(d1):
            barrier -= (1.0 - a[i]) * logsafe(z[i])
        return barrier

    def update_dual_grad_H(self, z):  # :343-396
        d1, a = self.dim1, self.alpha
        phi = 1.0
        for i in range(d1):
            phi *= (z[i] / a[i]) ** (2 * a[i])
        w = z[d1:]
        norm2w = float(np.sum(w * w))
        zeta = phi - norm2w
        assert zeta > 0
        tau = 2 * a / z[:d1]
        self.grad[:d1] = -tau * phi / zeta - (1 - a) / z[:d1]
        self.grad[d1:] = 2 * w / zeta
        p0 = math.sqrt(phi * (phi + norm2w) / 2)
        p1 = -2 * phi / p0
        q0 = math.sqrt(zeta * phi / 2)
        r1 = 2 * math.sqrt(zeta / (phi + norm2w))
        self.d1[:] = tau * phi / (zeta * z[:d1]) + (1 - a) / (z[:d1] * z[:d1])
        self.d2 = 2 / zeta
        self.p[:d1] = p0 * tau / zeta
        self.p[d1:] = p1 * w / zeta
        self.q[:] = tau * (q0 / zeta)
        self.r[:] = r1 * w / zeta

    def gradient_primal(self, g, s):  # :400-435
        d1, a = self.dim1, self.alpha
        phi = 1.0
        for i in range(d1):
            phi *= s[i] ** (2 * a[i])
        p = s[:d1]
        r = s[d1:]
        norm_r = float(np.linalg.norm(r))
        if norm_r > _EPS:
            g1 = self._newton_raphson(norm_r, p, phi, a, self.psi)
            g[d1:] = g1 * r / norm_r
            g[:d1] = -(1 + a + a * g1 * norm_r) / p
        else:
            g[d1:] = 0.0
            g[:d1] = -(1 + a) / p

    @staticmethod
    def _newton_raphson(norm_r, p, phi, a, psi):  # :446-482
        x0 = -1.0 / norm_r + (psi * norm_r + math.sqrt((phi / norm_r / norm_r + psi * psi - 1.0) * phi)) / (phi - norm_r * norm_r)

        def f0(x):
            f = -logsafe(2 * x / norm_r + x * x)
            for i in range(len(a)):
                f += 2 * a[i] * (logsafe(x * norm_r + (1 + a[i]) / a[i]) - logsafe(p[i]))
            return f

        def f1(x):
            f = -(2 * x + 2 / norm_r) / (x * x + 2 * x / norm_r)
            for i in range(len(a)):
                f += 2 * a[i] * norm_r / (norm_r * x + (1 + a[i]) / a[i])
            return f

        return _newton_raphson_onesided(x0, f0, f1)
