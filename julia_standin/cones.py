"""Cone objects: the INPUT CONTRACT of the KKT path (``get_Hs``, SOC sparse data ``eta,u,v,d``)
plus the cone algebra the stand-in IPM caller (ipm.py) needs.

This is host-side caller code that the reference keeps in Julia and that stays on the CPU
(SURVEY.md §2 rows 10/11): it is NOT the accelerated path.  It mirrors, for the symmetric cones
used by the benchmark configs (Zero, Nonnegative, SecondOrder, PSDTriangle):

  src/cones/coneops_zerocone.jl, coneops_nncone.jl, coneops_socone.jl,
  src/cones/coneops_psdtrianglecone.jl, coneops_symmetric_common.jl,
  src/cones/coneops_compositecone.jl, compositecone_type.jl, cone_api.jl:96-153

All vectors are numpy float64; indices 0-based.
"""
from __future__ import annotations

import math

import numpy as np

FLOATMAX = float(np.finfo(np.float64).max)
SOC_NO_EXPANSION_MAX_SIZE = 4  # cone_types.jl:101


def _logsafe(v):  # mathutils.jl:12-18
    if v < 0:
        return -FLOATMAX
    return math.log(v) if v > 0 else -math.inf


from clarabel_jl_amd.cone_api import (ExponentialConeT, GenPowerConeT, NonnegativeConeT, PowerConeT, PSDTriangleConeT,  # noqa: F401
                                       SecondOrderConeT, ZeroConeT, cones_new_collapsed, nvars, triangular_number)
from .cones_nonsym import ExponentialCone, GenPowerCone, PowerCone  # noqa: E402


# ------------------------------------------------------------------ concrete cones
class ZeroCone:
    """coneops_zerocone.jl"""

    is_sparse_expandable = False
    hs_is_diagonal = True

    kind_code = 0   # include/hipkkt.h hipkkt_set_cone_types

    def __init__(self, dim):
        self.dim = dim
        self.numel = dim
        self.degree = 0

    def rectify_equilibration(self, delta, e):
        delta[:] = 1.0
        return False

    def margins(self, z, pd):
        return FLOATMAX, 0.0

    def scaled_unit_shift(self, z, alpha, pd):
        if pd == "primal":
            z[:] = 0.0

    def set_identity_scaling(self):
        pass

    def update_scaling(self, s, z, mu):
        return True

    def get_Hs(self, block):
        block[:] = 0.0

    def mul_Hs(self, y, x, work):
        y[:] = 0.0

    def affine_ds(self, ds, s):
        ds[:] = 0.0

    def combined_ds_shift(self, shift, step_z, step_s, sigma_mu):
        shift[:] = 0.0

    def ds_from_dz_offset(self, out, ds, work, z):
        out[:] = 0.0

    def step_length(self, dz, ds, z, s, alpha_max):
        return alpha_max, alpha_max

    def unit_initialization(self, z, s):  # coneops_zerocone.jl unit_initialization!
        s[:] = 0.0
        z[:] = 0.0

    def compute_barrier(self, z, s, dz, ds, alpha):
        return 0.0


class NonnegativeCone:
    """coneops_nncone.jl"""

    is_sparse_expandable = False
    hs_is_diagonal = True

    kind_code = 1   # include/hipkkt.h hipkkt_set_cone_types

    def __init__(self, dim):
        self.dim = dim
        self.numel = dim
        self.degree = dim
        self.w = np.zeros(dim)
        self.lam = np.zeros(dim)

    def rectify_equilibration(self, delta, e):
        delta[:] = 1.0
        return False

    def margins(self, z, pd):  # :19-37
        alpha = float(z.min()) if z.size else FLOATMAX
        beta = float(z[z > 0].sum())
        return alpha, beta

    def scaled_unit_shift(self, z, alpha, pd):
        z += alpha

    def set_identity_scaling(self):
        self.w[:] = 1.0

    def update_scaling(self, s, z, mu):  # :77-89
        np.sqrt(s * z, out=self.lam)
        np.sqrt(s / z, out=self.w)
        return True

    def get_Hs(self, block):  # :91-101
        np.multiply(self.w, self.w, out=block)

    def mul_Hs(self, y, x, work):  # :104-113
        y[:] = self.w * (self.w * x)

    def affine_ds(self, ds, s):
        ds[:] = self.lam * self.lam

    def mul_W(self, y, x):
        y[:] = x * self.w

    def mul_Winv(self, y, x):
        y[:] = x / self.w

    def circ_op(self, x, y, z):
        x[:] = y * z

    def combined_ds_shift(self, shift, step_z, step_s, sigma_mu):
        _combined_ds_shift_symmetric(self, shift, step_z, step_s, sigma_mu)

    def ds_from_dz_offset(self, out, ds, work, z):  # :140-148
        out[:] = ds / z

    def step_length(self, dz, ds, z, s, alpha_max):  # :151-170
        az = alpha_max
        as_ = alpha_max
        neg = dz < 0
        if neg.any():
            az = min(az, float((-z[neg] / dz[neg]).min()))
        neg = ds < 0
        if neg.any():
            as_ = min(as_, float((-s[neg] / ds[neg]).min()))
        return az, as_

    def unit_initialization(self, z, s):  # coneops_nncone.jl unit_initialization!
        s[:] = 1.0
        z[:] = 1.0

    def compute_barrier(self, z, s, dz, ds, alpha):  # coneops_nncone.jl compute_barrier: -sum log(s_i z_i) at the shifted point
        barrier = 0.0
        for i in range(self.dim):
            barrier -= _logsafe((s[i] + alpha * ds[i]) * (z[i] + alpha * dz[i]))
        return barrier


def _soc_residual(z):  # coneops_socone.jl:395-399
    z1 = float(np.linalg.norm(z[1:]))
    return (z[0] - z1) * (z[0] + z1)


def _sqrt_soc_residual(z):
    r = _soc_residual(z)
    return math.sqrt(r) if r > 0.0 else 0.0


class SecondOrderCone:
    """coneops_socone.jl; sparse rank-2 expansion data for dim > 4 (cone_types.jl:86-118)."""

    kind_code = 2   # include/hipkkt.h hipkkt_set_cone_types

    def __init__(self, dim):
        if dim < 2:
            raise ValueError("dimension must be >= 2")
        self.dim = dim
        self.numel = dim
        self.degree = 1
        self.w = np.zeros(dim)
        self.lam = np.zeros(dim)
        self.eta = 0.0
        self.is_sparse_expandable = dim > SOC_NO_EXPANSION_MAX_SIZE
        self.hs_is_diagonal = self.is_sparse_expandable  # :194-198
        if self.is_sparse_expandable:
            self.u = np.zeros(dim)
            self.v = np.zeros(dim)
            self.d = 0.0

    def rectify_equilibration(self, delta, e):  # coneops_defaults.jl:32-44
        delta[:] = e.mean() / e
        return True

    def margins(self, z, pd):
        alpha = float(z[0] - np.linalg.norm(z[1:]))
        return alpha, max(0.0, alpha)

    def scaled_unit_shift(self, z, alpha, pd):
        z[0] += alpha

    def set_identity_scaling(self):  # :57-73
        self.w[:] = 0.0
        self.w[0] = 1.0
        self.eta = 1.0
        if self.is_sparse_expandable:
            self.d = 0.5
            self.u[:] = 0.0
            self.u[0] = math.sqrt(0.5)
            self.v[:] = 0.0

    def update_scaling(self, s, z, mu):  # :75-157
        zscale = _sqrt_soc_residual(z)
        sscale = _sqrt_soc_residual(s)
        if zscale == 0.0 or sscale == 0.0:
            return False
        self.eta = math.sqrt(sscale / zscale)
        w = self.w
        w[:] = s / sscale
        w[0] += z[0] / zscale
        w[1:] -= z[1:] / zscale
        wscale = _sqrt_soc_residual(w)
        if wscale == 0.0:
            return False
        w /= wscale
        w1sq = float(np.dot(w[1:], w[1:]))
        w[0] = math.sqrt(1.0 + w1sq)
        gamma = 0.5 * wscale
        lam = self.lam
        lam[0] = gamma
        lam[1:] = ((gamma + z[0] / zscale) / sscale) * s[1:] + ((gamma + s[0] / sscale) / zscale) * z[1:]
        lam[1:] *= 1.0 / (s[0] / sscale + z[0] / zscale + 2.0 * gamma)
        lam *= math.sqrt(sscale * zscale)
        if self.is_sparse_expandable:
            alpha = 2.0 * w[0]
            wsq = w[0] * w[0] + w1sq
            wsqinv = 1.0 / wsq
            self.d = wsqinv / 2.0
            u0 = math.sqrt(wsq - self.d)
            u1 = alpha / u0
            v1 = math.sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv))
            self.u[0] = u0
            self.u[1:] = u1 * w[1:]
            self.v[0] = 0.0
            self.v[1:] = v1 * w[1:]
        return True

    def get_Hs(self, block):  # :159-192
        eta2 = self.eta * self.eta
        if self.is_sparse_expandable:
            block[:] = eta2
            block[0] *= self.d
        else:
            w = self.w
            r2 = math.sqrt(2.0)
            block[0] = (r2 * w[0] - 1.0) * (r2 * w[0] + 1.0)
            h = 1
            for col in range(1, self.dim):
                wc = w[col]
                for row in range(col + 1):
                    block[h] = 2.0 * w[row] * wc
                    h += 1
                block[h - 1] += 1.0
            block *= eta2

    def mul_Hs(self, y, x, work):  # :200-215
        c = 2.0 * float(np.dot(self.w, x))
        y[:] = x
        y[0] = -x[0]
        y += c * self.w
        y *= self.eta * self.eta

    def circ_op(self, x, y, z):  # :364-378
        x0 = float(np.dot(y, z))
        y0, z0 = y[0], z[0]
        x[1:] = y0 * z[1:] + z0 * y[1:]
        x[0] = x0

    def affine_ds(self, ds, s):
        self.circ_op(ds, self.lam, self.lam)

    def mul_W(self, y, x):  # :300-322
        w = self.w
        zeta = float(np.dot(w[1:], x[1:]))
        c = x[0] + zeta / (1.0 + w[0])
        y0 = self.eta * (w[0] * x[0] + zeta)
        y[1:] = self.eta * (x[1:] + c * w[1:])
        y[0] = y0

    def mul_Winv(self, y, x):  # :324-347
        w = self.w
        zeta = float(np.dot(w[1:], x[1:]))
        c = -x[0] + zeta / (1.0 + w[0])
        ei = 1.0 / self.eta
        y0 = ei * (w[0] * x[0] - zeta)
        y[1:] = ei * (x[1:] + c * w[1:])
        y[0] = y0

    def combined_ds_shift(self, shift, step_z, step_s, sigma_mu):
        _combined_ds_shift_symmetric(self, shift, step_z, step_s, sigma_mu)

    def ds_from_dz_offset(self, out, ds, work, z):  # :241-268
        resz = _soc_residual(z)
        lam, w = self.lam, self.w
        l1ds1 = float(np.dot(lam[1:], ds[1:]))
        w1ds1 = float(np.dot(w[1:], ds[1:]))
        out[:] = -z
        out[0] = z[0]
        c = lam[0] * ds[0] - l1ds1
        out *= c / resz
        out[0] += self.eta * w1ds1
        out[1:] += self.eta * (ds[1:] + w1ds1 / (1.0 + w[0]) * w[1:])
        out *= 1.0 / lam[0]

    def step_length(self, dz, ds, z, s, alpha_max):  # :270-286
        return (_step_length_soc_component(z, dz, alpha_max),
                _step_length_soc_component(s, ds, alpha_max))

    def unit_initialization(self, z, s):  # coneops_socone.jl unit_initialization!
        s[:] = 0.0
        z[:] = 0.0
        self.scaled_unit_shift(s, 1.0, "primal")
        self.scaled_unit_shift(z, 1.0, "dual")

    def compute_barrier(self, z, s, dz, ds, alpha):  # :288-305
        res_s = _soc_residual(s + alpha * ds)
        res_z = _soc_residual(z + alpha * dz)
        if res_s > 0 and res_z > 0:
            return -_logsafe(res_s * res_z) / 2
        return math.inf


def _step_length_soc_component(x, y, alpha_max):  # coneops_socone.jl:443-512
    if x[0] >= 0 and y[0] < 0:
        alpha_max = min(alpha_max, -x[0] / y[0])
    a = _soc_residual(y)
    b = 2.0 * (x[0] * y[0] - float(np.dot(x[1:], y[1:])))
    c = max(0.0, _soc_residual(x))
    d = b * b - 4.0 * a * c
    if (a > 0 and b > 0) or d < 0:
        return alpha_max
    if a == 0:
        return alpha_max
    if c == 0:
        return alpha_max if a >= 0 else 0.0
    t = (-b - math.sqrt(d)) if b >= 0 else (-b + math.sqrt(d))
    r1 = (2.0 * c) / t
    r2 = t / (2.0 * a)
    r1 = FLOATMAX if r1 < 0 else r1
    r2 = FLOATMAX if r2 < 0 else r2
    return min(alpha_max, r1, r2)


_ISQRT2 = 1.0 / math.sqrt(2.0)


class PSDTriangleCone:
    """coneops_psdtrianglecone.jl (scaled upper-triangle vectorisation, column-major packed)."""

    is_sparse_expandable = False
    hs_is_diagonal = False

    kind_code = 3   # include/hipkkt.h hipkkt_set_cone_types

    def __init__(self, n):
        self.n = n
        self.numel = triangular_number(n)
        self.degree = n
        self.lam = np.zeros(n)
        self.lisqrt = np.zeros(n)
        self.R = np.zeros((n, n))
        self.Rinv = np.zeros((n, n))
        self.Hs = np.zeros((self.numel, self.numel))
        self.RRt = np.eye(n)          # W = R R^T of the current scaling (what skron! is applied to)
        self._hs_valid = False        # Hs is formed lazily: the HIP path builds the block on the device from RRt
        # packed index helpers: element k <-> (row[k], col[k]), row <= col, column-major packed
        # (= row-major packed lower triangle with the roles of row/col swapped)
        il = np.tril_indices(n)
        self._r = il[1].astype(np.int64)
        self._c = il[0].astype(np.int64)
        ilh = np.tril_indices(self.numel)
        self._hs_r = ilh[1]
        self._hs_c = ilh[0]
        self._isdiag = self._r == self._c
        self._diagidx = np.array([triangular_number(k + 1) - 1 for k in range(n)], dtype=np.int64)
        self._f = np.where(self._isdiag, 1.0, math.sqrt(2.0))

    # -- svec helpers (:468-500)
    def svec_to_mat(self, x):
        M = np.zeros((self.n, self.n))
        vals = np.where(self._isdiag, x, x * _ISQRT2)
        M[self._r, self._c] = vals
        M[self._c, self._r] = vals
        return M

    def mat_to_svec(self, M):
        return np.where(self._isdiag, M[self._r, self._c], (M[self._r, self._c] + M[self._c, self._r]) * _ISQRT2)

    def rectify_equilibration(self, delta, e):
        delta[:] = e.mean() / e
        return True

    def margins(self, z, pd):  # :8-28
        if z.size == 0:
            return FLOATMAX, 0.0
        ev = np.linalg.eigvalsh(self.svec_to_mat(z))
        return float(ev.min()), float(ev[ev > 0].sum())

    def scaled_unit_shift(self, z, alpha, pd):  # :31-45
        z[self._diagidx] += alpha

    def set_identity_scaling(self):  # :66-75
        self.R[:] = np.eye(self.n)
        self.Rinv[:] = np.eye(self.n)
        self.RRt[:] = np.eye(self.n)
        self.Hs[:] = np.eye(self.numel)
        self._hs_valid = True

    def update_scaling(self, s, z, mu):  # :78-143
        if s.size == 0:
            return True
        S = self.svec_to_mat(s)
        Z = self.svec_to_mat(z)
        try:
            L1 = np.linalg.cholesky(S)
            L2 = np.linalg.cholesky(Z)
        except np.linalg.LinAlgError:
            return False
        U, sv, Vt = np.linalg.svd(L2.T @ L1)
        self.lam[:] = sv
        self.lisqrt[:] = 1.0 / np.sqrt(sv)
        self.R[:] = (L1 @ Vt.T) * self.lisqrt[None, :]
        self.Rinv[:] = self.lisqrt[:, None] * (U.T @ L2.T)
        self.RRt[:] = self.R @ self.R.T
        self._hs_valid = False        # skron! (:153-161) is deferred until somebody asks for the block
        return True

    def _skron(self, A):
        """triu(A (x)_s A), :502-540, vectorised with the reference's four cases and association:
        i != j, k != l: A_ik A_jl + A_il A_jk;  i == j, k != l: (sqrt2 A_jl) A_jk;
        i != j, k == l: (sqrt2 A_il) A_jk;      i == j, k == l: A_jl A_jl.
        Only row <= col is meaningful (the reference fills only the upper triangle); get_Hs packs that part."""
        i, j = self._r[:, None], self._c[:, None]
        k, l = self._r[None, :], self._c[None, :]
        ij_eq, kl_eq = self._isdiag[:, None], self._isdiag[None, :]
        sqrt2 = math.sqrt(2.0)
        Ajl, Ajk, Ail, Aik = A[j, l], A[j, k], A[i, l], A[i, k]
        self.Hs[:] = np.where(ij_eq, np.where(kl_eq, Ajl * Ajl, (sqrt2 * Ajl) * Ajk),
                              np.where(kl_eq, (sqrt2 * Ail) * Ajk, Aik * Ajl + Ail * Ajk))

    def get_Hs(self, block):  # :153-161 -> pack_triu (mathutils.jl:402-412)
        if not self._hs_valid:
            self._skron(self.RRt)
            self._hs_valid = True
        block[:] = self.Hs[self._hs_r, self._hs_c]

    def _mul_Wx_inner(self, transpose, x, Rx):  # :404-432
        X = self.svec_to_mat(x)
        if transpose:
            Y = Rx @ (X @ Rx.T)
        else:
            Y = (Rx.T @ X) @ Rx
        return self.mat_to_svec(Y)

    def mul_W(self, y, x, transpose=False):
        y[:] = self._mul_Wx_inner(transpose, x, self.R)

    def mul_Winv(self, y, x, transpose=False):
        y[:] = self._mul_Wx_inner(transpose, x, self.Rinv)

    def mul_Hs(self, y, x, work):  # :164-186
        work[:] = self._mul_Wx_inner(False, x, self.R)
        y[:] = self._mul_Wx_inner(True, work, self.R)

    def affine_ds(self, ds, s):  # :189-205
        ds[:] = 0.0
        ds[self._diagidx] = self.lam * self.lam

    def circ_op(self, x, y, z):  # :343-364
        Y = self.svec_to_mat(y)
        Z = self.svec_to_mat(z)
        X = 0.5 * (Y @ Z + Z @ Y)
        x[:] = self.mat_to_svec(X)

    def lambda_inv_circ_op(self, x, z):  # :318-336
        Z = self.svec_to_mat(z)
        X = 2.0 * Z / (self.lam[:, None] + self.lam[None, :])
        x[:] = self.mat_to_svec(X)

    def combined_ds_shift(self, shift, step_z, step_s, sigma_mu):
        _combined_ds_shift_symmetric(self, shift, step_z, step_s, sigma_mu)

    def ds_from_dz_offset(self, out, ds, work, z):  # coneops_symmetric_common.jl:39-52
        self.lambda_inv_circ_op(work, ds)
        self.mul_W(out, work, transpose=True)

    def _step_component(self, d, alpha_max):  # :434-466
        if d.size == 0:
            return alpha_max
        D = self.svec_to_mat(d)
        D = self.lisqrt[:, None] * D * self.lisqrt[None, :]
        g = float(np.linalg.eigvalsh(D).min())
        if g < 0:
            return min(1.0 / (-g), alpha_max)
        return alpha_max

    def step_length(self, dz, ds, z, s, alpha_max):  # :230-254
        d = np.empty(self.numel)
        self.mul_W(d, dz, transpose=False)
        az = self._step_component(d, alpha_max)
        self.mul_Winv(d, ds, transpose=True)
        as_ = self._step_component(d, alpha_max)
        return az, as_

    def unit_initialization(self, z, s):  # coneops_psdtrianglecone.jl unit_initialization!
        s[:] = 0.0
        z[:] = 0.0
        self.scaled_unit_shift(s, 1.0, "primal")
        self.scaled_unit_shift(z, 1.0, "dual")

    def _logdet_barrier(self, x, dx, alpha):  # :272-290: log det by Cholesky; +Inf when the shifted point is not positive definite
        try:
            L = np.linalg.cholesky(self.svec_to_mat(x + alpha * dx))
        except np.linalg.LinAlgError:
            return math.inf
        return 2.0 * float(np.sum(np.log(np.diag(L))))

    def compute_barrier(self, z, s, dz, ds, alpha):  # :256-270
        barrier = 0.0
        barrier -= self._logdet_barrier(z, dz, alpha)
        barrier -= self._logdet_barrier(s, ds, alpha)
        return barrier


def _combined_ds_shift_symmetric(K, shift, step_z, step_s, sigma_mu):
    """coneops_symmetric_common.jl:1-36: shift = W^-1 ds o W dz - sigma*mu*e (step_z/step_s overwritten)."""
    tmp = shift
    tmp[:] = step_z
    if isinstance(K, PSDTriangleCone):
        K.mul_W(step_z, tmp, transpose=False)
        tmp[:] = step_s
        K.mul_Winv(step_s, tmp, transpose=True)
    else:
        K.mul_W(step_z, tmp.copy())
        tmp[:] = step_s
        K.mul_Winv(step_s, tmp.copy())
    K.circ_op(shift, step_s, step_z)
    K.scaled_unit_shift(shift, -sigma_mu, "primal")


def make_cone(spec):
    if isinstance(spec, ZeroConeT):
        return ZeroCone(spec.dim)
    if isinstance(spec, NonnegativeConeT):
        return NonnegativeCone(spec.dim)
    if isinstance(spec, SecondOrderConeT):
        return SecondOrderCone(spec.dim)
    if isinstance(spec, PSDTriangleConeT):
        return PSDTriangleCone(spec.dim)
    if isinstance(spec, ExponentialConeT):
        return ExponentialCone()
    if isinstance(spec, PowerConeT):
        return PowerCone(spec.alpha)
    if isinstance(spec, GenPowerConeT):
        return GenPowerCone(spec.alpha, spec.dim2)
    raise TypeError(f"unsupported cone spec {spec!r}")


class CompositeCone:
    """compositecone_type.jl:28-66 + coneops_compositecone.jl."""

    def __init__(self, specs):
        self.cones = [make_cone(s) for s in specs]
        self.numel = sum(c.numel for c in self.cones)
        self.degree = sum(c.degree for c in self.cones)
        self.rng_cones = []
        self.rng_blocks = []
        a = 0
        b = 0
        for c in self.cones:
            self.rng_cones.append(slice(a, a + c.numel))
            a += c.numel
            nb = c.numel if c.hs_is_diagonal else triangular_number(c.numel)
            self.rng_blocks.append(slice(b, b + nb))
            b += nb
        self.nnz_Hs = b
        self._is_symmetric = all(getattr(c, "is_symmetric", True) for c in self.cones)      # compositecone_type.jl:54-60
        from clarabel_jl_amd.settings import Settings   # line-search constants of the non-symmetric cones: the defaults until Solver passes its own
        self._settings = Settings()

    def __iter__(self):
        return iter(self.cones)

    def __len__(self):
        return len(self.cones)

    def is_symmetric(self):
        return self._is_symmetric

    def allows_primal_dual_scaling(self):  # coneops_compositecone.jl:23-25
        return all(getattr(c, "allows_primal_dual_scaling", True) for c in self.cones)

    def use_settings(self, settings):
        self._settings = settings

    def unit_initialization(self, z, s):  # :78-89
        for c, r in zip(self.cones, self.rng_cones):
            c.unit_initialization(z[r], s[r])

    def compute_barrier(self, z, s, dz, ds, alpha):  # :254-266
        barrier = 0.0
        for c, r in zip(self.cones, self.rng_cones):
            barrier += c.compute_barrier(z[r], s[r], dz[r], ds[r], alpha)
        return barrier

    def rectify_equilibration(self, delta, e):  # :26-45
        any_changed = False
        delta[:] = 1.0
        for c, r in zip(self.cones, self.rng_cones):
            any_changed |= c.rectify_equilibration(delta[r], e[r])
        return any_changed

    def margins(self, z, pd):
        alpha, beta = FLOATMAX, 0.0
        for c, r in zip(self.cones, self.rng_cones):
            a, b = c.margins(z[r], pd)
            alpha = min(alpha, a)
            beta += b
        return alpha, beta

    def scaled_unit_shift(self, z, alpha, pd):
        for c, r in zip(self.cones, self.rng_cones):
            c.scaled_unit_shift(z[r], alpha, pd)

    def set_identity_scaling(self):
        for c in self.cones:
            c.set_identity_scaling()

    def update_scaling(self, s, z, mu, strategy="primal_dual"):  # :103-120
        for c, r in zip(self.cones, self.rng_cones):
            if getattr(c, "is_symmetric", True):
                ok = c.update_scaling(s[r], z[r], mu)
            else:
                ok = c.update_scaling(s[r], z[r], mu, strategy)
            if not ok:
                return False
        return True

    def get_Hs(self, hsblocks, skip=()):  # :122-131; `skip`: cones whose block the caller forms elsewhere
        for c, r in zip(self.cones, self.rng_blocks):
            if c not in skip:
                c.get_Hs(hsblocks[r])

    def mul_Hs(self, y, x, work):
        for c, r in zip(self.cones, self.rng_cones):
            c.mul_Hs(y[r], x[r], work[r])

    def affine_ds(self, ds, s):
        for c, r in zip(self.cones, self.rng_cones):
            c.affine_ds(ds[r], s[r])

    def combined_ds_shift(self, shift, step_z, step_s, sigma_mu):
        for c, r in zip(self.cones, self.rng_cones):
            c.combined_ds_shift(shift[r], step_z[r], step_s[r], sigma_mu)

    def ds_from_dz_offset(self, out, ds, work, z):
        for c, r in zip(self.cones, self.rng_cones):
            c.ds_from_dz_offset(out[r], ds[r], work[r], z[r])

    def step_length(self, dz, ds, z, s, alpha_max):  # :216-252: the symmetric cones first, then (from a step slightly short
        alpha = alpha_max                                # of 1, so that the logarithms stay finite) the non-symmetric ones
        for c, r in zip(self.cones, self.rng_cones):
            if getattr(c, "is_symmetric", True):
                az, as_ = c.step_length(dz[r], ds[r], z[r], s[r], alpha)
                alpha = min(alpha, az, as_)
        if not self._is_symmetric:
            alpha = min(alpha, 1.0 - math.sqrt(float(np.finfo(np.float64).eps)))
            for c, r in zip(self.cones, self.rng_cones):
                if not getattr(c, "is_symmetric", True):
                    az, as_ = c.step_length(dz[r], ds[r], z[r], s[r], alpha, self._settings)
                    alpha = min(alpha, az, as_)
        return alpha, alpha

    # ---- what the KKT structure needs to know (directldl_kkt_assembly.jl:66-86)
    def kkt_descriptors(self):
        """Per cone (numel, hs_dense, sparse_kind, dim1) for the C ABI / the oracle."""
        numel = np.array([c.numel for c in self.cones], dtype=np.int64)
        hs_dense = np.array([0 if c.hs_is_diagonal else 1 for c in self.cones], dtype=np.int32)
        sparse_kind = np.array([getattr(c, "sparse_kind", 1) if c.is_sparse_expandable else 0 for c in self.cones], dtype=np.int32)
        dim1 = np.array([getattr(c, "dim1", 0) if getattr(c, "sparse_kind", 1) == 2 else 0 for c in self.cones], dtype=np.int64)
        return numel, hs_dense, sparse_kind, dim1

    def kkt_cone_kinds(self):
        """cone type codes for the plugin's on-device scaling"""
        return np.array([c.kind_code for c in self.cones], dtype=np.int32)
