"""Stand-in for the reference's UNTOUCHED Julia caller of the KKT path.

north_star keeps ``solver.jl`` / ``variables.jl`` / ``residuals.jl`` / ``kktsystem.jl`` in Julia; there
is no ``julia`` in this image, so to (a) pin the oracle on the reference's end-to-end known
answers and (b) measure "IPM iterations/s" at all, this module restates that caller on the host
in numpy.  It is NOT the accelerated path: every call into the hot path goes through the
``kktsolver_*`` interface of src/kktsolvers/kktsolver_defaults.jl:2-47, served either by
``HipKKTSolver`` (product, HIP kernels behind the C ABI) or, in tests / the CPU baseline, by the
oracle.  Symmetric cones only (Zero, NN, SOC, PSD) — all the benchmark configs need.

Mirrors (paths relative to /root/reference/src):
  solver.jl:89-153 (setup), :189-380 (main loop), :383-514 (start, step length, checkpoints)
  kktsystem.jl (reduced-system algebra around the KKT solves)
  variables.jl, residuals.jl, info.jl, solution.jl:1-50
  problemdata.jl:3-88 (copy / triu / b cap), :133-243 (Ruiz equilibration)
  data_updating.jl:56-130 (update_P!/A!/q!/b!)
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .cones import CompositeCone, _logsafe, cones_new_collapsed
from clarabel_jl_amd.settings import Settings

INFINITY = 1e20  # Clarabel.jl:15
_EPS = float(np.finfo(np.float64).eps)
FLOATMAX = float(np.finfo(np.float64).max)

# statuscodes.jl
UNSOLVED = "UNSOLVED"
SOLVED = "SOLVED"
PRIMAL_INFEASIBLE = "PRIMAL_INFEASIBLE"
DUAL_INFEASIBLE = "DUAL_INFEASIBLE"
ALMOST_SOLVED = "ALMOST_SOLVED"
ALMOST_PRIMAL_INFEASIBLE = "ALMOST_PRIMAL_INFEASIBLE"
ALMOST_DUAL_INFEASIBLE = "ALMOST_DUAL_INFEASIBLE"
MAX_ITERATIONS = "MAX_ITERATIONS"
MAX_TIME = "MAX_TIME"
NUMERICAL_ERROR = "NUMERICAL_ERROR"
INSUFFICIENT_PROGRESS = "INSUFFICIENT_PROGRESS"

_INFEASIBLE = {PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, ALMOST_PRIMAL_INFEASIBLE, ALMOST_DUAL_INFEASIBLE}
_ERRORED = {NUMERICAL_ERROR, INSUFFICIENT_PROGRESS}


def _norm_inf(v):
    return float(np.max(np.abs(v))) if v.size else 0.0


def _norm_scaled(x, v):  # mathutils.jl:58-80 (2-norm of x.*v)
    return float(np.linalg.norm(x * v))


def _clip(s, lo, hi):
    return np.minimum(np.maximum(s, lo), hi)


def _quad_form(x, Ptriu, y):
    """x' * Symmetric(P,:U) * y, mathutils.jl:299-337."""
    Py = Ptriu @ y
    Pty = Ptriu.T @ y
    return float(np.dot(x, Py) + np.dot(x, Pty) - np.dot(x, Ptriu.diagonal() * y))


def _symv(Ptriu, x):
    return Ptriu @ x + Ptriu.T @ x - Ptriu.diagonal() * x


class ProblemData:
    """problemdata.jl:3-88 — local copy, triu(P), cone collapse, b capped at INFINITY.

    The presolver (presolver.jl, drops NN rows with b >= 1e20) and chordal decomposition are
    out of scope; none of the configs trigger them."""

    def __init__(self, P, q, A, b, cone_specs, settings: Settings):
        self.cone_specs = cones_new_collapsed(list(cone_specs))
        P = sp.csc_matrix(P, dtype=np.float64)
        P = sp.triu(P, format="csc")
        P.sort_indices()
        A = sp.csc_matrix(A, dtype=np.float64).copy()
        A.sort_indices()
        self.P = P
        self.A = A
        self.q = np.array(q, dtype=np.float64).copy()
        self.b = np.minimum(np.array(b, dtype=np.float64), INFINITY)
        self.m, self.n = A.shape
        self.d = np.ones(self.n)
        self.dinv = np.ones(self.n)
        self.e = np.ones(self.m)
        self.einv = np.ones(self.m)
        self.c = 1.0
        self.normq = _norm_inf(self.q)
        self.normb = _norm_inf(self.b)

    # -- Ruiz equilibration, problemdata.jl:133-221
    def equilibrate(self, cones: CompositeCone, settings: Settings):
        if not settings.equilibrate_enable:
            return
        P, A = self.P, self.A
        smin, smax = settings.equilibrate_min_scaling, settings.equilibrate_max_scaling
        n, m = self.n, self.m
        Pcol = np.repeat(np.arange(n), np.diff(P.indptr))
        Acol = np.repeat(np.arange(n), np.diff(A.indptr))
        for _ in range(settings.equilibrate_max_iter):
            # kkt_col_norms!, mathutils.jl:129-141
            dwork = np.zeros(n)
            absP = np.abs(P.data)
            np.maximum.at(dwork, Pcol, absP)
            np.maximum.at(dwork, P.indices, absP)
            absA = np.abs(A.data)
            np.maximum.at(dwork, Acol, absA)
            ework = np.zeros(m)
            np.maximum.at(ework, A.indices, absA)
            dwork[dwork == 0.0] = 1.0
            ework[ework == 0.0] = 1.0
            dwork = 1.0 / np.sqrt(dwork)
            ework = 1.0 / np.sqrt(ework)
            dwork = _clip(dwork, smin / self.d, smax / self.d)
            ework = _clip(ework, smin / self.e, smax / self.e)
            self._scale_data(dwork, ework)
            self.d *= dwork
            self.e *= ework
            # cost scaling uses the NON-symmetric column norms of the stored triu(P) (:187)
            cn = np.zeros(n)
            np.maximum.at(cn, Pcol, np.abs(P.data))
            mean_col_norm_P = float(cn.sum() / n) if n else 0.0
            inf_norm_q = _norm_inf(self.q)
            if mean_col_norm_P != 0.0 and inf_norm_q != 0.0:
                ctmp = 1.0 / max(inf_norm_q, mean_col_norm_P)
                ctmp = float(_clip(ctmp, smin / self.c, smax / self.c))
                P.data *= ctmp
                self.q *= ctmp
                self.c *= ctmp
        ework = np.ones(m)
        if cones.rectify_equilibration(ework, self.e):
            self._scale_data(None, ework)
            self.e *= ework
        self.dinv = 1.0 / self.d
        self.einv = 1.0 / self.e

    def _scale_data(self, d, e):  # problemdata.jl:224-243
        P, A = self.P, self.A
        if d is not None:
            Pcol = np.repeat(np.arange(self.n), np.diff(P.indptr))
            P.data *= d[P.indices] * d[Pcol]
            Acol = np.repeat(np.arange(self.n), np.diff(A.indptr))
            A.data *= e[A.indices] * d[Acol]
            self.q *= d
        else:
            A.data *= e[A.indices]
        self.b *= e

    def get_normq(self):  # problemdata.jl:90-99
        if self.normq is None:
            self.normq = _norm_inf(self.q * self.dinv) / self.c
        return self.normq

    def get_normb(self):
        if self.normb is None:
            self.normb = _norm_inf(self.b * self.einv)
        return self.normb


@dataclass
class Variables:
    x: np.ndarray
    s: np.ndarray
    z: np.ndarray
    tau: float = 1.0
    kappa: float = 1.0

    @classmethod
    def zeros(cls, n, m):
        return cls(np.zeros(n), np.zeros(m), np.zeros(m), 1.0, 1.0)

    def copy_from(self, o):
        self.x[:] = o.x
        self.s[:] = o.s
        self.z[:] = o.z
        self.tau, self.kappa = o.tau, o.kappa


@dataclass
class Residuals:
    rx: np.ndarray
    rz: np.ndarray
    rtau: float
    rx_inf: np.ndarray
    rz_inf: np.ndarray
    Px: np.ndarray
    dot_qx: float = 0.0
    dot_bz: float = 0.0
    dot_sz: float = 0.0
    dot_xPx: float = 0.0


@dataclass
class Info:
    status: str = UNSOLVED
    iterations: int = 0
    mu: float = 0.0
    sigma: float = 0.0
    step_length: float = 0.0
    cost_primal: float = 0.0
    cost_dual: float = 0.0
    res_primal: float = 0.0
    res_dual: float = 0.0
    res_primal_inf: float = 0.0
    res_dual_inf: float = 0.0
    gap_abs: float = 0.0
    gap_rel: float = 0.0
    ktratio: float = 0.0
    prev_cost_primal: float = 0.0
    prev_cost_dual: float = 0.0
    prev_res_primal: float = 0.0
    prev_res_dual: float = 0.0
    prev_gap_abs: float = 0.0
    prev_gap_rel: float = 0.0
    solve_time: float = 0.0
    timers: dict = field(default_factory=dict)


@dataclass
class Solution:
    x: np.ndarray
    z: np.ndarray
    s: np.ndarray
    status: str = UNSOLVED
    obj_val: float = float("nan")
    obj_val_dual: float = float("nan")
    iterations: int = 0
    r_prim: float = float("nan")
    r_dual: float = float("nan")
    solve_time: float = 0.0


class KKTSystem:
    """kktsystem.jl: DefaultKKTSystem.  ``kktsolver`` implements kktsolver_defaults.jl:2-47."""

    def __init__(self, kktsolver, m, n):
        self.kktsolver = kktsolver
        self.x1 = np.zeros(n)
        self.z1 = np.zeros(m)
        self.x2 = np.zeros(n)
        self.z2 = np.zeros(m)
        self.workx = np.zeros(n)
        self.workz = np.zeros(m)
        self.work_conic = np.zeros(m)
        self._multi = hasattr(kktsolver, "kktsolver_solve_multi") and getattr(kktsolver, "batch_constant_rhs", True)
        self._const_pending = False
        self._have_const_dev = False      # the plugin holds (x2, z2) of the current factorisation (device_reduced)
        self._device_scaling = bool(getattr(getattr(kktsolver, "settings", None), "device_scaling", False)) and \
            hasattr(kktsolver, "kktsolver_update_scaled")
        # SURVEY section 8(f) row N2, second half: d tau, dx, dz of kkt_solve! formed by the plugin (one PCIe round trip per kkt_solve!)
        self._device_reduced = bool(getattr(getattr(kktsolver, "settings", None), "device_reduced", False)) and \
            hasattr(kktsolver, "kktsolver_kkt_solve_reduced")
        self._rx2, self._rz2 = np.zeros((2, n)), np.zeros((2, m))
        self._lx2, self._lz2 = np.zeros((2, n)), np.zeros((2, m))

    def kkt_update(self, data, cones, sz=None):  # :62-78
        if sz is not None and self._device_scaling:
            # the plugin forms the Hs blocks / sparse-cone terms of K from the iterate (s, z) itself
            ok = self.kktsolver.kktsolver_update_scaled(cones, sz[0], sz[1])
        else:
            ok = self.kktsolver.kktsolver_update(cones)
        if not ok:
            return False
        self._have_const_dev = False
        if self._multi:
            # the constant-rhs solve of :80-92 is left pending and batched with the first
            # kkt_solve of the iteration (both right-hand sides are known by then; INTEGRATION.md shows the Julia side)
            self._const_pending = True
            return True
        return self._solve_constant_rhs(data)

    def _solve_constant_rhs(self, data):  # :80-92
        self.workx[:] = -data.q
        self.kktsolver.kktsolver_setrhs(self.workx, data.b)
        return self.kktsolver.kktsolver_solve(self.x2, self.z2)

    def kkt_solve_initial_point(self, variables, data):  # :95-132
        ks = self.kktsolver
        if data.P.nnz == 0:
            self.workx[:] = 0.0
            self.workz[:] = data.b
            ks.kktsolver_setrhs(self.workx, self.workz)
            ok = ks.kktsolver_solve(variables.x, variables.s)
            variables.s *= -1.0
            if not ok:
                return ok
            self.workx[:] = -data.q
            self.workz[:] = 0.0
            ks.kktsolver_setrhs(self.workx, self.workz)
            ok = ks.kktsolver_solve(None, variables.z)
        else:
            self.workx[:] = -data.q
            self.workz[:] = data.b
            ks.kktsolver_setrhs(self.workx, self.workz)
            ok = ks.kktsolver_solve(variables.x, variables.z)
            variables.s[:] = -variables.z
        return ok

    def kkt_solve(self, lhs, rhs, data, variables, cones, steptype):  # :135-215
        x1, z1, x2, z2 = self.x1, self.z1, self.x2, self.z2
        workx, workz = self.workx, self.workz
        workx[:] = rhs.x
        ds_const = self.work_conic
        if steptype == "affine":
            ds_const[:] = variables.s
        else:
            cones.ds_from_dz_offset(ds_const, rhs.s, lhs.z, variables.z)
        workz[:] = ds_const - rhs.z
        if self._device_reduced and (self._const_pending or self._have_const_dev):
            ok, lhs.tau = self.kktsolver.kktsolver_kkt_solve_reduced(workx, workz, variables.x, variables.tau, variables.kappa, rhs.tau,
                                                                    rhs.kappa, self._const_pending, lhs.x, lhs.z)
            if not ok:
                return False
            self._have_const_dev = True
            self._const_pending = False
            cones.mul_Hs(lhs.s, lhs.z, workz)
            lhs.s[:] = -(lhs.s + ds_const)
            lhs.kappa = -(rhs.kappa + variables.kappa * lhs.tau) / variables.tau
            return True
        if self._const_pending:      # [-q; b] and this step's right-hand side on one factorisation, concurrently
            self._rx2[0], self._rz2[0] = -data.q, data.b
            self._rx2[1], self._rz2[1] = workx, workz
            if not self.kktsolver.kktsolver_solve_multi(self._rx2, self._rz2, self._lx2, self._lz2):
                return False
            x2[:], z2[:] = self._lx2[0], self._lz2[0]
            x1[:], z1[:] = self._lx2[1], self._lz2[1]
            self._const_pending = False
        else:
            self.kktsolver.kktsolver_setrhs(workx, workz)
            if not self.kktsolver.kktsolver_solve(x1, z1):
                return False
        xi = workx
        xi[:] = variables.x / variables.tau
        P = data.P
        tau_num = (rhs.tau - rhs.kappa / variables.tau + float(np.dot(data.q, x1))
                   + float(np.dot(data.b, z1)) + 2.0 * _quad_form(xi, P, x1))
        xi -= x2
        tau_den = variables.kappa / variables.tau - float(np.dot(data.q, x2)) - float(np.dot(data.b, z2))
        tau_den += _quad_form(xi, P, xi) - _quad_form(x2, P, x2)
        lhs.tau = tau_num / tau_den
        lhs.x[:] = x1 + lhs.tau * x2
        lhs.z[:] = z1 + lhs.tau * z2
        cones.mul_Hs(lhs.s, lhs.z, workz)
        lhs.s[:] = -(lhs.s + ds_const)
        lhs.kappa = -(rhs.kappa + variables.kappa * lhs.tau) / variables.tau
        return True


class Solver:
    """solver.jl: ``Solver(P,q,A,b,cones,settings)`` + ``solve!``.

    ``kktsolver_factory(P, A, cones, m, n, settings)`` builds the AbstractKKTSolver (the plugin
    point the reference hard-wires at kktsystem.jl:33); default = the HIP solver."""

    def __init__(self, P, q, A, b, cone_specs, settings: Settings | None = None, kktsolver_factory=None):
        t0 = time.perf_counter()
        self.settings = settings if settings is not None else Settings()
        st = self.settings
        self.data = ProblemData(P, q, A, b, cone_specs, st)
        data = self.data
        if data.A.shape[0] != data.b.size or data.A.shape[1] != data.q.size or data.P.shape != (data.n, data.n):
            raise ValueError("dimension mismatch")  # solver.jl:157-186
        self.cones = CompositeCone(data.cone_specs)
        if self.cones.numel != data.m:
            raise ValueError("cone dimensions do not match the rows of A")
        self.cones.use_settings(st)
        data.equilibrate(self.cones, st)
        m, n = data.m, data.n
        self.variables = Variables.zeros(n, m)
        self.residuals = Residuals(np.zeros(n), np.zeros(m), 1.0, np.zeros(n), np.zeros(m), np.zeros(n))
        if kktsolver_factory is None:
            from clarabel_jl_amd.kktsolver import HipKKTSolver

            kktsolver_factory = HipKKTSolver
        t1 = time.perf_counter()
        self.kktsystem = KKTSystem(kktsolver_factory(data.P, data.A, self.cones, m, n, st), m, n)
        ks = self.kktsystem.kktsolver
        self._device_residuals = bool(getattr(st, "device_residuals", False)) and hasattr(ks, "residuals_update")
        self._needs_qb = self._device_residuals or (bool(getattr(st, "device_reduced", False)) and hasattr(ks, "kktsolver_kkt_solve_reduced"))
        if self._needs_qb:               # q, b resident in the plugin (N4 residuals, N2 reduced-system algebra)
            ks.set_problem_vectors(data.q, data.b)
        self.info = Info()
        self.info.timers["kkt init"] = time.perf_counter() - t1
        self.step_lhs = Variables.zeros(n, m)
        self.step_rhs = Variables.zeros(n, m)
        self.prev_vars = Variables.zeros(n, m)
        self.solution = Solution(np.zeros(n), np.zeros(m), np.zeros(m))
        self.info.timers["setup!"] = time.perf_counter() - t0
        self.trace = None  # optional list of per-iteration records (tests / bench)

    # ------------------------------------------------------------- data updating (data_updating.jl)
    def update_P(self, Pvals):
        data = self.data
        P = data.P
        Pcol = np.repeat(np.arange(data.n), np.diff(P.indptr))
        P.data[:] = np.asarray(Pvals, dtype=np.float64) * data.d[P.indices] * data.d[Pcol] * data.c
        self.kktsystem.kktsolver.kktsolver_update_P(P)

    def update_A(self, Avals):
        data = self.data
        A = data.A
        Acol = np.repeat(np.arange(data.n), np.diff(A.indptr))
        A.data[:] = np.asarray(Avals, dtype=np.float64) * data.e[A.indices] * data.d[Acol]
        self.kktsystem.kktsolver.kktsolver_update_A(A)

    def update_q(self, q):
        data = self.data
        data.q[:] = np.asarray(q, dtype=np.float64) * data.d * data.c
        data.normq = None
        if self._needs_qb:
            self.kktsystem.kktsolver.set_problem_vectors(data.q, data.b)

    def update_b(self, b):
        data = self.data
        data.b[:] = np.minimum(np.asarray(b, dtype=np.float64), INFINITY) * data.e
        data.normb = None
        if self._needs_qb:
            self.kktsystem.kktsolver.set_problem_vectors(data.q, data.b)

    # ------------------------------------------------------------- residuals.jl:1-37
    def _residuals_update(self):
        r, v, d = self.residuals, self.variables, self.data
        if self._device_residuals:       # the SpMVs and dots of residuals.jl on the device
            self.kktsystem.kktsolver.residuals_update(r, v)
            return
        qx = float(np.dot(d.q, v.x))
        bz = float(np.dot(d.b, v.z))
        sz = float(np.dot(v.s, v.z))
        r.Px[:] = _symv(d.P, v.x)
        xPx = float(np.dot(v.x, r.Px))
        r.rx_inf[:] = -(d.A.T @ v.z)
        r.rz_inf[:] = v.s + d.A @ v.x
        r.rx[:] = r.rx_inf - r.Px - d.q * v.tau
        r.rz[:] = r.rz_inf - d.b * v.tau
        r.rtau = qx + bz + v.kappa + xPx / v.tau
        r.dot_qx, r.dot_bz, r.dot_sz, r.dot_xPx = qx, bz, sz, xPx

    # ------------------------------------------------------------- info.jl:1-60
    def _info_update(self, t_start):
        info, data, v, r = self.info, self.data, self.variables, self.residuals
        tauinv = 1.0 / v.tau
        normb = data.get_normb()
        normq = data.get_normq()
        d, dinv, e, einv = data.d, data.dinv, data.e, data.einv
        cinv = 1.0 / data.c
        xPx2 = r.dot_xPx * tauinv * tauinv / 2.0
        info.cost_primal = (r.dot_qx * tauinv + xPx2) * cinv
        info.cost_dual = (-r.dot_bz * tauinv - xPx2) * cinv
        normx = _norm_scaled(d, v.x)
        normz = _norm_scaled(e, v.z) * cinv
        norms = _norm_scaled(einv, v.s)
        info.res_primal_inf = (_norm_scaled(dinv, r.rx_inf) * cinv) / max(1.0, normz)
        info.res_dual_inf = max(_norm_scaled(dinv, r.Px) / max(1.0, normx),
                                _norm_scaled(einv, r.rz_inf) / max(1.0, normx + norms))
        normx *= tauinv
        normz *= tauinv
        norms *= tauinv
        info.res_primal = _norm_scaled(einv, r.rz) * tauinv / max(1.0, normb + normx + norms)
        info.res_dual = _norm_scaled(dinv, r.rx) * tauinv * cinv / max(1.0, normq + normx + normz)
        info.gap_abs = abs(info.cost_primal - info.cost_dual)
        info.gap_rel = info.gap_abs / max(1.0, min(abs(info.cost_primal), abs(info.cost_dual)))
        info.ktratio = v.kappa * tauinv
        info.solve_time = time.perf_counter() - t_start

    # ------------------------------------------------------------- info.jl:62-120, 230-331
    def _check_convergence(self, tol_gap_abs, tol_gap_rel, tol_feas, tol_infeas_abs, tol_infeas_rel,
                           tol_ktratio, solved, pinf, dinf):
        info, r = self.info, self.residuals
        is_solved = ((info.gap_abs < tol_gap_abs or info.gap_rel < tol_gap_rel)
                     and info.res_primal < tol_feas and info.res_dual < tol_feas)
        if info.ktratio <= 1.0 and is_solved:
            info.status = solved
        elif info.ktratio > 1000.0 / tol_ktratio:
            if r.dot_bz < -tol_infeas_abs and info.res_primal_inf < -tol_infeas_rel * r.dot_bz:
                info.status = pinf
            elif r.dot_qx < -tol_infeas_abs and info.res_dual_inf < -tol_infeas_rel * r.dot_qx:
                info.status = dinf

    def _check_termination(self, it):
        info, st = self.info, self.settings
        info.status = UNSOLVED
        self._check_convergence(st.tol_gap_abs, st.tol_gap_rel, st.tol_feas, st.tol_infeas_abs,
                                st.tol_infeas_rel, st.tol_ktratio, SOLVED, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE)
        if info.status == UNSOLVED and it > 1 and (info.res_dual > info.prev_res_dual
                                                    or info.res_primal > info.prev_res_primal):
            if info.ktratio < 100 * _EPS and (info.prev_gap_abs < st.tol_gap_abs
                                              or info.prev_gap_rel < st.tol_gap_rel):
                info.status = INSUFFICIENT_PROGRESS
            if info.ktratio < 1.0:
                if ((info.res_dual > 100 * st.tol_feas and info.res_dual > 100 * info.prev_res_dual)
                        or (info.res_primal > 100 * st.tol_feas and info.res_primal > 100 * info.prev_res_primal)):
                    info.status = INSUFFICIENT_PROGRESS
        if info.status == UNSOLVED:
            if st.max_iter == info.iterations:
                info.status = MAX_ITERATIONS
            elif info.solve_time > st.time_limit:
                info.status = MAX_TIME
        return info.status != UNSOLVED

    def _save_prev_iterate(self):
        i = self.info
        i.prev_cost_primal, i.prev_cost_dual = i.cost_primal, i.cost_dual
        i.prev_res_primal, i.prev_res_dual = i.res_primal, i.res_dual
        i.prev_gap_abs, i.prev_gap_rel = i.gap_abs, i.gap_rel
        self.prev_vars.copy_from(self.variables)

    def _reset_to_prev_iterate(self):
        i = self.info
        i.cost_primal, i.cost_dual = i.prev_cost_primal, i.prev_cost_dual
        i.res_primal, i.res_dual = i.prev_res_primal, i.prev_res_dual
        i.gap_abs, i.gap_rel = i.prev_gap_abs, i.prev_gap_rel
        self.variables.copy_from(self.prev_vars)

    # ------------------------------------------------------------- variables.jl
    def _calc_step_length(self, steptype):  # :14-43
        v, step = self.variables, self.step_lhs
        a_tau = -v.tau / step.tau if step.tau < 0 else FLOATMAX
        a_kap = -v.kappa / step.kappa if step.kappa < 0 else FLOATMAX
        alpha = min(a_tau, a_kap, 1.0)
        az, as_ = self.cones.step_length(step.z, step.s, v.z, v.s, alpha)
        alpha = min(az, as_)
        if steptype == "combined":
            alpha *= self.settings.max_step_fraction
        return alpha

    def _get_step_length(self, steptype, scaling):  # solver.jl:408-424
        alpha = self._calc_step_length(steptype)
        if not self.cones.is_symmetric() and steptype == "combined" and scaling == "dual":
            alpha = self._backtrack_step_to_barrier(alpha)
        return alpha

    def _backtrack_step_to_barrier(self, alpha_init):  # solver.jl:427-445: distance to the boundary of the non-symmetric cones
        step = self.settings.linesearch_backtrack_step
        alpha = alpha_init
        for _ in range(50):
            if self._variables_barrier(alpha) < 1.0:
                return alpha
            alpha = step * alpha
        return alpha

    def _variables_barrier(self, alpha):  # variables.jl:46-74
        v, step, cones = self.variables, self.step_lhs, self.cones
        central_coef = cones.degree + 1
        cur_tau = v.tau + alpha * step.tau
        cur_kappa = v.kappa + alpha * step.kappa
        sz = float(np.dot(v.z + alpha * step.z, v.s + alpha * step.s))        # dot_shifted, mathutils.jl:23-43
        mu = (sz + cur_tau * cur_kappa) / central_coef
        barrier = central_coef * _logsafe(mu) - _logsafe(cur_tau) - _logsafe(cur_kappa)
        barrier += cones.compute_barrier(v.z, v.s, step.z, step.s, alpha)
        return barrier

    def _shift_to_cone_interior(self, z, pd):  # :181-208
        cones = self.cones
        min_margin, pos_margin = cones.margins(z, pd)
        target = max(1.0, 0.1 * pos_margin / cones.degree) if cones.degree > 0 else 1.0
        if min_margin <= 0:
            cones.scaled_unit_shift(z, -min_margin, pd)
            cones.scaled_unit_shift(z, target, pd)
        elif min_margin < target:
            cones.scaled_unit_shift(z, target - min_margin, pd)
        else:
            cones.scaled_unit_shift(z, 0.0, pd)

    def _default_start(self):  # solver.jl:383-405
        if not self.cones.is_symmetric():
            # variables_unit_initialization!, variables.jl:211-225: unit (z, s) along the central rays, x = 0
            v = self.variables
            self.cones.unit_initialization(v.z, v.s)
            v.x[:] = 0.0
            v.tau = 1.0
            v.kappa = 1.0
            return
        self.cones.set_identity_scaling()
        self.kktsystem.kkt_update(self.data, self.cones)
        self.kktsystem.kkt_solve_initial_point(self.variables, self.data)
        self._shift_to_cone_interior(self.variables.s, "primal")
        self._shift_to_cone_interior(self.variables.z, "dual")
        self.variables.tau = 1.0
        self.variables.kappa = 1.0

    def solve(self):
        """The reference's host-side vector algebra is single-threaded Julia; numpy's OpenBLAS instead spins up a thread
        team for every 20k-element ``dot`` (4 ms instead of 4 us on an 8-core box), so the BLAS pool is limited to one
        thread while the loop runs."""
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:          # pragma: no cover - threadpoolctl ships with the image
            return self._solve()
        with threadpool_limits(limits=1, user_api="blas"):
            return self._solve()

    # ------------------------------------------------------------- solver.jl:189-380
    def _solve(self):
        st, info, data, cones = self.settings, self.info, self.data, self.cones
        v, r = self.variables, self.residuals
        lhs, rhs = self.step_lhs, self.step_rhs
        tm = info.timers
        for k in ("kkt update", "kkt solve", "scale cones", "default start"):
            tm[k] = 0.0
        it = 0
        sigma, alpha, mu = 1.0, 0.0, FLOATMAX
        info.status = UNSOLVED
        info.iterations = 0
        t_start = time.perf_counter()
        t0 = time.perf_counter()
        self._default_start()
        tm["default start"] = time.perf_counter() - t0
        t_loop = time.perf_counter()
        nonsym = not cones.is_symmetric()
        scaling = "primal_dual" if cones.allows_primal_dual_scaling() else "dual"      # solver.jl:222
        while True:
            self._residuals_update()
            mu = (r.dot_sz + v.tau * v.kappa) / (cones.degree + 1)  # variables.jl:2-11
            info.mu, info.step_length, info.sigma, info.iterations = mu, alpha, sigma, it
            self._info_update(t_start)
            if st.verbose:
                print(f"{it:3d}  pcost {info.cost_primal: .4e}  dcost {info.cost_dual: .4e}  gap {info.gap_abs:.2e}"
                      f"  pres {info.res_primal:.2e}  dres {info.res_dual:.2e}  k/t {info.ktratio:.2e}"
                      f"  mu {mu:.2e}  step {alpha:.2e}")
            if self.trace is not None:
                self.trace.append(dict(iter=it, mu=mu, alpha=alpha, sigma=sigma, cost_primal=info.cost_primal,
                                       cost_dual=info.cost_dual, res_primal=info.res_primal,
                                       res_dual=info.res_dual, ktratio=info.ktratio))
            if self._check_termination(it):
                # _strategy_checkpoint_insufficient_progress (:453-473)
                if info.status == INSUFFICIENT_PROGRESS:
                    self._reset_to_prev_iterate()
                    if nonsym and scaling == "primal_dual":      # continue with the dual-only scaling
                        info.status = UNSOLVED
                        scaling = "dual"
                        continue
                break
            t0 = time.perf_counter()
            ok_scaling = cones.update_scaling(v.s, v.z, mu, scaling) if nonsym else cones.update_scaling(v.s, v.z, mu)
            tm["scale cones"] += time.perf_counter() - t0
            if not ok_scaling:
                info.status = NUMERICAL_ERROR
                break
            it += 1
            t0 = time.perf_counter()
            ok = self.kktsystem.kkt_update(data, cones, sz=(v.s, v.z))
            tm["kkt update"] += time.perf_counter() - t0
            # variables_affine_step_rhs!, variables.jl:107-121
            rhs.x[:] = r.rx
            rhs.z[:] = r.rz
            cones.affine_ds(rhs.s, v.s)
            rhs.tau = r.rtau
            rhs.kappa = v.tau * v.kappa
            t0 = time.perf_counter()
            ok = ok and self.kktsystem.kkt_solve(lhs, rhs, data, v, cones, "affine")
            tm["kkt solve"] += time.perf_counter() - t0
            if ok:
                alpha = self._get_step_length("affine", scaling)
                sigma = (1.0 - alpha) ** 3  # :446-449
                mcorr = 1.0 if it > 1 else alpha
                # variables_combined_step_rhs!, variables.jl:124-162
                sm = sigma * mu
                rhs.x[:] = (1.0 - sigma) * r.rx
                rhs.tau = (1.0 - sigma) * r.rtau
                rhs.kappa = -sm + mcorr * lhs.tau * lhs.kappa + v.tau * v.kappa
                if mcorr != 1.0:
                    lhs.z *= mcorr
                cones.combined_ds_shift(rhs.z, lhs.z, lhs.s, sm)
                rhs.s += rhs.z
                rhs.z[:] = (1.0 - sigma) * r.rz
                t0 = time.perf_counter()
                ok = self.kktsystem.kkt_solve(lhs, rhs, data, v, cones, "combined")
                tm["kkt solve"] += time.perf_counter() - t0
            if not ok:  # _strategy_checkpoint_numerical_error (:476-490)
                alpha = 0.0
                if nonsym and scaling == "primal_dual":
                    scaling = "dual"
                    continue
                info.status = NUMERICAL_ERROR
                break
            alpha = self._get_step_length("combined", scaling)
            # _strategy_checkpoint_small_step (:493-506)
            if nonsym and scaling == "primal_dual" and alpha < st.min_switch_step_length:
                scaling = "dual"
                alpha = 0.0
                continue
            if alpha <= max(0.0, st.min_terminate_step_length):
                info.status = INSUFFICIENT_PROGRESS
                alpha = 0.0
                break
            self._save_prev_iterate()
            v.x += alpha * lhs.x
            v.s += alpha * lhs.s
            v.z += alpha * lhs.z
            v.tau += alpha * lhs.tau
            v.kappa += alpha * lhs.kappa
        tm["IP iteration"] = time.perf_counter() - t_loop
        if alpha == 0.0:
            info.mu, info.step_length, info.sigma, info.iterations = mu, alpha, sigma, it
        # info_post_process!, info.jl:173-189
        if info.status in _ERRORED or info.status in (MAX_ITERATIONS, MAX_TIME):
            self._check_convergence(st.reduced_tol_gap_abs, st.reduced_tol_gap_rel, st.reduced_tol_feas,
                                    st.reduced_tol_infeas_abs, st.reduced_tol_infeas_rel, st.reduced_tol_ktratio,
                                    ALMOST_SOLVED, ALMOST_PRIMAL_INFEASIBLE, ALMOST_DUAL_INFEASIBLE)
        # solution_post_process!, solution.jl:2-45 + variables_unscale!, variables.jl:247-275
        sol = self.solution
        sol.status = info.status
        infeasible = info.status in _INFEASIBLE
        if infeasible:
            sol.obj_val = sol.obj_val_dual = float("nan")
        else:
            sol.obj_val, sol.obj_val_dual = info.cost_primal, info.cost_dual
        sol.iterations = info.iterations
        sol.r_prim, sol.r_dual = info.res_primal, info.res_dual
        scaleinv = 1.0 / v.kappa if infeasible else 1.0 / v.tau
        cinv = 1.0 / data.c
        sol.x[:] = v.x * data.d * scaleinv
        sol.z[:] = v.z * data.e * (scaleinv * cinv)
        sol.s[:] = v.s * data.einv * scaleinv
        info.solve_time = time.perf_counter() - t_start
        tm["solve!"] = info.solve_time
        sol.solve_time = info.solve_time
        return sol
