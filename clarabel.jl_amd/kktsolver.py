"""``HipKKTSolver`` — host-side mirror of the reference's KKT-solver plugin for seam L1.

It implements the ``AbstractKKTSolver`` contract (src/kktsolvers/kktsolver_defaults.jl:2-47) with
the same names, argument meaning and success semantics as the reference's ``DirectLDLKKTSolver``
(src/kktsolvers/kktsolver_directldl.jl); everything below the method boundary runs in the HIP
library through the C ABI (include/hipkkt.h).  The Julia file a maintainer would add is shown in
INTEGRATION.md; this class is its Python twin so that parity tests read like the reference's."""
from __future__ import annotations

import os

import numpy as np

from . import hipkkt
from .settings import Settings


_DEBUG = os.environ.get("HIPKKT_DEBUG", "0") == "1"


class HipKKTSolver:
    def __init__(self, P, A, cones, m, n, settings: Settings, **optkw):
        """ref: DirectLDLKKTSolver{T}(P,A,cones,m,n,settings), kktsolver_directldl.jl:46-92.
        P: n x n triu CSC (scipy), A: m x n CSC, cones: CompositeCone."""
        self.settings = settings
        self.m, self.n = m, n
        numel, hs_dense, sparse_kind, dim1 = cones.kkt_descriptors()
        self.h = hipkkt.Handle.from_parts(
            P, A, numel, hs_dense, sparse_kind, dim1, device=settings.device_id,
            dynamic_reg_eps=settings.dynamic_regularization_eps,
            dynamic_reg_delta=settings.dynamic_regularization_delta, **optkw)
        self.p = self.h.p
        self.Hsblocks = np.zeros(self.h.nHs)          # ref: _allocate_kkt_Hsblocks
        # sparse-expandable cones in sparse-map order: SOC (rank-2 expansion, batched upload) and GenPow (rank-3 expansion)
        sparse = [c for c in cones if c.is_sparse_expandable]
        self._soc = [c for c in sparse if getattr(c, "sparse_kind", 1) == 1]
        self._genpow = [(i, c) for i, c in enumerate(sparse) if getattr(c, "sparse_kind", 1) == 2]
        self._soc_total = sum(c.dim for c in self._soc)
        self._u = np.zeros(self._soc_total)
        self._v = np.zeros(self._soc_total)
        self._eta2 = np.zeros(len(self._soc))
        # PSD cones: the Hs block (packed triu of W (x)_s W, numel^2/2 values) is formed on the device from the
        # n x n matrix W = R R^T (hipkkt_set_hs_psd): no host skron!, no upload of the block
        self._psd = [(c, r.start) for c, r in zip(cones.cones, cones.rng_blocks) if hasattr(c, "RRt")]
        self._psd_off = np.array([off for _, off in self._psd], dtype=np.int64)
        self._psd_dim = np.array([c.n for c, _ in self._psd], dtype=np.int64)
        self._psd_cones = tuple(c for c, _ in self._psd)
        # N1: cone types for the on-device update_scaling! / get_Hs! (hipkkt_set_cone_types); optional in the cones object
        # (the on-device scaling knows the symmetric cones only: include/hipkkt.h hipkkt_set_cone_types)
        kinds = cones.kkt_cone_kinds() if hasattr(cones, "kkt_cone_kinds") else None
        self._has_cone_kinds = kinds is not None and bool(np.all(np.asarray(kinds) >= 0))
        if self._has_cone_kinds:
            self.h.set_cone_types(kinds)
        self.scaling_w = self.scaling_lambda = self.scaling_soc_eta = None
        self.diagonal_regularizer = 0.0
        self.last_ir_steps = 0
        self.total_ir_steps = 0
        self.nsolves = 0
        self.last_nreg = 0

    # ref: kktsolver_update!, kktsolver_directldl.jl:197-245
    def kktsolver_update(self, cones) -> bool:
        # PSD cones whose block is not formed yet (i.e. after update_scaling!; the identity scaling sets Hs = I exactly,
        # :66-75, and goes the host way) get it from the device-side skron
        dev = [k for k, c in enumerate(self._psd_cones) if not c._hs_valid]
        cones.get_Hs(self.Hsblocks, skip=tuple(self._psd_cones[k] for k in dev))   # :223 (host cone algebra)
        self.h.set_hs(self.Hsblocks)                   # :225-228 negate + scatter, on the device
        if dev:
            self.h.set_hs_psd(self._psd_off[dev], self._psd_dim[dev],
                              np.concatenate([self._psd_cones[k].RRt.ravel() for k in dev]))
        if self._soc:                                  # :235-241 sparse-cone expansion columns
            off = 0
            for i, c in enumerate(self._soc):
                self._u[off:off + c.dim] = c.u
                self._v[off:off + c.dim] = c.v
                self._eta2[i] = c.eta * c.eta
                off += c.dim
            self.h.set_soc_batch(self._eta2, self._u, self._v)
        for i, c in self._genpow:                      # _csc_update_sparsecone(::GenPowerCone), directldl_datamaps.jl:146-167
            self.h.set_genpow(i, float(np.sqrt(c.mu)), c.p, c.q, c.r)
        return self._refactor()

    def _refactor(self) -> bool:
        st = self.settings                             # :243, :247-294
        ok, eps, nreg = self.h.refactor(st.static_regularization_enable, st.static_regularization_constant,
                                        st.static_regularization_proportional)
        self.diagonal_regularizer = eps
        self.last_nreg = nreg
        if _DEBUG:
            print(f"[hipkkt] refactor ok={ok} eps={eps:.3e} dynamic_regularisations={nreg}")
        return ok

    # kktsolver_update! fed with the iterate instead of the cones' scaling -- update_scaling! + get_Hs! of
    # the Zero / Nonnegative / SecondOrder cones and skron(R R^T) of the PSD cones run on the device (hipkkt_update_scaling); nothing
    # but (s, z) and the PSD cones' R factors crosses PCIe.  The device's (w, lambda, eta) are kept for the caller.
    def kktsolver_update_scaled(self, cones, s, z) -> bool:
        if not self._has_cone_kinds:
            raise hipkkt.HipKKTError("kktsolver_update_scaled: the cones object does not provide kkt_cone_kinds()")
        R = np.concatenate([c.R.ravel(order="F") for c in self._psd_cones]) if self._psd_cones else None
        ok, self.scaling_w, self.scaling_lambda, self.scaling_soc_eta = self.h.update_scaling(s, z, R)
        if not ok:
            return False
        return self._refactor()

    # ref: kktsolver_setrhs!, :313-327
    def kktsolver_setrhs(self, rhsx, rhsz):
        self.h.setrhs(rhsx, rhsz)

    # ref: kktsolver_solve!, :346-371 (lhsx / lhsz may be None = Julia `nothing`)
    def kktsolver_solve(self, lhsx, lhsz) -> bool:
        st = self.settings
        ok, steps = self.h.solve(lhsx, lhsz, st.iterative_refinement_enable, st.iterative_refinement_reltol,
                                 st.iterative_refinement_abstol, st.iterative_refinement_max_iter,
                                 st.iterative_refinement_stop_ratio)
        self.last_ir_steps = steps
        self.total_ir_steps += steps
        self.nsolves += 1
        if _DEBUG:
            print(f"[hipkkt] solve ok={ok} ir_steps={steps}")
        return ok

    # several right-hand sides on the current factorisation (rhsx [k, n], rhsz [k, m]), refined
    # like kktsolver_solve! refines one, two at a time on concurrent device contexts
    def kktsolver_solve_multi(self, rhsx, rhsz, lhsx, lhsz) -> bool:
        st = self.settings
        ok, steps = self.h.solve_multi(rhsx, rhsz, lhsx, lhsz, st.iterative_refinement_enable, st.iterative_refinement_reltol,
                                       st.iterative_refinement_abstol, st.iterative_refinement_max_iter,
                                       st.iterative_refinement_stop_ratio)
        self.last_ir_steps = int(steps[-1]) if len(steps) else 0
        self.total_ir_steps += int(np.sum(steps))
        self.nsolves += len(steps)
        if _DEBUG:
            print(f"[hipkkt] solve_multi ok={ok} ir_steps={list(steps)}")
        return ok

    # SURVEY section 8(f) row N2, second half: kkt_solve! (kktsystem.jl:135-215) between the caller's cone algebra and mul_Hs! --
    # the solve for (x1, z1), the d tau numerator / denominator (dots with q, b, quad_form with P) and dx, dz on the device.
    # const_pending: the constant-rhs solve that kkt_update! left pending runs in the same call (its solution stays resident).
    # Needs set_problem_vectors(q, b).  Returns (ok, dtau).
    def kktsolver_kkt_solve_reduced(self, rhs_x, workz, var_x, tau, kappa, rhs_tau, rhs_kappa, const_pending, lhs_x, lhs_z):
        st = self.settings
        ok, dtau, scal, steps = self.h.kkt_solve_reduced(rhs_x, workz, var_x, tau, kappa, rhs_tau, rhs_kappa, const_pending, lhs_x, lhs_z,
                                                         st.iterative_refinement_enable, st.iterative_refinement_reltol,
                                                         st.iterative_refinement_abstol, st.iterative_refinement_max_iter,
                                                         st.iterative_refinement_stop_ratio)
        self.last_ir_steps = int(steps[0])
        self.total_ir_steps += int(steps[0]) + (int(steps[1]) if const_pending else 0)
        self.nsolves += 2 if const_pending else 1
        self.last_reduced_scalars = scal
        if _DEBUG:
            print(f"[hipkkt] kkt_solve_reduced ok={ok} dtau={dtau:.6e} ir_steps={list(steps)}")
        return ok, dtau

    # residuals_update!(residuals, variables, data) (residuals.jl:1-37) from the resident P, A
    def set_problem_vectors(self, q, b):
        self.h.set_qb(q, b)
        self._has_qb = True

    def residuals_update(self, r, v):
        """fills the residual object `r` (rx, rz, rx_inf, rz_inf, Px, rtau, dot_*) from the variables `v` (x, z, s, tau, kappa)"""
        r.dot_qx, r.dot_bz, r.dot_sz, r.dot_xPx, r.rtau = self.h.residuals(v.x, v.z, v.s, v.tau, v.kappa, r.rx, r.rz, r.rx_inf,
                                                                          r.rz_inf, r.Px)

    # ref: kktsolver_update_P!/A!, :374-386
    def kktsolver_update_P(self, P):
        self.h.update_P(P.data)

    def kktsolver_update_A(self, A):
        self.h.update_A(A.data)

    # ref: kktsolver_linear_solver_info -> LinearSolverInfo(name,threads,direct,nnzA,nnzL), types.jl:198-206
    def kktsolver_linear_solver_info(self):
        return dict(name="hip", threads=1, direct=True, nnzA=self.h.nnzK, nnzL=self.h.nnzL)
