"""CPU-side evidence for the refined block solves of the HIP path (kernels.hip k_invert_diag_wide / polish_fwd / polish_bwd, round 5).

Batch seed 324 was the one problem of cfg 4 whose HIP-vs-oracle difference exceeded 1e-10 (rounds 3 - 4: |dobj| 1.3e-5, iterations 24 vs
23): at IPM iteration 19 the first, unrefined LDL solve of the HIP path left a residual 1.6x the oracle's and the stop-ratio branch of the
reference's iterative refinement (kktsolver_directldl.jl:437-444) went the other way.  The host interpreter of the product's plan
(tests/support/plan_check.cpp: the same supernodal factorisation, panels and update lists) reproduces the cause without a GPU:
* diagonal-block solves by SUBSTITUTION give the oracle's residual (the scalar QDLDL restatement: same value to 4 digits);
* diagonal-block solves as PRODUCTS WITH THE EXPLICIT INVERSES -- what the solve kernels do -- leave a residual several times larger;
* one refinement step  y += Linv (b - L y)  on the WIDE blocks alone (> 16 columns: 16 of the 1361 supernodes) restores substitution's
  residual exactly; on the narrow blocks alone it changes nothing -- which is the rule the kernels implement (with a threshold on the
  largest |entry| of a wide block's inverse)."""
import os

import numpy as np
import scipy.sparse as sp

import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd import problems
from oracle.kkt_oracle import OracleKKT, OracleKKTSolver
from tests import plan_support as ps


class _Capture(OracleKKTSolver):
    it, want, caps = 0, 19, []

    def kktsolver_update(self, cones):
        _Capture.it += 1
        return super().kktsolver_update(cones)

    def kktsolver_setrhs(self, rx, rz):
        self._b = np.concatenate([rx, rz, np.zeros(self.k.N - len(rx) - len(rz))])
        super().kktsolver_setrhs(rx, rz)

    def kktsolver_solve(self, lx, lz):
        ok = super().kktsolver_solve(lx, lz)
        if _Capture.it == _Capture.want and not _Capture.caps:
            _Capture.caps.append(dict(b=self._b.copy(), nz=self.k.nzval, eps=self.diagonal_regularizer, norms=self.last_norms.copy(), k=self.k))
        return ok


def test_wide_block_refinement_restores_the_substitution_residual(monkeypatch):
    P, q, A, b, cones = problems.batch_problem(324)
    cc = cl.CompositeCone(cl.cones_new_collapsed(cones))
    Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
    Ac = sp.csc_matrix(A); Ac.sort_indices()
    k0 = OracleKKT(Pt, Ac, *cc.kkt_descriptors())
    rc, _, perm, _ = ps.run(k0.N, k0.colptr, k0.rowval, k0.nzval.copy(), k0.map("dsigns"), symbolic_only=True)
    assert rc == 0
    _Capture.it, _Capture.caps = 0, []
    sol = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: _Capture(*a, ordering=perm)).solve()
    assert sol.status == "SOLVED" and sol.iterations == 23 and len(_Capture.caps) == 1
    c = _Capture.caps[0]
    k = c["k"]
    ds, diag = k.map("dsigns"), k.map("map_diag_full")
    nz_reg = c["nz"].copy()
    nz_reg[diag] += c["eps"] * ds                       # what was factored: K + eps * Dsigns (the image itself stays unregularised)
    Ku = sp.csc_matrix((c["nz"], k.rowval, k.colptr), shape=(k.N, k.N))
    Ks = Ku + sp.triu(Ku, 1).T

    def residual(mode, wmin=0, wmax=1 << 20, tau=0.0):
        monkeypatch.setenv("PLANCHECK_EXPLICIT_INV", str(mode))
        monkeypatch.setenv("PLANCHECK_INV_TAU", repr(float(tau)))
        monkeypatch.setenv("PLANCHECK_INV_WMIN", str(wmin))
        monkeypatch.setenv("PLANCHECK_INV_WMAX", str(wmax))
        rc_, x, _, _ = ps.run(k.N, k.colptr, k.rowval, nz_reg, ds, b=c["b"])
        assert rc_ == 0
        return float(np.max(np.abs(c["b"] - Ks @ x)))

    e_oracle = float(c["norms"][0])                      # the oracle's own first residual norm at that solve
    e_sub, e_inv = residual(0), residual(1)
    assert abs(e_sub - e_oracle) <= 1e-3 * e_oracle      # the plan with substitution = the scalar oracle
    assert e_inv >= 2.0 * e_sub                          # explicit inverses: several times less accurate (measured 3.8x)
    assert abs(residual(2) - e_sub) <= 1e-3 * e_sub      # one refinement step on every block: back to substitution
    assert abs(residual(2, wmin=17) - e_sub) <= 1e-3 * e_sub          # ... on the wide blocks alone: the same
    assert abs(residual(2, wmin=17, tau=64.0) - e_sub) <= 1e-3 * e_sub  # ... with the kernels' threshold on max |Linv|: the same
    assert residual(2, wmax=16) >= 2.0 * e_sub           # ... on the narrow blocks alone: nothing gained
