#!/usr/bin/env python
"""Batch seed 324, CPU only: is the less accurate FIRST (unrefined) LDL solve of the HIP path at IPM iteration 19 (||e|| 2.3e-5 against
the oracle's 1.46e-5) made by the EXPLICIT INVERSES the solve kernels multiply by?  The oracle drives the IPM in the
product's elimination order; at the chosen iteration every solve's regularised K, right-hand side and the oracle's own first residual
are captured, and the host interpreter of the product's plan (tests/support/plan_check.cpp: the supernodal factorisation with the
product's panels and update lists) solves the same system twice: diagonal blocks by substitution, and by products with their explicit
inverses (PLANCHECK_EXPLICIT_INV=1, the kernels' form).  Prints ||b - K x||_inf (unregularised K, what the refinement measures).
usage: python tools/seed324_inverse_vs_substitution.py [seed [iteration]]"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd import problems
from oracle.kkt_oracle import OracleKKTSolver
import tests.plan_support as ps


class Capture(OracleKKTSolver):
    it = 0
    want = 19
    caps = []

    def kktsolver_update(self, cones):
        Capture.it += 1
        return super().kktsolver_update(cones)

    def kktsolver_setrhs(self, rx, rz):
        self._b = np.concatenate([rx, rz, np.zeros(self.k.N - len(rx) - len(rz))])
        super().kktsolver_setrhs(rx, rz)

    def kktsolver_solve(self, lx, lz):
        ok = super().kktsolver_solve(lx, lz)
        if Capture.it == Capture.want:
            Capture.caps.append(dict(b=self._b.copy(), nz=self.k.nzval, eps=self.diagonal_regularizer, norms=self.last_norms.copy(),
                                     steps=self.last_ir_steps, k=self.k))
        return ok


def main(seed, want):
    P, q, A, b, cones = problems.batch_problem(seed)
    Capture.want = want
    cc = cl.CompositeCone(cl.cones_new_collapsed(cones))
    Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
    Ac = sp.csc_matrix(A); Ac.sort_indices()
    from oracle.kkt_oracle import OracleKKT
    k0 = OracleKKT(Pt, Ac, *cc.kkt_descriptors())
    rc, _, perm, st = ps.run(k0.N, k0.colptr, k0.rowval, k0.nzval.copy(), k0.map("dsigns"), symbolic_only=True)
    assert rc == 0
    sol = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: Capture(*a, ordering=perm)).solve()
    print(f"[seed {seed}] oracle in the product's order: {sol.status} in {sol.iterations} iterations; N {k0.N}, supernodes {int(st['nsuper'])}, fronts {int(st['nfronts'])}")
    for ci, c in enumerate(Capture.caps):
        k = c["k"]
        ds = k.map("dsigns")
        diag = k.map("map_diag_full")
        nz_unreg = c["nz"].copy()
        # is the captured image regularised?  (the oracle un-shifts after the factorisation, kktsolver_directldl.jl:285-292)
        nz_reg = nz_unreg.copy()
        nz_reg[diag] += c["eps"] * ds
        Ku = sp.csc_matrix((nz_unreg, k.rowval, k.colptr), shape=(k.N, k.N))
        Ks = Ku + sp.triu(Ku, 1).T
        out = []
        for mode, tau, wmin, wmax in (("0", 0, 0, 99), ("1", 0, 0, 99), ("2", 0, 0, 99), ("2", 8, 0, 99), ("2", 64, 0, 99), ("2", 1024, 0, 99), ("2", 1e6, 0, 99),
                                      ("2", 0, 0, 4), ("2", 0, 0, 8), ("2", 0, 0, 16), ("2", 0, 9, 99), ("2", 0, 17, 99)):
            os.environ["PLANCHECK_EXPLICIT_INV"] = mode
            os.environ["PLANCHECK_INV_TAU"] = repr(float(tau))
            os.environ["PLANCHECK_INV_WMIN"] = str(wmin)
            os.environ["PLANCHECK_INV_WMAX"] = str(wmax)
            rc, x, _, stt = ps.run(k.N, k.colptr, k.rowval, nz_reg, ds, b=c["b"])
            assert rc == 0
            out.append((float(np.max(np.abs(c["b"] - Ks @ x))), int(stt["max_group_tasks"]) if mode == "2" else 0))
        print(f"[seed {seed}] iteration {want}, solve {ci}: oracle refinement ||e|| per step {np.array2string(c['norms'], precision=4)} ({c['steps']} steps); "
              f"first-solve ||b - K x||_inf of the supernodal plan: diagonal blocks by SUBSTITUTION {out[0][0]:.4e}; by EXPLICIT INVERSES {out[1][0]:.4e}; "
              f"inverses + one block-level refinement step on every block {out[2][0]:.4e} ({out[2][1]} block solves); only where max|Linv| > 8: "
              f"{out[3][0]:.4e} ({out[3][1]}); > 64: {out[4][0]:.4e} ({out[4][1]}); > 1024: {out[5][0]:.4e} ({out[5][1]}); > 1e6: {out[6][0]:.4e} ({out[6][1]}); "
              f"eps {c['eps']:.3e}, max|K| {np.max(np.abs(nz_unreg)):.2e}; refinement only on blocks of width <= 4: {out[7][0]:.4e} ({out[7][1]}), <= 8: {out[8][0]:.4e} ({out[8][1]}), "
              f"<= 16: {out[9][0]:.4e} ({out[9][1]}), >= 9: {out[10][0]:.4e} ({out[10][1]}), >= 17: {out[11][0]:.4e} ({out[11][1]})")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 324, int(sys.argv[2]) if len(sys.argv) > 2 else 19)
