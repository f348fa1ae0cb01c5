#!/usr/bin/env python
"""Batch seed 324 (cfg 4), the one problem whose HIP-vs-oracle difference exceeds 1e-10 (|dobj| 1.3e-5, iterations
24 vs 23): is the cause the reference's OWN sensitivity?  CPU only, no HIP code involved: the oracle (the :qdldl restatement) drives
the IPM while a second oracle is handed identical inputs at every KKT call; per solve the two solutions and refinement-step counts
are compared.  Two kinds of pairs: (a) different elimination ORDERS (the product's -- host symbolic analysis, tests/support/plan_check
--, SuperLU's MMD on K, reverse Cuthill-McKee): same residual arithmetic, different LDL rounding; (b) the SAME order, but the second
oracle sums the refinement residual e = b - K x in the opposite association order (oracle_kkt_set_residual_order: the same products,
columns swept downwards) -- what any parallel SpMV changes.  Prints one line per pair: the first solve whose refinement takes a
different number of steps, the residual norms of both at that solve, the agreement of the solutions before it and AT it.
usage: python tools/seed324_cpu_vs_cpu.py [seed ...]        (writes nothing; redirect into profiles/*_parity_causes.txt)"""
import os, sys
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd import problems
from oracle.kkt_oracle import OracleKKTSolver, OracleKKT
import tests.plan_support as ps


class ShadowPair:
    """oracle `a` drives, oracle `b` shadows on identical inputs"""
    batch_constant_rhs = False

    def __init__(self, oa, ob, *args, resid_b=0):
        self.c = OracleKKTSolver(*args, ordering=oa)
        self.g = OracleKKTSolver(*args, ordering=ob)
        self.g.k.L.oracle_kkt_set_residual_order(self.g.k.h, resid_b)
        self.settings = self.c.settings
        self.it, self.log = 0, []

    def kktsolver_update(self, cones):
        self.it += 1
        ok = self.c.kktsolver_update(cones)
        self.g.kktsolver_update(cones)
        return ok

    def kktsolver_setrhs(self, rx, rz):
        self.c.kktsolver_setrhs(rx, rz)
        self.g.kktsolver_setrhs(rx, rz)

    def kktsolver_solve(self, lx, lz):
        n, m = self.c.n, self.c.m
        gx, gz = np.zeros(n), np.zeros(m)
        self.g.kktsolver_solve(gx, gz)
        cx, cz = (lx if lx is not None else np.zeros(n)), (lz if lz is not None else np.zeros(m))
        ok = self.c.kktsolver_solve(cx, cz)
        xc, xg = np.concatenate([cx, cz]), np.concatenate([gx, gz])
        self.log.append((self.it, float(np.max(np.abs(xg - xc)) / max(1.0, np.max(np.abs(xc)))), int(self.g.last_ir_steps), int(self.c.last_ir_steps),
                         self.g.last_norms, self.c.last_norms))
        return ok

    def __getattr__(self, k):
        return getattr(self.c, k)


def product_order(P, A, cones_spec):
    cones = cl.CompositeCone(cl.cones_new_collapsed(cones_spec))
    Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
    Ac = sp.csc_matrix(A); Ac.sort_indices()
    k = OracleKKT(Pt, Ac, *cones.kkt_descriptors())
    rc, _, perm, _ = ps.run(k.N, k.colptr, k.rowval, k.nzval.copy(), k.map("dsigns"), symbolic_only=True)
    assert rc == 0
    U = sp.csc_matrix((np.ones(len(k.rowval)), k.rowval, k.colptr), shape=(k.N, k.N))
    rcm = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(U + U.T), symmetric_mode=True), dtype=np.int64)
    return np.asarray(perm, dtype=np.int64), rcm


def main(seeds):
    for seed in seeds:
        P, q, A, b, cones = problems.batch_problem(seed)
        prod, rcm = product_order(P, A, cones)
        base = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: OracleKKTSolver(*a, ordering=prod)).solve()
        print(f"[cpu-vs-cpu seed {seed}] oracle in the product's order: {base.status} in {base.iterations} iterations, obj {base.obj_val:.12e}")
        for name, oa, ob, rb in (("product's order vs MMD", prod, "mmd", 0), ("product's order vs RCM", prod, rcm, 0), ("MMD vs RCM", "mmd", rcm, 0),
                                 ("product's order, residual summed forwards vs backwards", prod, prod, 1),
                                 ("MMD order, residual summed forwards vs backwards", "mmd", "mmd", 1)):
            s = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: ShadowPair(oa, ob, *a, resid_b=rb))
            sol = s.solve()
            log = s.kktsystem.kktsolver.log
            first = next((k for k, r in enumerate(log) if r[2] != r[3]), None)
            before = max([r[1] for r in log[:first]] or [0.0])
            if first is None:
                print(f"[cpu-vs-cpu seed {seed}] {name}: no solve with different refinement step counts in {sol.iterations} iterations; max rel_dx {before:.2e}")
            else:
                it, rel, sb, sa, nb, na = log[first]
                print(f"[cpu-vs-cpu seed {seed}] {name}: first solve with different refinement step counts at IPM iteration {it} "
                      f"(steps {sa} vs {sb}); residual norms driver {np.array2string(na, precision=3)} shadow {np.array2string(nb, precision=3)}; "
                      f"the two solutions of THAT solve differ by rel_dx {rel:.3e}; max rel_dx over the "
                      f"{first} solves before it {before:.2e}; driver ends {sol.status} in {sol.iterations} iterations")


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [324])
