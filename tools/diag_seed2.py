#!/usr/bin/env python
"""Developer tool: the oracle drives the IPM of one cfg-4 batch problem; at every KKT call the HIP solver gets the very same inputs
(same elimination order) and its outputs are compared (which solve parts first, by how much, with what refinement history).
usage: diag_seed2.py <seed>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver
from oracle.kkt_oracle import OracleKKTSolver
seed = int(sys.argv[1])
P, q, A, b, cones = problems.batch_problem(seed)


class Both:
    batch_constant_rhs = False

    def __init__(self, *a):
        self.g = HipKKTSolver(*a)
        self.c = OracleKKTSolver(*a, ordering=self.g.h.perm())
        self.settings = self.c.settings
        self.it = 0

    def kktsolver_update(self, cones_):
        self.it += 1
        okc = self.c.kktsolver_update(cones_)
        okg = self.g.kktsolver_update(cones_)
        kg, kc = self.g.h.kkt()[2], self.c.k.nzval
        print(f"D2 it {self.it:2d} update ok {okg}/{okc} nreg {self.g.last_nreg}/{self.c.k.L.oracle_kkt_nreg(self.c.k.h)} eps {self.g.diagonal_regularizer:.3e}/{self.c.diagonal_regularizer:.3e} "
              f"Kdiff {np.max(np.abs(kg - kc)):.1e} max|K| {np.max(np.abs(kc)):.2e} min|Hs| {np.min(np.abs(self.g.Hsblocks)):.2e} max|Hs| {np.max(np.abs(self.g.Hsblocks)):.2e}")
        return okc

    def kktsolver_setrhs(self, rx, rz):
        self.c.kktsolver_setrhs(rx, rz); self.g.kktsolver_setrhs(rx, rz)
        self.rhs = np.concatenate([rx, rz])

    def kktsolver_solve(self, lx, lz):
        n, m = self.g.n, self.g.m
        gx, gz = np.zeros(n), np.zeros(m)
        okg = self.g.kktsolver_solve(gx, gz)
        cx, cz = (lx if lx is not None else np.zeros(n)), (lz if lz is not None else np.zeros(m))
        okc = self.c.kktsolver_solve(cx, cz)
        xc, xg = np.concatenate([cx, cz]), np.concatenate([gx, gz])
        full = np.concatenate([xc, np.zeros(self.c.k.N - n - m)])
        rc = np.max(np.abs(self.rhs - self.c.k.symv(full)[:n + m])) if self.c.k.N == n + m else float('nan')
        fullg = np.concatenate([xg, np.zeros(self.c.k.N - n - m)])
        rg = np.max(np.abs(self.rhs - self.c.k.symv(fullg)[:n + m])) if self.c.k.N == n + m else float('nan')
        print(f"D2 it {self.it:2d} solve ok {okg}/{okc} ir {self.g.last_ir_steps}/{self.c.last_ir_steps} rel_dx {np.max(np.abs(xg - xc)) / max(1, np.max(np.abs(xc))):.2e} |x| {np.max(np.abs(xc)):.2e} "
              f"res_trueK hip {rg:.2e} oracle {rc:.2e} |b| {np.max(np.abs(self.rhs)):.2e}")
        return okc

    def kktsolver_linear_solver_info(self):
        return self.c.kktsolver_linear_solver_info()

    def __getattr__(self, k):
        return getattr(self.c, k)


s = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: Both(*a))
sol = s.solve()
print("D2 done", sol.status, sol.iterations)
