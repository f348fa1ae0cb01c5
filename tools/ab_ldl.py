#!/usr/bin/env python
"""Developer tool (A/B timing): device time of plain LDL solves (no refinement) of a bench config under the current HIPKKT_* environment.
usage: ab_ldl.py <cfg> <label> [reps]"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
(P, q, A, b, specs), name = bench.make_problem(sys.argv[1])
label = sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
hk = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings())
fx = __import__("tests.fixtures", fromlist=["scale_cones"])
fx.scale_cones(cones, np.random.default_rng(0))
ok = hk.kktsolver_update(cones)
tf = hk.h.timing()["last_factor_ms"]
ts = []
for i in range(reps):
    x = hk.h.ldl_solve(np.random.default_rng(i).standard_normal(hk.h.N))
    ts.append(hk.h.timing()["last_solve_ms"])
print(f"LDL {sys.argv[1]:>3} {label:<24} ok={ok} factor_ms {tf:.3f} | ldl_solve_ms min {min(ts):.4f} med {sorted(ts)[len(ts)//2]:.4f} | |x| {np.linalg.norm(x):.6e}")
