#!/usr/bin/env python
"""Developer tool: a few refined solves of a bench config after one factorisation (front-sweep experiments).
usage: time_solve.py <cfg> [nsolves]"""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
(P, q, A, b, specs), name = bench.make_problem(sys.argv[1])
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
hk = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings())
cones.set_identity_scaling()
cones.get_Hs(hk.Hsblocks)
hk.h.set_hs(hk.Hsblocks)
ok, _, _ = hk.h.refactor(True, 1e-8, 4.9e-32)
n, m = A.shape[1], A.shape[0]
rng = np.random.default_rng(0)
x = np.zeros(n); z = np.zeros(m)
for i in range(ns):
    hk.h.setrhs(rng.standard_normal(n), rng.standard_normal(m))
    t0 = time.perf_counter()
    r = hk.h.solve(x, z)
    t1 = time.perf_counter()
    print("solve", i, "ok" if r is not False else r, "%.3f ms wall" % ((t1 - t0) * 1e3), hk.h.timing())
