#!/usr/bin/env python
"""Developer tool: solve one bench config end-to-end on the HIP path, verbose.  usage: run_cfg.py <cfg>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
(P, q, A, b, cones), name = bench.make_problem(sys.argv[1])
t0 = time.time()
s = cl.Solver(P, q, A, b, cones, cl.Settings(verbose=True))
print("setup", time.time() - t0, "ordering/levels", s.kktsystem.kktsolver.h.nlevels, "nnzL", s.kktsystem.kktsolver.h.nnzL)
sol = s.solve()
ks = s.kktsystem.kktsolver
print(sol.status, sol.iterations, sol.obj_val, sol.r_prim, sol.r_dual, "ir steps total", ks.total_ir_steps, "solves", ks.nsolves)
