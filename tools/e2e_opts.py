"""Developer tool: end-to-end IPM rate of the numpy stand-in caller with the widened plugin rows switched on one by one
(N4 device_residuals, N1 device_scaling, N2 device_reduced).  usage: e2e_opts.py [cfg ...]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import clarabel_jl_amd, bench
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
for cfg in (sys.argv[1:] or ("2a", "3", "2b")):
    (P, q, A, b, cones), name = bench.make_problem(cfg)
    for kw in ({}, {"device_residuals": True}, {"device_reduced": True}, {"device_reduced": True, "device_residuals": True},
               {"device_reduced": True, "device_residuals": True, "device_scaling": True}):
        best = None
        for rep in range(2):
            s = cl.Solver(P, q, A, b, cones, cl.Settings(**kw), kktsolver_factory=lambda *a: HipKKTSolver(*a))
            sol = s.solve()
            t = s.info.timers["IP iteration"]
            best = t if best is None else min(best, t)
        print("E2E", cfg, kw, sol.status, sol.iterations, "it/s", round(sol.iterations / best, 1), "obj %.12e" % sol.obj_val,
              {k: round(v * 1e3 / sol.iterations, 3) for k, v in s.info.timers.items() if k != "IP iteration"})
