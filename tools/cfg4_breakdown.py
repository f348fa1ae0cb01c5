import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import clarabel_jl_amd
import julia_standin as cl
from clarabel_jl_amd import problems
tot = dict(create=0.0, solve=0.0, it=0)
for seed in range(100, 124):
    P, q, A, b, cones = problems.batch_problem(seed)
    t0 = time.time()
    s = cl.Solver(P, q, A, b, cones, cl.Settings())
    t1 = time.time()
    sol = s.solve()
    t2 = time.time()
    tot["create"] += t1 - t0; tot["solve"] += t2 - t1; tot["it"] += sol.iterations
    if seed < 106:
        h = s.kktsystem.kktsolver.h
        t = h.timing()
        print(seed, "n", P.shape[0], "m", A.shape[0], "create ms", round(1e3*(t1-t0),1), "solve ms", round(1e3*(t2-t1),1), "it", sol.iterations, "per it ms", round(1e3*(t2-t1)/sol.iterations,2),
              "gpu factor ms/it", round(t["acc_factor_ms"]/max(1,t["n_factor"]),3), "gpu solve ms/call", round(t["acc_solve_ms"]/max(1,t["n_solve_calls"]),3), "calls", t["n_solve_calls"])
print("24 problems: create", round(tot["create"],3), "s, solve", round(tot["solve"],3), "s, iterations", tot["it"], "-> single process", round(tot["it"]/(tot["create"]+tot["solve"]),1), "it/s; solve-only", round(tot["it"]/tot["solve"],1))
