#!/usr/bin/env python
"""Developer tool: device time of one refactorisation of a bench config and of its update kernels
(HIP events, profiling mode).  usage: time_refactor.py <cfg> [reps]"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
(P, q, A, b, specs), name = bench.make_problem(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
st = cl.Settings()
hk = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], st)
cones.set_identity_scaling()
cones.get_Hs(hk.Hsblocks)
hk.h.set_hs(hk.Hsblocks)
cm = hk.h.cost_model()
for mode in (False, True):
    hk.h.set_profiling(mode)
    ts = []
    for _ in range(reps):
        ok, _, _ = hk.h.refactor(True, 1e-8, 4.9e-32)
        t = hk.h.timing()
        ts.append((t["last_factor_ms"], t["last_update_ms"]))
    print("profiling" if mode else "graph    ", "ok", ok, "factor ms", [round(x[0], 3) for x in ts], "update ms", [round(x[1], 3) for x in ts])
print("update flops %.3e -> %.2f TFLOP/s" % (cm["flops_update"], cm["flops_update"] / (ts[-1][1] * 1e-3) / 1e12))
