#!/usr/bin/env python
"""Developer tool: factorisation time of a bench config for several values of the update batch (plan option).
usage: time_batch.py <cfg> <batch> [<batch> ...]"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
(P, q, A, b, specs), name = bench.make_problem(sys.argv[1])
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
for ub in [int(v) for v in sys.argv[2:]]:
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    hk = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings(), update_batch=ub)
    cones.set_identity_scaling()
    cones.get_Hs(hk.Hsblocks)
    hk.h.set_hs(hk.Hsblocks)
    ts = []
    for _ in range(4):
        ok, _, _ = hk.h.refactor(True, 1e-8, 4.9e-32)
        ts.append(hk.h.timing()["last_factor_ms"])
    print("update_batch", ub, "ok", ok, "factor ms", [round(t, 3) for t in ts])
