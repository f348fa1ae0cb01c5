#!/usr/bin/env python
"""developer tool: dumps the stream records / scratch tiles / D of one factorisation of cfg 2a to gpurun_out/<label>.npz"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
from tests.fixtures import scale_cones
(P, q, A, b, specs), name = bench.make_problem("2a")
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
h = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings())
scale_cones(cones, np.random.default_rng(0))
assert h.kktsolver_update(cones)
np.savez_compressed(f"gpurun_out/{sys.argv[1]}.npz", stream=h.h.debug_dump(15).view(np.uint64), scratch=h.h.debug_dump(16), D=h.h.debug_dump(5))
print("dumped", sys.argv[1])
