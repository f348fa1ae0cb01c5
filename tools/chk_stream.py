#!/usr/bin/env python
"""developer tool: unrefined LDL solve of the streamed pivot chain against the whole-tile hand-off (HIPKKT_FB_STREAM=0) on a bench config,
over several factorisations with DIFFERENT values.   usage: chk_stream.py <cfg> <label>"""
import os, sys
if any(k.startswith("HIPKKT_") and k not in ("HIPKKT_VERBOSE", "HIPKKT_FB_TRACE") for k in os.environ) or __file__.endswith("chk_stream.py"):
    os.environ.setdefault("CLARABEL_HIPKKT_TESTING", "1")   # switches exist in the testing build of the library only
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
from tests.fixtures import scale_cones
(P, q, A, b, specs), name = bench.make_problem(sys.argv[1])
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
os.environ["HIPKKT_PLAN_CACHE"] = "0"
os.environ["HIPKKT_FB_STREAM"] = "0"
h0 = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings())
os.environ["HIPKKT_FB_STREAM"] = "1"
h1 = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings())
rng = np.random.default_rng(0)
rhs = rng.standard_normal(h0.h.N)
out = []
for rep in range(4):
    scale_cones(cones, rng)
    assert h0.kktsolver_update(cones) and h1.kktsolver_update(cones)
    x0, x1 = h0.h.ldl_solve(rhs), h1.h.ldl_solve(rhs)
    d0, d1 = h0.h.debug_dump(5), h1.h.debug_dump(5)
    out.append((float(np.max(np.abs(x1 - x0)) / max(1.0, np.max(np.abs(x0)))), float(np.max(np.abs(d1 - d0) / np.abs(d0))), int(np.argmax(np.abs(d1 - d0) / np.abs(d0)))))
print("CHK", sys.argv[1], sys.argv[2], " ".join(f"[dx {a:.2e} dD {b_:.2e} @{k}]" for a, b_, k in out), "timeouts", h1.h.counters()["sweep_timeouts"], "N", h0.h.N)
