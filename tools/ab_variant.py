#!/usr/bin/env python
"""Developer tool (A/B timing): device times of refactorisations and refined solves of a bench config under the environment the
process was started with (HIPKKT_* switches are read at handle creation).  Prints ONE line.
usage: [HIPKKT_...=..] ab_variant.py <cfg> <label> [reps]"""
import os, sys
if any(k.startswith("HIPKKT_") and k not in ("HIPKKT_VERBOSE", "HIPKKT_FB_TRACE") for k in os.environ) or __file__.endswith("chk_stream.py"):
    os.environ.setdefault("CLARABEL_HIPKKT_TESTING", "1")   # switches exist in the testing build of the library only
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd  # noqa: F401
import julia_standin as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
(P, q, A, b, specs), name = bench.make_problem(sys.argv[1])
label = sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
hk = HipKKTSolver(Pt, A, cones, A.shape[0], A.shape[1], cl.Settings())
rng = np.random.default_rng(0)
fx = __import__("tests.fixtures", fromlist=["scale_cones"])
fx.scale_cones(cones, rng)
cones.get_Hs(hk.Hsblocks)
ok = hk.kktsolver_update(cones)
n, m = A.shape[1], A.shape[0]
tf, ts = [], []
x = np.zeros(n); z = np.zeros(m)
xs = []
for i in range(reps):
    ok = hk._refactor() and ok
    tf.append(hk.h.timing()["last_factor_ms"])
    r = np.random.default_rng(100 + i)
    hk.kktsolver_setrhs(r.standard_normal(n), r.standard_normal(m))
    ok = hk.kktsolver_solve(x, z) and ok
    ts.append(hk.h.timing()["last_solve_ms"])
    xs.append(float(np.linalg.norm(x) + np.linalg.norm(z)))
cnt = hk.h.counters() if hasattr(hk.h, "counters") else {}
print(f"AB {sys.argv[1]:>3} {label:<28} ok={ok} factor_ms min {min(tf):.3f} med {sorted(tf)[len(tf)//2]:.3f} | solve_ms min {min(ts):.3f} med {sorted(ts)[len(ts)//2]:.3f} "
      f"| ir_steps {hk.last_ir_steps} | checksum {xs[-1]:.12e} | timeouts {cnt.get('sweep_timeouts', '?')}")
