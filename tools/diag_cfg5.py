"""diagnostic: linearity / determinism / oracle agreement of the unrefined LDL solve on cfg 5"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import clarabel_jl_amd
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver
from oracle.kkt_oracle import OracleKKTSolver
from tests.fixtures import scale_cones
rng = np.random.default_rng(11)
P, q, A, b, specs = problems.sdp_blocks(seed=5)
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
m, n = A.shape
st = cl.Settings()
hk = HipKKTSolver(Pt, A, cones, m, n, st)
scale_cones(cones, rng)
print("update", hk.kktsolver_update(cones), "nreg", hk.last_nreg, "eps", hk.diagonal_regularizer, hk.h.counters())
N = hk.h.N
b1, b2 = rng.standard_normal(N), rng.standard_normal(N)
x1, x1b, x2, x12 = hk.h.ldl_solve(b1), hk.h.ldl_solve(b1), hk.h.ldl_solve(b2), hk.h.ldl_solve(2.0 * b1 - 3.0 * b2)
print("determinism", np.max(np.abs(x1 - x1b)), "linearity", np.max(np.abs(x12 - (2 * x1 - 3 * x2))), "scale", np.max(np.abs(x1)))
if "oracle" in sys.argv:
    o = OracleKKTSolver(Pt, A, cones, m, n, st, ordering=hk.h.perm())
    print("oracle update", o.kktsolver_update(cones), "nreg", o.k.L.oracle_kkt_nreg(o.k.h))
    y1, y2, y12 = o.k.ldl_solve(b1), o.k.ldl_solve(b2), o.k.ldl_solve(2.0 * b1 - 3.0 * b2)
    print("oracle linearity", np.max(np.abs(y12 - (2 * y1 - 3 * y2))), "hip-oracle", np.max(np.abs(x1 - y1)))
    K = o.k
    r = b1 - K.symv(x1); r2 = b1 - K.symv(y1)
    print("residual hip", np.max(np.abs(r)), "oracle", np.max(np.abs(r2)))
