import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver
from oracle.kkt_oracle import OracleKKTSolver
from tests.fixtures import scale_cones
prob = problems.random_sparse_qp(1000, 2000, 1, 4, 2)
P, q, A, b, specs = prob
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
m, n = A.shape
st = cl.Settings()
hk = HipKKTSolver(Pt, A, cones, m, n, st)
os.environ["HIPKKT_NO_FRONT"] = "1"
hn = HipKKTSolver(Pt, A, cones, m, n, st)
ok_ = OracleKKTSolver(Pt, A, cones, m, n, st, ordering=hk.h.perm())
o = ok_.k
rng = np.random.default_rng(5)
for rep in range(3):
    if rep == 0: cones.set_identity_scaling()
    else: scale_cones(cones, rng)
    assert hk.kktsolver_update(cones); assert hn.kktsolver_update(cones); assert ok_.kktsolver_update(cones)
    b = rng.standard_normal(o.N)
    xg = hk.h.ldl_solve(b); xn = hn.h.ldl_solve(b); xc = o.ldl_solve(b)
    print("rep", rep, "front-oracle", np.abs(xg - xc).max(), "nofront-oracle", np.abs(xn - xc).max(), "front-nofront", np.abs(xg - xn).max())
    rx, rz = rng.standard_normal(n), rng.standard_normal(m)
    lx, lz = np.zeros(n), np.zeros(m)
    hk.kktsolver_setrhs(rx, rz); print(" refined solve ok:", hk.kktsolver_solve(lx, lz), hk.last_ir_steps)
