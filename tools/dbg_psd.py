import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clarabel_jl_amd as cl
from clarabel_jl_amd.kktsolver import HipKKTSolver
from oracle.kkt_oracle import OracleKKTSolver
from tests import fixtures as fx
P, q, A, b, specs = fx.basic_sdp()
specs = cl.cones_new_collapsed(specs); cones = cl.CompositeCone(specs)
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices(); A = sp.csc_matrix(A); A.sort_indices()
m, n = A.shape; st = cl.Settings()
hk = HipKKTSolver(Pt, A, cones, m, n, st)
ok_ = OracleKKTSolver(Pt, A, cones, m, n, st, ordering=hk.h.perm())
rng = np.random.default_rng(1)
fx.scale_cones(cones, rng)
hk.kktsolver_update(cones); ok_.kktsolver_update(cones)
_, _, kv = hk.h.kkt()
d = np.nonzero(kv != ok_.k.nzval)[0]
print("differ", len(d), "of", len(kv), "max abs", np.max(np.abs(kv - ok_.k.nzval)))
for i in d[:6]: print(i, kv[i], ok_.k.nzval[i], kv[i] - ok_.k.nzval[i])
print("psd off/dim", hk._psd_off, hk._psd_dim)
