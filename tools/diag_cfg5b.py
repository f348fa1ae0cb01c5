"""diagnostic: which entries of the LDL solve differ between the persistent and the per-level path (cfg 5)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import clarabel_jl_amd
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver
from tests.fixtures import scale_cones
def run(env):
    for k in ("HIPKKT_NO_PERSIST", "HIPKKT_SEG_TICKET"):
        os.environ.pop(k, None)
    os.environ.update(env)
    rng = np.random.default_rng(11)
    P, q, A, b, specs = problems.sdp_blocks(seed=5)
    cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
    Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
    A = sp.csc_matrix(A); A.sort_indices()
    m, n = A.shape
    hk = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
    scale_cones(cones, rng)
    assert hk.kktsolver_update(cones)
    b1 = rng.standard_normal(hk.h.N)
    xs = [hk.h.ldl_solve(b1) for _ in range(3)]
    return xs, hk.h.perm()
ref, perm = run({"HIPKKT_NO_PERSIST": "1"})
iperm = np.argsort(perm)
for mode in sys.argv[1:]:
    xs, _ = run({"HIPKKT_SEG_TICKET": mode})
    for i, x in enumerate(xs):
        d = np.abs(x - ref[0])
        bad = np.nonzero(d > 1e-9)[0]
        pos = np.sort(iperm[bad]) if len(bad) else bad
        print(f"ticket mode {mode} solve {i}: max diff {d.max():.3e}, #bad {len(bad)}, permuted positions of bad entries: {pos[:6]} ... {pos[-6:]}")
