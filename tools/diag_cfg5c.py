"""diagnostic: first (lowest-level) supernodes whose forward result differs between per-level and ticket mode"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import clarabel_jl_amd
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver
from tests.fixtures import scale_cones
def run(env, nsolve=1):
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
    out = []
    for _ in range(nsolve):
        x = hk.h.ldl_solve(b1)
        out.append((x, hk.h.debug_dump(1), hk.h.debug_dump(3)))
    tabs = [hk.h.debug_dump(w).astype(np.int64) for w in (10, 11, 12, 13, 14)]
    return out, tabs
ref, tabs = run({"HIPKKT_NO_PERSIST": "1"})
first, level, rows, parent, member = tabs
print("nsuper", len(level), "levels", level.max() + 1, "persistent supernodes", member.sum())
for mode in sys.argv[1:]:
    outs, _ = run({"HIPKKT_SEG_TICKET": mode}, nsolve=3)
    for i, (x, y, u) in enumerate(outs):
        dy = np.abs(y - ref[0][1])
        bad_sn = [s for s in range(len(level)) if dy[first[s]:first[s + 1]].max() > 1e-10]
        du = np.abs(u - ref[0][2])
        print(f"mode {mode} solve {i}: max |dx| {np.abs(x - ref[0][0]).max():.2e} |dy| {dy.max():.2e} |dubuf| {du.max():.2e} bad supernodes {len(bad_sn)}")
        if bad_sn:
            lv = level[bad_sn]
            lo = lv.min()
            for s in [s for s in bad_sn if level[s] == lo][:6]:
                w = first[s + 1] - first[s]
                print(f"   lowest bad: sn {s} level {level[s]} w {w} rows {rows[s]} blocks {max(1, -(-(rows[s] - w) // 64))} parent {parent[s]} member {member[s]} "
                      f"max dy {dy[first[s]:first[s+1]].max():.2e} at col {np.argmax(dy[first[s]:first[s+1]])}")
