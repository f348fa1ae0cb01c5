"""developer tool: phase breakdown of k_front_block2 (front_block2.hip) on cfg 2a from its stamps (HIPKKT_FB_TRACE=1)"""
import os, sys
os.environ["HIPKKT_FB_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import clarabel_jl_amd  # noqa
import julia_standin as cl
from clarabel_jl_amd import problems
from clarabel_jl_amd.kktsolver import HipKKTSolver
from tests.fixtures import scale_cones
P, q, A, b, specs = problems.random_sparse_qp()
cones = cl.CompositeCone(cl.cones_new_collapsed(specs))
Pt = sp.triu(sp.csc_matrix(P), format="csc"); Pt.sort_indices()
A = sp.csc_matrix(A); A.sort_indices()
m, n = A.shape
hk = HipKKTSolver(Pt, A, cones, m, n, cl.Settings())
scale_cones(cones, np.random.default_rng(1))
for _ in range(3):
    assert hk.kktsolver_update(cones)
print("factor ms", {k: round(float(v), 4) for k, v in hk.h.timing().items()})
raw = hk.h.debug_dump(9).view(np.int64).reshape(-1, 8, 16)
t = raw * 0.01   # microseconds
names = ["start", "loaded", "steps done", "first record", "streamed", "pivots start", "pivots end", "minv published", "end",
         "last step: minv seen", "X done", "diag tile done", "L flag raised", "L(i-1, j) seen"]
periods = []
for bi in range(len(t)):
    t0 = t[bi, 0, 0]
    piv = [t[bi, i, 5] - t0 for i in range(5) if t[bi, i, 5] > 0]
    if len(piv) > 1:
        periods.append(np.mean(np.diff(piv)))
    if bi not in (1, 8, 15):
        continue
    print(f"batch {bi}: stamps relative to workgroup 0's start (us)")
    for i in range(5):
        row = t[bi, i]
        print("  wg", i, " ".join(f"{nm}={row[k]-t0:7.2f}" for k, nm in enumerate(names) if row[k] > 0))
    tb = raw[bi, 5:8].reshape(-1)[:32].reshape(8, 4)
    if tb[0, 0] > 0:
        print("  wg 0 pivot loop, shader cycles per block [barrier A -> pivots done | -> barrier B | -> rank-8 done | -> next barrier A]:")
        for Bk in range(8):
            nxt = tb[Bk + 1, 0] - tb[Bk, 3] if Bk < 7 else 0
            print(f"    block {Bk}: {tb[Bk,1]-tb[Bk,0]:6d} {tb[Bk,2]-tb[Bk,1]:6d} {tb[Bk,3]-tb[Bk,2]:6d} {nxt:6d}")
        print(f"    total {tb[7,3]-tb[0,0]} cycles for 8 blocks; wall clock pivots {t[bi,0,6]-t[bi,0,5]:.2f} us")
    if len(piv) > 1:
        print(f"  chain: pivots start at {[round(float(v), 2) for v in piv]} -> {np.mean(np.diff(piv)):.2f} us per panel; pivots themselves "
              f"{np.mean([t[bi, i, 6] - t[bi, i, 5] for i in range(len(piv))]):.2f} us")
        for i in range(1, len(piv)):
            if t[bi, i, 3] > 0 and t[bi, i, 4] > 0:
                print(f"    wg {i}: first record {t[bi, i, 3] - t[bi, i - 1, 5]:5.2f} us after wg {i-1}'s pivot start, {(t[bi, i, 4] - t[bi, i, 3]) / 8.0:4.2f} us per record, "
                      f"streamed step done {t[bi, i, 4] - t[bi, i - 1, 6]:5.2f} us after wg {i-1}'s last pivot block (incl. its T and publish)")
if periods:
    print(f"all {len(periods)} batches: mean chain period {np.mean(periods):.2f} us per panel (min {np.min(periods):.2f}, max {np.max(periods):.2f})")
