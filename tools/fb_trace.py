"""developer tool: phase breakdown of k_front_block (front_block.hip) on cfg 2a from its wall-clock stamps (HIPKKT_FB_TRACE=1)"""
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
print("factor ms", hk.h.timing())
t = hk.h.debug_dump(9).view(np.int64).reshape(-1, 8, 16) * 0.01   # microseconds
names = ["start", "loaded", "minv seen", "minv in LDS", "trsm", "stores+pub", "updates", "tile->regs", "pivoted", "inverse", "published", "end",
         "first record", "streamed"]
for bi in (1, 8, 15):
    if bi >= len(t): continue
    t0 = t[bi, 0, 0]
    print(f"batch {bi}: stamps relative to workgroup 0's start (us)")
    for i in range(5):
        row = t[bi, i]
        print("  wg", i, " ".join(f"{nm}={row[k]-t0:7.2f}" for k, nm in enumerate(names) if k < 14 and row[k] > 0))
    tb = t[bi, 5].reshape(-1) / 0.01          # raw shader-clock stamps of workgroup 0's pivot loop
    tb = hk.h.debug_dump(9).view(np.int64).reshape(-1, 8, 16)[bi, 5:8].reshape(-1)[:32].reshape(8, 4)
    if tb[0, 0] > 0:
        print("  wg 0 pivot loop, shader cycles per block [barrier A -> pivots done | -> barrier B | -> rank-8 done | -> next barrier A]:")
        for Bk in range(8):
            nxt = tb[Bk + 1, 0] - tb[Bk, 3] if Bk < 7 else 0
            print(f"    block {Bk}: {tb[Bk,1]-tb[Bk,0]:6d} {tb[Bk,2]-tb[Bk,1]:6d} {tb[Bk,3]-tb[Bk,2]:6d} {nxt:6d}")
        print(f"    total {tb[7,3]-tb[0,0]} cycles for 8 blocks; wall clock pivots {t[bi,0,8]-t[bi,0,7]:.2f} us")
    # the chain: start of the pivots of consecutive diagonal workgroups ("tile->regs" = stamp 7), end of the last one's pivots
    piv = [t[bi, i, 7] - t0 for i in range(5) if t[bi, i, 7] > 0]
    if len(piv) > 1:
        print(f"  chain: pivots start at {[round(v, 2) for v in piv]} -> {np.mean(np.diff(piv)):.2f} us per panel; pivots themselves "
              f"{np.mean([t[bi, i, 8] - t[bi, i, 7] for i in range(len(piv))]):.2f} us")
        # streamed chain: what one panel of the chain is made of: lead = the consumer's first record after the
        # producer's pivot start (it needs the tile L(i, i-1) first), then 8 records; tail = streamed -> its own pivot start
        for i in range(2, len(piv)):
            if t[bi, i, 12] > 0 and t[bi, i, 13] > 0:
                lead = t[bi, i, 12] - t[bi, i - 1, 7]
                per_rec = (t[bi, i, 13] - t[bi, i, 12]) / 8.0
                behind = t[bi, i, 13] - t[bi, i - 1, 8]
                print(f"    wg {i}: first record {lead:5.2f} us after wg {i-1}'s pivot start, {per_rec:4.2f} us per record, last record done "
                      f"{behind:5.2f} us after wg {i-1}'s last pivot (producer: {(t[bi, i-1, 8] - t[bi, i-1, 7]) / 8.0:4.2f} us per block)")
