"""Stamps of the forward front sweep's workgroups (development; library built with HIPKKT_EXTRA_FLAGS=-DHIPKKT_SWEEP_TRACE).
Per workgroup (ticket): 0 start, 1 before the last hop's poll, 2 after that hop, 3 after the first reduction, 4 after the exchange of
the partial right-hand sides, 5 after the second reduction, 6 after publishing y.  Times in us relative to the first start."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "2a"
    import clarabel_jl_amd  # noqa: F401
    import julia_standin as cl
    from clarabel_jl_amd import hipkkt
    from clarabel_jl_amd.kktsolver import HipKKTSolver

    (P, q, A, b, cones), workload = bench.make_problem(cfg)
    solver = cl.Solver(P, q, A, b, cones, cl.Settings(device_id=0, max_iter=3), kktsolver_factory=lambda *a: HipKKTSolver(*a))
    solver.solve()
    L = hipkkt.lib()
    out = np.zeros(8 * 512, dtype=np.int64)
    rc = L.hipkkt_debug_sweep_trace(out.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    t = out.reshape(512, 8).astype(np.float64)
    used = t[:, 0] > 0
    t0 = t[used, 0].min()
    print("ticket  start  pre-last-poll  after-last-hop  red1  exchange  red2  published   (us)")
    for k in np.nonzero(used)[0]:
        r = [(x - t0) / 100.0 if x > 0 else float("nan") for x in t[k, :7]]
        print("%5d " % k + " ".join("%9.2f" % x for x in r))


if __name__ == "__main__":
    main()
