"""What the dense update launches of a configuration are made of: tiles per stage, tasks per tile, source widths, share of tasks that
go through a tile map, full tiles (development tool; `hipkkt_debug_dump(20)`).  usage: python tools/dense_stage_stats.py [config]"""
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
    from clarabel_jl_amd.kktsolver import HipKKTSolver

    (P, q, A, b, cones), workload = bench.make_problem(cfg)
    solver = cl.Solver(P, q, A, b, cones, cl.Settings(device_id=0), kktsolver_factory=lambda *a: HipKKTSolver(*a))
    h = solver.kktsystem.kktsolver.h
    v = h.debug_dump(20).reshape(-1, 6)
    print(workload, ":", len(v), "dense tiles")
    print(" stage  tiles  full  tasks/tile(mean max)  sumK/tile(mean)  K/task  mapped%  Gflop  fill%")
    for l in np.unique(v[:, 0]):
        s = v[v[:, 0] == l]
        nt = s[:, 1].sum()
        print("%6d %6d %5d  %8.1f %5d  %12.1f  %7.1f  %6.1f  %6.2f  %5.1f" % (
            l, len(s), s[:, 4].sum(), s[:, 1].mean(), s[:, 1].max(), s[:, 2].mean(), s[:, 2].sum() / nt, 100 * s[:, 3].sum() / nt,
            2 * s[:, 5].sum() / 1e9, 100 * s[:, 5].sum() / (4096 * s[:, 2].sum())))
    g = h.debug_dump(21).reshape(-1, 5)
    print("persistent segment sweeps:", len(g), "supernodes")
    print(" level  supernodes  width(mean)  rows(mean max)  longest gather list (mean max)  entries per row slot")
    for l in np.unique(g[:, 0]):
        s = g[g[:, 0] == l]
        print("%6d %8d %10.1f %10.1f %6d %14.1f %6d %14.2f" % (l, len(s), s[:, 1].mean(), s[:, 2].mean(), s[:, 2].max(), s[:, 3].mean(), s[:, 3].max(), s[:, 4].mean()))


if __name__ == "__main__":
    main()
