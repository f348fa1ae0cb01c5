"""developer tool: cProfile of the end-to-end IPM loop (host stand-in + HIP KKT path) for one bench config"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import clarabel_jl_amd
import julia_standin as cl
(P, q, A, b, cones), name = bench.make_problem(sys.argv[1])
s = cl.Solver(P, q, A, b, cones, cl.Settings())
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); sol = s.solve(); t1 = time.perf_counter()
pr.disable()
print(name, sol.status, sol.iterations, f"{t1 - t0:.3f}s", {k: round(v, 4) for k, v in s.info.timers.items()})
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
