"""developer tool: where does the time of one cfg-4 batch problem go (construction + IPM loop)?"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clarabel_jl_amd
import julia_standin as cl
from clarabel_jl_amd import problems
seeds = [int(a) for a in sys.argv[1:]] or [100, 137, 201]
for sd in seeds:   # warm-up of the library / device
    P, q, A, b, cones = problems.batch_problem(sd)
    cl.Solver(P, q, A, b, cones, cl.Settings()).solve()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
its = 0
for sd in seeds:
    P, q, A, b, cones = problems.batch_problem(sd)
    t1 = time.perf_counter()
    s = cl.Solver(P, q, A, b, cones, cl.Settings())
    t2 = time.perf_counter()
    sol = s.solve()
    t3 = time.perf_counter()
    its += sol.iterations
    h = s.kktsystem.kktsolver.h
    print(f"seed {sd}: n={P.shape[0]} m={A.shape[0]} N={h.N} nnzL={h.nnzL} levels={h.nlevels} gen {t1-t0:.4f}s ctor {t2-t1:.4f}s solve {t3-t2:.4f}s its {sol.iterations} {sol.status}  device factor {h.timing()['acc_factor_ms']:.2f} ms solve {h.timing()['acc_solve_ms']:.2f} ms")
    t0 = time.perf_counter()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
