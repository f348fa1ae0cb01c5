#!/usr/bin/env python
"""Developer tool: where does the time of the cfg-4 problems go?  Solves batch problems one after the other in each of <procs> processes
(each its own HIP context, like bench.py's workers) and prints wall ms of construct / solve next to the handle's own device clocks.
usage: [HIPKKT_...=..] cfg4_probe.py <label> <procs> [nseeds]"""
import os, sys, time
if any(k.startswith("HIPKKT_") and k not in ("HIPKKT_VERBOSE", "HIPKKT_FB_TRACE") for k in os.environ) or __file__.endswith("chk_stream.py"):
    os.environ.setdefault("CLARABEL_HIPKKT_TESTING", "1")   # switches exist in the testing build of the library only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def work(arg):
    label, seeds = arg
    import clarabel_jl_amd  # noqa: F401
    import julia_standin as cl
    from clarabel_jl_amd import problems as pr
    P, q, A, b, cones = pr.batch_problem(100)
    cl.Solver(P, q, A, b, cones, cl.Settings()).solve()        # warm-up: context, library, pools
    out = dict(construct=0.0, solve=0.0, dev_factor=0.0, dev_solve=0.0, its=0, timeouts=0, wall_factor=0.0, wall_solve=0.0, wall_update=0.0)
    for sd in seeds:
        P, q, A, b, cones = pr.batch_problem(sd)
        t0 = time.perf_counter()
        S = cl.Solver(P, q, A, b, cones, cl.Settings())
        t1 = time.perf_counter()
        k = S.kktsystem.kktsolver
        # wall clocks around the plugin's entry points
        for name, key in (("kktsolver_update", "wall_update"), ("kktsolver_solve", "wall_solve")):
            f = getattr(k, name)
            def wrap(*a, _f=f, _key=key, **kw):
                t = time.perf_counter(); r = _f(*a, **kw); out[_key] += time.perf_counter() - t; return r
            setattr(k, name, wrap)
        sol = S.solve()
        t2 = time.perf_counter()
        tm = k.h.timing(); c = k.h.counters()
        out["construct"] += t1 - t0; out["solve"] += t2 - t1; out["its"] += sol.iterations
        out["dev_factor"] += tm.get("acc_factor_ms", 0) * 1e-3; out["dev_solve"] += tm.get("acc_solve_ms", 0) * 1e-3
        out["timeouts"] += c.get("sweep_timeouts", 0)
    return out


if __name__ == "__main__":
    import multiprocessing as mp
    label, procs = sys.argv[1], int(sys.argv[2])
    ns = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    jobs = [(label, [100 + p + procs * i for i in range(ns)]) for p in range(procs)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(work, jobs)
    wall = time.perf_counter() - t0
    tot = {k: sum(r[k] for r in res) for k in res[0]}
    n = procs * ns
    print(f"CFG4 {label:<10} procs {procs} problems {n} iterations {tot['its']} | per problem (ms): construct {1e3*tot['construct']/n:.1f} solve {1e3*tot['solve']/n:.1f} "
          f"[update calls {1e3*tot['wall_update']/n:.1f} solve calls {1e3*tot['wall_solve']/n:.1f}] | device clocks: factor {1e3*tot['dev_factor']/n:.2f} solves {1e3*tot['dev_solve']/n:.2f} | "
          f"timeouts {tot['timeouts']} | pool wall {wall:.1f} s")
