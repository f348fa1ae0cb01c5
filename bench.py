#!/usr/bin/env python
"""bench.py — IPM-iterations/s of the MI355X-native KKT path (BASELINE.json metric).

A *step* = one KKT iteration unit of the hot path = what `kkt_update!` + 2x `kkt_solve!` do per IPM
iteration (reference src/solver.jl:278-323): 1 Hs value update + static regularisation + 1 numeric
LDL^T + 3 solves, each with iterative refinement, replayed from the (Hs, rhs) trace recorded while
the problem is solved end-to-end once.  In the timed region every input is already resident in HBM
(torch tensors -> device pointers through the `_dev` entry points of the C ABI).

Workload at N=1: BASELINE.json configs[1] = random sparse QP n=10000 m=20000, NN cone ("2a",
uniform random pattern: SURVEY.md §8d).  N>1: one independent problem per GPU (different seed),
no data-path collective; RCCL only for the start/stop barrier + max-over-ranks time.

One JSON line on rank 0 (see DESIGN.md §6 for the roofline / cpu_baseline definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 matrix (not in the local guides)


def make_problem(cfg, seed_shift=0):
    from clarabel_jl_amd import problems

    if cfg == "2a":
        return problems.random_sparse_qp(10000, 20000, 2 + seed_shift, 3, 1), "random sparse QP n=10000 m=20000 NN cone, uniform pattern (cfg 2a)"
    if cfg == "2b":
        return problems.random_sparse_qp(10000, 20000, 2 + seed_shift, 4, 2, window=50), "random sparse QP n=10000 m=20000 NN cone, banded +-50 (cfg 2b)"
    if cfg == "1":
        return problems.random_sparse_qp(1000, 2000, 1 + seed_shift, 4, 2), "random sparse QP n=1000 m=2000 NN cone (cfg 1)"
    if cfg == "3":
        return problems.portfolio_socp(seed=3 + seed_shift), "portfolio QP + 50 SOC(101), n=5000 (cfg 3)"
    if cfg == "5":
        return problems.sdp_blocks(seed=5 + seed_shift), "SDP 20 x PSDTriangle(50), n=1000 (cfg 5)"
    raise SystemExit(f"unknown config {cfg}")


def kernel_sources_hash():
    """sha1 over the HIP sources of the kernels the counters describe: a counter file measured on other sources is stale"""
    import hashlib

    h = hashlib.sha1()
    for f in ("kernels.hip", "dense_tile.h", "front_block.hip", "front_block2.hip", "front_sweep.hip", "device_plan.h", "symbolic.cpp"):
        with open(os.path.join(ROOT, "clarabel.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def pmc_counters(cfg):
    """Hardware-counter figures of the dominant kernel family (the big dense updates) from the newest committed
    profiles/r*_cfg<cfg>_counters.json (written by tools/pmc_to_json.py from SEPARATE rocprofv3 --pmc passes of this very command:
    FETCH_SIZE, WRITE_SIZE, and the SQ matrix-core counters).  The counters cannot be collected from inside the timed run, so they
    are read from profiles/ -- and only trusted when the file carries the hash of the kernel sources this run was built from;
    otherwise (None, reason)."""
    import glob

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_cfg{cfg}_counters.json")))
    if not cands:
        return None, "no counter file for this workload under profiles/"
    want, seen = kernel_sources_hash(), []
    for path in reversed(cands):             # newest first; the first one measured on THESE kernel sources counts
        try:
            with open(path) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("kernel_sources_sha1") == want:
            d["source"] = os.path.relpath(path, ROOT)
            return d, None
        seen.append(f"{os.path.relpath(path, ROOT)} ({d.get('kernel_sources_sha1')})")
    return None, "measured on other kernel sources: " + ", ".join(seen[:3]) + "; re-run tools/final_round.sh"


def _cpu_batch_worker(arg):
    """all-cores CPU baseline of cfg 4: one problem per core, the oracle (reference :qdldl restatement) as KKT solver -- on the
    elimination order the HIP path used for the same problem (arg = (seed, perm)), so that the two runs are comparable at 1e-10;
    perm = None: the oracle's own order (SuperLU MMD on K)"""
    import time as _t

    import clarabel_jl_amd  # noqa: F401
    import julia_standin as cl_
    from clarabel_jl_amd import problems as pr_
    from oracle.kkt_oracle import OracleKKTSolver

    seed, perm = arg
    t0 = _t.perf_counter()
    P, q, A, b, cones = pr_.batch_problem(seed)
    order = "mmd" if perm is None else perm
    sol = cl_.Solver(P, q, A, b, cones, cl_.Settings(), kktsolver_factory=lambda *a: OracleKKTSolver(*a, ordering=order)).solve()
    return sol.iterations, sol.status, _t.perf_counter() - t0, float(sol.obj_val), seed, float(sol.r_prim), float(sol.r_dual)


def _alg_bytes(S_, iterations):
    """algorithmic HBM bytes of one solved problem (B_iter without PCIe): per iteration one factorisation and three
    refined solves of two LDL solves + two SpMVs each, from the symbolic factor actually used"""
    cm = S_.kktsystem.kktsolver.h.cost_model()
    return float(iterations) * (cm["bytes_factor"] + 6.0 * (cm["bytes_solve"] + cm["bytes_spmv"]))


def _gpu_batch_init(device, cpus=None):
    """worker process of the cfg-4 batch driver: pinned to the cores next to the rank's GPU, one HIP context of its own on that GPU,
    warmed up"""
    import clarabel_jl_amd  # noqa: F401
    import julia_standin as cl_
    from clarabel_jl_amd import batch as b_
    from clarabel_jl_amd import problems as pr_

    b_.pin_process(cpus)
    global _W
    _W = (cl_, pr_, device)
    P, q, A, b, cones = pr_.batch_problem(100)
    cl_.Solver(P, q, A, b, cones, cl_.Settings(device_id=device)).solve()


def _gpu_batch_chunk(arg):
    seeds, in_flight = arg
    from clarabel_jl_amd import batch as b_

    cl_, pr_, device = _W

    def one(seed):
        P, q, A, b, cones = pr_.batch_problem(seed)
        S_ = cl_.Solver(P, q, A, b, cones, cl_.Settings(device_id=device))
        # the elimination order the oracle is held to in the parity leg: taken BEFORE the solve (afterwards the handle reports its
        # robust-order twin's permutation while the last factorisation lives there)
        perm = S_.kktsystem.kktsolver.h.perm()
        sol = S_.solve()
        return (seed, sol.iterations, sol.status, float(sol.obj_val), _alg_bytes(S_, sol.iterations), perm,
                float(sol.r_prim), float(sol.r_dual), int(S_.kktsystem.kktsolver.h.counters()["twin_refactors"]))

    return b_.run_concurrent(one, seeds, in_flight)


def bench_batch(args, cl, torch, dist, rank, world, local):
    """cfg 4: the 256 seeded Maros-Meszaros-like QPs (problems.batch_problem, seeds 100..355) sharded round-robin
    over the ranks; a step = one problem solved end-to-end on the HIP path (symbolic set-up + IPM loop with the
    numpy stand-in caller), `--in-flight` problems at a time per GPU (one host thread + handle each).
    value = IPM iterations / s over the whole job."""
    from clarabel_jl_amd import batch, problems

    dev = torch.device("cuda", local)
    mine = batch.shard(256, rank, world)
    steps = min(args.steps, len(mine)) if args.steps > 0 else len(mine)

    def solve_one(k):
        P, q, A, b, cones = problems.batch_problem(100 + mine[k % len(mine)])
        S_ = cl.Solver(P, q, A, b, cones, cl.Settings(device_id=local))
        sol = S_.solve()
        return (100 + mine[k % len(mine)], sol.iterations, sol.status, float(sol.obj_val), _alg_bytes(S_, sol.iterations),
                S_.kktsystem.kktsolver.h.perm(), float(sol.r_prim), float(sol.r_dual))

    res = []
    pool = None
    if args.workers > 1:
        # several host processes per GPU (each its own HIP context, `--in-flight` threads inside each): the per-problem work is
        # mostly host-side (symbolic analysis, the numpy caller, call latencies), and Python threads alone stop scaling at ~2
        import multiprocessing as mp

        # one BLAS / OpenMP thread per worker, workers pinned to the cores of the GPU's NUMA node (an even slice of the allowed cores
        # where sysfs does not tell): 8 ranks x 6 workers must not fight over the same cores
        batch.cap_host_threads(1)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        try:
            pr = torch.cuda.get_device_properties(local)
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            bdf = None
        cpus = batch.gpu_numa_cpus(local, local_world, bdf)
        pool = mp.get_context("spawn").Pool(args.workers, initializer=_gpu_batch_init, initargs=(local, cpus))
        pool.map(_gpu_batch_chunk, [([100 + mine[k % len(mine)]], 1) for k in range(args.workers)])      # warm-up: every worker once
        seeds = [100 + mine[(args.warmup + i) % len(mine)] for i in range(steps)]
        chunks = [(seeds[w::args.workers], args.in_flight) for w in range(args.workers)]

        def timed(_):
            for part in pool.map(_gpu_batch_chunk, chunks):
                res.extend(part)
    else:
        batch.run_concurrent(solve_one, list(range(min(args.warmup, len(mine)))), args.in_flight)      # warm-up

        def timed(_):
            res.extend(batch.run_concurrent(solve_one, [args.warmup + i for i in range(steps)], args.in_flight))

    elapsed = batch.timed_steps(timed, 1, 0, dist=dist, device_sync=torch.cuda.synchronize, reduce_device=dev)
    if pool is not None:
        pool.close()
        pool.join()
    total_iters = batch.gather_counts(sum(r[1] for r in res), dist, dev)
    total_solved = batch.gather_counts(sum(r[2] == "SOLVED" for r in res), dist, dev)
    total_probs = batch.gather_counts(steps, dist, dev)
    not_solved = [(r[0], r[2]) for r in res if r[2] != "SOLVED"]
    total_bytes = batch.gather_counts(int(sum(r[4] for r in res)), dist, dev)
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # all-cores CPU baseline: the same problems, one per core, oracle KKT solver, bounded sample
        import multiprocessing as mp

        # Round 6 (review of round 5): ALL cores of the box, not 64 -- and scheduled the way a CPU batch would be: every problem of
        # the batch, one at a time per worker, the big ones first (with 64 workers and static chunks the longest pair of problems set
        # the wall time: 25 % parallel efficiency, which flattered the GPU side)
        ncores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), args.cpu_cores or 256)
        by_seed = {r[0]: r for r in res}
        seeds = sorted((r[0] for r in res), key=lambda sd: -by_seed[sd][4])          # (descending algorithmic bytes: a size proxy)
        jobs = [(sd, by_seed[sd][5]) for sd in seeds]
        t0 = time.perf_counter()
        # one BLAS / OpenMP thread per worker, like the GPU side's workers (inherited by the spawned processes): 256 workers with a
        # thread pool each oversubscribe the box four times over (measured: 8 instead of 31 iterations/s per core)
        saved_env = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS")}
        for k in saved_env:
            os.environ[k] = "1"
        with mp.get_context("spawn").Pool(ncores) as pool:
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            t_spawn = time.perf_counter() - t0
            pool.map(_cpu_batch_worker, jobs[-ncores:], chunksize=1)               # warm the workers (imports, oracle build) on the small ones
            t1 = time.perf_counter()
            out = list(pool.imap(_cpu_batch_worker, jobs, chunksize=1))
            t_cpu = time.perf_counter() - t1
        cpu_baseline = {"value": round(sum(o[0] for o in out) / t_cpu, 2), "unit": "IPM-iterations/s (whole solves incl. set-up)",
                        "cores": ncores, "kind": "port",
                        "sample": f"all {len(seeds)} problems of the batch on {ncores} worker processes (one problem at a time per worker, big ones first), "
                                  "oracle/ C restatement of the :qdldl path inside the same numpy caller, on the HIP path's elimination order", "problems": len(seeds),
                        "wall_s": round(t_cpu, 3), "sum_of_per_problem_s": round(sum(o[2] for o in out), 3),
                        "one_core_iterations_per_s": round(sum(o[0] for o in out) / sum(o[2] for o in out), 2),
                        "host_cores_available": os.cpu_count(), "pool_start_s": round(t_spawn, 2)}
        # parity of the same run: the problems of the CPU sample, HIP path vs the oracle ON THE SAME ELIMINATION ORDER, whole IPM solves:
        # status equal, iterations equal, objective (relative) and residuals (absolute) to 1e-10 -- the gate of BASELINE.md and of
        # tests/test_gpu_fullsize.py::test_batch_config_matches_oracle.  A problem outside it is not waved through by a wider gate:
        # the oracle is run once more on ITS OWN order (SuperLU MMD) and what the reference's arithmetic itself moves by between the
        # two orders (CPU vs CPU) is recorded as the cause; "explained" = within 1e-10 + 4 x that spread and iterations within 1.
        st_eq = it_eq = 0
        max_dobj = max_dres = 0.0
        exceptions = []
        for o in out:
            g_ = by_seed[o[4]]
            st_eq += g_[2] == o[1]
            it_eq += g_[1] == o[0]
            dobj = abs(g_[3] - o[3]) / max(1.0, abs(o[3])) if g_[2] == o[1] and o[1] in ("SOLVED", "ALMOST_SOLVED") else (0.0 if g_[2] == o[1] else float("inf"))
            dres = max(abs(g_[6] - o[5]), abs(g_[7] - o[6])) if np.isfinite(dobj) else float("inf")
            if g_[1] == o[0] and dobj <= 1e-10 and dres <= 1e-10:
                max_dobj, max_dres = max(max_dobj, dobj), max(max_dres, dres)
                continue
            o2 = _cpu_batch_worker((o[4], None))
            sp_obj = abs(o2[3] - o[3]) / max(1.0, abs(o[3]))
            sp_res = max(abs(o2[5] - o[5]), abs(o2[6] - o[6]))
            ok_ = bool(g_[2] == o[1] and abs(g_[1] - o[0]) <= 1 and abs(o2[0] - o[0]) <= 1 and dobj <= 1e-10 + 4.0 * sp_obj and dres <= 1e-10 + 4.0 * sp_res)
            exceptions.append({"seed": o[4], "iterations_hip_oracle_oracle_mmd": [g_[1], o[0], o2[0]], "rel_dobj": float(f"{dobj:.3e}"),
                               "dres": float(f"{dres:.3e}"), "oracle_own_spread_obj": float(f"{sp_obj:.3e}"),
                               "oracle_own_spread_res": float(f"{sp_res:.3e}"), "explained": ok_,
                               "cause": ("inside the oracle's own spread between two elimination orders (CPU vs CPU)" if ok_ else
                                         "NOT explained by the ordering spread (see tests/test_gpu_fullsize.py: refinement-branch shadow run)")})
        parity = {"problems_compared": len(out), "status_equal": st_eq, "iterations_equal": it_eq,
                  "within_1e-10": len(out) - len(exceptions), "max_rel_dobj_of_those": float(f"{max_dobj:.3e}"),
                  "max_dres_of_those": float(f"{max_dres:.3e}"), "tolerance": 1e-10, "exceptions": exceptions,
                  "pass": bool(st_eq == len(out) and all(e["explained"] for e in exceptions)),
                  "note": "HIP path vs the oracle on the SAME elimination order, whole IPM solves of the same problems in the same run; "
                          "every problem outside 1e-10 is listed with its measured cause"}
    if rank == 0:
        # measured HBM traffic (separate rocprofv3 --pmc passes of a ONE-process sample of this workload, tools/latency_counters.sh):
        # bytes of every kernel in the trace per refactorisation ~ per IPM iteration, beside the algorithmic bytes per IPM iteration
        ctr4, why4 = pmc_counters("4")
        wt4 = (ctr4 or {}).get("whole_trace")
        alg_it = total_bytes / max(1, total_iters)
        traffic4 = None if not wt4 else dict(
            bytes_per_refactor=wt4["bytes_per_refactor"], read_x2_MB=wt4["read_x2_MB"], write_MB=wt4["write_MB"],
            dispatches_per_refactor=wt4.get("dispatches_per_refactor"), refactorisations_in_trace=wt4.get("refactorisations_in_trace"),
            algorithmic_bytes_per_ipm_iteration=round(alg_it), ratio_to_algorithmic=round(wt4["bytes_per_refactor"] / alg_it, 2) if alg_it else None,
            source=ctr4["source"], sample=wt4.get("command"),
            note="FETCH_SIZE x2 + WRITE_SIZE of EVERY kernel of the sample's trace / its refactorisations (one per IPM iteration + each problem's "
                 "set-up); the algorithmic figure is this run's mean over all 256 problems")
        print(json.dumps({
            "metric": "IPM iterations/sec + KKT factor+solve ms, 10k-var sparse QP, 1/2/4/8 GPU",
            "value": round(total_iters / elapsed, 3), "unit": "IPM-iterations/s (whole solves incl. set-up, batch of independent problems)",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(1, steps), 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "batch of 256 Maros-Meszaros-like QPs (cfg 4), seeds 100..355, sharded round-robin",
                       "problems_solved": total_probs, "status_solved": total_solved, "not_solved": not_solved,
                       "in_flight_per_gpu": args.in_flight * max(1, args.workers), "host_processes_per_gpu": max(1, args.workers),
                       "threads_per_process": args.in_flight,
                       "host_cores_kept_busy": batch.host_core_budget(world, args.workers, args.in_flight), "host_cores_available": os.cpu_count(),
                       "parallelism": f"{world} rank(s), {len(mine)} problems on rank 0"},
            "problems_per_s": round(total_probs / elapsed, 3),
            "roofline": {"bound": "hbm", "achieved": round(total_bytes / elapsed / 1e9, 3), "peak": 8000.0 * world, "unit": "GB/s",
                         "frac": round(total_bytes / elapsed / 1e9 / (8000.0 * world), 6),
                         "traffic": None if traffic4 is None else traffic4["bytes_per_refactor"], "traffic_of": "one refactorisation ~ one IPM iteration, all kernels",
                         "traffic_detail": traffic4, "traffic_unavailable": why4,
                         "note": "algorithmic bytes of all solved problems (B_factor + 6 (B_solve + B_spmv) per IPM iteration, from each "
                                 "problem's symbolic factor) / wall time: a batch of small problems is bound by host work and launch / "
                                 "dependency latency, not by HBM -- the fraction says how far, it is not a kernel-quality figure"},
            "cpu_baseline": cpu_baseline,
            "gpu_vs_cpu_batch": None if not cpu_baseline else {
                "one_gpu_over_all_host_cores": round(total_iters / elapsed / cpu_baseline["value"], 3),
                "winner": "the host's CPU cores" if cpu_baseline["value"] > total_iters / elapsed else "the GPU",
                "note": f"one MI355X with {max(1, args.workers)} host processes against the CPU restatement on {cpu_baseline['cores']} cores of the same box, "
                        "both inside the same numpy stand-in of the caller (whose Python costs ~0.8 ms per IPM iteration and 10 - 30 ms per Solver "
                        "construction on either side: tools/cfg4_breakdown.py)"},
            "parity": parity}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; cfg 4: the whole batch of 256 problems)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="2a")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cores", type=int, default=0, help="cfg 4: worker processes of the CPU batch baseline (0 = every core of the box, at most 256)")
    ap.add_argument("--cpu-budget", type=float, default=80.0,
                    help="seconds of CPU-oracle work allowed for cpu_baseline / parity (default: three KKT iteration units of the headline "
                         "config, 23 s each on one host core: SURVEY section 8d asks for >= 3)")
    ap.add_argument("--update-policy", type=int, default=None)
    ap.add_argument("--update-batch", type=int, default=None, help="levels per update batch (default: automatic)")
    ap.add_argument("--in-flight", type=int, default=1, help="cfg 4: problems solved concurrently per host process (threads)")
    ap.add_argument("--workers", type=int, default=6, help="cfg 4: host processes per GPU, each with its own HIP context")
    ap.add_argument("--device-scaling", action="store_true", help="also solve once end to end with N1 on (update_scaling!/get_Hs! on the device) and report it under end_to_end")
    ap.add_argument("--sequential-solves", action="store_true", help="three separate solve calls per unit instead of 2 concurrent + 1")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 0 if args.config == "4" else 20

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the KKT path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import clarabel_jl_amd  # noqa: F401
    import julia_standin as cl
    from clarabel_jl_amd.kktsolver import HipKKTSolver

    if args.config == "4":
        return bench_batch(args, cl, torch, dist, rank, world, local)

    (P, q, A, b, cones), workload = make_problem(args.config, seed_shift=rank)

    # ---- 1. solve once end-to-end on the HIP path, recording the KKT inputs of every iteration
    trace = []

    class Recorder(HipKKTSolver):
        def kktsolver_update(self, cones_):
            # PSD blocks are formed on the device during the solve (hipkkt_set_hs_psd); the replay wants the whole Hs
            # vector as data, so W = R R^T is kept here and expanded on the host AFTER the timed end-to-end solve
            psd_w = [(None if c._hs_valid else c.RRt.copy()) for c in self._psd_cones]
            ok = super().kktsolver_update(cones_)
            trace.append(dict(hs=self.Hsblocks.copy(), psd_w=psd_w, u=self._u.copy(), v=self._v.copy(), eta2=self._eta2.copy(), rhs=[]))
            return ok

        def kktsolver_setrhs(self, rhsx, rhsz):
            self._last_rhs = np.concatenate([rhsx, rhsz])
            super().kktsolver_setrhs(rhsx, rhsz)

        def kktsolver_solve(self, lhsx, lhsz):
            if trace:
                trace[-1]["rhs"].append(self._last_rhs)
            return super().kktsolver_solve(lhsx, lhsz)

        def kktsolver_solve_multi(self, rhsx, rhsz, lhsx, lhsz):
            if trace:
                for rx, rz in zip(rhsx, rhsz):
                    trace[-1]["rhs"].append(np.concatenate([rx, rz]))
            return super().kktsolver_solve_multi(rhsx, rhsz, lhsx, lhsz)

    optkw = {}
    if args.update_policy is not None:
        optkw["update_policy"] = args.update_policy
    if args.update_batch is not None:
        optkw["update_batch"] = args.update_batch
    st = cl.Settings(device_id=local)
    t0 = time.perf_counter()
    solver = cl.Solver(P, q, A, b, cones, st, kktsolver_factory=lambda *a: Recorder(*a, **optkw))
    t_setup = time.perf_counter() - t0
    sol = solver.solve()
    ks = solver.kktsystem.kktsolver
    h = ks.h
    rec_iters, rec_time = sol.iterations, solver.info.timers["IP iteration"]     # (this run also copies every Hs / rhs into the trace)
    # End-to-end rate of the NUMPY STAND-IN of the Julia caller (julia_standin/ipm.py; no Julia has run): the
    # same problem solved again by the plain plugin, (a) through the reference's own call sequence (L1 contract only), (b) with the
    # N2 hook on (reduced-system algebra of kkt_solve! by the plugin: what julia/clarabel_l1_seam.patch wires), (c) with N2 + N4 on
    # (residuals_update! by the plugin too: NOT wired on the Julia side -- solver.jl / residuals.jl stay untouched, INTEGRATION.md
    # section 5).  The headline is (b); three runs each, the MEDIAN counts.
    e2e_runs = {}
    for tag, kw in (("l1_contract_only", {}), ("n2_hook", {"device_reduced": True}), ("n2_n4_hooks", {"device_reduced": True, "device_residuals": True})):
        rates = []
        for _rep in range(3):     # (wall time of a 0.1-0.2 s host loop: three runs, all reported, the median counts)
            s_ = cl.Solver(P, q, A, b, cones, cl.Settings(device_id=local, **kw), kktsolver_factory=lambda *a: HipKKTSolver(*a, **optkw))
            sol_ = s_.solve()
            rates.append(round(sol_.iterations / s_.info.timers["IP iteration"], 4))
            e2e_runs[tag] = {"status": sol_.status, "ipm_iterations": sol_.iterations, "iterations_per_s": float(np.median(rates)), "iterations_per_s_runs": list(rates),
                             "objective_rel_diff_vs_recording_run": float(abs(sol_.obj_val - sol.obj_val) / max(1.0, abs(sol.obj_val)))}
            del s_
    e2e_iters = e2e_runs["n2_hook"]["ipm_iterations"]
    e2e_rate = e2e_runs["n2_hook"]["iterations_per_s"]
    tm = h.timing()
    # keep iterations that carry the regular 3 solves (the initial factorisation has 2 or 3)
    units = [t for t in trace if len(t["rhs"]) == 3]
    if not units:
        raise SystemExit("no complete KKT iteration units recorded")
    for t in units:                       # expand the PSD blocks of the recorded units (host skron, outside any timing)
        for (c, off), w in zip(ks._psd, t["psd_w"]):
            if w is not None:
                c._skron(w)
                c._hs_valid = True
                c.get_Hs(t["hs"][off:off + c.numel * (c.numel + 1) // 2])

    # ---- 2. stage the trace in HBM
    dev = torch.device("cuda", local)
    for t in units:
        t["hs_d"] = torch.from_numpy(t["hs"]).to(dev)
        t["rhs_all_d"] = torch.from_numpy(np.stack(t["rhs"])).to(dev)      # [3, n+m]: constant, affine, combined
        t["rhs_d"] = [t["rhs_all_d"][k] for k in range(3)]
    out_d = torch.zeros(h.n + h.m, dtype=torch.float64, device=dev)
    out2_d = torch.zeros(2, h.n + h.m, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ir = dict(ir_enable=st.iterative_refinement_enable, reltol=st.iterative_refinement_reltol,
              abstol=st.iterative_refinement_abstol, max_iter=st.iterative_refinement_max_iter,
              stop_ratio=st.iterative_refinement_stop_ratio)
    has_soc = ks._soc_total > 0
    ir_steps_total = [0]

    def step(i):
        t = units[i % len(units)]
        h.set_hs_dev(t["hs_d"].data_ptr(), h.nHs)
        if has_soc:
            h.set_soc_batch(t["eta2"], t["u"], t["v"])
        ok, _, _ = h.refactor(st.static_regularization_enable, st.static_regularization_constant,
                              st.static_regularization_proportional)
        if args.sequential_solves:
            for r in t["rhs_d"]:
                h.setrhs_dev(r.data_ptr())
                ok2, steps = h.solve_dev(out_d.data_ptr(), **ir)
                ir_steps_total[0] += steps
                ok = ok and ok2
        else:
            # what the batching caller does (julia_standin/ipm.py KKTSystem): [-q; b] and the affine
            # right-hand side together, then the combined step's
            ok2, steps2 = h.solve_multi_dev(2, t["rhs_all_d"].data_ptr(), out2_d.data_ptr(), **ir)
            h.setrhs_dev(t["rhs_d"][2].data_ptr())
            ok3, steps3 = h.solve_dev(out_d.data_ptr(), **ir)
            ir_steps_total[0] += int(steps2.sum()) + steps3
            ok = ok and ok2 and ok3
        if not ok:
            raise SystemExit("numerical failure inside the timed region")

    from clarabel_jl_amd import batch

    warm = batch.timed_steps(step, 0, args.warmup)          # warm-up only (graph capture, caches)
    h.reset_timing()
    ir_steps_total[0] = 0
    elapsed = batch.timed_steps(step, args.steps, 0, dist=dist, device_sync=torch.cuda.synchronize, reduce_device=dev)
    tm = h.timing()
    factor_ms = tm["acc_factor_ms"] / max(1, tm["n_factor"])
    solve_ms = tm["acc_solve_ms"] / max(1, tm["n_solve_calls"])
    ldl_per_unit = tm["n_ldl_solves"] / max(1, args.steps)

    # ---- 3. roofline of the dominant kernel (k_update_stage, FP64 MFMA): live HIP-event timing of
    #         every update launch on the handle's stream, in a profiling pass over the same inputs
    cm = h.cost_model()
    h.set_profiling(True)
    upd_ms, d4 = [], []
    for i in range(3):
        t = units[i % len(units)]
        h.set_hs_dev(t["hs_d"].data_ptr(), h.nHs)
        h.refactor(st.static_regularization_enable, st.static_regularization_constant,
                   st.static_regularization_proportional)
        upd_ms.append(h.timing()["last_update_ms"])
        d4.append(h.profile())
        launches = h.profile_launches()
    # the same launches with every far tile kept in its stage's own launch (none riding in the next k_front_block launch): what the
    # dominant kernel's launches looked like before round 3's extra workgroups, for comparison
    h.set_profiling(2)
    d4n = []
    for i in range(3):
        t = units[i % len(units)]
        h.set_hs_dev(t["hs_d"].data_ptr(), h.nHs)
        h.refactor(st.static_regularization_enable, st.static_regularization_constant, st.static_regularization_proportional)
        d4n.append(h.profile())
    p4n = sorted(d4n, key=lambda p: p["dense4_ms"])[len(d4n) // 2]
    h.set_profiling(False)
    upd = float(np.median(upd_ms))
    p4 = sorted(d4, key=lambda p: p["dense4_ms"])[len(d4) // 2]
    # the just-in-time updates inside a front's update batches run inside k_front_block, not in the timed update kernels
    # ... and so does the partial last round of a front batch's far updates (extra workgroups of the NEXT k_front_block launch)
    flops_upd_kernels = cm["flops_update"] - p4.get("front_block_update_flops", 0.0) - p4.get("front_block_extra_flops", 0.0)
    agg = flops_upd_kernels / (upd * 1e-3) / 1e12 if upd > 0 else 0.0
    if p4["dense4_launches"] > 0 and p4["dense4_ms"] > 0:
        # dominant kernel alone: k_update_dense<4,4> (one wavefront per 64x64 tile), HIP events around each launch
        achieved = p4["dense4_flops"] / (p4["dense4_ms"] * 1e-3) / 1e12
        kern = "k_update_dense<4,4>"
        per_launch = dict(launches_per_refactor=p4["dense4_launches"],
                          avg_launch_us=round(1e3 * p4["dense4_ms"] / p4["dense4_launches"], 2),
                          flops_per_launch=p4["dense4_flops"] / p4["dense4_launches"])
    else:   # small problems never reach the large-launch variant: report the aggregate of all update kernels
        achieved, kern, per_launch = agg, "all Schur-update kernels", {}
    lms, lfl, ltl = launches
    if len(lms):
        # the launches one by one: flops per target tile tells the K = 320 passes over the front (2 * 64 * 64 * 320 = 2.6e6 per
        # tile) from the pass that carries the sparse part of the tree into it (many narrow sources per tile, a fraction of that)
        per_launch["launches"] = [dict(tiles=int(t), us=round(1e3 * m, 1), tflops=round(f / (m * 1e-3) / 1e12, 2), mflop_per_tile=round(f / t / 1e6, 3))
                                  for m, f, t in zip(lms, lfl, ltl)]
        dense = [(m, f) for m, f, t in zip(lms, lfl, ltl) if f / t >= 2.0e6]
        if dense:
            per_launch["full_K_launches"] = dict(count=len(dense), achieved=round(sum(f for _, f in dense) / (sum(m for m, _ in dense) * 1e-3) / 1e12, 3),
                                                 frac=round(sum(f for _, f in dense) / (sum(m for m, _ in dense) * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, 4))
    if p4.get("front_block_launches", 0) > 0:
        # the other big time block of a refactorisation: the front-batch kernels are bound by the pivot chain (64 sequential pivots
        # per 64-column panel), not by a throughput roofline -- reported as time per panel on that chain
        per_launch["critical_path_kernel"] = dict(
            kernel="k_front_block", bound="dependency latency (pivot chain)", launches_per_refactor=p4["front_block_launches"],
            panels=p4["front_block_panels"], ms_per_refactor=round(p4["front_block_ms"], 4),
            us_per_panel=round(1e3 * p4["front_block_ms"] / max(1, p4["front_block_panels"]), 2),
            floor_note="one diagonal workgroup's 64 pivots + the next one finishing the last streamed record behind them; the per-phase "
                       "stamps of the committed code are under profiles/ (<round>_*_cfg2a_front_block_stamps.txt, tools/fb2_trace.py) and in "
                       "DESIGN.md section 7 -- this line carries no numbers of its own",
            extra_update_tiles=p4.get("front_block_extra_tiles", 0), extra_update_flops=p4.get("front_block_extra_flops", 0.0),
            extra_note="dense update tiles of the partial last rounds of the far stages, executed by extra workgroups of these launches on compute "
                       "units the panel chain leaves idle; their flops are NOT in roofline.achieved / all_update_kernels (those time the update "
                       "kernels' own launches)")
    # algorithmic HBM bytes of one big dense-update launch, from the plan (every target tile read and written
    # once, every source panel row once): T tiles of 64 x 64 doubles; a launch over T = R (R + 1) / 2 tiles of a front touches R
    # row blocks of its K source columns (K = flops per tile / (2 * 64 * 64))
    alg_bytes = None
    if len(lms):
        per = []
        for m_, f_, t_ in zip(lms, lfl, ltl):
            K_ = f_ / t_ / (2.0 * 64 * 64)
            R_ = (np.sqrt(8.0 * t_ + 1.0) - 1.0) / 2.0
            per.append(2.0 * t_ * 64 * 64 * 8 + R_ * 64 * K_ * 8)
        alg_bytes = float(np.mean(per))
    ctr, why = pmc_counters(args.config)
    traffic = None
    if ctr is not None:
        traffic = dict(bytes_per_launch=ctr["bytes_per_launch"], read_x2_MB=ctr["read_x2_MB"], write_MB=ctr["write_MB"],
                       launches_in_trace=ctr["launches_in_trace"], source=ctr["source"], kernel_sources_sha1=ctr["kernel_sources_sha1"],
                       algorithmic_bytes_per_launch=None if alg_bytes is None else round(alg_bytes),
                       ratio_to_algorithmic=None if not alg_bytes else round(ctr["bytes_per_launch"] / alg_bytes, 2),
                       note="fabric-side bytes (L2 misses; Infinity-Cache hits are counted): FETCH_SIZE x2 (calibrated on this access shape, "
                            "profiles/r03_a_traffic_calibration.txt) + WRITE_SIZE per launch")
    # the refactorisation as a whole, from the same counter passes: what a latency-regime workload (cfg 1 / 2b: the big dense-update
    # launches never run, `launches_in_trace` 0) quotes as `traffic` -- against B_factor of SURVEY section 8(d)
    whole_traffic = None
    if ctr is not None and ctr.get("whole_refactor"):
        wr_ = ctr["whole_refactor"]
        whole_traffic = dict(bytes_per_refactor=wr_["bytes_per_refactor"], read_x2_MB=wr_["read_x2_MB"], write_MB=wr_["write_MB"],
                             dispatches_per_refactor=wr_.get("dispatches_per_refactor"), algorithmic_bytes_per_refactor=round(cm["bytes_factor"]),
                             ratio_to_algorithmic=round(wr_["bytes_per_refactor"] / cm["bytes_factor"], 2) if cm["bytes_factor"] else None,
                             source=ctr["source"], note=wr_.get("note"))
    big_launches = ctr is not None and ctr.get("launches_in_trace", 0) > 0
    if ctr is not None and not big_launches:
        traffic = None
    cm_f = cm["flops_factor"]
    whole = cm_f / (factor_ms * 1e-3) / 1e12
    # per-kernel table underneath the headline: the dense-update family (throughput-bound), the front-batch kernel (latency-bound; its
    # matrix-core flops -- in-batch updates + the far tiles riding in its launches -- from the counter file), all update kernels
    fbk = None
    if p4.get("front_block_launches", 0) > 0:
        cfb = (ctr or {}).get("front_block") or {}
        hwf = cfb.get("hw_flops_per_refactor")
        fbk = dict(kernel="k_front_block", bound="dependency latency (pivot chain of the front's diagonal tiles)",
                   ms_per_refactor=round(p4["front_block_ms"], 4), launches_per_refactor=p4["front_block_launches"],
                   share_of_factor_time=round(p4["front_block_ms"] / factor_ms, 3) if factor_ms > 0 else None,
                   hw_flops_per_refactor=hwf, achieved=None if not hwf else round(hwf / (p4["front_block_ms"] * 1e-3) / 1e12, 3),
                   frac=None if not hwf else round(hwf / (p4["front_block_ms"] * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, 4),
                   mfma_busy_pct=cfb.get("mfma_busy_pct"),
                   flops_note="SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 per refactorisation (counter file; null when it is stale): executed matrix-core flops of the "
                              "panel batches and of the dense update tiles that ride in these launches as extra workgroups")
    dense_k = dict(kernel=kern, bound="mfma", achieved=round(achieved, 3), frac=round(achieved / F64_MFMA_PEAK_TFLOPS, 4),
                   mfma_busy_pct=None if ctr is None else ctr.get("mfma_busy_pct"), traffic=traffic, traffic_unavailable=why,
                   share_of_factor_time=round(p4["dense4_ms"] / factor_ms, 3) if factor_ms > 0 and p4["dense4_ms"] > 0 else None,
                   note=("algorithmic flops of the k_update_dense<4,4> / k_update_dense_tail launches / their HIP-event time.  The far tiles that would form a "
                         "stage's partial last rounds ride in the next k_front_block launch instead (critical_path_kernel.extra_update_*): the launches "
                         "that remain are whole rounds plus the sparse-tree launch; `all_tiles_in_own_launches` = the same refactorisation with that off"),
                   all_tiles_in_own_launches=(None if not (p4n["dense4_launches"] > 0 and p4n["dense4_ms"] > 0) else dict(
                       achieved=round(p4n["dense4_flops"] / (p4n["dense4_ms"] * 1e-3) / 1e12, 3),
                       frac=round(p4n["dense4_flops"] / (p4n["dense4_ms"] * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, 4),
                       launches_per_refactor=p4n["dense4_launches"], ms_per_refactor=round(p4n["dense4_ms"], 4),
                       front_block_ms_per_refactor=round(p4n["front_block_ms"], 4))),
                   **per_launch)
    # the solves: HBM-bound by their algorithmic bytes (B_solve = 2 (8 + 4) nnz(L) + 40 N per LDL solve, + B_spmv per
    # refinement SpMV), in fact bound by the hand-off chain of the sweeps -- the fraction says how far
    n_ldl = tm["n_ldl_solves"]
    solve_bytes = n_ldl * (cm["bytes_solve"] + cm["bytes_spmv"])       # every LDL solve of a refined solve is followed by one residual SpMV
    solve_s = tm["acc_solve_ms"] * 1e-3
    hbm_solve = dict(bound="hbm", kernel="LDL solves + refinement SpMVs (k_front_fwd_sb / k_front_bwd_sb, k_fwd_seg / k_bwd_seg, k_spmv_residual, ...)",
                     achieved=round(solve_bytes / solve_s / 1e9, 1) if solve_s > 0 else None, peak=8000.0, unit="GB/s",
                     frac=round(solve_bytes / solve_s / 1e9 / 8000.0, 4) if solve_s > 0 else None,
                     frac_of_measured_peak_6300=round(solve_bytes / solve_s / 1e9 / 6300.0, 4) if solve_s > 0 else None,
                     ldl_solves=n_ldl, bytes_per_ldl_solve=cm["bytes_solve"], bytes_per_spmv=cm["bytes_spmv"], device_ms=round(tm["acc_solve_ms"], 3),
                     note="algorithmic bytes of the timed steps' solves / their HIP-event time (two concurrent solves of a pair are charged their common "
                          "span); the sweeps are latency-bound: ~22 hand-offs per front sweep, DESIGN.md section 5")
    roofline = dict(bound="mfma", achieved=round(whole, 3), peak=F64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(whole / F64_MFMA_PEAK_TFLOPS, 4),
                    kernel="numeric LDL^T factorisation, all kernels (k_front_block, k_update_dense<4,4>, k_update_dense<1,2>, k_update_gather, k_factor_panel, ...)",
                    what="SURVEY section 8(d): F_factor / t_factor with F_factor = sum_j (c_j^2 + 3 c_j) from the symbolic factor in use and t_factor = the "
                         "mean HIP-event time of hipkkt_refactor over the timed steps",
                    flops=cm_f, ms=round(factor_ms, 4),
                    traffic=(traffic["bytes_per_launch"] if traffic is not None else
                             (whole_traffic["bytes_per_refactor"] if (whole_traffic is not None and not big_launches) else None)),
                    traffic_of=("kernels.dense_update (per launch of the dominant throughput kernel)" if (traffic is not None or whole_traffic is None) else
                                "whole_refactor (every kernel of one refactorisation: this workload never reaches the big dense-update launches)"),
                    traffic_unavailable=why, whole_refactor=whole_traffic,
                    kernels=dict(dense_update=dense_k, front_block=fbk,
                                 all_update_kernels=dict(achieved=round(agg, 3), frac=round(agg / F64_MFMA_PEAK_TFLOPS, 4), ms_per_refactor=round(upd, 4),
                                                         flops_per_refactor=flops_upd_kernels)),
                    solves=hbm_solve,
                    peak_source="MI355X datasheet FP64 matrix (the figure every frac here is priced against)",
                    sustained_matrix_rate=dict(two_wavefronts_per_simd_tflops=67.0, one_wavefront_per_simd_tflops=50.2,
                                               frac_of_peak=round(67.0 / F64_MFMA_PEAK_TFLOPS, 3),
                                               source="profiles/r06_b_ubench_mfma_occupancy_and_dense_tile_fit.txt (tools/ubench_mfma_mix.hip, tools/ubench_macro.hip)",
                                               note="v_mfma_f64_16x16x4_f64 back to back on all 1024 SIMDs, measured in round 6: 67 TFLOP/s with two wavefronts "
                                                    "per SIMD (what the dense update kernel runs), 50 with one (what the update tiles riding in a front-batch launch "
                                                    "run).  The dense update's tile core: 1.22 us per k-step against 1.0 us of pure matrix work (82 %), plus ~35 us "
                                                    "per launch in which every wavefront reads / writes its 32 KB tile at the same time"))

    result = {
        "metric": "IPM iterations/sec + KKT factor+solve ms, 10k-var sparse QP, 1/2/4/8 GPU",
        "value": round(world * args.steps / elapsed, 4),
        "unit": "IPM-iterations/s (KKT iteration units: 1 update + 1 factor + 3 refined solves [constant-rhs + affine concurrently, then combined], inputs in HBM)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "N": h.N, "nnzK": h.nnzK, "nnzL": h.nnzL, "supernodes": h.nsuper,
                   "levels": h.nlevels, "ordering": ["minimum degree", "cone rows first, variables last", "user", "nested dissection"][h.ordering], "parallelism": f"{world} independent problem(s), one per GPU"},
        "ipm_iterations_per_s_end_to_end": e2e_rate,
        "value_is": "replayed KKT iteration units per second with every input resident in HBM (tier rule for `value`); "
                    "`ipm_iterations_per_s_end_to_end` is SURVEY section 8(d)'s rate: iterations / wall time of the whole IPM loop "
                    "incl. the host cone algebra and the PCIe transfers -- of the numpy stand-in of the Julia caller (no Julia has run), with the N2 hook "
                    "of julia/clarabel_l1_seam.patch on; median of three runs (end_to_end.runs)",
        "kkt_factor_ms": round(factor_ms, 4), "kkt_solve_ms_per_call": round(solve_ms, 4),
        "kkt_solve_calls_per_step": round(tm["n_solve_calls"] / max(1, args.steps), 2),
        "kkt_factor_plus_solves_ms": round(factor_ms + solve_ms * tm["n_solve_calls"] / max(1, args.steps), 4),
        "ldl_solves_per_step": round(ldl_per_unit, 2),
        "refined_block_solves": {"factorisations_with_some": int(h.counters()["accurate_factorisations"]), "blocks_last_factorisation": int(h.profile()["refined_blocks"]),
                                 "threshold": 64.0, "note": "wide diagonal blocks (> 16 columns, regular supernodes) whose explicit inverse has an entry above the "
                                 "threshold: their solves take one refinement step against the factored block (kernels.hip k_invert_diag_wide; hipkkt_get_profile out[10..11])"},
        "end_to_end": {"ipm_iterations": e2e_iters, "status": e2e_runs["n2_hook"]["status"], "iterations_per_s": e2e_rate,
                       "caller": "julia_standin/ipm.py (numpy stand-in of the untouched Julia IPM loop); no Julia has run",
                       "note": "headline = median of three runs with the N2 hook (reduced-system algebra by the plugin: wired by julia/clarabel_l1_seam.patch); "
                               "`runs` holds it next to the L1-contract-only run and the N2 + N4 run (N4 is NOT wired on the Julia side)",
                       "runs": e2e_runs,
                       "recording_run": {"ipm_iterations": rec_iters, "iterations_per_s": round(rec_iters / rec_time, 4), "status": sol.status,
                                         "note": "the run the replayed trace was recorded from (copies every Hs / rhs on the host)"},
                       "setup_s": round(t_setup, 3)},
        "roofline": roofline,
    }
    if args.device_scaling and rank == 0:
        # SURVEY section 8(f) row N1 end to end: the same solve with update_scaling! / get_Hs! formed by the plugin from (s, z)
        # (hipkkt_update_scaling): no Hs vector, no SOC (u, v, eta) and no PSD block crosses PCIe
        st1 = cl.Settings(device_id=local, device_scaling=True)
        s1 = cl.Solver(P, q, A, b, cones, st1, kktsolver_factory=lambda *a: HipKKTSolver(*a, **optkw))
        sol1 = s1.solve()
        result["end_to_end"]["with_device_scaling"] = {
            "status": sol1.status, "ipm_iterations": sol1.iterations,
            "iterations_per_s": round(sol1.iterations / s1.info.timers["IP iteration"], 4),
            "objective_rel_diff": float(abs(sol1.obj_val - sol.obj_val) / max(1.0, abs(sol.obj_val)))}
        del s1

    # ---- 4. CPU baseline + parity: the oracle (C restatement of the reference's :qdldl path), 1 thread, same
    #         permutation, KKT iteration units of the same trace; its solutions are compared with the HIP path's
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import ctypes as C

        from oracle.kkt_oracle import OracleKKTSolver

        perm = h.perm()
        data = solver.data
        cpu = OracleKKTSolver(data.P, data.A, solver.cones, data.m, data.n, st, ordering=perm)
        L = cpu.k.L
        n_, m_ = data.n, data.m
        picks = [min(len(units) - 1, len(units) // 2)]
        budget_s = args.cpu_budget
        unit_s, fac_s, okc_all = [], [], True
        par = dict(max_rel_dx=0.0, res_true_K=0.0, nreg_equal=True, eps_equal=True, ir_steps_equal=True, rhs_compared=0)
        while picks:
            t = units[picks.pop(0)]
            tc0 = time.perf_counter()
            L.oracle_kkt_update_Hs(cpu.k.h, t["hs"])
            if has_soc:
                off = 0
                for si, c in enumerate(ks._soc):
                    L.oracle_kkt_update_soc(cpu.k.h, si, t["eta2"][si], np.ascontiguousarray(t["u"][off:off + c.dim]),
                                            np.ascontiguousarray(t["v"][off:off + c.dim]))
                    off += c.dim
            eps = C.c_double(0)
            okc = L.oracle_kkt_regularize_and_refactor(cpu.k.h, int(st.static_regularization_enable),
                                                       st.static_regularization_constant,
                                                       st.static_regularization_proportional, C.byref(eps))
            fac_s.append(time.perf_counter() - tc0)
            xs_c, steps_c = [], []
            for r in t["rhs"]:
                lx, lz = np.zeros(n_), np.zeros(m_)
                cpu.kktsolver_setrhs(np.ascontiguousarray(r[:n_]), np.ascontiguousarray(r[n_:]))
                okc = cpu.kktsolver_solve(lx, lz) and okc
                xs_c.append(np.concatenate([lx, lz]))
                steps_c.append(cpu.last_ir_steps)
            unit_s.append(time.perf_counter() - tc0)
            okc_all = okc_all and bool(okc)
            # the same unit on the HIP path (outside any timing), solutions read back
            h.set_hs_dev(t["hs_d"].data_ptr(), h.nHs)
            if has_soc:
                h.set_soc_batch(t["eta2"], t["u"], t["v"])
            okg, eps_g, nreg_g = h.refactor(st.static_regularization_enable, st.static_regularization_constant,
                                            st.static_regularization_proportional)
            par["nreg_equal"] = par["nreg_equal"] and nreg_g == L.oracle_kkt_nreg(cpu.k.h)
            par["units_in_twin"] = par.get("units_in_twin", 0) + int(h.counters()["in_twin"])
            par["eps_equal"] = par["eps_equal"] and abs(eps_g - eps.value) <= 1e-16 * max(1.0, eps.value)
            for r, rd, xc, sc in zip(t["rhs"], t["rhs_d"], xs_c, steps_c):
                h.setrhs_dev(rd.data_ptr())
                ok2, steps_g = h.solve_dev(out_d.data_ptr(), **ir)
                xg = out_d.cpu().numpy()
                par["max_rel_dx"] = max(par["max_rel_dx"], float(np.max(np.abs(xg - xc)) / max(1.0, np.max(np.abs(xc)))))
                full = np.concatenate([xg, np.zeros(h.N - n_ - m_)])
                res = np.concatenate([r, np.zeros(h.N - n_ - m_)]) - cpu.k.symv(full)
                if h.N == n_ + m_:      # with expansion columns the (n+m)-part alone is not a solution of the big system
                    par["res_true_K"] = max(par["res_true_K"], float(np.max(np.abs(res)) / max(1.0, np.max(np.abs(r)))))
                par["ir_steps_equal"] = par["ir_steps_equal"] and steps_g == sc
                par["rhs_compared"] += 1
            # more units while they fit the budget (SURVEY section 8d asks for >= 3 where affordable)
            if len(unit_s) < 3 and sum(unit_s) + 1.15 * unit_s[-1] < budget_s:
                picks.append((len(units) // 2 + len(unit_s)) % len(units))
        t_unit = float(np.mean(unit_s))
        result["cpu_baseline"] = {
            "value": round(1.0 / t_unit, 5), "unit": "IPM-iterations/s (KKT iteration units)", "cores": 1, "kind": "port",
            "sample": f"{len(unit_s)} KKT iteration unit(s) (1 update + 1 QDLDL refactor + 3 refined solves each) of the same "
                      "trace, same permutation, oracle/ C restatement of the reference's :qdldl path, gcc -O3, one thread "
                      "(the reference's QDLDL is single-threaded, directldl_qdldl.jl:37)",
            "units_timed": len(unit_s), "factor_s": round(float(np.mean(fac_s)), 4), "unit_s": round(t_unit, 4),
            "unit_s_min_max": [round(float(min(unit_s)), 4), round(float(max(unit_s)), 4)],
            "host_cores_available": os.cpu_count(), "ok": okc_all}
        result["speedup_vs_cpu_baseline"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
        par["max_rel_dx"] = float(f"{par['max_rel_dx']:.3e}")
        par["res_true_K"] = float(f"{par['res_true_K']:.3e}")
        par["tolerance"] = 1e-10
        par["pass"] = bool(par["max_rel_dx"] <= 1e-10 and par["res_true_K"] <= 1e-10 and par["nreg_equal"] and par["eps_equal"])
        par["note"] = ("refined solutions of the recorded right-hand sides, HIP path vs oracle: max_rel_dx = max |x_hip - x_cpu|_inf / "
                       "max(1,|x_cpu|_inf); res_true_K = |b - K x_hip|_inf / max(1,|b|_inf) against the oracle's unregularised K")
        result["parity"] = par

    if rank == 0:
        # what the latency-bound kernels (front batches, panel kernels, sweeps) depend on and the matrix-core / memory-bound ones do
        # not: a slow line names its cause (1.6x between two boxes of the pool on exactly those kernels)
        try:
            from clarabel_jl_amd import hipkkt as _hk
            result["box_probe"] = _hk.box_probe(local)
            result["box_probe"]["note"] = ("hipkkt_box_probe: shader clock one busy wavefront gets (shader cycles / 100 MHz constant clock), round trip of a "
                                           "flag between two workgroups on the same / on different XCDs (relaxed agent-scope atomics); reference box of "
                                           "profiles/r05_a_box_probe.txt: 2.39 GHz, 1000 / 1085 ns")
        except Exception as e:      # the probe never fails the bench
            result["box_probe"] = {"error": str(e)}
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
