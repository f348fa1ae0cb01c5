"""N>1 control path of bench.py (one independent problem set per rank, barrier + max-over-ranks
timing, no data-path collective) with 2 gloo ranks on CPU.  The per-rank "device work" is the CPU
oracle here (test infrastructure) -- what is under test is sharding, the barrier bracket and the
reductions, which are backend-independent."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import clarabel_jl_amd  # noqa: F401  (registers the dotted package directory)
import julia_standin as cl
from clarabel_jl_amd import batch, problems


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nprob, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.kkt_oracle import OracleKKTSolver

    mine = batch.shard(nprob, rank, world)
    sols = {}

    def step(i):
        k = mine[i % len(mine)]
        P, q, A, b, cones = problems.random_sparse_qp(40 + 5 * k, 70 + 7 * k, seed=100 + k, kA=3, kP=1)
        s = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: OracleKKTSolver(*a))
        sol = s.solve()
        sols[k] = (sol.status, sol.iterations, float(sol.obj_val))

    if rank == 0:   # rank 0 also exercises the several-problems-in-flight driver (host threads, one solver each)
        step(0)     # warm-up, like timed_steps does
        done = []

        def timed(_):
            done.extend(batch.run_concurrent(lambda i: step(i), list(range(len(mine))), in_flight=2))

        elapsed = batch.timed_steps(timed, steps=1, warmup=0, dist=dist)
        assert len(done) == len(mine)
    else:
        elapsed = batch.timed_steps(step, steps=len(mine), warmup=1, dist=dist)
    total = batch.gather_counts(len(mine), dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, mine, sols, elapsed, total))
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing():
    world, nprob = 2, 5
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nprob, out)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = sorted(k for _, mine, _, _, _ in gathered for k in mine)
    assert seen == list(range(nprob))                       # every problem exactly once
    assert all(t == nprob for *_, t in gathered)            # whole-job unit count agrees on all ranks
    el = [e for *_, e, _ in gathered]
    assert el[0] == el[1] and el[0] > 0                     # max-over-ranks is common to all ranks
    for _, mine, sols, _, _ in gathered:
        for k in mine:
            assert sols[k][0] == "SOLVED"
    # a problem's result does not depend on which rank solved it
    from oracle.kkt_oracle import OracleKKTSolver
    k = 3
    P, q, A, b, cones = problems.random_sparse_qp(40 + 5 * k, 70 + 7 * k, seed=100 + k, kA=3, kP=1)
    ref = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: OracleKKTSolver(*a)).solve()
    got = next(s[k] for _, mine, s, _, _ in gathered if k in mine)
    assert got[1] == ref.iterations and got[2] == float(ref.obj_val)


def test_shard_partition_properties():
    for n in (0, 1, 7, 256):
        for w in (1, 2, 4, 8):
            parts = [batch.shard(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_run_concurrent_keeps_order_and_results():
    """several problems in flight on one device (bench.py cfg 4): results come back in item order and equal the sequential ones"""
    from oracle.kkt_oracle import OracleKKTSolver

    def solve(k):
        P, q, A, b, cones = problems.random_sparse_qp(30 + 4 * k, 50 + 6 * k, seed=200 + k, kA=3, kP=1)
        sol = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: OracleKKTSolver(*a)).solve()
        return k, sol.status, sol.iterations, float(sol.obj_val)

    seq = batch.run_concurrent(solve, list(range(6)), 1)
    par = batch.run_concurrent(solve, list(range(6)), 3)
    assert seq == par and [r[0] for r in par] == list(range(6)) and all(r[1] == "SOLVED" for r in par)


def _worker8(rank, world, port, nprob, nworkers, out):
    """one rank of the cfg-4 batch driver on CPU: `nworkers` worker processes (bench.py --workers), pinned like the GPU run pins them"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import multiprocessing as pmp

    batch.cap_host_threads(1)
    cpus = batch.gpu_numa_cpus(rank, world, None)
    mine = batch.shard(nprob, rank, world)
    pool = pmp.get_context("spawn").Pool(nworkers, initializer=_pool_init, initargs=(cpus,))
    chunks = [mine[w::nworkers] for w in range(nworkers)]
    res = []

    def timed(_):
        for part in pool.map(_pool_chunk, chunks):
            res.extend(part)

    elapsed = batch.timed_steps(timed, steps=1, warmup=0, dist=dist)
    pool.close()
    pool.join()
    total = batch.gather_counts(len(res), dist)
    iters = batch.gather_counts(sum(r[2] for r in res), dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (rank, mine, res, elapsed, total, iters, cpus))
    if rank == 0:
        out.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def _pool_init(cpus):
    batch.pin_process(cpus)


def _pool_chunk(seeds):
    from oracle.kkt_oracle import OracleKKTSolver

    outp = []
    for k in seeds:
        P, q, A, b, cones = problems.random_sparse_qp(20 + 3 * (k % 7), 30 + 5 * (k % 7), seed=300 + k, kA=3, kP=1)
        sol = cl.Solver(P, q, A, b, cones, cl.Settings(), kktsolver_factory=lambda *a: OracleKKTSolver(*a)).solve()
        outp.append((k, sol.status, sol.iterations, os.getpid(), sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []))
    return outp


def test_eight_ranks_two_workers_each_solve_every_problem_once():
    """bench.py --config 4 --gpus 8 on CPU: 8 gloo ranks x 2 worker processes, 40 problems sharded round-robin over the ranks and
    then over each rank's workers.  Every problem is solved exactly once, the whole-job counts and the max-over-ranks time are common
    to all ranks, and each rank's workers run on the cores the placement helper gave that rank."""
    world, nprob, nworkers = 8, 40, 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, nprob, nworkers, out)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    solved = sorted(r[0] for _, _, res, *_ in gathered for r in res)
    assert solved == list(range(nprob))                              # every problem exactly once over ranks x workers
    assert all(t == nprob for *_, t, _, _ in gathered)               # whole-job unit count agrees on all ranks
    its = {i for *_, i, _ in gathered}
    assert len(its) == 1 and its.pop() == sum(r[2] for _, _, res, *_ in gathered for r in res)
    el = {e for _, _, _, e, *_ in gathered}
    assert len(el) == 1 and el.pop() > 0                             # the max over ranks is common
    for rank, mine, res, _, _, _, cpus in gathered:
        assert sorted(r[0] for r in res) == sorted(mine)
        assert all(r[1] == "SOLVED" for r in res)
        assert len({r[3] for r in res}) <= nworkers                  # at most `nworkers` processes worked for this rank
        for r in res:
            assert set(r[4]) <= set(cpus) or not r[4]                 # ... on this rank's cores
    assert batch.host_core_budget(8, 6) == 56                        # the figure DESIGN.md section 8 states for --gpus 8 --workers 6


def test_gpu_numa_cpus_partitions_the_allowed_cores():
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    parts = [batch.gpu_numa_cpus(r, 8, None) for r in range(8)]
    assert all(parts) and all(set(p) <= set(allowed) for p in parts)
    if len(allowed) >= 8:
        assert sum(len(p) for p in parts) <= len(allowed) and len({c for p in parts for c in p}) == sum(len(p) for p in parts)
