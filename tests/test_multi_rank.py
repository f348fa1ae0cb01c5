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
